// Temporal (3 x 1 x 1) convolution of the factorised Conv3d (guided_diffusion/nn.py:53-87: `temporal_conv` over 'b c f h w -> (b h w) c f')
// for the fp32 (parity) configuration of the video UNet: the three-bf16-plane counterpart of csrc/igemm_h3.hip conv_frames_h3.
//
// conv_igemm_f32x3 runs these layers tap by tap: the input is fetched AND split into its hi / mid / lo planes three times (once per
// tap), the weights once per 128-row tile -- 105-170 TFLOP/s where the spatial halo kernel (one split per nine taps) reaches 200.
// Here a 512-thread workgroup owns ALL F frames of 64 consecutive pixels of one sample (F * 64 output rows) x 128 output channels.
// The reduction runs in PHASES of 16 input channels (one k-step of v_mfma_f32_32x32x16_bf16): a phase's operands -- the F x 64 x 16
// input block (three planes) and the three taps' 128 x 16 weight tiles (three planes) -- are split ONCE on their way into LDS and
// serve 114 MFMAs per wave: output frame f, tap t <- input frame f + t - 1 is a shifted row window of the same block, and the
// products against the out-of-range frames -1 and F are simply not issued.  Conversions per MAC: a third of the tap-by-tap
// kernel's on the input side, 1 / 3.5 on the weight side (448-row tiles); LDS operand bytes per MFMA 0.57 KB instead of 0.75.
//
// Pipeline (one barrier per phase, nothing else synchronises): phase g multiplies out of input half g & 1 (the 16-B pieces of a
// 64-B LDS row that carry channels 16 (g & 1) ... + 15) and weight stage g & 1; behind its first tap it splits the registers that hold
// phase g + 1's operands into the OTHER half / stage (whose readers passed the previous barrier) and requests phase g + 2 into the
// same registers -- a full phase of MFMAs for the loads to land.  K = 3 C is short (8 phases at C = 128), so the launch is PERSISTENT:
// one workgroup per CU walks tiles lin, lin + G, ... as one phase stream; the next tile's first two phases are already in LDS when a
// tile's epilogue starts, and its third is requested behind the epilogue's stores.
// Epilogue straight from the accumulators (a lane holds 16 rows of ONE output column: a store instruction covers two full 128-B
// lines): bias + the per-sample embedding row (`rowvec`, unet.py:248-257 emb_out) + the ResBlock's residual; per-64-row-block sums /
// sums of squares for the next GroupNorm, the two pixel halves of a block added in a fixed order through 4 KB of LDS.
// LDS: 3 x 28 KB input planes + 2 x 36 KB weight stages + 4 KB = 160 KB.  Arithmetic: the six plane products of conv_igemm_f32x3
// (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi, smallest first), fp32-equivalent (profiles/r04_f32x3_accuracy_speed.txt).
#include "common.h"
#include "x3t.h"

typedef __attribute__((address_space(1))) f32x4 gf32x4_x3t;
typedef __attribute__((ext_vector_type(8))) __bf16 bfx8_x3t;

struct ConvDescX3T {
    const float* x;          // [B, F, HW, C]
    const float* w;          // [Cout][3][C]
    const float* bias;       // [Cout] or null
    const float* rowvec;     // [B][Cout] or null (one row per sample)
    const float* residual;   // [B, F, HW, Cout] or null
    float* y;                // [B, F, HW, Cout]
    float* stats;            // optional [B * F * HW / 64][2][Cout]
    const float* zeros;
    int B, HW, C, Cout, K, tiles_b;
};

__device__ __forceinline__ int xcd_remap_x3t(int bid, int nblk) {
    int q = nblk >> 3, r = nblk & 7;
    int xcd = bid & 7, slot = bid >> 3;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

template <int F>
__global__ __launch_bounds__(512, 1) void conv_frames_x3(const ConvDescX3T p) {
    constexpr int PX = 64, BM = F * PX, BN = 128, NT = 512;
    constexpr int PA = BM * 64;                              // bytes of one plane of the input block: BM rows x 64 B (32 channels, both halves)
    constexpr int TAPB = BN * 32, PWB = 3 * TAPB, WST = 3 * PWB;   // weight stage: [plane][tap][128 rows x 32 B]
    constexpr int W_OFF = 3 * PA, ST_OFF = W_OFF + 2 * WST;
    constexpr int SMEM = ST_OFF + 4096;
    constexpr int AJ = (BM * 4 + NT - 1) / NT;               // float4 per thread of an input half (BM rows x 4); the last pass covers
    constexpr int ALAST_WAVES = (BM * 4 - (AJ - 1) * NT) / 64;   // only the first ALAST_WAVES waves
    static_assert(SMEM <= 160 * 1024 && (BM * 4) % 64 == 0, "LDS budget / whole-wave tail");
    static_assert(4 * 4 * 2 * 32 * 4 <= 4096 && F <= 8, "statistics exchange: four frames per round");
    __shared__ __attribute__((aligned(128))) unsigned char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = p.Cout / BN;
    const int total = p.B * p.tiles_b * tiles_n;
    const int G = gridDim.x;
    int lin = xcd_remap_x3t(blockIdx.x, G);
    if (lin >= total) return;
    const int nchunks = p.C >> 5;
    const bool a_tail = wid < ALAST_WAVES;                   // wave-uniform
    const float* zsrc = p.zeros;

    // ---- loader: item q = j * 512 + tid of an input half -> row q >> 2 (frame (q >> 2) / 64, pixel (q >> 2) % 64), float4 q & 3 of its 16 channels
    uint32_t a_src[AJ];                                      // element offset inside the tile's [F, HW, C] block
    int a_dst[AJ];                                           // LDS byte offset for half 0 (half 1: ^ 32)
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int q = j * NT + tid;
        const int row = (q >> 2) < BM ? (q >> 2) : 0, c4 = q & 3;
        const int f = row >> 6, px = row & 63;
        a_src[j] = ((uint32_t)f * (uint32_t)p.HW + (uint32_t)px) * (uint32_t)p.C + (uint32_t)c4 * 4u;
        a_dst[j] = row * 64 + ((((c4 >> 1) ^ ((row >> 2) & 3)) << 4) | ((c4 & 1) << 3));
    }
    // weights of a phase: tap j, row n = tid >> 2, float4 tid & 3 of its 16 channels.  LDS: 32-B rows in pairs, the pair's four 16-B
    // pieces XOR-ed with (n >> 3) & 3 -- a 16-lane group of a ds_read_b128 then touches every bank once
    const int wn_ld = tid >> 2, wc4 = tid & 3;
    const uint32_t w_src = (uint32_t)wn_ld * (uint32_t)p.K + (uint32_t)wc4 * 4u;
    const int w_dst = (wn_ld >> 1) * 64 + ((((((wn_ld & 1) << 1) | (wc4 >> 1)) ^ ((wn_ld >> 3) & 3)) << 4) | ((wc4 & 1) << 3));

    // the phase stream being LOADED: tile lq_lin, chunk lq_c, half lq_h (two phases ahead of the one computed, across tile boundaries)
    int lq_lin = lin, lq_c = 0, lq_h = 0;
    size_t lq_x = 0, lq_w = 0;
    auto tile_base = [&](int l, size_t& tx, size_t& tw) {
        const int t2 = l / tiles_n;
        const int bb = t2 / p.tiles_b;
        tx = ((size_t)(bb * F) * p.HW + (size_t)(t2 - bb * p.tiles_b) * PX) * p.C;
        tw = (size_t)(l - t2 * tiles_n) * BN * p.K;
    };
    tile_base(lq_lin, lq_x, lq_w);
    f32x4 ra[AJ], rw[3];
    // (no branches around the loads: past the end of the stream they read the zero line -- a conditional load makes the compiler's
    // wait-count bookkeeping fall back to vmcnt(0) everywhere)
    auto issue_loads = [&]() {
        const bool live = lq_lin < total;
        const float* xb = p.x + lq_x + lq_c * 32 + lq_h * 16;
        const float* wb = p.w + lq_w + lq_c * 32 + lq_h * 16;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const float* g = (live && (j < AJ - 1 || a_tail)) ? xb + a_src[j] : zsrc;
            ra[j] = *(const gf32x4_x3t*)(uint64_t)g;
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const float* g = live ? wb + (size_t)t * p.C + w_src : zsrc;
            rw[t] = *(const gf32x4_x3t*)(uint64_t)g;
        }
        lq_h ^= 1;
        if (lq_h == 0 && ++lq_c == nchunks) {
            lq_c = 0;
            lq_lin += G;
            if (lq_lin < total) tile_base(lq_lin, lq_x, lq_w);
        }
    };
    // split the registers into the planes of input half `hd` and weight stage `hd`
    auto store_regs = [&](int hd) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            if (j == AJ - 1 && !a_tail) break;
            uint32_t h0, m0, l0, h1, m1, l1;
            v2a_split3x2(ra[j][0], ra[j][1], h0, m0, l0);
            v2a_split3x2(ra[j][2], ra[j][3], h1, m1, l1);
            unsigned char* d = smem + (a_dst[j] ^ (hd << 5));      // half 1: the row's other two 16-B pieces
            *reinterpret_cast<uint2*>(d) = uint2{h0, h1};
            *reinterpret_cast<uint2*>(d + PA) = uint2{m0, m1};
            *reinterpret_cast<uint2*>(d + 2 * PA) = uint2{l0, l1};
        }
        unsigned char* wbs = smem + W_OFF + hd * WST + w_dst;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            uint32_t h0, m0, l0, h1, m1, l1;
            v2a_split3x2(rw[t][0], rw[t][1], h0, m0, l0);
            v2a_split3x2(rw[t][2], rw[t][3], h1, m1, l1);
            unsigned char* d = wbs + t * TAPB;
            *reinterpret_cast<uint2*>(d) = uint2{h0, h1};
            *reinterpret_cast<uint2*>(d + PWB) = uint2{m0, m1};
            *reinterpret_cast<uint2*>(d + 2 * PWB) = uint2{l0, l1};
        }
    };

    // ---- compute mapping: wave = (pixel half w, 32-channel group); sub-tile i = frame i
    const int w = wid >> 2, wn = (wid & 3) * 32;
    const int lr = lane & 31, lk = lane >> 5;
    int a_base = (w * 32 + lr) * 64;
    const int a_swz = (lr >> 2) & 3;
    const int b_off = ((wn + lr) >> 1) * 64 + (((((lr & 1) << 1) | lk) ^ ((lr >> 3) & 3)) << 4);

    f32x16 acc[F];
    // one tap of a phase: the three weight-plane fragments, then the frames in NB batches of at most FB (fragments of a batch requested
    // together, its MFMAs interleaved over the batch's accumulators: consecutive MFMAs never depend on each other)
    constexpr int FB = 3, NB = (F + FB - 1) / FB;
    auto tap = [&](int h, int t) {
        const unsigned char* wb = smem + W_OFF + h * WST + t * TAPB + b_off;
        const int pos = (((h << 1) | lk) ^ a_swz) << 4;
        bfx8_x3t b[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) b[q] = *reinterpret_cast<const bfx8_x3t*>(wb + q * PWB);
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) {
            const int i0 = bi * F / NB, i1 = (bi + 1) * F / NB;      // F = 7: frames {0, 1}, {2, 3}, {4, 5, 6}
            bfx8_x3t a[FB][3];
#pragma unroll
            for (int k = 0; k < FB; ++k) {
                const int i = i0 + k, src = i + t - 1;
                if (i < i1 && src >= 0 && src < F) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) a[k][q] = *reinterpret_cast<const bfx8_x3t*>(smem + q * PA + a_base + src * (PX * 64) + pos);
                }
            }
            asm volatile("" ::: "memory");
#define V2A_X3T_PROD(QA, QB)                                                                                            \
    _Pragma("unroll") for (int k = 0; k < FB; ++k) {                                                                    \
        const int i = i0 + k, src = i + t - 1;                                                                          \
        if (i < i1 && src >= 0 && src < F) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k][QA], b[QB], acc[i], 0, 0, 0); \
    }
            V2A_X3T_PROD(2, 0)      // lo  * hi
            V2A_X3T_PROD(0, 2)      // hi  * lo
            V2A_X3T_PROD(1, 1)      // mid * mid
            V2A_X3T_PROD(1, 0)      // mid * hi
            V2A_X3T_PROD(0, 1)      // hi  * mid
            V2A_X3T_PROD(0, 0)      // hi  * hi
#undef V2A_X3T_PROD
        }
    };

    // ---- prologue: phases 0 and 1 of the first tile into LDS, phase 2 requested
    issue_loads();
    store_regs(0);
    issue_loads();
    store_regs(1);
    issue_loads();
    __syncthreads();

    for (; lin < total; lin += G) {
        const int tmi = lin / tiles_n;
        const int b = tmi / p.tiles_b, p0 = (tmi - b * p.tiles_b) * PX, n0 = (lin - tmi * tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < F; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

        // invariant at a tile's start: LDS holds its phases 0 and 1, phase 2 is in flight into the registers
        for (int c = 0; c < nchunks; ++c) {
            asm volatile("" : "+v"(a_base));
            tap(0, 0);
            if (c > 0) {                                     // (phase 1 of the tile went into LDS before its epilogue / in the prologue)
                store_regs(1);
                issue_loads();
            }
            tap(0, 1);
            tap(0, 2);
            __syncthreads();
            tap(1, 0);
            store_regs(0);                                   // the next chunk's (or the next tile's) first phase
            issue_loads();
            tap(1, 1);
            tap(1, 2);
            __syncthreads();
        }
        // the next tile's phase 1 (requested behind the last phase's first tap) goes into LDS BEFORE this tile's stores are issued: a
        // wait for loads that has stores in the queue behind them costs the stores' acknowledgements; phase 2 is requested behind the stores
        store_regs(1);

        // ---- epilogue: lane = output column n0 + wn + lr, rows (r & 3) + 8 (r >> 2) + 4 lk of frame i's 32 pixels
        const int ncol = n0 + wn + lr;
        const float colb = (p.bias ? p.bias[ncol] : 0.f) + (p.rowvec ? p.rowvec[(size_t)b * p.Cout + ncol] : 0.f);
        float* st = reinterpret_cast<float*>(smem + ST_OFF);          // [4 frames][4 channel groups][2][32]
        float keep_s[F], keep_q[F];
#pragma unroll
        for (int i = 0; i < F; ++i) {
            const size_t m0 = ((size_t)(b * F + i) * p.HW) + p0 + w * 32;
            const size_t o0 = m0 * p.Cout + ncol;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][r] + colb;
            if (p.residual) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
                    v[r] += p.residual[o0 + (size_t)row * p.Cout];
                }
            }
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
                p.y[o0 + (size_t)row * p.Cout] = v[r];
                s += v[r];
                q += v[r] * v[r];
            }
            s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 32, 64);
            keep_s[i] = s;
            keep_q[i] = q;
        }
        if (p.stats) {
            // the two pixel halves of frame i's 64-row block: half 1 hands its sums over through LDS, half 0 adds (fixed order) and writes
#pragma unroll
            for (int i0 = 0; i0 < F; i0 += 4) {
                if (w == 1 && lane < 32) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (i0 + k < F) {
                            float* d = st + ((k * 4 + (wid & 3)) * 2) * 32 + lr;
                            d[0] = keep_s[i0 + k];
                            d[32] = keep_q[i0 + k];
                        }
                }
                __syncthreads();
                if (w == 0 && lane < 32) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (i0 + k < F) {
                            const float* d = st + ((k * 4 + (wid & 3)) * 2) * 32 + lr;
                            const size_t blk = (((size_t)(b * F + i0 + k) * p.HW) + p0) >> 6;
                            p.stats[blk * 2 * p.Cout + ncol] = keep_s[i0 + k] + d[0];
                            p.stats[blk * 2 * p.Cout + p.Cout + ncol] = keep_q[i0 + k] + d[32];
                        }
                }
                __syncthreads();
            }
        }
        issue_loads();                                       // the next tile's phase 2
        __syncthreads();                                     // its phase 1 (stored above) visible
    }
}

// 1 when conv_frames_x3 takes the problem: the temporal tap of the factorised Conv3d seen as a 2-d conv over [B, F, HW, C] with a 3 x 1
// filter, stride 1, pad (1, 0); F = 7 (Libero: seven predicted frames), HW % 64 == 0, C % 32 == 0, Cout % 128 == 0, one embedding row
// per sample (rows_per_batch = F * HW) if any, enough tiles for the chip.
int conv_frames_x3_eligible(int B, int F, int HW, int C, int Cout, int rows_per_batch, bool has_rowvec) {
    if (F != 7 || HW % 64 || C % 32 || Cout % 128) return 0;
    if (has_rowvec && rows_per_batch != F * HW) return 0;
    if ((long)B * (HW / 64) * (Cout / 128) < 208) return 0;
    if ((double)F * HW * C >= 4294967296.0 || (double)128 * 3 * C >= 4294967296.0) return 0;
    return 1;
}

int conv_frames_x3_launch(const float* x, const float* w_packed, const float* bias, const float* rowvec, const float* residual, float* y,
                          const void* zeros, int B, int F, int HW, int C, int Cout, int rows_per_batch, float* stats, hipStream_t stream) {
    if (!x || !w_packed || !zeros || !y || B <= 0) return V2A_ERR_ARG;
    if (!conv_frames_x3_eligible(B, F, HW, C, Cout, rows_per_batch, rowvec != nullptr)) return V2A_ERR_ARG;
    if ((((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)zeros | (uintptr_t)y) & 15) != 0) return V2A_ERR_ARG;
    ConvDescX3T p;
    p.x = x; p.w = w_packed; p.bias = bias; p.rowvec = rowvec; p.residual = residual; p.y = y; p.stats = stats;
    p.zeros = (const float*)zeros;
    p.B = B; p.HW = HW; p.C = C; p.Cout = Cout; p.K = 3 * C;
    p.tiles_b = HW / 64;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (ncu <= 0) ncu = 256;
    }
    const int total = B * p.tiles_b * (Cout / 128);
    hipLaunchKernelGGL((conv_frames_x3<7>), dim3(total < ncu ? total : ncu), dim3(512), 0, stream, p);      // one persistent workgroup per CU
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

extern "C" {
// 1 when v2a_conv2d_fwd_dma_f32 / _d run this (3 x 1) temporal conv over x [B, F, HW, C] on the frame-stack kernel (three-plane mode)
int v2a_conv2d_x3t_eligible(int B, int F, int HW, int C, int Cout, int rows_per_batch, int has_rowvec) {
    return conv_frames_x3_eligible(B, F, HW, C, Cout, rows_per_batch, has_rowvec != 0);
}
}
