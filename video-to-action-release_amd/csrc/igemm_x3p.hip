// 3 x 3 / stride 1 / pad 1 convolution (optionally behind a nearest x2 upsample) for the fp32 (parity) configuration of the video UNet:
// the spatial half of the factorised Conv3d (guided_diffusion/nn.py:53-87 `spatial_conv`, unet.py:105-115 Upsample) on three bf16
// planes per fp32 operand, in the phase structure of csrc/igemm_x3t.hip conv_frames_x3.
//
// conv_halo_x3 (csrc/igemm_h.hip) -- 128 pixels x 64 channels per 256-thread workgroup, a weight tile split per (chunk, tap) step of 24
// MFMAs per wave, a barrier per step -- holds 190-205 TFLOP/s on these layers; the temporal kernel, with 448 x 128 tiles, 114 MFMAs per
// wave and barrier, reaches 225-280.  Same recipe here: a 512-thread workgroup owns a 16 x 16 pixel patch of one image (256 output rows)
// x 128 output channels, wave = 64 rows (four patch rows) x 64 channels.  The reduction runs in PHASES (32-channel chunk c, 16-channel
// half h, filter row kh): a phase multiplies the three taps (kh, 0 .. 2) -- shifted row windows of the patch's 18 x 18 HALO, held in LDS
// as three bf16 plane images -- against three 128 x 16 weight tiles (three planes each): 72 MFMAs per wave, one barrier.
//   * weights: registers -> split -> LDS stage (phase + 1) & 1 behind the phase's first tap, then the request for phase + 2 into the same
//     registers (a full phase of MFMAs for the loads to land);
//   * halo: the 16-B pieces of the 64-B LDS rows that carry channels 16 h .. + 15 form "half h"; while the three phases of (c, h) read
//     half h, the halo of the next (c, h) is requested (filter row 0) and split into the other half (filter row 2).
// Conversions per MAC: 1/9 of the tap-by-tap kernel's on the input side, half of conv_halo_x3's on the weight side (256-row tiles).
// K = 9 C gives 24 phases at C = 128; the launch is PERSISTENT (one workgroup per CU walks tiles lin, lin + G, ... as one phase stream):
// the next tile's halo and first two weight phases are in LDS when a tile's epilogue starts, the third is requested behind its stores.
// Epilogue straight from the accumulators (a lane holds 16 rows of one output column; a store instruction covers two 128-B lines):
// bias + optional fp32 residual.  LDS: 3 x 20.3 KB halo planes + 2 x 36 KB weight stages = 133 KB.
// Arithmetic: conv_igemm_f32x3's six plane products, smallest first (fp32-equivalent, profiles/r04_f32x3_accuracy_speed.txt).
#include "common.h"
#include "x3t.h"

typedef __attribute__((address_space(1))) f32x4 gf32x4_x3p;
typedef __attribute__((ext_vector_type(8))) __bf16 bfx8_x3p;

struct ConvDescX3P {
    const float* x;          // [N, H, W, C] (ups: [N, H/2, W/2, C]; H, W below are the conv's map)
    const float* w;          // [Cout][3][3][C]
    const float* bias;       // [Cout] or null
    const float* residual;   // [N, H, W, Cout] or null
    float* y;                // [N, H, W, Cout]
    const float* zeros;
    int N, H, W, C, Cout, K, tiles_x, tiles_img, ups;
    // GN = true: the conv reads act(GroupNorm(x)) -- z = (x - mean[s, g]) * rstd[s, g] * gamma[c] + beta[c], the arithmetic of
    // gn_apply_fwd_rows (csrc/norm.hip) -- applied to the halo in registers on its way into LDS (padding stays zero): the normalised
    // tensor is never written (GroupNorm32 + SiLU in front of the ResBlock convs, guided_diffusion/unet.py:181-197, nn.py:95-97)
    const float* mean;       // [N / fps][G]
    const float* rstd;
    const float* gamma;      // [C]
    const float* beta;
    int G, fps, act;         // groups; images (frames) per GroupNorm sample; ACT_SILU / ACT_NONE
    // KS = 2 ("upsample classes"): Upsample (nearest x2) + 3 x 3 conv as FOUR 2 x 2 convs over the SOURCE map, one per parity class of
    // the output pixel (2a + ph, 2b + pw).  Of the three filter rows, two read the same source row (kh = 1, 2 for ph = 0; kh = 0, 1 for
    // ph = 1), so their weights are summed in advance (v2a_pack_weight_ups4: [4 classes][Cout][2][2][C]) and a class needs 4 of the 9
    // products per output.  A tile = 16 x 16 pixels (a, b) of one class: 17 x 17 source window starting at (a0 - 1 + ph, b0 - 1 + pw),
    // outputs scattered with stride 2.  x = the source [N, H/2, W/2, C]; H, W stay the OUTPUT map; tiles_x / tiles_img count 16 x 16
    // patches of the source map.
};

__device__ __forceinline__ int xcd_remap_x3p(int bid, int nblk) {
    int q = nblk >> 3, r = nblk & 7;
    int xcd = bid & 7, slot = bid >> 3;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}
// MFMA row m of a 32-row sub-tile (two patch rows of 16 pixels) carries pixel patch16_perm_x3p(m): every 16-lane group of a ds_read_b128
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}: MI355X_MICROARCH.md, LDS) then reads ONE patch row = 16 consecutive halo slots, which the
// (slot >> 2) & 3 piece swizzle spreads over all 64 banks under every tap shift (csrc/igemm_h3.hip lds_group_perm3 is the same map)
__device__ __forceinline__ int patch16_perm_x3p(int m) {
    const int qd = m >> 2;
    return ((__builtin_popcount(qd) & 1) << 4) | ((qd >> 1) << 2) | (m & 3);
}

template <bool GN, int KS>
__global__ __launch_bounds__(512, 1) void conv_patch_x3(const ConvDescX3P p) {
    constexpr int BN = 128, NT = 512;
    constexpr bool UPS4 = KS == 2;                           // the four 2 x 2 class convs of Upsample + 3 x 3 (see ConvDescX3P)
    static_assert(KS == 3 || KS == 2, "filter size");
    constexpr int HWD = 16 + KS - 1, HS = HWD * HWD;         // halo: 18 x 18 (17 x 17) slots of 64 B (32 channels, both halves)
    constexpr int PHB = HS * 64;                             // bytes of one plane of the halo image
    constexpr int TAPB = BN * 32, PWB = KS * TAPB, WST = 3 * PWB;  // weight stage: [plane][tap kw][128 rows x 32 B]
    constexpr int W_OFF = 3 * PHB;
    constexpr int SMEM = W_OFF + 2 * WST;
    constexpr int AJ = (HS * 4 + NT - 1) / NT;               // float4 per thread of a halo half (324 slots x 4)
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(128))) unsigned char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = p.Cout / BN;
    const int ncls = UPS4 ? 4 : 1;
    const int total = p.N * ncls * p.tiles_img * tiles_n;
    const int G = gridDim.x;
    int lin = xcd_remap_x3p(blockIdx.x, G);
    if (lin >= total) return;
    const int nchunks = p.C >> 5;
    const float* zsrc = p.zeros;
    const int srcH = (p.ups || UPS4) ? p.H >> 1 : p.H, srcW = (p.ups || UPS4) ? p.W >> 1 : p.W;
    // tile index -> (image, class, patch row, patch column); class 0 without UPS4
    auto tile_of = [&](int l, int& img, int& cls, int& ty, int& tx) {
        const int tm = l / tiles_n;
        const int per_img = ncls * p.tiles_img;
        img = tm / per_img;
        const int rem = tm - img * per_img;
        cls = rem / p.tiles_img;
        const int t = rem - cls * p.tiles_img;
        ty = t / p.tiles_x;
        tx = t - ty * p.tiles_x;
    };

    // ---- halo loader: item q = j * 512 + tid -> slot q >> 2 (halo pixel (slot / 18, slot % 18)), float4 q & 3 of the half's 16 channels
    int a_dst[AJ];                                           // LDS byte offset for half 0 (half 1: ^ 32); -1: no such slot
    int a_hy[AJ], a_hx[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int q = j * NT + tid;
        const int slot = q >> 2, c4 = q & 3;
        a_hy[j] = slot / HWD;
        a_hx[j] = slot - a_hy[j] * HWD;
        a_dst[j] = slot < HS ? slot * 64 + ((((c4 >> 1) ^ ((slot >> 2) & 3)) << 4) | ((c4 & 1) << 3)) : -1;
    }
    // the halo stream being LOADED (one (chunk, half) period ahead of the one computed, across tile boundaries)
    int la_lin = lin, la_c = 0, la_h = 0;
    uint32_t a_off[AJ];                                      // element offset of the slot's pixel in x (+ float4 index), 0xffffffff: zero line
    int la_ns = 0;                                           // GN: GroupNorm sample of the tile being loaded
    auto a_tile_setup = [&](int l) {
        int img, cls, ty, tx;
        tile_of(l, img, cls, ty, tx);
        if (GN) la_ns = img / p.fps;
        // window origin: one pixel up / left of the patch; the lower / right classes of UPS4 start AT the patch
        const int oy = UPS4 ? (cls >> 1) - 1 : -1, ox = UPS4 ? (cls & 1) - 1 : -1;
        const int limH = UPS4 ? srcH : p.H, limW = UPS4 ? srcW : p.W;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int ih = ty * 16 + a_hy[j] + oy, iw = tx * 16 + a_hx[j] + ox;
            const bool ok = a_dst[j] >= 0 && (unsigned)ih < (unsigned)limH && (unsigned)iw < (unsigned)limW;
            const int ihs = p.ups ? ih >> 1 : ih, iws = p.ups ? iw >> 1 : iw;
            a_off[j] = ok ? ((uint32_t)(img * srcH + ihs) * (uint32_t)srcW + (uint32_t)iws) * (uint32_t)p.C + (uint32_t)((tid & 3) * 4) : 0xffffffffu;
        }
    };
    a_tile_setup(la_lin);
    f32x4 ra[AJ], rw[3];
    // GN: scale / shift of this thread's four channels of the period in flight, and which of its slots carry image pixels
    f32x4 gn_gm = {0.f, 0.f, 0.f, 0.f}, gn_bt = {0.f, 0.f, 0.f, 0.f};
    float gn_mu = 0.f, gn_rs = 0.f;
    int ra_ok = 0;
    const int cg = GN ? p.C / p.G : 1;
    // (no branches around the loads: past the end of the stream and outside the image they read the zero line -- a conditional load makes
    // the compiler's wait-count bookkeeping fall back to vmcnt(0) everywhere)
    auto issue_a = [&]() {
        const bool live = la_lin < total;
        const float* xb = p.x + la_c * 32 + la_h * 16;
        ra_ok = 0;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const bool ok = live && a_off[j] != 0xffffffffu;
            const float* g = ok ? xb + a_off[j] : zsrc;
            ra[j] = *(const gf32x4_x3p*)(uint64_t)g;
            ra_ok |= ok ? (1 << j) : 0;
        }
        if (GN) {
            const int c0 = la_c * 32 + la_h * 16 + (tid & 3) * 4;
            const int sg = (live ? la_ns : 0) * p.G + c0 / cg;
            gn_gm = *(const gf32x4_x3p*)(uint64_t)(p.gamma + c0);
            gn_bt = *(const gf32x4_x3p*)(uint64_t)(p.beta + c0);
            gn_mu = p.mean[sg];
            gn_rs = p.rstd[sg];
        }
        la_h ^= 1;
        if (la_h == 0 && ++la_c == nchunks) {
            la_c = 0;
            la_lin += G;
            if (la_lin < total) a_tile_setup(la_lin);
        }
    };
    auto store_a = [&](int hd) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            if (a_dst[j] < 0) continue;
            if (GN && ((ra_ok >> j) & 1)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float z = (ra[j][e] - gn_mu) * gn_rs * gn_gm[e] + gn_bt[e];
                    ra[j][e] = act_fwd(z, p.act);
                }
            }
            uint32_t h0, m0, l0, h1, m1, l1;
            v2a_split3x2(ra[j][0], ra[j][1], h0, m0, l0);
            v2a_split3x2(ra[j][2], ra[j][3], h1, m1, l1);
            unsigned char* d = smem + (a_dst[j] ^ (hd << 5));      // half 1: the row's other two 16-B pieces
            *reinterpret_cast<uint2*>(d) = uint2{h0, h1};
            *reinterpret_cast<uint2*>(d + PHB) = uint2{m0, m1};
            *reinterpret_cast<uint2*>(d + 2 * PHB) = uint2{l0, l1};
        }
    };
    // ---- weight loader of a phase (c, h, kh): tap kw = j, row n = tid >> 2, float4 tid & 3 of its 16 channels.  LDS: 32-B rows in pairs,
    // the pair's four 16-B pieces XOR-ed with (n >> 3) & 3 -- a 16-lane group of a ds_read_b128 then touches every bank once
    const int wn_ld = tid >> 2, wc4 = tid & 3;
    const uint32_t w_src = (uint32_t)wn_ld * (uint32_t)p.K + (uint32_t)wc4 * 4u;
    const int w_dst = (wn_ld >> 1) * 64 + ((((((wn_ld & 1) << 1) | (wc4 >> 1)) ^ ((wn_ld >> 3) & 3)) << 4) | ((wc4 & 1) << 3));
    int lw_lin = lin, lw_c = 0, lw_h = 0, lw_kh = 0;
    auto w_base_of = [&](int l) -> size_t {                  // first weight row of the tile: [class][Cout][KS][KS][C]
        int img, cls, ty, tx;
        tile_of(l, img, cls, ty, tx);
        return ((size_t)cls * p.Cout + (size_t)(l % tiles_n) * BN) * p.K;
    };
    size_t lw_base = w_base_of(lin);
    auto issue_w = [&]() {
        const bool live = lw_lin < total;
        const float* wb = p.w + lw_base + (size_t)(lw_kh * KS) * p.C + lw_c * 32 + lw_h * 16 + w_src;
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            const float* g = live ? wb + (size_t)t * p.C : zsrc;
            rw[t] = *(const gf32x4_x3p*)(uint64_t)g;
        }
        if (++lw_kh == KS) {
            lw_kh = 0;
            lw_h ^= 1;
            if (lw_h == 0 && ++lw_c == nchunks) {
                lw_c = 0;
                lw_lin += G;
                if (lw_lin < total) lw_base = w_base_of(lw_lin);
            }
        }
    };
    auto store_w = [&](int stage) {
        unsigned char* wbs = smem + W_OFF + stage * WST + w_dst;
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            uint32_t h0, m0, l0, h1, m1, l1;
            v2a_split3x2(rw[t][0], rw[t][1], h0, m0, l0);
            v2a_split3x2(rw[t][2], rw[t][3], h1, m1, l1);
            unsigned char* d = wbs + t * TAPB;
            *reinterpret_cast<uint2*>(d) = uint2{h0, h1};
            *reinterpret_cast<uint2*>(d + PWB) = uint2{m0, m1};
            *reinterpret_cast<uint2*>(d + 2 * PWB) = uint2{l0, l1};
        }
    };

    // ---- compute mapping: wave = (64-row group wm: patch rows 4 wm .. + 3, 64-channel group wn); sub-tile (i, j) = 32 rows x 32 channels
    const int wm = wid >> 1, wn = (wid & 1) * 64;
    const int lr = lane & 31, lk = lane >> 5;
    int slot0[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pr = wm * 64 + i * 32 + patch16_perm_x3p(lr);
        slot0[i] = (pr >> 4) * HWD + (pr & 15);
    }
    int b_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) b_off[j] = ((wn + j * 32 + lr) >> 1) * 64 + (((((lr & 1) << 1) | lk) ^ ((lr >> 3) & 3)) << 4);

    f32x16 acc[2][2];
    // one tap (kh, kw) of a phase: all twelve operand fragments requested together, then 24 MFMAs interleaved over the four accumulators
    auto tap = [&](int h, int stage, int kh, int kw) {
        const unsigned char* wb = smem + W_OFF + stage * WST + kw * TAPB;
        const int kp = (h << 1) | lk;
        bfx8_x3p a[2][3], b[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int sl = slot0[i] + kh * HWD + kw;
            const unsigned char* ab = smem + sl * 64 + ((kp ^ ((sl >> 2) & 3)) << 4);
#pragma unroll
            for (int q = 0; q < 3; ++q) a[i][q] = *reinterpret_cast<const bfx8_x3p*>(ab + q * PHB);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q) b[j][q] = *reinterpret_cast<const bfx8_x3p*>(wb + q * PWB + b_off[j]);
        asm volatile("" ::: "memory");
#define V2A_X3P_PROD(QA, QB)                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                               \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                           \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][QA], b[j][QB], acc[i][j], 0, 0, 0);
        V2A_X3P_PROD(2, 0)      // lo  * hi
        V2A_X3P_PROD(0, 2)      // hi  * lo
        V2A_X3P_PROD(1, 1)      // mid * mid
        V2A_X3P_PROD(1, 0)      // mid * hi
        V2A_X3P_PROD(0, 1)      // hi  * mid
        V2A_X3P_PROD(0, 0)      // hi  * hi
#undef V2A_X3P_PROD
    };

    // ---- prologue: halo (0, 0) and weight phases 0 and 1 of the first tile into LDS, weight phase 2 requested
    issue_a();
    issue_w();
    store_a(0);
    store_w(0);
    issue_w();
    store_w(1);
    issue_w();
    __syncthreads();

    for (; lin < total; lin += G) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // invariant at a tile's start: LDS holds its halo (0, 0) and weight phases 0 and 1; weight phase 2 is in flight into the registers
        for (int c = 0; c < nchunks; ++c) {
#pragma unroll
            for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(slot0[i]));
#define V2A_X3P_PHASE(H_, KH_)                                                                                                  \
    {                                                                                                                           \
        constexpr int stage_ = ((H_) * KS + (KH_)) & 1; /* 2 KS phases per chunk (even): the stage of phase (c, h, kh) */            \
        tap(H_, stage_, KH_, 0);                                                                                                \
        if (!((H_) == 0 && (KH_) == 0) || c > 0) {      /* (weight phase 1 of a tile went into LDS before its epilogue) */        \
            store_w(stage_ ^ 1);                                                                                                \
            issue_w();                                                                                                          \
        }                                                                                                                       \
        if ((KH_) == 0) issue_a();                       /* the halo of the next (chunk, half) ... */                             \
        if ((KH_) == KS - 1) store_a((H_) ^ 1);          /* ... into the half whose readers passed the barriers of the last (c, h) */ \
        tap(H_, stage_, KH_, 1);                                                                                                \
        if (KS == 3) tap(H_, stage_, KH_, 2);                                                                                   \
        __syncthreads();                                                                                                        \
    }
            V2A_X3P_PHASE(0, 0) V2A_X3P_PHASE(0, 1)
            if constexpr (KS == 3) V2A_X3P_PHASE(0, 2)
            V2A_X3P_PHASE(1, 0) V2A_X3P_PHASE(1, 1)
            if constexpr (KS == 3) V2A_X3P_PHASE(1, 2)
#undef V2A_X3P_PHASE
        }
        // the next tile's weight phase 1 (requested behind the last phase's first tap) goes into LDS BEFORE this tile's stores are issued: a
        // wait for loads that has stores in the queue behind them costs the stores' acknowledgements; phase 2 is requested behind the stores
        store_w(1);

        // ---- epilogue: lane = output column n0 + wn + j * 32 + lr; register r = row (r & 3) + 8 (r >> 2) + 4 lk of sub-tile i
        const int n0 = (lin % tiles_n) * BN;
        int img, cls, ty, tx;
        tile_of(lin, img, cls, ty, tx);
        // output pixel of patch pixel (py, px): (ty * 16 + py, tx * 16 + px), or its class twin (2 (..) + ph, 2 (..) + pw)
        const int ost = UPS4 ? 2 : 1;
        const size_t pix0 = ((size_t)img * p.H + (size_t)(ty * 16 * ost + (UPS4 ? (cls >> 1) : 0))) * p.W + tx * 16 * ost + (UPS4 ? (cls & 1) : 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ncol = n0 + wn + j * 32 + lr;
            const float colb = p.bias ? p.bias[ncol] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pr = wm * 64 + i * 32 + patch16_perm_x3p((r & 3) + 8 * (r >> 2) + 4 * lk);
                    const size_t o = (pix0 + (size_t)((pr >> 4) * ost) * p.W + (pr & 15) * ost) * p.Cout + ncol;
                    float v = acc[i][j][r] + colb;
                    if (p.residual) v += p.residual[o];
                    p.y[o] = v;
                }
            }
        }
        issue_w();                                           // the next tile's weight phase 2
        __syncthreads();                                     // its phase 1 (stored above) visible
    }
}

// 1 when conv_patch_x3 takes the problem: 3 x 3 / stride 1 / pad 1 over an H x W map (H, W the conv's map: twice the source's when `ups`)
// of 16 x 16 patches, C % 32 == 0, Cout % 128 == 0, enough tiles that the persistent launch's last round wastes little.
int conv_patch_x3_eligible(int N, int H, int W, int C, int Cout, int ncu) {
    if (H % 16 || W % 16 || C % 32 || Cout % 128) return 0;
    const long tiles = (long)N * (H / 16) * (W / 16) * (Cout / 128);
    if (ncu <= 0) ncu = 256;
    const long rounds = (tiles + ncu - 1) / ncu;
    if (tiles < 208 || tiles * 100 < rounds * ncu * 85) return 0;      // >= 85 % of the persistent launch's slots carry a tile
    if ((double)N * H * W * C >= 4294967296.0 || (double)Cout * 9 * C >= 4294967296.0) return 0;
    return 1;
}

static int x3p_ncu() {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (ncu <= 0) ncu = 256;
    }
    return ncu;
}

int conv_patch_x3_ups4_eligible(int N, int H, int W, int C, int Cout, int ncu);
static int conv_patch_x3_launch_gn(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, const void* zeros,
                                   int N, int H, int W, int C, int Cout, int ups, const float* mean, const float* rstd, const float* gamma,
                                   const float* beta, int G, int fps, int act, hipStream_t stream);
int conv_patch_x3_launch(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, const void* zeros, int N,
                         int H, int W, int C, int Cout, int ups, hipStream_t stream) {
    return conv_patch_x3_launch_gn(x, w_packed, bias, residual, y, zeros, N, H, W, C, Cout, ups, nullptr, nullptr, nullptr, nullptr, 0, 1, 0, stream);
}
static int conv_patch_x3_launch_gn(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, const void* zeros,
                                   int N, int H, int W, int C, int Cout, int ups, const float* mean, const float* rstd, const float* gamma,
                                   const float* beta, int G, int fps, int act, hipStream_t stream) {
    if (!x || !w_packed || !zeros || !y || N <= 0) return V2A_ERR_ARG;
    const int ncu = x3p_ncu();
    const bool ups4 = ups == 2;                   // ups: 0 none, 1 nearest x2 folded into the gather, 2 the four class convs (pre-summed pack)
    if (ups4 ? !conv_patch_x3_ups4_eligible(N, H, W, C, Cout, ncu) : !conv_patch_x3_eligible(N, H, W, C, Cout, ncu)) return V2A_ERR_ARG;
    if ((((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)zeros) & 15) != 0 || (((uintptr_t)y | (uintptr_t)residual | (uintptr_t)bias) & 3) != 0)
        return V2A_ERR_ARG;
    ConvDescX3P p;
    p.x = x; p.w = w_packed; p.bias = bias; p.residual = residual; p.y = y; p.zeros = (const float*)zeros;
    p.N = N; p.H = H; p.W = W; p.C = C; p.Cout = Cout; p.K = (ups4 ? 4 : 9) * C; p.ups = ups == 1 ? 1 : 0;
    p.tiles_x = (ups4 ? W / 2 : W) / 16;
    p.tiles_img = ((ups4 ? H / 2 : H) / 16) * p.tiles_x;
    const int total = N * (ups4 ? 4 : 1) * p.tiles_img * (Cout / 128);
    const dim3 grid(total < ncu ? total : ncu);   // one persistent workgroup per CU
    p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.beta = beta; p.G = G; p.fps = fps > 0 ? fps : 1; p.act = act;
    if (mean) {
        if (ups4 || !rstd || !gamma || !beta || G <= 0 || C % G || (C / G) % 4 || N % p.fps || (act != ACT_NONE && act != ACT_SILU) ||
            (((uintptr_t)gamma | (uintptr_t)beta) & 15))
            return V2A_ERR_ARG;
        hipLaunchKernelGGL((conv_patch_x3<true, 3>), grid, dim3(512), 0, stream, p);
    } else if (ups4) {
        hipLaunchKernelGGL((conv_patch_x3<false, 2>), grid, dim3(512), 0, stream, p);
    } else {
        hipLaunchKernelGGL((conv_patch_x3<false, 3>), grid, dim3(512), 0, stream, p);
    }
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

// Upsample + 3 x 3 weights [Cout][3][3][C] -> the four class filters [4][Cout][2][2][C]: class (ph, pw), tap (r, c) = the sum of the 3 x 3
// taps (kh, kw) whose source row is a - 1 + ph + r: ph = 0: r = 0 <- {0}, r = 1 <- {1, 2};  ph = 1: r = 0 <- {0, 1}, r = 1 <- {2}  (columns alike)
__global__ void pack_ups4_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int C) {
    const size_t total = (size_t)4 * Cout * 4 * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t t = i / C;
        const int tap = (int)(t & 3);
        t >>= 2;
        const int co = (int)(t % Cout), cls = (int)(t / Cout);
        const int ph = cls >> 1, pw = cls & 1, r = tap >> 1, cc = tap & 1;
        const int kh0 = ph == 0 ? (r == 0 ? 0 : 1) : (r == 0 ? 0 : 2), kh1 = ph == 0 ? (r == 0 ? 0 : 2) : (r == 0 ? 1 : 2);
        const int kw0 = pw == 0 ? (cc == 0 ? 0 : 1) : (cc == 0 ? 0 : 2), kw1 = pw == 0 ? (cc == 0 ? 0 : 2) : (cc == 0 ? 1 : 2);
        const float* wr = w + (size_t)co * 9 * C + c;
        float acc = 0.f;
        for (int kh = kh0; kh <= kh1; ++kh)
            for (int kw = kw0; kw <= kw1; ++kw) acc += wr[(size_t)(kh * 3 + kw) * C];
        out[i] = acc;
    }
}

// 1 when the four-class form takes the problem (H, W: the OUTPUT map, twice the source's): source map of 16 x 16 patches
int conv_patch_x3_ups4_eligible(int N, int H, int W, int C, int Cout, int ncu) {
    if (H % 32 || W % 32 || C % 32 || Cout % 128) return 0;
    const long tiles = (long)N * 4 * (H / 32) * (W / 32) * (Cout / 128);
    if (ncu <= 0) ncu = 256;
    const long rounds = (tiles + ncu - 1) / ncu;
    if (tiles < 208 || tiles * 100 < rounds * ncu * 85) return 0;
    if ((double)N * H * W * C >= 4294967296.0 || (double)Cout * 16 * C >= 4294967296.0) return 0;
    return 1;
}

extern "C" {
// 1 when v2a_conv2d_fwd_dma_f32 / _d run this 3 x 3 / stride 1 / pad 1 conv (H, W: the conv's map) on the patch kernel (three-plane mode)
int v2a_conv2d_x3p_eligible(int N, int H, int W, int C, int Cout) { return conv_patch_x3_eligible(N, H, W, C, Cout, x3p_ncu()); }

// Upsample (nearest x2, unet.py:105-115) + 3 x 3 conv as four 2 x 2 class convs over the source map (conv_patch_x3<.., 2>): 4 of the 9
// products per output.  v2a_pack_weight_ups4: forward pack [Cout][3][3][C] -> [4][Cout][2][2][C] (sums of the taps that read the same
// source pixel; rounded once more in fp32).  v2a_conv2d_fwd_x3p_ups4: x = SOURCE [N, H/2, W/2, C] -> y [N, H, W, Cout], bias only.
int v2a_conv2d_x3p_ups4_eligible(int N, int H, int W, int C, int Cout) { return conv_patch_x3_ups4_eligible(N, H, W, C, Cout, x3p_ncu()); }
int v2a_pack_weight_ups4(const float* w_packed, float* out, int Cout, int C, hipStream_t stream) {
    if (!w_packed || !out || Cout <= 0 || C <= 0) return V2A_ERR_ARG;
    const size_t total = (size_t)16 * Cout * C;
    int g = (int)((total + 255) / 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(pack_ups4_kernel, dim3(g), dim3(256), 0, stream, w_packed, out, Cout, C);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_conv2d_fwd_x3p_ups4(const float* x, const float* w_ups4, const float* bias, float* y, const void* zeros, int N, int H, int W, int C,
                            int Cout, hipStream_t stream) {
    return conv_patch_x3_launch_gn(x, w_ups4, bias, nullptr, y, zeros, N, H, W, C, Cout, 2, nullptr, nullptr, nullptr, nullptr, 0, 1, 0, stream);
}

// conv_patch_x3 over act(GroupNorm(x)) without materialising the normalised tensor: x [N, H, W, C] fp32 (N images = N / gn_frames GroupNorm
// samples of gn_frames images each), mean / rstd [N / gn_frames][G] (v2a_groupnorm_stats_f32), gamma / beta [C], act 0 none / 1 SiLU.
// 3 x 3 / stride 1 / pad 1, bias, no upsample; only where v2a_conv2d_x3p_eligible(N, H, W, C, Cout) says 1 and (C / G) % 4 == 0.
int v2a_conv2d_fwd_x3p_gn(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int G, int gn_frames,
                          int act, const float* w_packed, const float* bias, float* y, const void* zeros, int N, int H, int W, int C, int Cout,
                          hipStream_t stream) {
    if (!mean) return V2A_ERR_ARG;
    return conv_patch_x3_launch_gn(x, w_packed, bias, nullptr, y, zeros, N, H, W, C, Cout, 0, mean, rstd, gamma, beta, G, gn_frames, act, stream);
}
}
