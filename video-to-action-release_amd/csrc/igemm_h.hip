// Implicit-GEMM convolution over bf16 activations / bf16 weights with fp32 accumulation: the "bf16 storage" configuration of the
// video UNet (the reference's GPU path runs these layers under fp16 autocast: flowdiffusion/flowdiffusion/goal_diffusion.py +
// diffuser/libero/lb_online_trainer_v7.py:889 `accelerator.autocast()`; guided_diffusion/guided_diffusion/nn.py:53-87 Conv3d).
//
//   y[m][n] = sum_k A[m][k] * W[n][k] + bias[n] + rowvec[m / rows_per_batch][n] + residual[m][n]
//   A[m][k] = x[img, oh*s - p + kh, ow*s - p + kw, ci]   (k = (kh, kw, ci); optional second source = channel concat; optional
//             nearest x2 upsample folded into the gather), x channels-last bf16, W = [Cout][KH][KW][Cin] bf16.
//
// Data movement (MI355X_MICROARCH.md / cdna_hip_programming.md section 5):
//   * 128 x 128 output tile per 256-thread workgroup, k tile = 64 bf16 = one full 128-B line per tile row, which by the layer
//     shapes (Cin % 64 == 0) always lies inside ONE filter tap -> every row segment is one contiguous 128-B global read;
//   * both operands go HBM/L2 -> LDS by LDS-DMA (`global_load_lds_dwordx4`, 16 B per lane, no staging VGPRs, no ds_write pass);
//     the LDS image of a DMA is lane-linear, so the bank swizzle is applied on the SOURCE side: LDS slot (row r, position p)
//     receives the row's 16-B chunk p ^ ((r >> 1) & 7).  The ds_read_b128 operand fetches of the 32x32x16 MFMA are then
//     conflict-free in every one of the instruction's four 16-lane groups;
//   * out-of-image taps (zero padding) and rows past M read a 128-B zero line in global memory instead of branching;
//   * two LDS buffers (64 KB -> two workgroups per CU): the DMA of tile t+1 is in flight while tile t is multiplied; one
//     barrier per k tile;
//   * epilogue in registers: bias, per-(batch, channel) embedding vector, residual, then bf16 (or fp32) stores; split-K writes
//     fp32 slabs and a second kernel finishes (only the small 8x8 / 16x16 levels need it).
#include "common.h"
#include "x3t.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
struct f16s { uint16_t v; };      // storage tag of the fp16 instances (tensors are raw 16-bit words either way)
thread_local int g_v2a_half_f16 = 0;   // 16-bit format of the calling host thread's `_h` launches: 0 bf16, 1 fp16 (v2a_set_half_format)
template <typename T> struct is_f16s { static constexpr bool value = false; };
template <> struct is_f16s<f16s> { static constexpr bool value = true; };
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct FastDivH {
    uint32_t d, m, s;
};
static inline FastDivH make_fastdiv_h(uint32_t d) {
    FastDivH f;
    f.d = d;
    if (d <= 1) { f.m = 0; f.s = 0; return f; }
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;
    f.s = s;
    f.m = (uint32_t)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
    return f;
}
__device__ __forceinline__ uint32_t fdivh(uint32_t n, const FastDivH& f) {
    if (f.d <= 1) return n;
    const uint32_t t = __umulhi(f.m, n);
    return (t + ((n - t) >> 1)) >> (f.s - 1);
}

struct ConvDescH {
    // T = storage type of the kernel instance: bf16 (uint16_t) or fp32 (float, exact-f32 MFMA: the parity configuration's kernels)
    const void* x;            // source 1: [N, H, W, C1] T
    const void* x2;           // source 2 (concat along C) or null
    const void* w;            // [Cout][KH*KW*(C1+C2)] T
    const float* bias;        // [Cout] or null
    const float* rowvec;      // [M / rows_per_batch][Cout] or null
    const void* residual;     // [M][Cout] T or null
    const float* residual_f;  // bf16 instances: [M][Cout] fp32 residual (alternative to `residual`) or null
    void* y;                  // [M][Cout] T (null when yf is used)
    float* yf;                // bf16 instances: [M][Cout] fp32 output instead of bf16
    float* partial;           // split-K slabs [splitk][M][Cout]
    const void* zeros;        // >= 128 B of zeros
    float* stats;             // optional [ceil(M/64)][2][Cout]: per 64-row block sum / sum of squares of the bf16 outputs (GroupNorm)
    int N, H, W, C1, C2, OH, OW, Cout, KH, KW, sh, sw, ph, pw, ups, HL, WL, M, K;
    int rows_per_batch, splitk, ktiles_per_split;
    int idil;                 // input dilation 1 | 2 (strided data gradient / transposed conv): logical input = zero-interleaved x
    int frame_tiles;          // > 0: frame-interleaved tile order for (3 x 1) convs, = tiles per frame (OW / BM); 0: row order
    FastDivH fd_ow, fd_oh;
    int split_xcd;            // > 0: 1-D grid, split-K slices pinned to XCDs (slices per XCD)              [conv_igemm_f32x3 only]
    size_t xps, x2ps, wps;    // conv_p3 (pre-split operands): elements between the hi / mid / lo bf16 planes of x, x2, w
    int xp1;                  // element pitch between consecutive pixels of source 1 (= C1; < C1: overlapping channel windows,
                              // v2a_conv2d_fwd_window_f32)                                      [conv_igemm_f32x3, non-GEN path only]
    // Parity classes of a zero-interleaved input (idil = 2: the data gradient of a stride-2 conv, a transposed conv).  Logical input
    // position (oh - ph + kh, ow - pw + kw) is a stored pixel only when both coordinates are even, i.e. for the taps kh = (ph + oh) mod 2
    // (mod 2), kw likewise: a 3 x 3 filter has 1 / 2 / 2 / 4 live taps for the four (oh & 1, ow & 1) classes -- 2.25 of 9 on average.
    // pcls = 1: tile rows are numbered class-major (class c = rows [c * cls_R, (c + 1) * cls_R), inside a class (image, oh / nch, ow / ncw)),
    // so that a tile's rows share ONE class and its K loop walks the live taps only; the epilogue maps rows back to NHWC order.
    int pcls, nch, ncw, chh, cwh, cls_R;                                                         // [conv_igemm_f32x3<GEN>]
    FastDivH fd_cw, fd_chw;
};

__device__ __forceinline__ int xcd_remap_h(int bid, int nblk) {
    int q = nblk >> 3, r = nblk & 7;
    int xcd = bid & 7, slot = bid >> 3;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

__device__ __forceinline__ uint16_t f2bf(float f) {
    return v2a_f2bf(f);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// STAGES = 2: two LDS buffers per workgroup (64 KB at 128x128 -> 2 workgroups / CU), the DMA of tile t+1 flies under the MFMAs of
// tile t.  STAGES = 1: one buffer (32 KB, <= 128 VGPRs -> 4 workgroups / CU): nothing overlaps inside a workgroup, but twice as
// many tiles are in flight per CU -- these kernels wait ~1.4 us for a k tile against ~0.2 us of MFMA work on it, so bytes in
// flight per CU (not per workgroup) is what buys throughput.
template <int BM, int BN, typename T, int STAGES>
__global__ __launch_bounds__(256, STAGES == 1 ? 4 : 2) void conv_igemm_h(const ConvDescH p) {
    constexpr int ROWB = 128;                       // bytes per tile row (64 bf16 / 32 fp32)
    constexpr int EPT = ROWB / (int)sizeof(T);      // elements per k tile
    constexpr int EPC = 16 / (int)sizeof(T);        // elements per 16-B chunk
    constexpr bool HALF = sizeof(T) == 2;
    constexpr bool F16 = is_f16s<T>::value;
    constexpr int AL = BM / 32, BL = BN / 32;       // DMA pieces per thread per tile (32 rows x 8 chunks per 256-thread pass)
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int BUF = (BM + BN) * ROWB;
    __shared__ __attribute__((aligned(128))) unsigned char smem[STAGES * BUF];      // ONE LDS object (A0 B0 [A1 B1])

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = (p.Cout + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int lin = xcd_remap_h(blockIdx.x, tiles_m * tiles_n);
    int tm = lin / tiles_n;
    const int n0 = (lin % tiles_n) * BN;
    if (p.frame_tiles > 0) {
        // (3 x 1) convs over frames ([B, F, HW, C] view: OH = F, OW = HW): tile (frame f, pixel block pb) reads blocks pb of frames
        // f-1, f, f+1.  In row order the three readers of a block are frame_tiles slots apart -- one XCD streams ~16 MB between
        // them through its 4 MB L2 and the block comes from the fabric three times (PMC: 2.7x the tensor).  Run the F frames of a
        // pixel block back to back instead: slot -> (sample, pixel block, frame).
        const int per_sample = p.OH * p.frame_tiles;
        const int img = tm / per_sample, rem = tm - img * per_sample;
        const int pb = rem / p.OH, f = rem - pb * p.OH;
        tm = img * per_sample + f * p.frame_tiles + pb;
    }
    const int m0 = tm * BM;
    const int split = blockIdx.y;
    const int Cin = p.C1 + p.C2;
    const int nkt = p.K / EPT;
    const int kt_begin = split * p.ktiles_per_split;
    const int kt_end = min(nkt, kt_begin + p.ktiles_per_split);

    // ---- DMA source state.  Pass j of the 256 threads fills rows j*32 .. j*32+31; this thread's slot is (row j*32 + tid/8,
    // position tid%8) and carries the row's chunk (tid%8) ^ swz with swz = ((row >> 1) & 7) = ((tid >> 4) & 7) for every j.
    const int lrow = tid >> 3;
    const int chunk = (tid & 7) ^ ((tid >> 4) & 7);
    int a_ihb[AL], a_iwb[AL], a_img[AL];
#pragma unroll
    for (int j = 0; j < AL; ++j) {
        const int m = m0 + j * 32 + lrow;
        const bool ok = m < p.M;
        const uint32_t mm = ok ? (uint32_t)m : 0u;
        const uint32_t t = fdivh(mm, p.fd_ow);
        const int ow = (int)(mm - t * p.OW);
        const uint32_t img = fdivh(t, p.fd_oh);
        const int oh = (int)(t - img * p.OH);
        a_img[j] = (int)img;
        a_ihb[j] = ok ? oh * p.sh - p.ph : -(1 << 28);       // rows past M fail the bounds test below
        a_iwb[j] = ow * p.sw - p.pw;
    }
    const T* zsrc = reinterpret_cast<const T*>(p.zeros) + chunk * EPC;
    const T* b_src[BL];
    bool b_ok[BL];
#pragma unroll
    for (int j = 0; j < BL; ++j) {
        const int n = n0 + j * 32 + lrow;
        b_ok[j] = n < p.Cout;
        b_src[j] = b_ok[j] ? reinterpret_cast<const T*>(p.w) + (size_t)n * p.K + chunk * EPC : zsrc;
    }

    // running (tap, channel) position of the next k tile to issue
    int ik0 = kt_begin * EPT;
    int itap = ik0 / Cin;
    int ic0 = ik0 - itap * Cin;
    int ikh = itap / p.KW, ikw = itap - ikh * p.KW;

    auto issue = [&](int buf) {
        unsigned char* abase = smem + buf * BUF;
        unsigned char* bbase = abase + BM * ROWB;
        const bool first = ic0 < p.C1;
        const T* src = reinterpret_cast<const T*>(first ? p.x : p.x2);
        const uint32_t Cs = (uint32_t)(first ? p.C1 : p.C2);
        const uint32_t cc = (uint32_t)((first ? ic0 : ic0 - p.C1) + chunk * EPC);
#pragma unroll
        for (int j = 0; j < AL; ++j) {
            int ih = a_ihb[j] + ikh, iw = a_iwb[j] + ikw;
            bool ok = (unsigned)ih < (unsigned)p.HL && (unsigned)iw < (unsigned)p.WL;
            if (p.ups) { ih >>= 1; iw >>= 1; }
            if (p.idil == 2) { ok = ok && !((ih | iw) & 1); ih >>= 1; iw >>= 1; }
            // offset computed unconditionally in 32 bits (the host checks the tensor has < 2^32 elements); select, do not branch
            const uint32_t off = ((uint32_t)(a_img[j] * p.H + ih) * (uint32_t)p.W + (uint32_t)iw) * Cs + cc;
            const T* g = ok ? src + off : zsrc;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(abase + (j * 256 + wid * 64) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < BL; ++j) {
            const T* g = b_src[j] + (b_ok[j] ? ik0 : 0);
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(bbase + (j * 256 + wid * 64) * 16), 16, 0, 0);
        }
        ik0 += EPT;
        ic0 += EPT;
        if (ic0 >= Cin) {
            ic0 = 0;
            if (++ikw == p.KW) { ikw = 0; ++ikh; }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm = (wid >> 1) * WM, wn = (wid & 1) * WN;
    const int lr = lane & 31, lk = lane >> 5;
    const int rswz = (lr >> 1) & 7;                 // (row >> 1) & 7 of every operand row this lane reads (wm, i*32 are multiples of 16)
    int a_off[TM], b_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_off[i] = (wm + i * 32 + lr) * ROWB;
#pragma unroll
    for (int j = 0; j < TN; ++j) b_off[j] = BM * ROWB + (wn + j * 32 + lr) * ROWB;

    auto compute = [&](const unsigned char* base) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int pos = (((h << 1) | lk) ^ rswz) << 4;
        if constexpr (HALF) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(base + a_off[i] + pos);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8*>(base + b_off[j] + pos);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = v2a_mfma_h<F16>(a[i], b[j], acc[i][j]);
        } else {
            // one b128 fetch per operand row feeds FOUR exact-f32 MFMA k-steps: lanes with lk = 0 supply k = 8h + e, the others
            // k = 8h + 4 + e, for A and B alike (the order of an exact fp32 sum is free)
            f32x4_t a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4_t*>(base + a_off[i] + pos);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4_t*>(base + b_off[j] + pos);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
    }
    };
    if constexpr (STAGES == 2) {
        if (kt_begin < kt_end) issue(0);
        int buf = 0;
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                        // tile kt landed everywhere; everyone is done reading the other buffer
            if (kt + 1 < kt_end) issue(buf ^ 1);
            compute(smem + buf * BUF);
            buf ^= 1;
        }
    } else {
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            issue(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                        // tile kt landed everywhere
            compute(smem);
            __syncthreads();                        // everyone is done reading before the next tile overwrites the buffer
        }
    }

    // ---- epilogue.  C layout of the 32x32 MFMA: lane -> column lr, rows (r & 3) + 8 * (r >> 2) + 4 * lk.  Each wave parks its
    // 64x64 fp32 sub-tile in its own 16 KB + pad of the (now free) LDS, then every lane owns 8 consecutive channels of a row:
    // bias / embedding vector / residual are applied on 8-wide vectors and the result leaves as one 16-B bf16 store (a full
    // 128-B line per row and wave) instead of 2-byte scatters.
    __syncthreads();
    constexpr int LDC = WN;                                       // floats per parked row; columns XOR-ed with 4 * (row & 1) so that
    static_assert(4 * 32 * LDC * 4 <= STAGES * BUF, "epilogue staging exceeds the operand buffers");   // the b128 read-back is conflict-free
    float* cw = reinterpret_cast<float*>(smem) + wid * 32 * LDC;  // one 32-row sub-tile of this wave at a time (wave-private region)
    constexpr int V = WN / 8;                                     // 8-channel vectors per row
    const int vrow = lane / V, vcol = (lane % V) * 8;
    const int n = n0 + wn + vcol;
    const bool vec_ok = (p.Cout % 8 == 0) && (n + 8 <= p.Cout);
    float bv[8], ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bv[e] = (p.bias && p.splitk == 1 && n + e < p.Cout) ? p.bias[n + e] : 0.f;
        ssum[e] = 0.f;
        ssq[e] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
                cw[row * LDC + ((j * 32 + lr) ^ ((row & 1) << 2))] = acc[i][j][r];
            }
        // LDS operations of one wave complete in order: only this wave's writes must have landed before its reads (and its reads
        // of sub-tile i are issued before the writes of sub-tile i + 1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int rr = 0; rr < 32; rr += 64 / V) {
            const int ml = rr + vrow;                                  // row inside the parked 32-row sub-tile
            const int m = m0 + wm + i * 32 + ml;
            if (m >= p.M || n >= p.Cout) continue;
            const int sx = (ml & 1) << 2;
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(&cw[ml * LDC + (vcol ^ sx)]);
            const f32x4 c1 = *reinterpret_cast<const f32x4*>(&cw[ml * LDC + ((vcol + 4) ^ sx)]);
            float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
            const size_t o = (size_t)m * p.Cout + n;
            if (p.splitk > 1) {
                float* dst = p.partial + (size_t)split * p.M * p.Cout + o;
                if (vec_ok) {
                    *reinterpret_cast<f32x4*>(dst) = c0;
                    *reinterpret_cast<f32x4*>(dst + 4) = c1;
                } else {
                    for (int e = 0; e < 8 && n + e < p.Cout; ++e) dst[e] = v[e];
                }
                continue;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bv[e];
            if (vec_ok) {
                if (p.rowvec) {
                    const float* rv = p.rowvec + (size_t)(m / p.rows_per_batch) * p.Cout + n;
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>(rv), r1 = *reinterpret_cast<const f32x4*>(rv + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[e + 4] += r1[e]; }
                }
                if (p.residual_f) {
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>(p.residual_f + o), r1 = *reinterpret_cast<const f32x4*>(p.residual_f + o + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[e + 4] += r1[e]; }
                }
                if (p.residual) {
                    if constexpr (HALF) {
                        const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(p.residual) + o);
                        float rr[8];
                        v2a_unpack_h8<F16>(u, rr);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += rr[e];
                    } else {
                        const float* rp = reinterpret_cast<const float*>(p.residual) + o;
                        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[e + 4] += r1[e]; }
                    }
                }
                if (!HALF || p.yf) {
                    float* yo = HALF ? p.yf : reinterpret_cast<float*>(p.y);
                    f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                    *reinterpret_cast<f32x4*>(yo + o) = o0;
                    *reinterpret_cast<f32x4*>(yo + o + 4) = o1;
                    if (!HALF) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { ssum[e] += v[e]; ssq[e] += v[e] * v[e]; }
                    }
                } else {
                    uint16_t h[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        h[e] = v2a_f2h<F16>(v[e]);
                        const float r = v2a_h2f<F16>(h[e]);           // statistics of what the consumer will read
                        ssum[e] += r;
                        ssq[e] += r * r;
                    }
                    uint4 u;
                    u.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
                    u.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
                    u.z = (uint32_t)h[4] | ((uint32_t)h[5] << 16);
                    u.w = (uint32_t)h[6] | ((uint32_t)h[7] << 16);
                    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.y) + o) = u;
                }
            } else {                                                  // ragged channel count (e.g. the 3-channel output head)
                for (int e = 0; e < 8 && n + e < p.Cout; ++e) {
                    float t = v[e];
                    if (p.rowvec) t += p.rowvec[(size_t)(m / p.rows_per_batch) * p.Cout + n + e];
                    if (p.residual) t += HALF ? v2a_h2f<F16>(reinterpret_cast<const uint16_t*>(p.residual)[o + e]) : reinterpret_cast<const float*>(p.residual)[o + e];
                    if (p.residual_f) t += p.residual_f[o + e];
                    if (!HALF) reinterpret_cast<float*>(p.y)[o + e] = t;
                    else if (p.yf) p.yf[o + e] = t;
                    else reinterpret_cast<uint16_t*>(p.y)[o + e] = v2a_f2h<F16>(t);
                }
            }
        }
    }
    if (p.stats && p.splitk == 1 && vec_ok) {
        // this wave's 64 rows x 64 channels: lanes with equal lane % V hold the same 8 channels for different rows
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o = V; o < 64; o <<= 1) {
                ssum[e] += __shfl_xor(ssum[e], o, 64);
                ssq[e] += __shfl_xor(ssq[e], o, 64);
            }
        }
        const int blk = (m0 + wm) >> 6;                                // 64-row block index of this wave's rows
        if (lane < V && (m0 + wm) < p.M) {
            float* dst = p.stats + (size_t)blk * 2 * p.Cout + n;
            f32x4 a0 = {ssum[0], ssum[1], ssum[2], ssum[3]}, a1 = {ssum[4], ssum[5], ssum[6], ssum[7]};
            f32x4 q0 = {ssq[0], ssq[1], ssq[2], ssq[3]}, q1 = {ssq[4], ssq[5], ssq[6], ssq[7]};
            *reinterpret_cast<f32x4*>(dst) = a0;
            *reinterpret_cast<f32x4*>(dst + 4) = a1;
            *reinterpret_cast<f32x4*>(dst + p.Cout) = q0;
            *reinterpret_cast<f32x4*>(dst + p.Cout + 4) = q1;
        }
    }
}

// Epilogue of the fp32 instances (conv_igemm_f32p, conv_igemm_f32x3): each wave parks one 32-row sub-tile at a time in its own LDS
// region (4 * 32 * BN/2 floats in all; the caller has drained every LDS user), then every lane owns 8 consecutive channels of a row --
// split-K slab, or bias / row vector / residual and two 16-B stores -- and the per-64-row-block GroupNorm statistics.
// (tile row -> output row: consecutive rows, or the 8 x 16 pixel patch of conv_halo_x3<0>)
struct RowsLinearH {
    int m0;
    __device__ __forceinline__ int operator()(int r) const { return m0 + r; }
};
// A `ds_read_b128` of a wave is served in four groups of 16 lanes, in the lower 32 lanes {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31}
// (MI355X_MICROARCH.md, LDS).  MFMA row m of a 32-row sub-tile of a 16-pixel-wide patch therefore carries pixel patch16_perm_h(m) of the
// sub-tile's 2 x 16 pixels: each hardware group then reads ONE patch row = 16 consecutive halo slots, which the (slot >> 2) & 3 piece
// swizzle spreads over all 64 banks under every tap shift (with m -> pixel m the groups straddle two patch rows, 18 slots apart: 2-way
// conflicts on every operand read).  csrc/igemm_h3.hip lds_group_perm3 is the same map.
__device__ __forceinline__ int patch16_perm_h(int m) {
    const int qd = m >> 2;
    return ((__builtin_popcount(qd) & 1) << 4) | ((qd >> 1) << 2) | (m & 3);
}
struct RowsPatch16H {
    int base, W;                // output row of the patch's pixel (0, 0); map width
    __device__ __forceinline__ int operator()(int r) const {
        const int q = (r & ~31) | patch16_perm_h(r & 31);
        return base + (q >> 4) * W + (q & 15);
    }
};
// (class-major tile rows of a zero-interleaved input, ConvDescH::pcls: row m' = cls * R + (img, a, b) -> NHWC row of pixel
// (a * nch + cph, b * ncw + cpw))
__device__ __forceinline__ int parity_row_h(const ConvDescH& p, int mp, int cls, int cph, int cpw, int& oh, int& ow, int& img) {
    const uint32_t r = (uint32_t)(mp - cls * p.cls_R);
    const uint32_t im = fdivh(r, p.fd_chw);
    const uint32_t rem = r - im * (uint32_t)(p.chh * p.cwh);
    const uint32_t a = fdivh(rem, p.fd_cw);
    const uint32_t b = rem - a * (uint32_t)p.cwh;
    img = (int)im;
    if (p.pcls == 2) {          // position classes: chh = OH, cwh = 1 -> a = oh, b = 0; the class IS the output column
        oh = (int)a;
        ow = cpw;
    } else {
        oh = (int)a * p.nch + cph;
        ow = (int)b * p.ncw + cpw;
    }
    return (img * p.OH + oh) * p.OW + ow;
}
struct RowsParityH {
    const ConvDescH* p;
    int m0, cls, cph, cpw;
    __device__ __forceinline__ int operator()(int r) const {
        const int mp = m0 + r;
        if (mp >= p->M) return p->M;
        int oh, ow, img;
        return parity_row_h(*p, mp, cls, cph, cpw, oh, ow, img);
    }
};
template <int BM, int BN, int WVM = 2, int WVN = 2, typename RM = RowsLinearH>
__device__ __forceinline__ void conv_f32_epilogue_rm(const ConvDescH& p, f32x16 (&acc)[BM / WVM / 32][BN / WVN / 32], unsigned char* smem,
                                                     const RM rowmap, const int n0, const int split, const float* bias_sel) {
    constexpr int WM = BM / WVM, WN = BN / WVN, TM = WM / 32, TN = WN / 32;      // WVM x WVN waves, each a (TM x 32) x (TN x 32) sub-tile
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = (wid / WVN) * WM, wn = (wid % WVN) * WN;
    const int lr = lane & 31, lk = lane >> 5;
    constexpr int LDC = WN;
    float* cw = reinterpret_cast<float*>(smem) + wid * 32 * LDC;
    constexpr int V = WN / 8;
    const int vrow = lane / V, vcol = (lane % V) * 8;
    const int n = n0 + wn + vcol;
    const bool vec_ok = (p.Cout % 8 == 0) && (n + 8 <= p.Cout);
    float bv[8], ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bv[e] = (bias_sel && p.splitk == 1 && n + e < p.Cout) ? bias_sel[n + e] : 0.f;
        ssum[e] = 0.f;
        ssq[e] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
                cw[row * LDC + ((j * 32 + lr) ^ ((row & 1) << 2))] = acc[i][j][r];
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int rr = 0; rr < 32; rr += 64 / V) {
            const int ml = rr + vrow;
            const int m = rowmap(wm + i * 32 + ml);
            if (m >= p.M || n >= p.Cout) continue;
            const int sx = (ml & 1) << 2;
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(&cw[ml * LDC + (vcol ^ sx)]);
            const f32x4 c1 = *reinterpret_cast<const f32x4*>(&cw[ml * LDC + ((vcol + 4) ^ sx)]);
            float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
            const size_t o = (size_t)m * p.Cout + n;
            if (p.splitk > 1) {
                float* dst = p.partial + (size_t)split * p.M * p.Cout + o;
                if (vec_ok) {
                    *reinterpret_cast<f32x4*>(dst) = c0;
                    *reinterpret_cast<f32x4*>(dst + 4) = c1;
                } else {
                    for (int e = 0; e < 8 && n + e < p.Cout; ++e) dst[e] = v[e];
                }
                continue;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bv[e];
            float* yo = reinterpret_cast<float*>(p.y);
            if (vec_ok) {
                if (p.rowvec) {
                    const float* rv = p.rowvec + (size_t)(m / p.rows_per_batch) * p.Cout + n;
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>(rv), r1 = *reinterpret_cast<const f32x4*>(rv + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[e + 4] += r1[e]; }
                }
                if (p.residual) {
                    const float* rp = reinterpret_cast<const float*>(p.residual) + o;
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[e + 4] += r1[e]; }
                }
                f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                *reinterpret_cast<f32x4*>(yo + o) = o0;
                *reinterpret_cast<f32x4*>(yo + o + 4) = o1;
#pragma unroll
                for (int e = 0; e < 8; ++e) { ssum[e] += v[e]; ssq[e] += v[e] * v[e]; }
            } else {
                for (int e = 0; e < 8 && n + e < p.Cout; ++e) {
                    float t = v[e];
                    if (p.rowvec) t += p.rowvec[(size_t)(m / p.rows_per_batch) * p.Cout + n + e];
                    if (p.residual) t += reinterpret_cast<const float*>(p.residual)[o + e];
                    yo[o + e] = t;
                }
            }
        }
    }
    if (p.stats && p.splitk == 1 && vec_ok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o = V; o < 64; o <<= 1) {
                ssum[e] += __shfl_xor(ssum[e], o, 64);
                ssq[e] += __shfl_xor(ssq[e], o, 64);
            }
        }
        const int blk = rowmap(wm) >> 6;
        if (lane < V && rowmap(wm) < p.M) {
            float* dst = p.stats + (size_t)blk * 2 * p.Cout + n;
            f32x4 a0 = {ssum[0], ssum[1], ssum[2], ssum[3]}, a1 = {ssum[4], ssum[5], ssum[6], ssum[7]};
            f32x4 q0 = {ssq[0], ssq[1], ssq[2], ssq[3]}, q1 = {ssq[4], ssq[5], ssq[6], ssq[7]};
            *reinterpret_cast<f32x4*>(dst) = a0;
            *reinterpret_cast<f32x4*>(dst + 4) = a1;
            *reinterpret_cast<f32x4*>(dst + p.Cout) = q0;
            *reinterpret_cast<f32x4*>(dst + p.Cout + 4) = q1;
        }
    }
}

template <int BM, int BN, int WVM = 2, int WVN = 2>
__device__ __forceinline__ void conv_f32_epilogue(const ConvDescH& p, f32x16 (&acc)[BM / WVM / 32][BN / WVN / 32], unsigned char* smem,
                                                  const int m0, const int n0, const int split, const float* bias_sel) {
    conv_f32_epilogue_rm<BM, BN, WVM, WVN, RowsLinearH>(p, acc, smem, RowsLinearH{m0}, n0, split, bias_sel);
}

template <int N> __device__ __forceinline__ void wait_vmcnt_h() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- Pipelined exact-f32 instance (round 4).  In-kernel stamps of conv_igemm_h<.., float> (tools/probes/conv_stamp_probe.py) showed its
// launches spend 1.3 us before the first DMA, 2.4 us in the epilogue -- and 45 us in a main loop whose MFMA work is 30.7 us: every k tile
// (32 floats) pays `s_waitcnt vmcnt(0)` + `__syncthreads()` + a ~150-instruction DMA-issue block (address arithmetic, exec-masked selects,
// scalar branches on p.ups / p.idil) during which the wave issues no MFMA, and co-resident workgroups fall into the same phase because they
// alternate on the SIMD's matrix pipe.  v_mfma_f32_32x32x2_f32 occupies the pipe for 64 cycles per instruction, i.e. ~16 issue slots of
// anything else are free per MFMA -- so here
//   * the DMA of k tile t + S - 1 is issued IN PIECES between the MFMA groups of tile t (branch-free address arithmetic: the nearest-x2 /
//     zero-interleave cases are a shift and a parity mask, the out-of-image select is a v_cndmask on the finished pointer),
//   * S LDS stages with a COUNTED `s_waitcnt vmcnt((S-2) * pieces)` + one raw s_barrier per k tile (never drained: past the end of the
//     slice the pieces read the zero line into a dead stage, so the count is a compile-time constant),
//   * operand fragments of k-step h + 1 are read while the MFMAs of step h issue (register double buffer).
// Same tiles, same LDS image (source-side swizzle), same k order per accumulator as conv_igemm_h<.., float>: results are bit-identical.
// GEN = false: plain gather (no nearest-x2 upsample, no zero-interleaved input) -- a piece's offset is a per-thread constant plus a
// per-tile SCALAR (no multiply in the loop); GEN = true: the general form.
template <int BM, int BN, int S, int MINW, bool GEN>
__global__ __launch_bounds__(256, MINW) void conv_igemm_f32p(const ConvDescH p) {
    typedef float T;
    constexpr int ROWB = 128, EPT = 32, EPC = 4;
    constexpr int AL = BM / 32, BL = BN / 32, NL = AL + BL;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int STAGE = (BM + BN) * ROWB;
    static_assert((S - 2) * NL <= 63 && S >= 2, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(128))) unsigned char smem[S * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = (p.Cout + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int lin = xcd_remap_h(blockIdx.x, tiles_m * tiles_n);
    int tm = lin / tiles_n;
    const int n0 = (lin % tiles_n) * BN;
    if (p.frame_tiles > 0) {
        const int per_sample = p.OH * p.frame_tiles;
        const int img = tm / per_sample, rem = tm - img * per_sample;
        const int pb = rem / p.OH, f = rem - pb * p.OH;
        tm = img * per_sample + f * p.frame_tiles + pb;
    }
    const int m0 = tm * BM;
    const int split = blockIdx.y;
    const int Cin = p.C1 + p.C2;
    const int nkt = p.K / EPT;
    const int kt_begin = split * p.ktiles_per_split;
    const int kt_end = min(nkt, kt_begin + p.ktiles_per_split);
    const T* wsel = reinterpret_cast<const T*>(p.w);
    const float* bias_sel = p.bias;

    const int lrow = tid >> 3;
    const int chunk = (tid & 7) ^ ((tid >> 4) & 7);
    int a_ihb[AL], a_iwb[AL], a_imgh[AL];
#pragma unroll
    for (int j = 0; j < AL; ++j) {
        const int m = m0 + j * 32 + lrow;
        const bool ok = m < p.M;
        const uint32_t mm = ok ? (uint32_t)m : 0u;
        const uint32_t t = fdivh(mm, p.fd_ow);
        const int ow = (int)(mm - t * p.OW);
        const uint32_t img = fdivh(t, p.fd_oh);
        const int oh = (int)(t - img * p.OH);
        a_imgh[j] = (int)img * p.H;
        a_ihb[j] = ok ? oh * p.sh - p.ph : -(1 << 28);       // rows past M fail the bounds test below
        a_iwb[j] = ow * p.sw - p.pw;
    }
    int a_lin1[AL], a_lin2[AL];                             // GEN = false: element offset of (row, tap (0,0), channel chunk) in source 1 / 2
#pragma unroll
    for (int j = 0; j < AL; ++j) {
        const int pix = (a_imgh[j] + a_ihb[j]) * p.W + a_iwb[j];
        a_lin1[j] = pix * p.C1 + chunk * EPC;
        a_lin2[j] = pix * p.C2 + chunk * EPC;
    }
    const T* zsrc = reinterpret_cast<const T*>(p.zeros) + chunk * EPC;
    const T* b_src[BL];
    bool b_ok[BL];
#pragma unroll
    for (int j = 0; j < BL; ++j) {
        const int n = n0 + j * 32 + lrow;
        b_ok[j] = n < p.Cout;
        b_src[j] = wsel + (size_t)(b_ok[j] ? n : 0) * p.K + chunk * EPC;
    }
    const int shift = (p.ups || p.idil == 2) ? 1 : 0;       // logical -> stored pixel: nearest x2 upsample / zero-interleaved input
    const int pmask = (p.idil == 2) ? 1 : 0;                // zero-interleaved input: odd logical positions are zeros

    // (tap, channel) position of the next k tile to issue, and whether it still belongs to this slice
    int ik0 = kt_begin * EPT;
    int itap = ik0 / Cin;
    int ic0 = ik0 - itap * Cin;
    int ikh = itap / p.KW, ikw = itap - ikh * p.KW;
    int it = kt_begin;

    // pieces [q0, q1) of the pending k tile go to stage `buf`; a piece past the slice's end reads the zero line (keeps vmcnt counts fixed)
    auto issue = [&](int buf, int q0, int q1) {
        unsigned char* abase = smem + buf * STAGE;
        unsigned char* bbase = abase + BM * ROWB;
        const bool live = it < kt_end;
        const bool first = ic0 < p.C1;
        const T* src = reinterpret_cast<const T*>(first ? p.x : p.x2);
        const uint32_t Cs = (uint32_t)(first ? p.C1 : p.C2);
        const uint32_t cc = (uint32_t)((first ? ic0 : ic0 - p.C1) + chunk * EPC);
        const int s_tap = (ikh * p.W + ikw) * (int)Cs + (first ? ic0 : ic0 - p.C1);      // scalar part of a GEN = false offset
#pragma unroll
        for (int q = q0; q < q1; ++q) {
            if (q < AL) {
                const int j = q;
                int ih = a_ihb[j] + ikh, iw = a_iwb[j] + ikw;
                bool ok;
                uint32_t off;
                if constexpr (GEN) {
                    ok = live & ((unsigned)ih < (unsigned)p.HL) & ((unsigned)iw < (unsigned)p.WL) & (((ih | iw) & pmask) == 0);
                    ih >>= shift; iw >>= shift;
                    off = ((uint32_t)(a_imgh[j] + ih) * (uint32_t)p.W + (uint32_t)iw) * Cs + cc;
                } else {
                    ok = live & ((unsigned)ih < (unsigned)p.H) & ((unsigned)iw < (unsigned)p.W);
                    off = (uint32_t)((first ? a_lin1[j] : a_lin2[j]) + s_tap);
                }
                asm volatile("" : "+v"(off));               // keep the offset arithmetic out of an exec-masked region (plain select below)
                const T* g = src + off;
                g = ok ? g : zsrc;
                __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(abase + (j * 256 + wid * 64) * 16), 16, 0, 0);
            } else {
                const int j = q - AL;
                uint64_t gi = (uint64_t)(b_src[j] + ik0);
                asm volatile("" : "+v"(gi));                // (a branch-free select: no scalar branch inside the MFMA stream)
                const T* g = (live & b_ok[j]) ? reinterpret_cast<const T*>(gi) : zsrc;
                __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(bbase + (j * 256 + wid * 64) * 16), 16, 0, 0);
            }
        }
    };
    auto advance = [&]() {
        ++it;
        ik0 += EPT;
        ic0 += EPT;
        const bool wrap = ic0 >= Cin;
        ic0 = wrap ? 0 : ic0;
        const int kw1 = ikw + (wrap ? 1 : 0);
        const bool wrap2 = kw1 == p.KW;
        ikw = wrap2 ? 0 : kw1;
        ikh += wrap2 ? 1 : 0;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm = (wid >> 1) * WM, wn = (wid & 1) * WN;
    const int lr = lane & 31, lk = lane >> 5;
    const int rswz = (lr >> 1) & 7;
    int a_off[TM], b_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_off[i] = (wm + i * 32 + lr) * ROWB;
#pragma unroll
    for (int j = 0; j < TN; ++j) b_off[j] = BM * ROWB + (wn + j * 32 + lr) * ROWB;
    int pos[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) pos[h] = (((h << 1) | lk) ^ rswz) << 4;

    // ---- prologue: S - 1 k tiles in flight
#pragma unroll
    for (int s = 0; s < S - 1; ++s) {
        issue(s, 0, NL);
        advance();
    }
    int cbuf = 0, ibuf = S - 1;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        wait_vmcnt_h<(S - 2) * NL>();                       // this wave's pieces of tile kt have landed; younger tiles stay in flight
        __builtin_amdgcn_s_barrier();                       // ... everyone's have, and everyone is done reading the stage issued into next
        const unsigned char* base = smem + cbuf * STAGE;
        f32x4_t a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = *reinterpret_cast<const f32x4_t*>(base + a_off[i] + pos[0]);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = *reinterpret_cast<const f32x4_t*>(base + b_off[j] + pos[0]);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int c = h & 1, nx = c ^ 1;
            if (h < 3) {                                     // fragments of k-step h + 1: requested before this step's MFMAs issue
#pragma unroll
                for (int i = 0; i < TM; ++i) a[nx][i] = *reinterpret_cast<const f32x4_t*>(base + a_off[i] + pos[h + 1]);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[nx][j] = *reinterpret_cast<const f32x4_t*>(base + b_off[j] + pos[h + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            issue(ibuf, NL * h / 4, NL * (h + 1) / 4);       // its address arithmetic and DMA instructions go between the MFMAs below
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i][e], b[c][j][e], acc[i][j], 0, 0, 0);
        }
        advance();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's operand reads are done before it reaches the next barrier
        cbuf = (cbuf + 1 == S) ? 0 : cbuf + 1;
        ibuf = (ibuf + 1 == S) ? 0 : ibuf + 1;
    }

    // ---- epilogue: park each wave's sub-tiles in LDS, leave as 2 x 16-B stores per lane
    wait_vmcnt_h<0>();                                       // the zero-line pieces past the end still land in the stages
    __syncthreads();
    static_assert(4 * 32 * WN * 4 <= S * STAGE, "epilogue staging exceeds the operand buffers");
    conv_f32_epilogue<BM, BN>(p, acc, smem, m0, n0, split, bias_sel);
}

// ---- fp32 convolution by three bf16 planes (round 4).  v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 matrix rate, so an exact-f32
// conv is bound by 64 issue cycles per 4096 FLOP whatever the schedule (conv_igemm_f32p reaches the loop's floor at the clock the chip
// sustains, tools/probes/conv_stamp_probe.py).  Here every fp32 operand value x is split while it is staged into LDS:
//     hi = bf16(x),  mid = bf16(x - hi),  lo = bf16(x - hi - mid)        (both differences are exact in fp32: 24 significant bits)
// and a product block is the sum of the six plane products of weight >= 2^-16 -- lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi, smallest
// first, each bf16 x bf16 product exact in the MFMA's fp32 accumulation; the three dropped products are <= 2^-24 |a||b|, one fp32 rounding.
// Six v_mfma_f32_32x32x16_bf16 = 192 cycles per 32x32x16 block against 512 for eight v_mfma_f32_32x32x2_f32.  Measured error against an
// fp64 reference on the same inputs: rms 3.6e-7 .. 7.3e-7 of the output's rms -- at or below the exact-f32 kernel's 4.2e-7 .. 8.5e-7
// (tools/probes/emu_probe.py); NOT bit-equal to an fmaf chain, so `V2A_F32_CONV=exact` keeps the exact kernels selectable.
// Structure: register-staged (the split needs the values in VGPRs): 16-B global loads two k tiles ahead (two register sets), split +
// ds_write_b64 of tile t + 1 between the MFMA groups of tile t, two LDS stages of 3 x (BM + BN) rows x 64 B with the 16-B chunk swizzle
// p ^ ((row >> 2) & 3) (conflict-free ds_read_b128 operand fetch), one barrier per k tile.
typedef __attribute__((address_space(1))) f32x4 gf32x4_t;
__device__ __forceinline__ void split3_pair_h(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) { v2a_split3x2(x0, x1, h, m, l); }

template <int BM, int BN, int WVM, int WVN, int MINW, bool GEN>
__global__ __launch_bounds__(64 * WVM * WVN, MINW) void conv_igemm_f32x3(const ConvDescH p) {
    constexpr int PF = 2;      // register sets of k tiles in flight (four were measured: no gain -- these launches are not load-latency-bound, tools/probes/r5)
    constexpr int EPT = 32;
    constexpr int NT = 64 * WVM * WVN, RP = NT / 8;             // threads; tile rows one loader pass covers (8 float4 per 32-float row)
    constexpr int AL = BM / RP, BL = BN / RP;
    constexpr int WM = BM / WVM, WN = BN / WVN, TM = WM / 32, TN = WN / 32;
    static_assert(AL >= 1 && BL >= 1 && TM >= 1 && TN >= 1, "tile shape");
    constexpr int ROWH = 64;                                    // bytes per plane row: 32 bf16
    constexpr int PA = BM * ROWH, PB = BN * ROWH, STG = 3 * (PA + PB);
    constexpr int S = 2, STAGE = STG;                           // (names the shared epilogue's size check uses)
    __shared__ __attribute__((aligned(128))) unsigned char smem[2 * STG];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = (p.Cout + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    int lin, split;
    if (p.split_xcd > 0) {
        // split-K slices pinned to XCDs (1-D grid of tiles x splits, splits = 8 x split_xcd): workgroup L runs on XCD L & 7, which owns
        // slices (L & 7) * split_xcd .. + split_xcd - 1 and ALL their output tiles -- its L2 then holds one K slice of both operands
        // (a few MB) and every byte is fetched over the fabric once; with slice = blockIdx.y each XCD streamed the WHOLE weight matrix
        // for its share of the row tiles (the deep small-M GEMMs of the ConditionalUnet1D ran at the fabric's ~8 TB/s, not the L2's)
        const int L = (int)blockIdx.x, xcd = L & 7, slot = L >> 3, tiles = tiles_m * tiles_n;
        const int sl = slot / tiles;
        split = xcd * p.split_xcd + sl;
        lin = slot - sl * tiles;
    } else {
        lin = xcd_remap_h(blockIdx.x, tiles_m * tiles_n);
        split = blockIdx.y;
    }
    int tm = lin / tiles_n;
    const int n0 = (lin % tiles_n) * BN;
    if (p.frame_tiles > 0) {
        const int per_sample = p.OH * p.frame_tiles;
        const int img = tm / per_sample, rem = tm - img * per_sample;
        const int pb = rem / p.OH, f = rem - pb * p.OH;
        tm = img * per_sample + f * p.frame_tiles + pb;
    }
    const int m0 = tm * BM;
    const int Cin = p.C1 + p.C2;
    // parity classes (GEN, p.pcls): this tile's class and the live taps kh0, kh0 + khs, ... / kw0, kw0 + kws, ...
    int cls = 0, cph = 0, cpw = 0, kh0 = 0, khs = 1, kw0 = 0, kws = 1, ntw = p.KW;
    int nkt = p.K / EPT;
    int kt_begin = split * p.ktiles_per_split;
    int kt_end = min(nkt, kt_begin + p.ktiles_per_split);
    if (p.pcls) {
        cls = m0 / p.cls_R;
        int nth = p.KH;
        if (p.pcls == 2) {
            // position classes of a short same-size 1-d conv (ncw = OW classes, one per output column ow = cls): the taps that fall into
            // the zero padding are skipped -- kw in [pw - ow, W - 1 + pw - ow] clipped to the filter
            cph = 0; cpw = cls;
            kw0 = max(0, p.pw - cls);
            const int kw1 = min(p.KW - 1, p.W - 1 + p.pw - cls);
            ntw = kw1 - kw0 + 1;
        } else {
            cph = p.ncw == 2 ? (cls >> 1) : cls;
            cpw = p.ncw == 2 ? (cls & 1) : 0;
            if (p.nch == 1) { cph = 0; cpw = cls; }
            if (p.nch == 2) { kh0 = (p.ph + cph) & 1; khs = 2; nth = (p.KH - kh0 + 1) >> 1; }
            if (p.ncw == 2) { kw0 = (p.pw + cpw) & 1; kws = 2; ntw = (p.KW - kw0 + 1) >> 1; }
        }
        nkt = nth * ntw * (Cin / EPT);
        kt_begin = (int)(((long)split * nkt) / p.splitk);           // the class's own k tiles, spread evenly over the slices
        kt_end = (int)(((long)(split + 1) * nkt) / p.splitk);
    }
    const float* wsel = reinterpret_cast<const float*>(p.w);
    const float* bias_sel = p.bias;

    // loader: thread -> (row lrow + 32 j, float4 c4 of the 32-float k tile)
    const int lrow = tid >> 3, c4 = tid & 7;
    int a_ihb[AL], a_iwb[AL], a_imgh[AL], a_lin1[AL], a_lin2[AL];
#pragma unroll
    for (int j = 0; j < AL; ++j) {
        const int m = m0 + j * RP + lrow;
        const bool ok = m < p.M;
        const uint32_t mm = ok ? (uint32_t)m : 0u;
        int ow, oh, img;
        if (p.pcls) {
            parity_row_h(p, ok ? m : m0, cls, cph, cpw, oh, ow, img);
        } else {
            const uint32_t t = fdivh(mm, p.fd_ow);
            ow = (int)(mm - t * p.OW);
            const uint32_t im = fdivh(t, p.fd_oh);
            oh = (int)(t - im * p.OH);
            img = (int)im;
        }
        a_imgh[j] = (int)img * p.H;
        a_ihb[j] = ok ? oh * p.sh - p.ph : -(1 << 28);
        a_iwb[j] = ow * p.sw - p.pw;
        const int pix = (a_imgh[j] + a_ihb[j]) * p.W + a_iwb[j];
        a_lin1[j] = pix * p.xp1 + c4 * 4;
        a_lin2[j] = pix * p.C2 + c4 * 4;
    }
    const float* zsrc = reinterpret_cast<const float*>(p.zeros);
    const float* b_src[BL];
    bool b_ok[BL];
#pragma unroll
    for (int j = 0; j < BL; ++j) {
        const int n = n0 + j * RP + lrow;
        b_ok[j] = n < p.Cout;
        b_src[j] = wsel + (size_t)(b_ok[j] ? n : 0) * p.K + c4 * 4;
    }
    const int shift = (p.ups || p.idil == 2) ? 1 : 0;
    const int pmask = (p.idil == 2) ? 1 : 0;
    // position of the next k tile: the itap-th LIVE tap (all taps without parity classes) and the channel offset inside it
    const int itap = (kt_begin * EPT) / Cin;
    int ic0 = kt_begin * EPT - itap * Cin;
    const int ntw1 = ntw > 0 ? ntw : 1;                         // (a class without live taps has no k tiles: its outputs are bias + residual)
    int ikh = kh0 + (itap / ntw1) * khs, ikw = kw0 + (itap % ntw1) * kws;
    int ik0 = (ikh * p.KW + ikw) * Cin + ic0;
    int it = kt_begin;

    // request the next k tile (16-B loads into the given register set; past the slice's end: the zero line) and advance the position
    auto load = [&](f32x4 (&ra)[AL], f32x4 (&rb)[BL]) {
        const bool live = it < kt_end;
        const bool first = ic0 < p.C1;
        const float* src = reinterpret_cast<const float*>(first ? p.x : p.x2);
        const uint32_t Cs = (uint32_t)(first ? p.C1 : p.C2);
        const uint32_t cc = (uint32_t)((first ? ic0 : ic0 - p.C1) + c4 * 4);
        const int s_tap = (ikh * p.W + ikw) * (first ? p.xp1 : p.C2) + (first ? ic0 : ic0 - p.C1);
#pragma unroll
        for (int j = 0; j < AL; ++j) {
            int ih = a_ihb[j] + ikh, iw = a_iwb[j] + ikw;
            bool ok;
            uint32_t off;
            if constexpr (GEN) {
                ok = live & ((unsigned)ih < (unsigned)p.HL) & ((unsigned)iw < (unsigned)p.WL) & (((ih | iw) & pmask) == 0);
                ih >>= shift; iw >>= shift;
                off = ((uint32_t)(a_imgh[j] + ih) * (uint32_t)p.W + (uint32_t)iw) * Cs + cc;
            } else {
                ok = live & ((unsigned)ih < (unsigned)p.H) & ((unsigned)iw < (unsigned)p.W);
                off = (uint32_t)((first ? a_lin1[j] : a_lin2[j]) + s_tap);
            }
            asm volatile("" : "+v"(off));
            const float* g = src + off;
            g = ok ? g : zsrc;
            ra[j] = *(const gf32x4_t*)(uint64_t)g;              // (address space 1: a global_load, never a flat_load -- flat loads also count on lgkmcnt)
        }
#pragma unroll
        for (int j = 0; j < BL; ++j) {
            uint64_t gi = (uint64_t)(b_src[j] + ik0);
            asm volatile("" : "+v"(gi));
            gi = (live & b_ok[j]) ? gi : (uint64_t)zsrc;
            rb[j] = *reinterpret_cast<const gf32x4_t*>(gi);
        }
        ++it;
        ic0 += EPT;
        const bool wrap = ic0 >= Cin;
        ic0 = wrap ? 0 : ic0;
        const int kw1 = ikw + (wrap ? kws : 0);
        const bool wrap2 = kw1 >= p.KW;
        ikw = wrap2 ? kw0 : kw1;
        ikh += wrap2 ? khs : 0;
        ik0 = (ikh * p.KW + ikw) * Cin + ic0;
    };
    // LDS position of this thread's 8-B piece (4 bf16) of plane row r: 16-B chunk (c4 >> 1) ^ ((r >> 2) & 3), half c4 & 1
    int w_off[AL > BL ? AL : BL];
#pragma unroll
    for (int j = 0; j < (AL > BL ? AL : BL); ++j) {
        const int r = j * RP + lrow;
        w_off[j] = r * ROWH + ((((c4 >> 1) ^ ((r >> 2) & 3)) << 4) | ((c4 & 1) << 3));
    }
    auto split_store_a = [&](const f32x4 (&ra)[AL], unsigned char* stage) {
#pragma unroll
        for (int j = 0; j < AL; ++j) {
            uint32_t h0, m0_, l0, h1, m1, l1;
            split3_pair_h(ra[j][0], ra[j][1], h0, m0_, l0);
            split3_pair_h(ra[j][2], ra[j][3], h1, m1, l1);
            *reinterpret_cast<uint2*>(stage + w_off[j]) = uint2{h0, h1};
            *reinterpret_cast<uint2*>(stage + PA + w_off[j]) = uint2{m0_, m1};
            *reinterpret_cast<uint2*>(stage + 2 * PA + w_off[j]) = uint2{l0, l1};
        }
    };
    auto split_store_b = [&](const f32x4 (&rb)[BL], unsigned char* stage) {
        unsigned char* bb = stage + 3 * PA;
#pragma unroll
        for (int j = 0; j < BL; ++j) {
            uint32_t h0, m0_, l0, h1, m1, l1;
            split3_pair_h(rb[j][0], rb[j][1], h0, m0_, l0);
            split3_pair_h(rb[j][2], rb[j][3], h1, m1, l1);
            *reinterpret_cast<uint2*>(bb + w_off[j]) = uint2{h0, h1};
            *reinterpret_cast<uint2*>(bb + PB + w_off[j]) = uint2{m0_, m1};
            *reinterpret_cast<uint2*>(bb + 2 * PB + w_off[j]) = uint2{l0, l1};
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm = (wid / WVN) * WM, wn = (wid % WVN) * WN;
    const int lr = lane & 31, lk = lane >> 5;
    const int rswz = (lr >> 2) & 3;                             // ((row >> 2) & 3) of every operand row this lane reads (wm, i * 32 are multiples of 16)
    int a_off[TM], b_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_off[i] = (wm + i * 32 + lr) * ROWH;
#pragma unroll
    for (int j = 0; j < TN; ++j) b_off[j] = 3 * PA + (wn + j * 32 + lr) * ROWH;
    int pos[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) pos[h] = (((h << 1) | lk) ^ rswz) << 4;

    typedef __attribute__((ext_vector_type(8))) __bf16 bfx8;
    auto mfma6 = [&](const unsigned char* base, int h) {
        bfx8 a[3][TM], b[3][TN];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[q][i] = *reinterpret_cast<const bfx8*>(base + q * PA + a_off[i] + pos[h]);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[q][j] = *reinterpret_cast<const bfx8*>(base + q * PB + b_off[j] + pos[h]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x16 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], c, 0, 0, 0);     // lo  * hi
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], c, 0, 0, 0);     // hi  * lo
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], c, 0, 0, 0);     // mid * mid
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], c, 0, 0, 0);     // mid * hi
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], c, 0, 0, 0);     // hi  * mid
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], c, 0, 0, 0);     // hi  * hi
                acc[i][j] = c;
            }
    };

    // ---- prologue: tile kt_begin staged, tiles kt_begin + 1 .. kt_begin + PF - 1 requested.  PF register sets of k tiles are in flight:
    // these launches wait ~1 us for a k tile against ~0.2 us of MFMA work on it, and with two sets a workgroup had 32 KB in flight --
    // 64 KB per CU at ~2 us of latency is the "8 TB/s load-path ceiling" the small-M GEMMs of the ConditionalUnet1D ran into (Little's
    // law, not a bandwidth limit): four sets double it (PF = 4: the 64 x 64 tile, 64 staging VGPRs).
    f32x4 ra0[AL], rb0[BL], ra1[AL], rb1[BL], ra2[PF == 4 ? AL : 1], rb2[PF == 4 ? BL : 1], ra3[PF == 4 ? AL : 1], rb3[PF == 4 ? BL : 1];
    load(ra0, rb0);
    load(ra1, rb1);
    if constexpr (PF == 4) {
        load(ra2, rb2);
        load(ra3, rb3);
    }
    split_store_a(ra0, smem);
    split_store_b(rb0, smem);
    __syncthreads();
    // one k tile: `cur` registers hold tile kt + 1 (requested PF - 1 steps ago), tile kt + PF is requested into `nxt` (the set whose
    // tile was split and stored one step ago)
    auto step = [&](f32x4 (&ra_c)[AL], f32x4 (&rb_c)[BL], f32x4 (&ra_n)[AL], f32x4 (&rb_n)[BL], int buf) {
        unsigned char* cur = smem + buf * STG;
        unsigned char* oth = smem + (buf ^ 1) * STG;
        load(ra_n, rb_n);
        mfma6(cur, 0);
        split_store_a(ra_c, oth);                               // (VALU of the split runs under the MFMAs around it)
        mfma6(cur, 1);
        split_store_b(rb_c, oth);
        __syncthreads();
    };
    // the register sets keep fixed roles at the loop header (no copies of in-flight load results): the loop body is PF steps, left
    // in the middle when the slice ends (a uniform forward branch; every wave has then passed that step's barrier after its last LDS
    // access, the trailing register loads hit the zero line)
    if constexpr (PF == 4) {
        for (int kt = kt_begin; kt < kt_end; kt += 4) {
            step(ra1, rb1, ra0, rb0, 0);
            if (kt + 1 >= kt_end) break;
            step(ra2, rb2, ra1, rb1, 1);
            if (kt + 2 >= kt_end) break;
            step(ra3, rb3, ra2, rb2, 0);
            if (kt + 3 >= kt_end) break;
            step(ra0, rb0, ra3, rb3, 1);
        }
    } else {
        for (int kt = kt_begin; kt < kt_end; kt += 2) {
            step(ra1, rb1, ra0, rb0, 0);
            if (kt + 1 >= kt_end) break;
            step(ra0, rb0, ra1, rb1, 1);
        }
    }
    // (every wave passed the loop's last barrier after its last LDS access; the trailing register loads hit the zero line)
    static_assert(WVM * WVN * 32 * WN * 4 <= S * STAGE, "epilogue staging exceeds the operand buffers");
    if (p.pcls) {
        conv_f32_epilogue_rm<BM, BN, WVM, WVN, RowsParityH>(p, acc, smem, RowsParityH{&p, m0, cls, cph, cpw}, n0, split, bias_sel);
        return;
    }
    conv_f32_epilogue<BM, BN, WVM, WVN>(p, acc, smem, m0, n0, split, bias_sel);
}

// ---- three-plane conv over PRE-SPLIT operands (round 5).  conv_igemm_f32x3<64, 64> on the ConditionalUnet1D's small-M /
// deep-K GEMMs (M <= 1024 rows, K up to 10 240) spends a k tile's time on splitting: every one of the Cout / 64 column tiles splits the
// same A rows again and every one of the M / 64 row tiles the same weights (per k tile and wave ~108 VALU + 6 ds_write2st64_b64 against
// 12 MFMAs; the LDS store path and the conversion VALU, not the matrix pipe or the loads, set its ~0.9 us per k tile --
// tools/probes/r5/ab_hook_probe.py: four register sets of k tiles in flight instead of two changed nothing).  Here the operands ARRIVE
// split: x / x2 / w point at three bf16 planes (hi, mid, lo; `xps` / `x2ps` / `wps` elements apart) written once by the launches that
// produce them -- the GroupNorm kernels for activations and gradients (v2a_groupnorm_fwd_s / _bwd_s, yh_plane_stride), the optimiser's
// update kernel and the transposing pack launch for the weights -- and both operands move HBM / L2 -> LDS by LDS-DMA (no staging VGPRs,
// no conversion, no ds_write): per k tile and thread six 16-B pieces.  Same tile, same LDS image (64-B plane rows, 16-B chunk swizzle
// p ^ ((row >> 2) & 3), applied on the source side), same six plane products in the same order, same split plan and epilogue as
// conv_igemm_f32x3<64, 64>: results are BIT-IDENTICAL to it (tests/test_ops_gpu.py).  S LDS stages of 24 KB with counted vmcnt + one raw
// barrier per k tile (conv_igemm_f32p's scheme); pieces past the slice's end read the zero line.
template <int BM, int BN, int S, int WVM = 2, int WVN = 2>
__global__ __launch_bounds__(64 * WVM * WVN, (S * 3 * (BM + BN) * 64 <= 53 * 1024) ? 3 : ((S * 3 * (BM + BN) * 64 <= 80 * 1024) ? 2 : 1)) void conv_p3(const ConvDescH p) {
    constexpr int EPT = 32, ROWH = 64;
    constexpr int NT = 64 * WVM * WVN, RP = NT / 4;                    // threads; rows of a plane image one pass of the threads fills (4 chunks per row)
    constexpr int AL = BM / RP, BL = BN / RP, NL = 3 * (AL + BL);      // DMA pieces per thread and k tile
    static_assert(AL >= 1 && BL >= 1, "tile rows per pass");
    constexpr int WM = BM / WVM, WN = BN / WVN, TM = WM / 32, TN = WN / 32;  // WVM x WVN waves, each TM x TN accumulators of 32 x 32
    constexpr int PA = BM * ROWH, PB = BN * ROWH, STG = 3 * (PA + PB), STAGE = STG;
    static_assert(S >= 2 && (S - 2) * NL <= 63, "vmcnt is a 6-bit counter");
    static_assert(S * STG <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(128))) unsigned char smem[S * STG];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = (p.Cout + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    int lin, split;
    if (p.split_xcd > 0) {                                      // split-K slices pinned to XCDs (see conv_igemm_f32x3)
        const int L = (int)blockIdx.x, xcd = L & 7, slot = L >> 3, tiles = tiles_m * tiles_n;
        const int sl = slot / tiles;
        split = xcd * p.split_xcd + sl;
        lin = slot - sl * tiles;
    } else {
        lin = xcd_remap_h(blockIdx.x, tiles_m * tiles_n);
        split = blockIdx.y;
    }
    const int tm = lin / tiles_n;
    const int n0 = (lin % tiles_n) * BN, m0 = tm * BM;
    const int Cin = p.C1 + p.C2;
    const int nkt = p.K / EPT;
    const int kt_begin = split * p.ktiles_per_split;
    const int kt_end = min(nkt, kt_begin + p.ktiles_per_split);
    typedef unsigned short hT;
    const hT* xs = reinterpret_cast<const hT*>(p.x);
    const hT* x2s = reinterpret_cast<const hT*>(p.x2);
    const hT* ws = reinterpret_cast<const hT*>(p.w);
    const hT* zsrc = reinterpret_cast<const hT*>(p.zeros);

    // DMA source state: pass j of the 256 threads fills rows 64 j .. 64 j + 63 of a plane image; this thread's piece is (row 64 j + tid / 4,
    // position tid % 4) and carries the row's 16-B chunk (tid % 4) ^ ((row >> 2) & 3) -- 8 bf16 of the 32-element k tile -- for every j
    const int lrow = tid >> 2;
    const int chunk = (tid & 3) ^ ((lrow >> 2) & 3);
    int a_ihb[AL], a_iwb[AL], a_lin1[AL], a_lin2[AL];
#pragma unroll
    for (int j = 0; j < AL; ++j) {
        const int m = m0 + j * RP + lrow;
        const bool ok = m < p.M;
        const uint32_t mm = ok ? (uint32_t)m : 0u;
        const uint32_t t = fdivh(mm, p.fd_ow);
        const int ow = (int)(mm - t * p.OW);
        const uint32_t img = fdivh(t, p.fd_oh);
        const int oh = (int)(t - img * p.OH);
        a_ihb[j] = ok ? oh * p.sh - p.ph : -(1 << 28);          // rows past M fail the bounds test below
        a_iwb[j] = ow * p.sw - p.pw;
        const int pix = ((int)img * p.H + a_ihb[j]) * p.W + a_iwb[j];
        a_lin1[j] = pix * p.C1 + chunk * 8;
        a_lin2[j] = pix * p.C2 + chunk * 8;
    }
    bool b_ok[BL];
    size_t b_row[BL];
#pragma unroll
    for (int j = 0; j < BL; ++j) {
        const int n = n0 + j * RP + lrow;
        b_ok[j] = n < p.Cout;
        b_row[j] = (size_t)(b_ok[j] ? n : 0) * p.K + chunk * 8;
    }
    int ik0 = kt_begin * EPT;
    int itap = ik0 / Cin;
    int ic0 = ik0 - itap * Cin;
    int ikh = itap / p.KW, ikw = itap - ikh * p.KW;
    int it = kt_begin;

    // pieces [q0, q1) of the pending k tile go to stage `buf`: piece = (plane, pass); first the 3 AL pieces of A, then the 3 BL of B
    auto issue = [&](int buf, int q0, int q1) {
        unsigned char* abase = smem + buf * STG + wid * 1024;
        const bool live = it < kt_end;
        const bool first = ic0 < p.C1;
        const hT* src = first ? xs : x2s;
        const size_t ps = first ? p.xps : p.x2ps;
        const int s_tap = (ikh * p.W + ikw) * (first ? p.C1 : p.C2) + (first ? ic0 : ic0 - p.C1);
#pragma unroll
        for (int q = q0; q < q1; ++q) {
            if (q < 3 * AL) {
                const int pl = q / AL, j = q % AL;
                const int ih = a_ihb[j] + ikh, iw = a_iwb[j] + ikw;
                const bool aok = live & ((unsigned)ih < (unsigned)p.H) & ((unsigned)iw < (unsigned)p.W);
                uint32_t aoff = (uint32_t)((first ? a_lin1[j] : a_lin2[j]) + s_tap);
                asm volatile("" : "+v"(aoff));                  // keep the offset arithmetic out of an exec-masked region (plain select below)
                const hT* g = src + (size_t)pl * ps + aoff;
                g = aok ? g : zsrc;
                __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(abase + pl * PA + j * RP * 64), 16, 0, 0);
            } else {
                const int pl = (q - 3 * AL) / BL, j = (q - 3 * AL) % BL;
                uint64_t gi = (uint64_t)(ws + (size_t)pl * p.wps + b_row[j] + ik0);
                asm volatile("" : "+v"(gi));
                const hT* g = (live & b_ok[j]) ? reinterpret_cast<const hT*>(gi) : zsrc;
                __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(abase + 3 * PA + pl * PB + j * RP * 64), 16, 0, 0);
            }
        }
    };
    auto advance = [&]() {
        ++it;
        ik0 += EPT;
        ic0 += EPT;
        const bool wrap = ic0 >= Cin;
        ic0 = wrap ? 0 : ic0;
        const int kw1 = ikw + (wrap ? 1 : 0);
        const bool wrap2 = kw1 == p.KW;
        ikw = wrap2 ? 0 : kw1;
        ikh += wrap2 ? 1 : 0;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wm = (wid / WVN) * WM, wn = (wid % WVN) * WN;
    const int lr = lane & 31, lk = lane >> 5;
    const int rswz = (lr >> 2) & 3;
    int a_off[TM], b_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_off[i] = (wm + i * 32 + lr) * ROWH;
#pragma unroll
    for (int j = 0; j < TN; ++j) b_off[j] = 3 * PA + (wn + j * 32 + lr) * ROWH;
    int pos[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) pos[h] = (((h << 1) | lk) ^ rswz) << 4;
    typedef __attribute__((ext_vector_type(8))) __bf16 bfx8;
    // one k half: every fragment first, then the six plane products of each accumulator (smallest first), the accumulators' dependent
    // chains interleaved (a dependent MFMA waits 16 passes for its predecessor)
    auto mfma6 = [&](const unsigned char* base, int h) {
        bfx8 a[3][TM], b[3][TN];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[q][i] = *reinterpret_cast<const bfx8*>(base + q * PA + a_off[i] + pos[h]);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[q][j] = *reinterpret_cast<const bfx8*>(base + q * PB + b_off[j] + pos[h]);
        }
        constexpr int PQ[6][2] = {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};      // (A plane, B plane): lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PQ[t][0]][i], b[PQ[t][1]][j], acc[i][j], 0, 0, 0);
    };

    // ---- prologue: S - 1 k tiles in flight
#pragma unroll
    for (int s_ = 0; s_ < S - 1; ++s_) {
        issue(s_, 0, NL);
        advance();
    }
    int cbuf = 0, ibuf = S - 1;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        wait_vmcnt_h<(S - 2) * NL>();                           // this wave's pieces of tile kt have landed; younger tiles stay in flight
        __builtin_amdgcn_s_barrier();                           // ... everyone's have, and everyone is done reading the stage issued into next
        const unsigned char* base = smem + cbuf * STG;
        issue(ibuf, 0, NL / 2);
        mfma6(base, 0);
        issue(ibuf, NL / 2, NL);
        mfma6(base, 1);
        advance();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's operand reads are done before it reaches the next barrier
        cbuf = (cbuf + 1 == S) ? 0 : cbuf + 1;
        ibuf = (ibuf + 1 == S) ? 0 : ibuf + 1;
    }
    wait_vmcnt_h<0>();                                          // the zero-line pieces past the end still land in the stages
    __syncthreads();
    static_assert(WVM * WVN * 32 * WN * 4 <= S * STAGE, "epilogue staging exceeds the operand buffers");
    conv_f32_epilogue<BM, BN, WVM, WVN>(p, acc, smem, m0, n0, split, p.bias);
}

// ---- three-plane 3x3 / stride 1 / pad 1 conv with a spatial halo tile in LDS (round 4): the forward / data-gradient counterpart of
// csrc/igemm.hip wgrad_x3h_body, for the square maps of the policy's ResNet-18 encoders (32 x 32, 16 x 16, 8 x 8, 4 x 4).
// conv_igemm_f32x3 gathers and splits the A tile once per filter tap and column tile: nine loads and nine fp32 -> plane conversions of
// what is, up to a one-pixel shift, the same patch (its launches are bound by that traffic through the vector L1 and by the conversion
// VALU, profiles/r04_x3_ablation.txt).  Here a workgroup owns 128 consecutive output pixels (whole map rows: 4 x 32, 8 x 16, two 8 x 8
// maps, eight 4 x 4 maps) x 64 output channels and walks the reduction as (32-channel chunk) x (tap): per chunk the zero-padded halo of
// its pixels is loaded and split ONCE into three bf16 plane images ([plane][slot][32 channels], 64-B rows, 16-B pieces XOR-swizzled by
// (slot >> 2) & 3 as in conv_igemm_f32x3), the nine taps read shifted slot windows of it; per (chunk, tap) only the 64 x 32 weight tile
// moves (registers -> split -> one of two LDS stages).  Conversions per MAC: 1/9 of the A side's; A bytes through L1: 1/9.
// One halo buffer + two weight stages = 59 ... 80 KB: two workgroups per CU.  Epilogue, split-K slabs (over channel chunks) and their
// consumers as conv_igemm_f32x3.  fp32-equivalent accuracy (the same six plane products per block, smallest first).
template <int OWC>
struct HX3 {
    static constexpr int BM = 128, BN = 64, CK = 32;
    static constexpr bool PATCH = OWC == 0;                           // OWC = 0: 8 x 16 pixel patches of an H x W map given at run time
    static constexpr int PIXI = PATCH ? 1 << 30 : OWC * OWC;          // pixels per (square) map
    static constexpr int SP = PIXI >= BM ? 1 : BM / PIXI;             // maps per tile
    static constexpr int PW = PATCH ? 16 : OWC;                       // tile width in pixels
    static constexpr int PH = PATCH ? 8 : (PIXI >= BM ? BM / OWC : OWC);   // map rows per tile and map
    static constexpr int HWD = PW + 2, HS = (PH + 2) * HWD, NS = SP * HS;   // halo row pitch, slots per map, slots per tile
    static constexpr int NHP = (NS * 8 + 255) / 256;                  // halo loader passes (32 slots x 8 float4 per pass of 256 threads)
    static constexpr int PHB = NS * 64;                               // bytes of one plane of the halo image
    static constexpr int PB = BN * 64;                                // bytes of one plane of a weight tile
    static constexpr int LDS = 3 * PHB + 2 * 3 * PB;
};
template <int OWC>
__global__ __launch_bounds__(256, 2) void conv_halo_x3(const ConvDescH p) {
    typedef HX3<OWC> G;
    constexpr int BM = G::BM, BN = G::BN, NHP = G::NHP, PHB = G::PHB, PB = G::PB, HWD = G::HWD, HS = G::HS, NS = G::NS;
    constexpr int S = 1, STAGE = G::LDS;                        // (names of the shared epilogue's size check)
    __shared__ __attribute__((aligned(128))) unsigned char smem[G::LDS];
    unsigned char* Hs = smem;
    unsigned char* Bs = smem + 3 * PHB;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = p.Cout / BN, tiles_m = p.M / BM;
    const int lin = xcd_remap_h(blockIdx.x, tiles_m * tiles_n);
    const int tm = lin / tiles_n;
    const int n0 = (lin - tm * tiles_n) * BN, m0 = tm * BM;
    const int split = blockIdx.y;
    const int Cin = p.C1;
    const int nchunks = Cin >> 5;
    const int ck_begin = split * p.ktiles_per_split;
    const int ck_end = min(nchunks, ck_begin + p.ktiles_per_split);
    // first map / first map row (and column) of the tile (row0 = 0 when a tile holds whole maps).  `ups`: the nearest x2 upsample in
    // front of the conv (unet.py:105-115) folded into the halo gather -- the source map is (mapH / 2) x (mapW / 2)
    int img0, row0, col0 = 0;
    const int mapH = G::PATCH ? p.OH : OWC, mapW = G::PATCH ? p.OW : OWC;
    if constexpr (G::PATCH) {
        const int tpr = mapW >> 4, tpi = (mapH >> 3) * tpr;
        img0 = tm / tpi;
        const int rem = tm - img0 * tpi, ty = rem / tpr;
        row0 = ty * 8;
        col0 = (rem - ty * tpr) * 16;
    } else {
        img0 = m0 / G::PIXI;
        row0 = (m0 % G::PIXI) / G::PW;
    }
    const int srcH = p.ups ? mapH >> 1 : mapH, srcW = p.ups ? mapW >> 1 : mapW;
    const float* xs = reinterpret_cast<const float*>(p.x);
    const float* zsrc = reinterpret_cast<const float*>(p.zeros);
    const int lrow = tid >> 3, c4 = tid & 7;

    // halo loader: pass q of the 256 threads covers slots 32 q .. 32 q + 31 (thread: slot 32 q + lrow, float4 c4 of its 32 channels)
    int h_src[NHP], h_dst[NHP];
#pragma unroll
    for (int q = 0; q < NHP; ++q) {
        const int slot = q * 32 + lrow;
        const int sp = slot / HS, rem = slot - sp * HS;
        const int hy = rem / HWD, hx = rem - hy * HWD;
        const int ih = row0 + hy - 1, iw = col0 + hx - 1;
        const bool valid = slot < NS;
        const bool ok = valid && (unsigned)ih < (unsigned)mapH && (unsigned)iw < (unsigned)mapW;
        const int ihs = p.ups ? ih >> 1 : ih, iws = p.ups ? iw >> 1 : iw;
        h_src[q] = ok ? (((img0 + sp) * srcH + ihs) * srcW + iws) * Cin + c4 * 4 : -1;
        h_dst[q] = valid ? slot * 64 + ((((c4 >> 1) ^ ((slot >> 2) & 3)) << 4) | ((c4 & 1) << 3)) : -1;
    }
    // (no branches around the loads, here and below: past the slice's end they read the zero line -- a conditional load makes the
    // compiler's wait-count bookkeeping fall back to vmcnt(0) at the top of every step, which serialises the whole prefetch)
    auto loadH = [&](int ck, bool live, f32x4 (&rh)[NHP]) {
#pragma unroll
        for (int q = 0; q < NHP; ++q) {
            uint32_t off = (uint32_t)(h_src[q] + ck * 32);
            asm volatile("" : "+v"(off));
            const float* g = xs + off;
            g = (live & (h_src[q] >= 0)) ? g : zsrc;
            rh[q] = *(const gf32x4_t*)(uint64_t)g;
        }
    };
    auto storeH = [&](const f32x4 (&rh)[NHP]) {
#pragma unroll
        for (int q = 0; q < NHP; ++q) {
            if (h_dst[q] < 0) continue;
            uint32_t h0, m0_, l0, h1, m1, l1;
            split3_pair_h(rh[q][0], rh[q][1], h0, m0_, l0);
            split3_pair_h(rh[q][2], rh[q][3], h1, m1, l1);
            unsigned char* d = Hs + h_dst[q];
            *reinterpret_cast<uint2*>(d) = uint2{h0, h1};
            *reinterpret_cast<uint2*>(d + PHB) = uint2{m0_, m1};
            *reinterpret_cast<uint2*>(d + 2 * PHB) = uint2{l0, l1};
        }
    };
    // weight tile of (chunk, tap): rows n0 + lrow, n0 + lrow + 32 of the [Cout][tap][Cin] pack
    const float* b_src[2];
    int w_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = j * 32 + lrow;
        b_src[j] = reinterpret_cast<const float*>(p.w) + (size_t)(n0 + r) * p.K + c4 * 4;
        w_off[j] = r * 64 + ((((c4 >> 1) ^ ((r >> 2) & 3)) << 4) | ((c4 & 1) << 3));
    }
    auto loadB = [&](int off, bool live, f32x4 (&rb)[2]) {      // off = tap * Cin + chunk * 32
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            uint64_t gi = (uint64_t)(b_src[j] + off);
            asm volatile("" : "+v"(gi));
            gi = live ? gi : (uint64_t)zsrc;
            rb[j] = *reinterpret_cast<const gf32x4_t*>(gi);
        }
    };
    auto storeB = [&](const f32x4 (&rb)[2], int stage) {
        unsigned char* bb = Bs + stage * 3 * PB;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            uint32_t h0, m0_, l0, h1, m1, l1;
            split3_pair_h(rb[j][0], rb[j][1], h0, m0_, l0);
            split3_pair_h(rb[j][2], rb[j][3], h1, m1, l1);
            *reinterpret_cast<uint2*>(bb + w_off[j]) = uint2{h0, h1};
            *reinterpret_cast<uint2*>(bb + PB + w_off[j]) = uint2{m0_, m1};
            *reinterpret_cast<uint2*>(bb + 2 * PB + w_off[j]) = uint2{l0, l1};
        }
    };

    f32x16 acc[2][1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    const int wm = (wid >> 1) * 64, wn = (wid & 1) * 32;
    const int lr = lane & 31, lk = lane >> 5;
    int slot0[2];                                               // halo slot of tap (0, 0) of the lane's two output pixels
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pl = wm + i * 32 + (G::PATCH ? patch16_perm_h(lr) : lr);
        const int sp = pl / (G::PH * G::PW), q = pl - sp * (G::PH * G::PW);
        const int py = q / G::PW, px = q - py * G::PW;
        slot0[i] = sp * HS + py * HWD + px;
    }
    const int b_off = (wn + lr) * 64, brs = (lr >> 2) & 3;
    typedef __attribute__((ext_vector_type(8))) __bf16 bfx8;
    auto compute = [&](int stage, int toff) {
        const unsigned char* bb = Bs + stage * 3 * PB;
        int ab[2], as[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int sl = slot0[i] + toff;
            ab[i] = sl * 64;
            as[i] = (sl >> 2) & 3;
        }
        // every operand fragment of the step is requested before its first MFMA (the fence keeps the compiler from sinking the
        // reads back between the MFMAs, where each one waited out a full LDS round trip)
        bfx8 a[2][3][2], b[2][3];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kp = (h << 1) | lk;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a[h][q][i] = *reinterpret_cast<const bfx8*>(Hs + q * PHB + ab[i] + ((kp ^ as[i]) << 4));
                b[h][q] = *reinterpret_cast<const bfx8*>(bb + q * PB + b_off + ((kp ^ brs) << 4));
            }
        }
        asm volatile("" ::: "memory");
        f32x16 c0 = acc[0][0], c1 = acc[1][0];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // the two accumulators' chains interleaved (a dependent MFMA waits 16 passes for its predecessor)
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[h][2][0], b[h][0], c0, 0, 0, 0);     // lo  * hi
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[h][2][1], b[h][0], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[h][0][0], b[h][2], c0, 0, 0, 0);     // hi  * lo
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[h][0][1], b[h][2], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[h][1][0], b[h][1], c0, 0, 0, 0);     // mid * mid
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[h][1][1], b[h][1], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[h][1][0], b[h][0], c0, 0, 0, 0);     // mid * hi
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[h][1][1], b[h][0], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[h][0][0], b[h][1], c0, 0, 0, 0);     // hi  * mid
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[h][0][1], b[h][1], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[h][0][0], b[h][0], c0, 0, 0, 0);     // hi  * hi
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[h][0][1], b[h][0], c1, 0, 0, 0);
        }
        acc[0][0] = c0;
        acc[1][0] = c1;
    };

    // weight tiles travel three steps ahead in three register sets with fixed roles (9 taps per chunk = 0 mod 3: the tile of tap t
    // always sits in set t % 3): tile u is requested at the start of step u - 3, split and stored into the free LDS stage during
    // step u - 1, multiplied at step u -- the request has two and a half steps of MFMAs to land
    f32x4 rh[NHP], rb0[2], rb1[2], rb2[2];
    const int nsteps = (ck_end - ck_begin) * 9;
    // running (chunk, tap) of the tile three steps ahead
    int l_ck = ck_begin, l_tap = 0, l_step = 0;
    auto loadB_next = [&](f32x4 (&rb)[2]) {
        loadB(l_tap * Cin + l_ck * 32, l_step < nsteps, rb);
        ++l_step;
        const bool wrap = l_tap == 8;
        l_tap = wrap ? 0 : l_tap + 1;
        l_ck += wrap ? 1 : 0;
    };
    loadH(ck_begin, true, rh);
    loadB_next(rb0);
    loadB_next(rb1);
    loadB_next(rb2);
    storeH(rh);
    storeB(rb0, 0);
    __syncthreads();
    int stage = 0;
    for (int ck = ck_begin; ck < ck_end; ++ck) {
        const bool more = ck + 1 < ck_end;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // set tap % 3 was stored during the previous step: request the tile three steps ahead into it
            if (tap % 3 == 0) loadB_next(rb0);
            else if (tap % 3 == 1) loadB_next(rb1);
            else loadB_next(rb2);
            if (tap == 4) loadH(ck + 1, more, rh);              // the next chunk's halo flies under taps 4 .. 8
            compute(stage, (tap / 3) * HWD + (tap % 3));
            // tile step + 1 -> the other stage (its readers passed the previous barrier; past the end: a zero tile nobody reads)
            if ((tap + 1) % 3 == 0) storeB(rb0, stage ^ 1);
            else if ((tap + 1) % 3 == 1) storeB(rb1, stage ^ 1);
            else storeB(rb2, stage ^ 1);
            if (tap == 8) {
                __syncthreads();                                // every wave is done with this chunk's halo
                storeH(rh);
            }
            __syncthreads();
            stage ^= 1;
        }
    }
    static_assert(4 * 32 * 32 * 4 <= S * STAGE, "epilogue staging exceeds the operand buffers");
    if constexpr (G::PATCH)
        conv_f32_epilogue_rm<BM, BN, 2, 2, RowsPatch16H>(p, acc, smem, RowsPatch16H{(img0 * mapH + row0) * mapW + col0, mapW}, n0, split, p.bias);
    else
        conv_f32_epilogue<BM, BN, 2, 2>(p, acc, smem, m0, n0, split, p.bias);
}

template <typename T>
__global__ void conv_splitk_reduce_h(const ConvDescH p) {
    constexpr bool HALF = sizeof(T) == 2;
    constexpr bool F16 = is_f16s<T>::value;
    const size_t total = (size_t)p.M * p.Cout;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / p.Cout), n = (int)(idx - (size_t)m * p.Cout);
        float v = 0.f;
        for (int s = 0; s < p.splitk; ++s) v += p.partial[(size_t)s * total + idx];
        if (p.bias) v += p.bias[n];
        if (p.rowvec) v += p.rowvec[(size_t)(m / p.rows_per_batch) * p.Cout + n];
        if (p.residual) v += HALF ? v2a_h2f<F16>(reinterpret_cast<const uint16_t*>(p.residual)[idx]) : reinterpret_cast<const float*>(p.residual)[idx];
        if (p.residual_f) v += p.residual_f[idx];
        if (!HALF) reinterpret_cast<float*>(p.y)[idx] = v;
        else if (p.yf) p.yf[idx] = v;
        else reinterpret_cast<uint16_t*>(p.y)[idx] = v2a_f2h<F16>(v);
    }
}

// torch [Cout][Cin][taps] fp32 -> [Cout][taps][Cin] bf16 (round to nearest even); taps == 1 is a plain cast
template <bool F16>
__global__ void pack_weight_h_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int Cout, int Cin, int taps) {
    const size_t total = (size_t)Cout * Cin * taps;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(idx % Cin);
        const size_t t = idx / Cin;
        const int tap = (int)(t % taps);
        const int co = (int)(t / taps);
        out[idx] = v2a_f2h<F16>(w[((size_t)co * Cin + ci) * taps + tap]);
    }
}

__global__ void split3_f32_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, size_t n4, size_t ps) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        uint32_t h0, m0, l0, h1, m1, l1;
        split3_pair_h(v[0], v[1], h0, m0, l0);
        split3_pair_h(v[2], v[3], h1, m1, l1);
        *reinterpret_cast<uint2*>(y + 4 * i) = uint2{h0, h1};
        *reinterpret_cast<uint2*>(y + ps + 4 * i) = uint2{m0, m1};
        *reinterpret_cast<uint2*>(y + 2 * ps + 4 * i) = uint2{l0, l1};
    }
}
template <bool F16>
__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        uint2 o;
        o.x = v2a_pack_h2<F16>(v[0], v[1]);
        o.y = v2a_pack_h2<F16>(v[2], v[3]);
        reinterpret_cast<uint2*>(y)[i] = o;
    }
}
// [M][Cin] fp32 -> [M][Cpad] bf16, channels Cin .. Cpad-1 zero: the 6-channel input of the sampler's stem padded to one 32-channel
// chunk so that the stem conv runs on the halo kernel.  Thread = one 8-channel (16-B) piece of a row.
template <bool F16>
__global__ void pad_cast_f32_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, size_t M, int Cin, int Cpad) {
    const int pieces = Cpad >> 3;
    const size_t total = M * (size_t)pieces;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / pieces;
        const int c0 = (int)(i - m * pieces) * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (c0 + e < Cin) ? x[m * Cin + c0 + e] : 0.f;
        uint4 o;
        o.x = v2a_pack_h2<F16>(v[0], v[1]);
        o.y = v2a_pack_h2<F16>(v[2], v[3]);
        o.z = v2a_pack_h2<F16>(v[4], v[5]);
        o.w = v2a_pack_h2<F16>(v[6], v[7]);
        reinterpret_cast<uint4*>(y)[i] = o;
    }
}
template <bool F16>
__global__ void cast_bf16_f32_kernel(const uint16_t* __restrict__ x, float* __restrict__ y, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const uint2 u = reinterpret_cast<const uint2*>(x)[i];
        f32x4 v = {v2a_lo_h2<F16>(u.x), v2a_hi_h2<F16>(u.x), v2a_lo_h2<F16>(u.y), v2a_hi_h2<F16>(u.y)};
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}

static void f32_conv_mode_init();
static int g_maps_on = 1;     // conv_maps_x3 for the small square maps (v2a_debug_set_maps_kernel: A/B hook)
static int g_pcls_on = 1;     // parity-class tiles for zero-interleaved inputs (v2a_debug_set_parity_classes: A/B hook of the tests)
static int g_f32x3 = -1;      // fp32 convs by three bf16 planes (conv_igemm_f32x3): V2A_F32_CONV=exact / v2a_set_f32_conv_mode(0) select the exact-f32 MFMA kernels
// Tile / split plan shared by the LDS-DMA conv families.  128-row tiles (64 output columns for 64-wide layers); problems that 128-row
// tiles cannot spread over the chip (< 128 tiles) take 64 x 64 tiles; K is split so that one round of about 512 workgroups covers the
// launch, every slice at least 6 k tiles deep, at most 16 slices.
static void conv_plan_h(int M, int Cout, int K, int ept, int* bm, int* bn, int* tiles, int* s) {
    *bm = 128;
    *bn = Cout <= 64 ? 64 : 128;                   // 64-wide layers (ResNet layer1) would waste half of a 128-column tile
    *tiles = cdiv(M, 128) * cdiv(Cout, *bn);
    if (*tiles < 128) {
        *bm = 64;
        *bn = 64;
        *tiles = cdiv(M, 64) * cdiv(Cout, 64);
    }
    int sp = 1;
    const int nkt = K / ept;                       // k tiles of 128 B: 64 bf16 or 32 fp32
    const int slots = 512;                         // workgroups aimed at per launch
    if (*tiles < slots - slots / 8) {
        sp = slots / *tiles;
        const int smax = nkt / 6 < 16 ? nkt / 6 : 16;
        if (sp > smax) sp = smax;
        if (sp < 1) sp = 1;
    }
    *s = sp;
}

// split (over 32-channel chunks) of the three-plane halo conv: about 512 workgroups, at most 8 slabs
static int conv_halo_x3_split(int M, int Cout, int Cin) {
    const int tiles = (M / 128) * (Cout / 64), nchunks = Cin / 32;
    int s = 1;
    if (tiles < 448) {
        s = 512 / (tiles < 1 ? 1 : tiles);
        if (s > 8) s = 8;
        if (s > nchunks) s = nchunks;
        if (s < 1) s = 1;
    }
    const int cps = cdiv(nchunks, s);
    return cdiv(nchunks, cps);
}
extern "C" {

size_t v2a_conv2d_h_workspace_bytes(int M, int Cout, int K) {
    int bm, bn, tiles, s;
    conv_plan_h(M, Cout, K, 64, &bm, &bn, &tiles, &s);
    return s > 1 ? (size_t)s * M * Cout * sizeof(float) : 0;
}
// 1 when v2a_conv2d_fwd_h accepts a `stats` buffer for this problem (single-pass epilogue on 128-row tiles)
int v2a_conv2d_h_can_emit_stats(int M, int Cout, int K) {
    int bm, bn, tiles, s;
    conv_plan_h(M, Cout, K, 64, &bm, &bn, &tiles, &s);
    return (s == 1 && bm == 128 && Cout % 8 == 0) ? 1 : 0;
}
// the same question for v2a_conv2d_fwd_dma_f32 (k tile = 32 floats)
int v2a_conv2d_dma_f32_can_emit_stats(int M, int Cout, int K) {
    int bm, bn, tiles, s;
    conv_plan_h(M, Cout, K, 32, &bm, &bn, &tiles, &s);
    return (s == 1 && bm == 128 && Cout % 8 == 0) ? 1 : 0;
}
size_t v2a_conv2d_dma_f32_workspace_bytes(int M, int Cout, int K) {
    int bm, bn, tiles, s;
    conv_plan_h(M, Cout, K, 32, &bm, &bn, &tiles, &s);
    if (K % 9 == 0 && (K / 9) % 32 == 0 && M % 128 == 0 && Cout % 64 == 0) {      // the halo kernel may take it (geometry permitting)
        const int sh = conv_halo_x3_split(M, Cout, K / 9);
        if (sh > s) s = sh;
        if (M % 256 == 0) {                                                          // ... or conv_maps_x3 (csrc/igemm_x3m.hip)
            const int sm = conv_maps_x3_split(M, Cout, K / 9);
            if (sm > s) s = sm;
        }
    }
    return s > 1 ? (size_t)s * M * Cout * sizeof(float) : 0;
}

}  // extern "C"

// shared launcher of the two storage types (ept = elements per 128-B k tile: 64 bf16 / 32 fp32)
template <typename T>
static int conv_dma_launch(const void* x, const void* x2, const void* w_packed, const float* bias, const float* rowvec, const void* residual,
                           const float* residual_f32, void* y, float* y_f32, const void* zeros, int N, int H, int W, int C1, int C2,
                           int Cout, int KH, int KW, int sh, int sw, int ph, int pw, int ups, int idil, int OH, int OW, int rows_per_batch,
                           float* stats, void* workspace, size_t workspace_bytes, hipStream_t stream, int* nslab_out = nullptr,
                           int xpitch = 0) {
    constexpr int ept = 128 / (int)sizeof(T);
    if (nslab_out) *nslab_out = 0;
    if (!x || !w_packed || !zeros || N <= 0 || Cout <= 0) return V2A_ERR_ARG;
    if (C1 <= 0 || C1 % ept || C2 < 0 || C2 % ept || (C2 > 0 && !x2)) return V2A_ERR_ARG;
    if ((idil != 1 && idil != 2) || (idil == 2 && ups) || (residual && residual_f32)) return V2A_ERR_ARG;
    if ((((uintptr_t)x | (uintptr_t)x2 | (uintptr_t)w_packed | (uintptr_t)zeros | (uintptr_t)y | (uintptr_t)y_f32 | (uintptr_t)residual) & 15) != 0)
        return V2A_ERR_ARG;
    if ((double)N * H * W * (C1 > C2 ? C1 : C2) >= 4294967296.0) return V2A_ERR_ARG;      // the gather uses 32-bit element offsets
    ConvDescH p;
    p.pcls = 0;
    p.x = x; p.x2 = x2; p.w = w_packed;
    p.bias = bias; p.rowvec = rowvec; p.residual = residual; p.residual_f = residual_f32; p.idil = idil;
    p.y = y; p.yf = y_f32; p.partial = (float*)workspace; p.zeros = zeros;
    p.N = N; p.H = H; p.W = W; p.C1 = C1; p.C2 = C2; p.OH = OH; p.OW = OW; p.Cout = Cout;
    p.KH = KH; p.KW = KW; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.ups = ups ? 1 : 0;
    p.stats = nullptr;
    p.HL = ups ? 2 * H : (idil == 2 ? 2 * H - 1 : H);
    p.WL = ups ? 2 * W : (idil == 2 ? 2 * W - 1 : W);
    p.M = N * OH * OW;
    p.K = KH * KW * (C1 + C2);
    p.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
    p.fd_ow = make_fastdiv_h((uint32_t)OW);
    p.fd_oh = make_fastdiv_h((uint32_t)OH);
    int bm, bn, tiles, s;
    conv_plan_h(p.M, Cout, p.K, ept, &bm, &bn, &tiles, &s);
    if (s > 1 && (size_t)s * p.M * Cout * sizeof(float) > workspace_bytes) return V2A_ERR_WORKSPACE;
    p.splitk = s;
    p.ktiles_per_split = cdiv(p.K / ept, s);
    p.frame_tiles = 0;
    if (KH == 3 && KW == 1 && sh == 1 && sw == 1 && ph == 1 && pw == 0 && !ups && idil == 1 && OH == H && OW == W &&
        OH > 1 && OW % bm == 0 && s == 1)
        p.frame_tiles = OW / bm;
    if constexpr (sizeof(T) == 4) {
        // temporal (3 x 1) conv of the factorised Conv3d over [B, F = 7, HW, C] in the three-plane mode: the frame-stack kernel
        // (csrc/igemm_x3t.hip; no split-K, statistics in its own epilogue)
        f32_conv_mode_init();
        if (g_f32x3 && xpitch == 0 && KH == 3 && KW == 1 && sh == 1 && sw == 1 && ph == 1 && pw == 0 && !ups && idil == 1 && !x2 && C2 == 0 &&
            OH == H && OW == W && y && conv_frames_x3_eligible(N, H, W, C1, Cout, p.rows_per_batch, rowvec != nullptr) &&
            (((uintptr_t)bias | (uintptr_t)rowvec) & 3) == 0)
            return conv_frames_x3_launch((const float*)x, (const float*)w_packed, bias, rowvec, (const float*)residual, (float*)y, zeros, N, H, W,
                                         C1, Cout, p.rows_per_batch, stats, stream);
    }
    // fused GroupNorm statistics need the single-pass epilogue, an output in the storage type and whole 8-channel vectors
    if (stats) {
        if (s > 1 || !y || Cout % 8 || bm != 128) return V2A_ERR_ARG;
        p.stats = stats;
    }
    p.split_xcd = 0;
    p.xp1 = xpitch > 0 ? xpitch : C1;
    if (xpitch > 0) {                                // channel windows: the three-plane kernel's plain loader only
        if (sizeof(T) != 4 || x2 || C2 || ups || idil != 1 || xpitch > C1 || xpitch % 4) return V2A_ERR_ARG;
        f32_conv_mode_init();
        if (!g_f32x3) return V2A_ERR_ARG;
    }
    if constexpr (sizeof(T) == 4) {
        f32_conv_mode_init();
        // 3x3 / stride 1 / pad 1 on maps of 16 x 16 patches with >= 128 output channels: the phase-structured patch kernel (csrc/igemm_x3p.hip)
        if (g_f32x3 && xpitch == 0 && KH == 3 && KW == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 && idil == 1 && !x2 && C2 == 0 &&
            (ups ? (OH == 2 * H && OW == 2 * W) : (OH == H && OW == W)) && !stats && !rowvec && y &&
            v2a_conv2d_x3p_eligible(N, OH, OW, C1, Cout))
            return conv_patch_x3_launch((const float*)x, (const float*)w_packed, bias, (const float*)residual, (float*)y, zeros, N, OH, OW, C1,
                                        Cout, ups, stream);
        // ... on small square maps (the policy's ResNet-18 encoders at batch 64): 256-row tiles, phases of 36 MFMAs (csrc/igemm_x3m.hip)
        if (g_f32x3 && g_maps_on && xpitch == 0 && KH == 3 && KW == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 && idil == 1 && !x2 && C2 == 0 &&
            !ups && OH == H && OW == W && OH == OW && !stats && !rowvec && y && conv_maps_x3_eligible(N, OW, C1, Cout)) {
            s = conv_maps_x3_split(p.M, Cout, C1);
            if (s > 1 && (size_t)s * p.M * Cout * sizeof(float) > workspace_bytes) return V2A_ERR_WORKSPACE;
            p.splitk = s;
            p.ktiles_per_split = cdiv(C1 / 32, s);
            p.frame_tiles = 0;
            const int rc = conv_maps_x3_launch((const float*)x, (const float*)w_packed, bias, (const float*)residual, (float*)y, p.partial, zeros,
                                               N, OW, C1, Cout, s, stream);
            if (rc != V2A_OK) return rc;
            goto launched;
        }
        const bool hx_square = OH == OW && (OW == 4 || OW == 8 || OW == 16 || OW == 32 || OW == 64);
        const bool hx_patch = !hx_square && OH % 8 == 0 && OW % 16 == 0;
        if (g_f32x3 && xpitch == 0 && KH == 3 && KW == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 && idil == 1 &&
            !x2 && C2 == 0 && (ups ? (OH == 2 * H && OW == 2 * W) : (OH == H && OW == W)) && (hx_square || hx_patch) && p.M % 128 == 0 &&
            Cout % 64 == 0 && !stats && (double)N * H * W * C1 < 2147483648.0) {
            // 3x3 / stride 1 / pad 1 (optionally behind a nearest x2 upsample): the halo kernel (its own split, over 32-channel chunks) --
            // whole rows of the small square maps (the policy's encoders, the UNet's inner levels), 8 x 16 patches of anything larger
            s = conv_halo_x3_split(p.M, Cout, C1);
            if (s > 1 && (size_t)s * p.M * Cout * sizeof(float) > workspace_bytes) return V2A_ERR_WORKSPACE;
            p.splitk = s;
            p.ktiles_per_split = cdiv(C1 / 32, s);
            p.frame_tiles = 0;
            const dim3 grid((p.M / 128) * (Cout / 64), s);
            if (hx_patch) hipLaunchKernelGGL(conv_halo_x3<0>, grid, dim3(256), 0, stream, p);
            else if (OW == 64) hipLaunchKernelGGL(conv_halo_x3<64>, grid, dim3(256), 0, stream, p);
            else if (OW == 32) hipLaunchKernelGGL(conv_halo_x3<32>, grid, dim3(256), 0, stream, p);
            else if (OW == 16) hipLaunchKernelGGL(conv_halo_x3<16>, grid, dim3(256), 0, stream, p);
            else if (OW == 8) hipLaunchKernelGGL(conv_halo_x3<8>, grid, dim3(256), 0, stream, p);
            else hipLaunchKernelGGL(conv_halo_x3<4>, grid, dim3(256), 0, stream, p);
            goto launched;
        }
        if (g_f32x3) {
            // tiles: 64 x 64 (small problems, the plan's split), 128 x 64 / 256 x 64 for 64-wide layers, 128 x 128 on 8 waves otherwise --
            // the wider the tile, the fewer fp32 -> plane conversions and LDS bytes per MFMA
            const bool gen = ups || idil == 2;
            p.split_xcd = (s >= 8 && s % 8 == 0 && p.frame_tiles == 0) ? s / 8 : 0;
            // zero-interleaved input (data gradient of a stride-2 conv, transposed conv): class-major tile rows, live taps only
            p.pcls = 0;
            if (idil == 2 && !x2 && C2 == 0 && (OH == 1 || OH % 2 == 0) && OW % 2 == 0 && !stats && g_pcls_on) {
                const int nch = OH > 1 ? 2 : 1, ncw = 2;
                const long R = (long)N * (OH / nch) * (OW / ncw);
                const int bm_eff = (bm == 64) ? 64 : ((bn == 64 && cdiv(p.M, 256) * cdiv(Cout, 64) >= 200 && p.M % 256 == 0) ? 256 : 128);
                if (R % bm_eff == 0 && R < 2147483647L) {
                    p.pcls = 1; p.nch = nch; p.ncw = ncw; p.chh = OH / nch; p.cwh = OW / ncw; p.cls_R = (int)R;
                    p.fd_cw = make_fastdiv_h((uint32_t)p.cwh);
                    p.fd_chw = make_fastdiv_h((uint32_t)(p.chh * p.cwh));
                }
            }
            // short same-size 1-d convs (ConditionalUnet1D: k = 5 over T = 4 / 8 / 16 with pad 2): 30 / 15 / 7.5 % of the tap x position
            // products meet the zero padding -- one class per output column, live taps only
            if (!p.pcls && !gen && g_pcls_on && KH == 1 && KW >= 3 && sh == 1 && sw == 1 && ph == 0 && pw > 0 && OH == 1 && H == 1 && OW == W &&
                OW <= 16 && !stats && xpitch == 0 && p.frame_tiles == 0) {
                const long R = (long)N;
                const int bm_eff = (bm == 64) ? 64 : ((bn == 64 && cdiv(p.M, 256) * cdiv(Cout, 64) >= 200 && p.M % 256 == 0) ? 256 : 128);
                if (R % bm_eff == 0) {
                    p.pcls = 2; p.nch = 1; p.ncw = OW; p.chh = 1; p.cwh = 1; p.cls_R = (int)R;
                    p.fd_cw = make_fastdiv_h(1u);
                    p.fd_chw = make_fastdiv_h(1u);
                }
            }
#define V2A_X3_LAUNCH(BM_, BN_, WM_, WN_, G_)                                                                                            \
    do {                                                                                                                                   \
        const dim3 grid_ = p.split_xcd > 0 ? dim3((G_) * s, 1) : dim3(G_, s);                                                              \
        if (gen) hipLaunchKernelGGL((conv_igemm_f32x3<BM_, BN_, WM_, WN_, 2, true>), grid_, dim3(64 * WM_ * WN_), 0, stream, p);           \
        else hipLaunchKernelGGL((conv_igemm_f32x3<BM_, BN_, WM_, WN_, 2, false>), grid_, dim3(64 * WM_ * WN_), 0, stream, p);             \
    } while (0)
            if (bm == 64) V2A_X3_LAUNCH(64, 64, 2, 2, tiles);
            else if (bn == 64) {
                const int t256 = cdiv(p.M, 256) * cdiv(Cout, 64);
                if (t256 >= 200 && p.M % 256 == 0 && p.frame_tiles == 0) V2A_X3_LAUNCH(256, 64, 4, 2, t256);
                else V2A_X3_LAUNCH(128, 64, 2, 2, tiles);
            } else V2A_X3_LAUNCH(128, 128, 2, 4, tiles);
#undef V2A_X3_LAUNCH
            goto launched;
        }
        {   // exact-f32 MFMA mode (V2A_F32_CONV=exact): the pipelined kernel, bit-equal to an fmaf chain per accumulator
            const bool gen = ups || idil == 2;
#define V2A_F32P_LAUNCH(BM_, BN_, S_, W_)                                                                                       \
    do {                                                                                                                          \
        if (gen) hipLaunchKernelGGL((conv_igemm_f32p<BM_, BN_, S_, W_, true>), dim3(tiles, s), dim3(256), 0, stream, p);          \
        else hipLaunchKernelGGL((conv_igemm_f32p<BM_, BN_, S_, W_, false>), dim3(tiles, s), dim3(256), 0, stream, p);            \
    } while (0)
            if (bm == 64) V2A_F32P_LAUNCH(64, 64, 4, 2);
            else if (bn == 64) V2A_F32P_LAUNCH(128, 64, 3, 2);
            else V2A_F32P_LAUNCH(128, 128, 2, 2);
#undef V2A_F32P_LAUNCH
            goto launched;
        }
    } else {
        // 16-bit storage: one LDS buffer (4 workgroups per CU) on launches of >= 768 workgroups, two buffers below
        if (bm == 64) hipLaunchKernelGGL((conv_igemm_h<64, 64, T, 2>), dim3(tiles, s), dim3(256), 0, stream, p);
        else if (bn == 64) hipLaunchKernelGGL((conv_igemm_h<128, 64, T, 2>), dim3(tiles, s), dim3(256), 0, stream, p);
        else if (tiles * s >= 768) hipLaunchKernelGGL((conv_igemm_h<128, 128, T, 1>), dim3(tiles, s), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((conv_igemm_h<128, 128, T, 2>), dim3(tiles, s), dim3(256), 0, stream, p);
    }
launched:
    V2A_CHECK_LAUNCH();
    if (s > 1) {
        if (nslab_out && !rowvec && !residual_f32) {       // the consumer (a GroupNorm launch) sums the slabs itself: no reduce launch
            *nslab_out = s;
            return V2A_OK;
        }
        const size_t total = (size_t)p.M * Cout;
        int g = (int)((total + 255) / 256);
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(conv_splitk_reduce_h<T>, dim3(g), dim3(256), 0, stream, p);
        V2A_CHECK_LAUNCH();
    }
    return V2A_OK;
}

extern "C" {

// 16-bit format of every `_h` entry point below and of csrc/igemm_h2.hip, igemm_h3.hip, norm_h.hip, the 16-bit attention: 0 = bf16
// (default), 1 = IEEE fp16 (v_mfma_f32_32x32x16_f16, v_cvt_f16_f32).  Returns the old value.  Process-wide: the Python wrappers set it
// from the dtype of the tensors they are handed before every call.
int v2a_set_half_format(int f16) {
    const int old = g_v2a_half_f16;
    if (f16 == 0 || f16 == 1) g_v2a_half_f16 = f16;
    return old;
}
int v2a_get_half_format(void) { return g_v2a_half_f16; }
// fp32 LDS-DMA convs: 1 = by three bf16 planes (conv_igemm_f32x3, default), 0 = exact-f32 MFMA kernels.  Returns the old value.
static void f32_conv_mode_init() {
    if (g_f32x3 < 0) {
        const char* e = getenv("V2A_F32_CONV");
        g_f32x3 = (e && e[0] == 'e') ? 0 : 1;
    }
}
// 1 when v2a_conv2d_fwd_dma_f32 / _d run this 3 x 3 / stride 1 / pad 1 conv over N square S x S maps on conv_maps_x3 (three-plane mode)
int v2a_conv2d_x3m_eligible(int N, int S, int C, int Cout) { return g_maps_on ? conv_maps_x3_eligible(N, S, C, Cout) : 0; }
int v2a_debug_set_maps_kernel(int on) {         // returns the old value; 0: the small square maps stay on conv_halo_x3 (round-5 form)
    const int old = g_maps_on;
    g_maps_on = on ? 1 : 0;
    return old;
}
int v2a_debug_set_parity_classes(int on) {      // returns the old value; 0: every tap of a zero-interleaved input is multiplied (round-5 form)
    const int old = g_pcls_on;
    g_pcls_on = on ? 1 : 0;
    return old;
}
int v2a_set_f32_conv_mode(int x3) {
    f32_conv_mode_init();
    const int old = g_f32x3;
    g_f32x3 = x3 ? 1 : 0;
    return old;
}
int v2a_get_f32_conv_mode(void) {
    f32_conv_mode_init();
    return g_f32x3;
}
// bf16-storage convolution forward.  x / x2 / residual / y: bf16; w_packed: bf16 [Cout][KH][KW][C1+C2]; bias / rowvec: fp32;
// exactly one of y (bf16) / y_f32 is non-null.  Requires C1 % 64 == 0, C2 % 64 == 0, 16-B aligned pointers; zeros: >= 128 zero bytes.
int v2a_conv2d_fwd_h(const void* x, const void* x2, const void* w_packed, const float* bias, const float* rowvec, const void* residual,
                     const float* residual_f32, void* y, float* y_f32, const void* zeros, int N, int H, int W, int C1, int C2, int Cout,
                     int KH, int KW, int sh, int sw, int ph, int pw, int ups, int idil, int OH, int OW, int rows_per_batch,
                     float* stats, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!y == !y_f32) return V2A_ERR_ARG;
    if (g_v2a_half_f16)
        return conv_dma_launch<f16s>(x, x2, w_packed, bias, rowvec, residual, residual_f32, y, y_f32, zeros, N, H, W, C1, C2, Cout, KH, KW,
                                     sh, sw, ph, pw, ups, idil, OH, OW, rows_per_batch, stats, workspace, workspace_bytes, stream);
    return conv_dma_launch<uint16_t>(x, x2, w_packed, bias, rowvec, residual, residual_f32, y, y_f32, zeros, N, H, W, C1, C2, Cout, KH, KW,
                                     sh, sw, ph, pw, ups, idil, OH, OW, rows_per_batch, stats, workspace, workspace_bytes, stream);
}

// v2a_conv2d_fwd_h with an fp32 output whose split-K reduce is left to the consuming GroupNorm launch (see v2a_conv2d_fwd_dma_f32_d)
int v2a_conv2d_fwd_h_d(const void* x, const void* x2, const void* w_packed, const float* bias, const void* residual_unused,
                       const float* residual_f32, float* y_f32, const void* zeros, int N, int H, int W, int C1, int C2, int Cout, int KH,
                       int KW, int sh, int sw, int ph, int pw, int ups, int idil, int OH, int OW, int rows_per_batch, int* nslab_out,
                       void* workspace, size_t workspace_bytes, hipStream_t stream) {
    (void)residual_unused;
    if (!y_f32 || !nslab_out) return V2A_ERR_ARG;
    // the fp32 residual of the epilogue travels with the slabs (the consumer adds it): pass it through the `residual` slot of the
    // reduce-less path by clearing it here when the plan splits -- conv_dma_launch only defers when residual_f32 is null
    int ns = 0;
    const int rc = g_v2a_half_f16
        ? conv_dma_launch<f16s>(x, x2, w_packed, bias, nullptr, nullptr, nullptr, nullptr, y_f32, zeros, N, H, W, C1, C2, Cout, KH, KW, sh, sw,
                                ph, pw, ups, idil, OH, OW, rows_per_batch, nullptr, workspace, workspace_bytes, stream, &ns)
        : conv_dma_launch<uint16_t>(x, x2, w_packed, bias, nullptr, nullptr, nullptr, nullptr, y_f32, zeros, N, H, W, C1, C2, Cout, KH, KW, sh,
                                    sw, ph, pw, ups, idil, OH, OW, rows_per_batch, nullptr, workspace, workspace_bytes, stream, &ns);
    *nslab_out = ns;
    if (rc != V2A_OK || ns > 0 || !residual_f32) return rc;
    return V2A_ERR_ARG;      // unsplit plan with a residual: the caller must use v2a_conv2d_fwd_h (it asked v2a_conv2d_h_splits first)
}
// number of split-K slices v2a_conv2d_fwd_h / _d will use for this problem (1 = the conv finishes its output itself)
int v2a_conv2d_h_splits(int M, int Cout, int K) {
    int bm, bn, tiles, s;
    conv_plan_h(M, Cout, K, 64, &bm, &bn, &tiles, &s);
    return s;
}

// The same LDS-DMA kernel over fp32 tensors with the exact-f32 MFMA (v_mfma_f32_32x32x2_f32): the parity configuration's
// convolution for layers whose channel counts are multiples of 32 (k tile = 32 floats = one 128-B line).  x / x2 / residual / y /
// w_packed ([Cout][KH][KW][C1+C2]) fp32.
int v2a_conv2d_fwd_dma_f32(const float* x, const float* x2, const float* w_packed, const float* bias, const float* rowvec,
                           const float* residual, float* y, const void* zeros, int N, int H, int W, int C1, int C2, int Cout, int KH, int KW,
                           int sh, int sw, int ph, int pw, int ups, int idil, int OH, int OW, int rows_per_batch, float* stats,
                           void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!y) return V2A_ERR_ARG;
    return conv_dma_launch<float>(x, x2, w_packed, bias, rowvec, residual, nullptr, y, nullptr, zeros, N, H, W, C1, C2, Cout, KH, KW, sh, sw,
                                  ph, pw, ups, idil, OH, OW, rows_per_batch, stats, workspace, workspace_bytes, stream);
}

// "Channel window" form of the conv above for few-channel inputs (the RGB stem of the policy's ResNet-18 encoders,
// diffuser/diffusion_policy/common/vision_nets.py:29-39 -> torchvision resnet18 conv1 7x7 / stride 2): logical pixel (ih, iw) is the
// C-channel window that starts at x + ((n * H + ih) * W + iw) * xpitch with xpitch <= C (windows overlap).  With the image stored
// zero-padded as [N, Hp, Wp, 4] (RGB + a zero channel), a 7 x 8-pixel patch row is 32 contiguous floats: the 7x7x3 filter padded
// to [Cout][7][8][4] is a KH = 7, KW = 1, C = 32 conv over windows of pitch sw * 4 floats -- every k tile one aligned 128-B line per
// output pixel, on the vector loader of the three-plane kernel (the scalar-gather kernel this replaces: 166 us per encoder).
// No padding (the caller's buffer carries the zero border), fp32 three-plane mode only (V2A_ERR_ARG otherwise).
int v2a_conv2d_fwd_window_f32(const float* x, const float* w_packed, const float* bias, float* y, const void* zeros, int N, int H, int W,
                              int xpitch, int C, int Cout, int KH, int KW, int sh, int sw, int OH, int OW, void* workspace,
                              size_t workspace_bytes, hipStream_t stream) {
    if (!y || xpitch <= 0) return V2A_ERR_ARG;
    if ((OH - 1) * sh + KH > H || ((OW - 1) * sw + KW - 1) * xpitch + C > W * xpitch) return V2A_ERR_ARG;      // a window past its image row
    return conv_dma_launch<float>(x, nullptr, w_packed, bias, nullptr, nullptr, nullptr, y, nullptr, zeros, N, H, W, C, 0, Cout, KH, KW, sh,
                                  sw, 0, 0, 0, 1, OH, OW, 1, nullptr, workspace, workspace_bytes, stream, nullptr, xpitch);
}

// The same conv with the split-K reduce left to the consumer: when the plan splits K, the fp32 slabs [nslab][M][Cout] stay in
// `workspace` (bias and residual NOT applied, y untouched) and *nslab_out = their number; a GroupNorm launch then sums them
// (v2a_groupnorm_fwd_s / _bwd_s).  *nslab_out = 0: the conv finished y itself (no split, or a rowvec epilogue).
int v2a_conv2d_fwd_dma_f32_d(const float* x, const float* x2, const float* w_packed, const float* bias, const float* rowvec,
                             const float* residual, float* y, const void* zeros, int N, int H, int W, int C1, int C2, int Cout, int KH, int KW,
                             int sh, int sw, int ph, int pw, int ups, int idil, int OH, int OW, int rows_per_batch, int* nslab_out,
                             void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!y || !nslab_out) return V2A_ERR_ARG;
    return conv_dma_launch<float>(x, x2, w_packed, bias, rowvec, residual, nullptr, y, nullptr, zeros, N, H, W, C1, C2, Cout, KH, KW, sh, sw,
                                  ph, pw, ups, idil, OH, OW, rows_per_batch, nullptr, workspace, workspace_bytes, stream, nslab_out);
}

// 1 when v2a_conv2d_fwd_p3 takes this problem: channel counts in 32N and a plan on 64 x 64 tiles (the small-M GEMMs the kernel is for)
int v2a_conv2d_p3_eligible(int M, int Cout, int K, int C1, int C2) {
    if (M <= 0 || Cout <= 0 || C1 <= 0 || C1 % 32 || C2 < 0 || C2 % 32 || K % 32) return 0;
    int bm, bn, tiles, s;
    conv_plan_h(M, Cout, K, 32, &bm, &bn, &tiles, &s);
    return bm == 64 ? 1 : 0;
}
// fp32 conv (three-plane products, the arithmetic of v2a_conv2d_fwd_dma_f32 in its default mode) whose operands are ALREADY split into
// bf16 planes: x3 / x2_3 / w3 point at the hi plane of x [N,H,W,C1] / x2 [N,H,W,C2] / the packed weight [Cout][KH][KW][C1+C2]; the mid
// and lo planes follow `*_plane_stride` ELEMENTS apart.  Who writes them: v2a_groupnorm_fwd_s / _bwd_s (yh_plane_stride), the
// optimiser's update kernel (v2a_opt_step_packed, twin format 2), v2a_pack_weights_multi (mode bit 8), v2a_split3_f32.  Same split plan,
// workspace (v2a_conv2d_dma_f32_workspace_bytes) and results -- bit for bit -- as v2a_conv2d_fwd_dma_f32_d on the unsplit tensors;
// nslab_out as there (null: the reduce launch runs here).  Only where v2a_conv2d_p3_eligible says 1; stride >= 1, no upsample / dilation.
int v2a_conv2d_fwd_p3(const void* x3, size_t x_plane_stride, const void* x2_3, size_t x2_plane_stride, const void* w3, size_t w_plane_stride,
                      const float* bias, const float* residual, float* y, const void* zeros, int N, int H, int W, int C1, int C2, int Cout,
                      int KH, int KW, int sh, int sw, int ph, int pw, int OH, int OW, int* nslab_out, void* workspace, size_t workspace_bytes,
                      hipStream_t stream) {
    if (nslab_out) *nslab_out = 0;
    if (!x3 || !w3 || !y || !zeros || N <= 0 || (C2 > 0 && !x2_3) || KH <= 0 || KW <= 0 || sh <= 0 || sw <= 0) return V2A_ERR_ARG;
    const int M = N * OH * OW, K = KH * KW * (C1 + C2);
    if (!v2a_conv2d_p3_eligible(M, Cout, K, C1, C2)) return V2A_ERR_ARG;
    if ((((uintptr_t)x3 | (uintptr_t)x2_3 | (uintptr_t)w3 | (uintptr_t)zeros | (uintptr_t)y | (uintptr_t)residual) & 15) != 0) return V2A_ERR_ARG;
    if (((x_plane_stride | x2_plane_stride | w_plane_stride) & 7) != 0) return V2A_ERR_ARG;      // planes 16-B aligned
    if (x_plane_stride == 0 || w_plane_stride == 0 || (C2 > 0 && x2_plane_stride == 0)) return V2A_ERR_ARG;
    if ((double)N * H * W * (C1 > C2 ? C1 : C2) >= 2147483648.0) return V2A_ERR_ARG;
    ConvDescH p = {};
    p.x = x3; p.x2 = x2_3; p.w = w3; p.xps = x_plane_stride; p.x2ps = x2_plane_stride; p.wps = w_plane_stride;
    p.bias = bias; p.residual = residual; p.y = y; p.partial = (float*)workspace; p.zeros = zeros;
    p.N = N; p.H = H; p.W = W; p.C1 = C1; p.C2 = C2; p.OH = OH; p.OW = OW; p.Cout = Cout;
    p.KH = KH; p.KW = KW; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.idil = 1;
    p.HL = H; p.WL = W; p.M = M; p.K = K; p.rows_per_batch = 1;
    p.fd_ow = make_fastdiv_h((uint32_t)OW);
    p.fd_oh = make_fastdiv_h((uint32_t)OH);
    p.xp1 = C1;
    int bm, bn, tiles, s;
    conv_plan_h(M, Cout, K, 32, &bm, &bn, &tiles, &s);
    if (s > 1 && (size_t)s * M * Cout * sizeof(float) > workspace_bytes) return V2A_ERR_WORKSPACE;
    p.splitk = s;
    p.ktiles_per_split = cdiv(K / 32, s);
    p.split_xcd = (s >= 8 && s % 8 == 0) ? s / 8 : 0;
    const dim3 grid = p.split_xcd > 0 ? dim3(tiles * s, 1) : dim3(tiles, s);
    // 64 x 64 tiles, two LDS stages of 24 KB (three workgroups per CU): the best of the shapes / stage counts measured on the ConditionalUnet1D
    // GEMMs (tools/probes/r5/conv_p3_tiles.py: 128 x 64, 64 x 128, 128 x 128 on four and eight waves, 3 ... 6 stages, a pre-tiled weight
    // layout -- all within 10 %: what binds these launches is the LDS-DMA issue rate per wave, ~1 KB per ~300 cycles, not the tile)
    hipLaunchKernelGGL((conv_p3<64, 64, 2>), grid, dim3(256), 0, stream, p);
    V2A_CHECK_LAUNCH();
    if (s > 1) {
        if (nslab_out) { *nslab_out = s; return V2A_OK; }
        const size_t total = (size_t)M * Cout;
        int g = (int)((total + 255) / 256);
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(conv_splitk_reduce_h<float>, dim3(g), dim3(256), 0, stream, p);
        V2A_CHECK_LAUNCH();
    }
    return V2A_OK;
}
// x fp32 [n] -> three bf16 planes hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) at y3, y3 + plane_stride, y3 + 2 plane_stride
// (elements): the operand format of v2a_conv2d_fwd_p3, for tensors no fused producer writes (n % 4 == 0, 16-B / 8-B aligned)
int v2a_split3_f32(const float* x, void* y3, size_t n, size_t plane_stride, hipStream_t stream) {
    if (!x || !y3 || n % 4 || plane_stride < n || (plane_stride & 3) || (((uintptr_t)x & 15) | ((uintptr_t)y3 & 7))) return V2A_ERR_ARG;
    if (n == 0) return V2A_OK;
    int g = (int)((n / 4 + 255) / 256);
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(split3_f32_kernel, dim3(g), dim3(256), 0, stream, x, (uint16_t*)y3, n / 4, plane_stride);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

// torch-layout fp32 weight [Cout][Cin][taps] -> bf16 [Cout][taps][Cin]
int v2a_pack_weight_h(const float* w, void* out, int Cout, int Cin, int taps, hipStream_t stream) {
    if (!w || !out || Cout <= 0 || Cin <= 0 || taps <= 0) return V2A_ERR_ARG;
    const size_t total = (size_t)Cout * Cin * taps;
    int g = (int)((total + 255) / 256);
    if (g > 8192) g = 8192;
    if (g_v2a_half_f16) hipLaunchKernelGGL(pack_weight_h_kernel<true>, dim3(g), dim3(256), 0, stream, w, (uint16_t*)out, Cout, Cin, taps);
    else hipLaunchKernelGGL(pack_weight_h_kernel<false>, dim3(g), dim3(256), 0, stream, w, (uint16_t*)out, Cout, Cin, taps);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

// elementwise casts between the two storage types (n % 4 == 0, 16-B / 8-B aligned).  The 16-bit format is an ARGUMENT here
// (f16: 0 bf16, 1 IEEE fp16), never the process-wide flag: callers that stage bf16 data (the data-parallel wire buffer, the policy's
// weight twins) must not inherit the format an fp16 video forward selected before them.
int v2a_cast_f32_h(const float* x, void* y, size_t n, int f16, hipStream_t stream) {
    if (!x || !y || n % 4 || (f16 != 0 && f16 != 1)) return V2A_ERR_ARG;
    if (n == 0) return V2A_OK;
    int g = (int)((n / 4 + 255) / 256);
    if (g > 16384) g = 16384;
    if (f16) hipLaunchKernelGGL(cast_f32_bf16_kernel<true>, dim3(g), dim3(256), 0, stream, x, (uint16_t*)y, n / 4);
    else hipLaunchKernelGGL(cast_f32_bf16_kernel<false>, dim3(g), dim3(256), 0, stream, x, (uint16_t*)y, n / 4);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
// fp32 -> bf16, always (whatever v2a_set_half_format says)
int v2a_cast_f32_bf16(const float* x, void* y, size_t n, hipStream_t stream) { return v2a_cast_f32_h(x, y, n, 0, stream); }
// x fp32 [M][Cin] -> y bf16 [M][Cpad], zero-padded channels (Cpad % 8 == 0, Cin <= Cpad, y 16-B aligned)
int v2a_pad_cast_f32_bf16(const float* x, void* y, size_t M, int Cin, int Cpad, hipStream_t stream) {
    if (!x || !y || Cin <= 0 || Cpad % 8 || Cin > Cpad || ((uintptr_t)y & 15)) return V2A_ERR_ARG;
    if (M == 0) return V2A_OK;
    const size_t total = M * (size_t)(Cpad / 8);
    int g = (int)((total + 255) / 256);
    if (g > 16384) g = 16384;
    if (g_v2a_half_f16) hipLaunchKernelGGL(pad_cast_f32_bf16_kernel<true>, dim3(g), dim3(256), 0, stream, x, (uint16_t*)y, M, Cin, Cpad);
    else hipLaunchKernelGGL(pad_cast_f32_bf16_kernel<false>, dim3(g), dim3(256), 0, stream, x, (uint16_t*)y, M, Cin, Cpad);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_cast_h_f32(const void* x, float* y, size_t n, int f16, hipStream_t stream) {
    if (!x || !y || n % 4 || (f16 != 0 && f16 != 1)) return V2A_ERR_ARG;
    if (n == 0) return V2A_OK;
    int g = (int)((n / 4 + 255) / 256);
    if (g > 16384) g = 16384;
    if (f16) hipLaunchKernelGGL(cast_bf16_f32_kernel<true>, dim3(g), dim3(256), 0, stream, (const uint16_t*)x, y, n / 4);
    else hipLaunchKernelGGL(cast_bf16_f32_kernel<false>, dim3(g), dim3(256), 0, stream, (const uint16_t*)x, y, n / 4);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
// bf16 -> fp32, always
int v2a_cast_bf16_f32(const void* x, float* y, size_t n, hipStream_t stream) { return v2a_cast_h_f32(x, y, n, 0, stream); }

}  // extern "C"
