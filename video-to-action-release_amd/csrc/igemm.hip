// Implicit-GEMM convolution family for gfx950 (fp32 storage, exact-f32 MFMA v_mfma_f32_32x32x2_f32).
//
// One kernel covers every contraction on the hot path:
//   * Conv3d spatial part   (3x3 / 1x1, stride 1|2) on [(B F), H, W, C]        reference nn.py:45,64-69
//   * Conv3d temporal part  (k=3 over frames, zero pad 1+1) viewed as a (3x1) conv on [B, F, (H W), C]   nn.py:46-51,76-85
//   * nearest x2 upsample folded into the loader (no 4x tensor)                reference unet.py:105-115
//   * channel concat of the decoder skip folded into the loader (two sources)  reference unet.py:681
//   * ResNet-18 convs (7x7 s2, 3x3, 1x1 s2), Conv1d k5/k3/k1 (H=1), Linear (1x1)
//   * data-gradient of all of those (same kernel: flipped/transposed weight pack + input dilation)
//   * ConvTranspose1d (k4 s2 p1) as an input-dilated conv
// and a sibling kernel computes weight gradients (dW = dY^T x im2col(X)).
//
// Layout: activations channels-last [N, H, W, C] fp32; weights packed [Cout][KH][KW][Cin] (K contiguous).
// GEMM view: Y[m, n] = sum_k A[m, k] W[n, k],  m = (img, oh, ow), k = (kh, kw, ci), n = cout.
// Tile: BM x BN x 16, 256 threads = 4 waves (2 x 2), each wave (BM/2) x (BN/2) as 32x32 MFMA tiles; LDS tiles
// are k-major ([16][BM+4]) so that the MFMA operand read (lane -> row, k = lane>>5) is bank-conflict free;
// global loads are float4 along the contiguous channel axis; register-staged double buffering.
#include "common.h"
#include <stdlib.h>

#define BK 16
#define LDS_PAD 4
#define LDK 20

// division by a launch-invariant divisor without the ~40-instruction integer divide (libdivide branch-free form)
struct FastDiv {
    uint32_t d, m, s;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    if (d <= 1) { f.m = 0; f.s = 0; return f; }
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;
    f.s = s;
    f.m = (uint32_t)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
    if (f.d <= 1) return n;
    const uint32_t t = __umulhi(f.m, n);
    return (t + ((n - t) >> 1)) >> (f.s - 1);
}

struct ConvDesc {
    const float* x;         // source 1: [N, H, W, C1]
    const float* x2;        // source 2 (concat along C): [N, H, W, C2] or null
    const float* w;         // [Cout][KH*KW*(C1+C2)]
    const float* bias;      // [Cout] or null
    const float* rowvec;    // [N / rows_per_batch ...][Cout] additive per (batch, channel) or null
    const float* residual;  // [M][Cout] or null
    float* y;               // [M][Cout]   (channels [0, csplit) when y2 != null: row stride csplit)
    float* y2;              // optional second output for channels [csplit, Cout): row stride Cout - csplit
    float* partial;         // split-K slabs [splitk][M][Cout]
    int N, H, W, C1, C2;
    int OH, OW, Cout;
    int KH, KW, sh, sw, ph, pw;
    int idil;               // input dilation (transposed conv / strided data-gradient); 1 = none
    int ups;                // 1: nearest x2 upsample folded in (logical input = 2H x 2W)
    int HL, WL;             // logical input extent used for the bounds test
    int M, K;
    int rows_per_batch;     // rowvec row = m / rows_per_batch
    int splitk, ktiles_per_split;
    int csplit;
    int bmode;              // 0: w = [Cout][K] (K contiguous).  1: data-gradient straight from the FORWARD pack [Cred][taps][Cout]
};

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    // blocks are dispatched round-robin over the 8 XCDs; give each XCD a contiguous run of tiles so that the
    // tiles sharing an activation panel hit the same L2 (bijective for any nblk).
    int q = nblk >> 3, r = nblk & 7;
    int xcd = bid & 7, slot = bid >> 3;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

template <int BM, int BN, int BKT, bool VEC, bool BNMAJ, int PF>
__global__ __launch_bounds__(256) void conv_igemm_f32(const ConvDesc p) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int KC = BKT / 4;                 // float4 chunks per tile row
    constexpr int RPP = 256 / KC;               // tile rows loaded per pass of the 256 threads
    constexpr int AL = BM / RPP, BL = BN / RPP; // float4 loads per thread per tile
    constexpr int LDR = BKT + 4;                // LDS row stride (words): 16-B aligned, conflict-free b128 reads
    constexpr int NB_ROWS = BKT * 4 / BN > 0 ? 1024 / BN : 1;   // BNMAJ: k-rows per pass = 256 / (BN/4)
    constexpr int NBL = BKT / (1024 / BN);      // BNMAJ: float4 loads per thread per tile
    // row-major tiles [rows][BKT k + 4 pad]: both the b128 store of a loaded float4 and the b128 operand read
    // (16 lanes x 4 words = all 64 banks) are conflict-free; BKT = 32 makes every row segment a full 128-B line.
    __shared__ __attribute__((aligned(16))) float As[2][BM * LDR];
    // BNMAJ (data gradient read from the forward pack): B rows are k, columns n contiguous -> k-major tile, b128 store, b32 reads
    __shared__ __attribute__((aligned(16))) float Bs[2][BNMAJ ? BKT * (BN + LDS_PAD) : BN * LDR];
    (void)NB_ROWS;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = (p.Cout + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int ntile = tiles_m * tiles_n;
    const int lin = xcd_remap(blockIdx.x, ntile);
    const int m0 = (lin / tiles_n) * BM, n0 = (lin % tiles_n) * BN;
    const int split = blockIdx.y;
    const int Cin = p.C1 + p.C2;
    const int nkt = (p.K + BKT - 1) / BKT;
    const int kt_begin = split * p.ktiles_per_split;
    const int kt_end = min(nkt, kt_begin + p.ktiles_per_split);

    // ---- per-thread loader state
    const int lrow = tid / KC, chunk = tid % KC;
    int a_ihb[AL], a_iwb[AL], a_img[AL];
    bool a_ok[AL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        int m = m0 + lrow + i * RPP;
        a_ok[i] = m < p.M;
        int mm = a_ok[i] ? m : 0;
        int ow = mm % p.OW;
        int t = mm / p.OW;
        int oh = t % p.OH;
        a_img[i] = t / p.OH;
        a_ihb[i] = oh * p.sh - p.ph;
        a_iwb[i] = ow * p.sw - p.pw;
    }

    // PF register stages: tile t+1 .. t+PF are in flight while tile t is multiplied (weight-streaming layers with a handful of
    // k-tiles per block are bound by the HBM latency of ONE outstanding tile otherwise)
    f32x4 ra[PF][AL], rb[PF][BNMAJ ? NBL : BL];

    auto pix_of = [&](int i, int kh, int kw, bool& ok) -> size_t {
        int ih = a_ihb[i] + kh, iw = a_iwb[i] + kw;
        ok = a_ok[i] && ih >= 0 && ih < p.HL && iw >= 0 && iw < p.WL;
        if (p.idil > 1) {
            ok = ok && (ih % p.idil == 0) && (iw % p.idil == 0);
            ih /= p.idil;
            iw /= p.idil;
        }
        if (p.ups) { ih >>= 1; iw >>= 1; }
        return ((size_t)a_img[i] * p.H + ih) * p.W + iw;
    };

    auto load_tile = [&](int kt, f32x4* ra, f32x4* rb) {
        const int k0 = kt * BKT;
        if (VEC) {
            const int tap = k0 / Cin;
            const int c = k0 - tap * Cin + chunk * 4;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                bool ok;
                size_t pix = pix_of(i, kh, kw, ok);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) {
                    const float* src = (c < p.C1) ? p.x + pix * p.C1 + c : p.x2 + pix * p.C2 + (c - p.C1);
                    v = *reinterpret_cast<const f32x4*>(src);
                }
                ra[i] = v;
            }
            if (BNMAJ) {
                // rows kk = tid / (BN/4) + i * (1024/BN): k = k0 + kk = (tap, cred = cbase + kk); 4 consecutive n per thread
                const int taps = p.KH * p.KW;
                const int cbase = k0 - tap * Cin;
#pragma unroll
                for (int i = 0; i < NBL; ++i) {
                    const int kk = tid / (BN / 4) + i * (1024 / BN);
                    const int n = n0 + (tid % (BN / 4)) * 4;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (n < p.Cout)
                        v = *reinterpret_cast<const f32x4*>(p.w + ((size_t)(cbase + kk) * taps + (taps - 1 - tap)) * p.Cout + n);
                    rb[i] = v;
                }
            } else {
#pragma unroll
                for (int i = 0; i < BL; ++i) {
                    int n = n0 + lrow + i * RPP;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (n < p.Cout) v = *reinterpret_cast<const f32x4*>(p.w + (size_t)n * p.K + k0 + chunk * 4);
                    rb[i] = v;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int k = k0 + chunk * 4 + j;
                    if (k < p.K) {
                        int tap = k / Cin, c = k - tap * Cin;
                        int kh = tap / p.KW, kw = tap - kh * p.KW;
                        bool ok;
                        size_t pix = pix_of(i, kh, kw, ok);
                        if (ok) v[j] = (c < p.C1) ? p.x[pix * p.C1 + c] : p.x2[pix * p.C2 + (c - p.C1)];
                    }
                }
                ra[i] = v;
            }
#pragma unroll
            for (int i = 0; i < BL; ++i) {
                int n = n0 + lrow + i * RPP;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (n < p.Cout) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int k = k0 + chunk * 4 + j;
                        if (k < p.K) v[j] = p.w[(size_t)n * p.K + k];
                    }
                }
                rb[i] = v;
            }
        }
    };

    auto store_tile = [&](int buf, const f32x4* ra, const f32x4* rb) {
#pragma unroll
        for (int i = 0; i < AL; ++i) *reinterpret_cast<f32x4*>(&As[buf][(lrow + i * RPP) * LDR + chunk * 4]) = ra[i];
        if (BNMAJ) {
#pragma unroll
            for (int i = 0; i < NBL; ++i)
                *reinterpret_cast<f32x4*>(&Bs[buf][(tid / (BN / 4) + i * (1024 / BN)) * (BN + LDS_PAD) + (tid % (BN / 4)) * 4]) = rb[i];
        } else {
#pragma unroll
            for (int i = 0; i < BL; ++i) *reinterpret_cast<f32x4*>(&Bs[buf][(lrow + i * RPP) * LDR + chunk * 4]) = rb[i];
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

    // prologue: tiles kt_begin .. kt_begin+PF-1 into the register stages, the first one on to LDS
#pragma unroll
    for (int d = 0; d < PF; ++d)
        if (kt_begin + d < kt_end) load_tile(kt_begin + d, ra[d], rb[d]);
    if (kt_begin < kt_end) store_tile(0, ra[0], rb[0]);
    __syncthreads();
    int buf = 0;
    for (int kt0 = kt_begin; kt0 < kt_end; kt0 += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int kt = kt0 + d;
            if (kt < kt_end) {
                const bool more = (kt + 1) < kt_end;
                const bool refill = (kt + PF) < kt_end;       // stage d is free again: fetch tile kt + PF into it
                // one ds_read_b128 per operand tile feeds FOUR MFMA k-steps: in step (h, q) the lanes with lk = 0 supply k = 8h + q
                // and the lanes with lk = 1 supply k = 8h + 4 + q, for A and B alike (the k-order of an exact-f32 sum is free).
                // The refill's address arithmetic + global loads are issued after the first MFMA batch so they run under the
                // matrix pipe instead of in front of it.
#pragma unroll
                for (int h = 0; h < BKT / 8; ++h) {
                    if (h == 1 && refill) load_tile(kt + PF, ra[d], rb[d]);
                    f32x4 a[TM], b[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(&As[buf][(wm + i * 32 + lr) * LDR + 8 * h + 4 * lk]);
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (BNMAJ) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) b[j][q] = Bs[buf][(8 * h + 4 * lk + q) * (BN + LDS_PAD) + wn + j * 32 + lr];
                        } else {
                            b[j] = *reinterpret_cast<const f32x4*>(&Bs[buf][(wn + j * 32 + lr) * LDR + 8 * h + 4 * lk]);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
                }
                if (more) store_tile(buf ^ 1, ra[(d + 1) % PF], rb[(d + 1) % PF]);
                __syncthreads();
                buf ^= 1;
            }
        }
    }

    // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn + j * 32 + lr;
            if (n >= p.Cout) continue;
            const float bv = (p.bias && p.splitk == 1) ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m >= p.M) continue;
                float v = acc[i][j][r];
                if (p.splitk > 1) {
                    p.partial[((size_t)split * p.M + m) * p.Cout + n] = v;
                } else {
                    v += bv;
                    if (p.rowvec) v += p.rowvec[(size_t)(m / p.rows_per_batch) * p.Cout + n];
                    if (p.residual) v += p.residual[(size_t)m * p.Cout + n];
                    if (p.y2 && n >= p.csplit) p.y2[(size_t)m * (p.Cout - p.csplit) + (n - p.csplit)] = v;
                    else p.y[(size_t)m * (p.y2 ? p.csplit : p.Cout) + n] = v;
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------ bf16-MFMA variant
// Same implicit GEMM with v_mfma_f32_32x32x16_bf16 (16x the f32 matrix rate).  HBM storage stays fp32: operands are rounded
// to bf16 (RNE) while being staged into LDS, accumulation is fp32.  This is the PERFORMANCE configuration (the reference's GPU
// path runs fp16 autocast, lb_online_trainer_v7.py:72-76,593); the exact-f32 kernel above remains the parity configuration.
// LDS rows hold 32 bf16 (+8 pad) = 80 B: b64 stores of 4 converted values, b128 operand reads (8 consecutive k per lane).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    return v2a_pack_bf16x2(a, b);
}

template <int BM, int BN, bool F16>
__global__ __launch_bounds__(256) void conv_igemm_bf16(const ConvDesc p) {
    constexpr int BKT = 32, KC = 8, RPP = 32, LDH = 40;         // 40 halves = 80-B rows
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int AL = BM / RPP, BL = BN / RPP;
    __shared__ __attribute__((aligned(16))) uint16_t As[2][BM * LDH];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[2][BN * LDH];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = (p.Cout + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int lin = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (lin / tiles_n) * BM, n0 = (lin % tiles_n) * BN;
    const int split = blockIdx.y;
    const int Cin = p.C1 + p.C2;
    const int nkt = (p.K + BKT - 1) / BKT;
    const int kt_begin = split * p.ktiles_per_split;
    const int kt_end = min(nkt, kt_begin + p.ktiles_per_split);
    const int lrow = tid / KC, chunk = tid % KC;
    int a_ihb[AL], a_iwb[AL], a_img[AL];
    bool a_ok[AL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        int m = m0 + lrow + i * RPP;
        a_ok[i] = m < p.M;
        int mm = a_ok[i] ? m : 0;
        int ow = mm % p.OW;
        int t = mm / p.OW;
        int oh = t % p.OH;
        a_img[i] = t / p.OH;
        a_ihb[i] = oh * p.sh - p.ph;
        a_iwb[i] = ow * p.sw - p.pw;
    }
    f32x4 ra[AL], rb[BL];
    auto load_tile = [&](int kt) {
        const int k0 = kt * BKT;
        const int tap = k0 / Cin;
        const int c = k0 - tap * Cin + chunk * 4;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            int ih = a_ihb[i] + kh, iw = a_iwb[i] + kw;
            bool ok = a_ok[i] && ih >= 0 && ih < p.HL && iw >= 0 && iw < p.WL;
            if (p.idil > 1) {
                ok = ok && (ih % p.idil == 0) && (iw % p.idil == 0);
                ih /= p.idil;
                iw /= p.idil;
            }
            if (p.ups) { ih >>= 1; iw >>= 1; }
            const size_t pix = ((size_t)a_img[i] * p.H + ih) * p.W + iw;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                const float* src = (c < p.C1) ? p.x + pix * p.C1 + c : p.x2 + pix * p.C2 + (c - p.C1);
                v = *reinterpret_cast<const f32x4*>(src);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            int n = n0 + lrow + i * RPP;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n < p.Cout) v = *reinterpret_cast<const f32x4*>(p.w + (size_t)n * p.K + k0 + chunk * 4);
            rb[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            uint2 u = {v2a_pack_h2<F16>(ra[i][0], ra[i][1]), v2a_pack_h2<F16>(ra[i][2], ra[i][3])};
            *reinterpret_cast<uint2*>(&As[buf][(lrow + i * RPP) * LDH + chunk * 4]) = u;
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            uint2 u = {v2a_pack_h2<F16>(rb[i][0], rb[i][1]), v2a_pack_h2<F16>(rb[i][2], rb[i][3])};
            *reinterpret_cast<uint2*>(&Bs[buf][(lrow + i * RPP) * LDH + chunk * 4]) = u;
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
    if (kt_begin < kt_end) {
        load_tile(kt_begin);
        store_tile(0);
    }
    __syncthreads();
    int buf = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = (kt + 1) < kt_end;
        if (more) load_tile(kt + 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(&As[buf][(wm + i * 32 + lr) * LDH + 16 * h + 8 * lk]);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8*>(&Bs[buf][(wn + j * 32 + lr) * LDH + 16 * h + 8 * lk]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = v2a_mfma_h<F16>(a[i], b[j], acc[i][j]);
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn + j * 32 + lr;
            if (n >= p.Cout) continue;
            const float bv = (p.bias && p.splitk == 1) ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m >= p.M) continue;
                float v = acc[i][j][r];
                if (p.splitk > 1) {
                    p.partial[((size_t)split * p.M + m) * p.Cout + n] = v;
                } else {
                    v += bv;
                    if (p.rowvec) v += p.rowvec[(size_t)(m / p.rows_per_batch) * p.Cout + n];
                    if (p.residual) v += p.residual[(size_t)m * p.Cout + n];
                    if (p.y2 && n >= p.csplit) p.y2[(size_t)m * (p.Cout - p.csplit) + (n - p.csplit)] = v;
                    else p.y[(size_t)m * (p.y2 ? p.csplit : p.Cout) + n] = v;
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------ fp32 by three bf16 planes
// x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): both differences are exact in fp32, so the three
// planes carry 24 significant bits.  a * b is then summed from the six plane products whose weight is >= 2^-16 (hi*hi, hi*mid, mid*hi,
// hi*lo, lo*hi, mid*mid; the three dropped ones are <= 2^-24 |a||b|, the size of one fp32 rounding), each exact in the bf16 MFMA's fp32
// accumulation, smallest first.  Six v_mfma_f32_32x32x16_bf16 (192 cycles per 32x32x16 block) replace eight v_mfma_f32_32x32x2_f32 (512).
__device__ __forceinline__ void split3_pair(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) { v2a_split3x2(x0, x1, h, m, l); }


// y[m][n] = sum_s partial[s][m][n] + bias + rowvec + residual   (split-K second pass)
__global__ void conv_splitk_reduce(const ConvDesc p) {
    const size_t total = (size_t)p.M * p.Cout;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / p.Cout), n = (int)(idx - (size_t)m * p.Cout);
        float v = 0.f;
        for (int s = 0; s < p.splitk; ++s) v += p.partial[(size_t)s * total + idx];
        if (p.bias) v += p.bias[n];
        if (p.rowvec) v += p.rowvec[(size_t)(m / p.rows_per_batch) * p.Cout + n];
        if (p.residual) v += p.residual[idx];
        if (p.y2 && n >= p.csplit) p.y2[(size_t)m * (p.Cout - p.csplit) + (n - p.csplit)] = v;
        else p.y[(size_t)m * (p.y2 ? p.csplit : p.Cout) + n] = v;
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[co][k'] = sum_r dY[r][co] * im2col(X)[r][k'],  r = conv output row, k' = (kh, kw, ci).
// GEMM: M_g = Cout (A = dY^T, co contiguous), N_g = K' (B = gathered X, ci contiguous), reduction over r.
struct WgradDesc {
    const float* x; const float* x2;   // conv input (two sources as in ConvDesc)
    const float* dy;                   // [M][Cout]
    const uint16_t* xh; const uint16_t* dyh;   // bf16 twins of x / dy (conv_wgrad_bf16h / conv_wgrad_tr_h) or null
    const uint16_t* x2h;               // twin of x2 (two-source gather of conv_wgrad_tr_h) or null
    float* dw;                         // torch layout [Cout][Cin][KH][KW]  (or [Cin][Cout][KH][KW]-free: see transposed)
    float* partial;                    // [splits][Cout][K'] followed by [splits][Cout] bias partials
    float* dbias;                      // optional: [Cout] = sum_r dY[r][co] (fused bias gradient) or null
    int N, H, W, C1, C2, OH, OW, Cout, KH, KW, sh, sw, ph, pw, idil, ups, HL, WL, M, K;
    int splits, rtiles_per_split;
    int accumulate;                    // 1: dw += result
    int f16;                           // twin-fed bodies: the 16-bit twins are IEEE fp16 (1) or bf16 (0)
    FastDiv fd_ow, fd_oh;
};

__device__ __forceinline__ void wgrad_store(const WgradDesc& p, int co, int k, float v) {
    const int Cin = p.C1 + p.C2;
    const int tap = k / Cin, ci = k - tap * Cin;
    float* dst = p.dw + ((size_t)co * Cin + ci) * (p.KH * p.KW) + tap;
    *dst = p.accumulate ? (*dst + v) : v;
}

template <int BM, int BN, bool VECA, bool VECB>
__global__ __launch_bounds__(256) void conv_wgrad_f32(const WgradDesc p) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int AL = BM / 64, BL = BN / 64;   // float4 loads per thread per tile: 16 rows x (BM/4) float4 / 256
    __shared__ float As[2][BK][BM + LDS_PAD];
    __shared__ float Bs[2][BK][BN + LDS_PAD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = (p.K + BN - 1) / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
    const int split = blockIdx.y;
    const int Cin = p.C1 + p.C2;
    const int nrt = (p.M + BK - 1) / BK;
    const int rt_begin = split * p.rtiles_per_split;
    const int rt_end = min(nrt, rt_begin + p.rtiles_per_split);

    // A: thread -> (row kr = tid / (BM/4) + i * (256/(BM/4)), col4 = tid % (BM/4))
    constexpr int ACOLS = BM / 4, AROWS = 256 / ACOLS;
    constexpr int BCOLS = BN / 4, BROWS = 256 / BCOLS;
    const int a_c4 = tid % ACOLS, a_r = tid / ACOLS;
    const int b_c4 = tid % BCOLS, b_r = tid / BCOLS;
    // column (k') of B is fixed per thread for the whole reduction
    int b_kh[4], b_kw[4], b_c[4];
    bool b_kok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int k = n0 + b_c4 * 4 + j;
        b_kok[j] = k < p.K;
        int kk = b_kok[j] ? k : 0;
        int tap = kk / Cin;
        b_c[j] = kk - tap * Cin;
        b_kh[j] = tap / p.KW;
        b_kw[j] = tap - b_kh[j] * p.KW;
    }
    f32x4 ra[AL], rb[BL];
    const bool do_bias = (p.dbias != nullptr) && (blockIdx.x % tiles_n == 0);
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    auto load_tile = [&](int rt) {
        const int r0 = rt * BK;
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            const int r = r0 + a_r + i * AROWS;
            const int co = m0 + a_c4 * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < p.M) {
                if (VECA) {
                    if (co < p.Cout) v = *reinterpret_cast<const f32x4*>(p.dy + (size_t)r * p.Cout + co);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (co + j < p.Cout) v[j] = p.dy[(size_t)r * p.Cout + co + j];
                }
            }
            ra[i] = v;
            if (do_bias) bsum += v;
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            const int r = r0 + b_r + i * BROWS;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < p.M) {
                const int t = (int)fdiv((uint32_t)r, p.fd_ow);
                const int ow = r - t * p.OW;
                const int img = (int)fdiv((uint32_t)t, p.fd_oh);
                const int oh = t - img * p.OH;
                const int ihb = oh * p.sh - p.ph, iwb = ow * p.sw - p.pw;
                auto fetch = [&](int j) -> const float* {
                    int ih = ihb + b_kh[j], iw = iwb + b_kw[j];
                    bool ok = b_kok[j] && ih >= 0 && ih < p.HL && iw >= 0 && iw < p.WL;
                    if (p.idil > 1) {
                        ok = ok && (ih % p.idil == 0) && (iw % p.idil == 0);
                        ih /= p.idil;
                        iw /= p.idil;
                    }
                    if (p.ups) { ih >>= 1; iw >>= 1; }
                    if (!ok) return nullptr;
                    size_t pix = ((size_t)img * p.H + ih) * p.W + iw;
                    int c = b_c[j];
                    return (c < p.C1) ? p.x + pix * p.C1 + c : p.x2 + pix * p.C2 + (c - p.C1);
                };
                if (VECB) {
                    const float* s = fetch(0);
                    if (s) v = *reinterpret_cast<const f32x4*>(s);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float* s = fetch(j);
                        if (s) v[j] = *s;
                    }
                }
            }
            rb[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AL; ++i) *reinterpret_cast<f32x4*>(&As[buf][a_r + i * AROWS][a_c4 * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < BL; ++i) *reinterpret_cast<f32x4*>(&Bs[buf][b_r + i * BROWS][b_c4 * 4]) = rb[i];
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

    if (rt_begin < rt_end) {
        load_tile(rt_begin);
        store_tile(0);
    }
    __syncthreads();
    int buf = 0;
    for (int rt = rt_begin; rt < rt_end; ++rt) {
        const bool more = (rt + 1) < rt_end;
        if (more) load_tile(rt + 1);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[buf][kk + lk][wm + i * 32 + lr];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[buf][kk + lk][wn + j * 32 + lr];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int k = n0 + wn + j * 32 + lr;
            if (k >= p.K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (co >= p.Cout) continue;
                if (p.splits > 1) p.partial[((size_t)split * p.Cout + co) * p.K + k] = acc[i][j][r];
                else wgrad_store(p, co, k, acc[i][j][r]);
            }
        }
    if (do_bias) {      // fused bias gradient: threads (a_r, a_c4) hold partial column sums of dY over their rows
        __syncthreads();
        float* red = &As[0][0][0];                       // [AROWS][BM] floats fit in the first buffer
#pragma unroll
        for (int j = 0; j < 4; ++j) red[a_r * BM + a_c4 * 4 + j] = bsum[j];
        __syncthreads();
        if (tid < BM) {
            float t = 0.f;
            for (int r = 0; r < AROWS; ++r) t += red[r * BM + tid];
            const int co = m0 + tid;
            if (co < p.Cout) {
                if (p.splits > 1) p.partial[(size_t)p.splits * p.Cout * p.K + (size_t)split * p.Cout + co] = t;
                else p.dbias[co] = p.accumulate ? p.dbias[co] + t : t;
            }
        }
    }
}

// Weight gradient with LDS-DMA staging (exact-f32 MFMA).  Both operands are k-major here -- tile rows are reduction rows r, the
// 128 columns are contiguous in HBM (output channels of dY; the channels of one or two filter taps of the gathered input) -- so a
// tile is 32 rows x 512 B, moved by `global_load_lds_dwordx4` (16 B per lane, no staging VGPRs), and the 32x32x2 MFMA operand
// fetch (one float per lane: row kk + lk, column lr) is a conflict-free `ds_read_b32` on the plain lane-linear image.  The
// column of every DMA piece (-> filter tap, input channel) is fixed per thread for the whole reduction; per k tile only the
// reduction row is decoded (two reciprocal multiplies).  Rows past M, padding taps and columns past the edge read a zero line.
// Needs Cout % 4 == 0, K % 4 == 0, Cin % 4 == 0 (whole 16-B pieces).  Epilogue / split layout identical to conv_wgrad_f32.
__device__ __attribute__((aligned(128))) float g_zero_line[64];

typedef const __attribute__((address_space(1))) void* gptr_w_t;
typedef __attribute__((address_space(3))) void* lptr_w_t;

template <int BM, int BN, int NST>
__device__ __forceinline__ void wgrad_dma_body(const WgradDesc& p, const int tile_id, const int split, unsigned char* smem) {
    // NST LDS stages of one 32-row k tile each (counted vmcnt: the DMAs of a tile complete in issue order).  Measured
    // (tools/wgrad_sweep.py, V2A_WGRAD_STAGES=4): four stages at two workgroups per CU are 5-15 % SLOWER than two stages at five
    // workgroups per CU on every 64x64 shape -- residency, not pipeline depth, is what hides the DMA latency here.  Default NST = 2.
    constexpr int BKR = 32;                                     // reduction rows per tile
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int ACH = BM / 4, BCH = BN / 4;                   // 16-B pieces per tile row
    constexpr int AROWS = 256 / ACH, BROWS = 256 / BCH;         // tile rows covered by one pass of the 256 threads
    constexpr int AL = BKR / AROWS, BL = BKR / BROWS;           // DMA pieces per thread per tile
    constexpr int ABYTES = BKR * BM * 4, BBYTES = BKR * BN * 4, BUF = ABYTES + BBYTES;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = (p.K + BN - 1) / BN;
    const int m0 = (tile_id / tiles_n) * BM, n0 = (tile_id % tiles_n) * BN;
    const int Cin = p.C1 + p.C2;
    const int nrt = (p.M + BKR - 1) / BKR;
    const int rt_begin = split * p.rtiles_per_split;
    const int rt_end = min(nrt, rt_begin + p.rtiles_per_split);
    const float* zline = g_zero_line;

    // A (dY^T): piece column fixed per thread
    const int a_col = tid % ACH, a_row = tid / ACH;
    const int a_co = m0 + a_col * 4;
    const bool a_cok = a_co < p.Cout;
    // B (gathered input): piece column -> (tap, channel) fixed per thread
    const int b_col = tid % BCH, b_row = tid / BCH;
    const int bk = n0 + b_col * 4;
    const bool b_kok = bk < p.K;
    const int btap = (b_kok ? bk : 0) / Cin;
    const int bci = (b_kok ? bk : 0) - btap * Cin;
    const int bkh = btap / p.KW, bkw = btap - bkh * p.KW;
    const bool bfirst = bci < p.C1;
    const float* bsrc = bfirst ? p.x : p.x2;
    const uint32_t bCs = (uint32_t)(bfirst ? p.C1 : p.C2);
    const uint32_t bcc = (uint32_t)(bfirst ? bci : bci - p.C1);

    auto issue = [&](int rt, int buf) {
        unsigned char* abase = smem + buf * BUF;
        unsigned char* bbase = abase + ABYTES;
        const int r0 = rt * BKR;
#pragma unroll
        for (int j = 0; j < AL; ++j) {
            const int r = r0 + a_row + j * AROWS;
            const float* g = (a_cok && r < p.M) ? p.dy + (size_t)r * p.Cout + a_co : zline;
            __builtin_amdgcn_global_load_lds((gptr_w_t)g, (lptr_w_t)(abase + (j * 256 + wid * 64) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < BL; ++j) {
            const int r = r0 + b_row + j * BROWS;
            const uint32_t rr = r < p.M ? (uint32_t)r : 0u;
            const uint32_t t = fdiv(rr, p.fd_ow);
            const int ow = (int)(rr - t * p.OW);
            const uint32_t img = fdiv(t, p.fd_oh);
            const int oh = (int)(t - img * p.OH);
            int ih = oh * p.sh - p.ph + bkh, iw = ow * p.sw - p.pw + bkw;
            bool ok = b_kok && r < p.M && (unsigned)ih < (unsigned)p.HL && (unsigned)iw < (unsigned)p.WL;
            if (p.idil > 1) {
                ok = ok && (ih % p.idil == 0) && (iw % p.idil == 0);
                ih /= p.idil;
                iw /= p.idil;
            }
            if (p.ups) { ih >>= 1; iw >>= 1; }
            const uint32_t off = ((uint32_t)((int)img * p.H + ih) * (uint32_t)p.W + (uint32_t)iw) * bCs + bcc;
            const float* g = ok ? bsrc + off : zline;
            __builtin_amdgcn_global_load_lds((gptr_w_t)g, (lptr_w_t)(bbase + (j * 256 + wid * 64) * 16), 16, 0, 0);
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
    const bool do_bias = (p.dbias != nullptr) && (tile_id % tiles_n == 0);
    float bsum = 0.f;                                           // thread tid < BM: column sum of dY over this block's rows

    constexpr int PER_TILE = AL + BL;                           // DMA instructions per thread per tile
#pragma unroll
    for (int j = 0; j < NST - 1; ++j)
        if (rt_begin + j < rt_end) issue(rt_begin + j, j);
    int buf = 0;
    for (int rt = rt_begin; rt < rt_end; ++rt) {
        // tile rt has landed once at most `ahead` younger tiles are still outstanding (ahead shrinks at the tail of the reduction)
        const int ahead = min(NST - 2, rt_end - 1 - rt);
        if (NST == 2 || ahead <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER_TILE) : "memory");
        __syncthreads();
        {   // the stage that held tile rt-1 is free (every wave passed the barrier after computing it): refill it with tile rt+NST-1
            int nb = buf + NST - 1;
            if (nb >= NST) nb -= NST;
            if (rt + NST - 1 < rt_end) issue(rt + NST - 1, nb);
        }
        const float* As = reinterpret_cast<const float*>(smem + buf * BUF);
        const float* Bs = As + BKR * BM;
#pragma unroll
        for (int kk = 0; kk < BKR; kk += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[(kk + lk) * BM + wm + i * 32 + lr];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[(kk + lk) * BN + wn + j * 32 + lr];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (do_bias && tid < BM) {
#pragma unroll 8
            for (int r = 0; r < BKR; ++r) bsum += As[r * BM + tid];
        }
        buf = (buf + 1 == NST) ? 0 : buf + 1;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int k = n0 + wn + j * 32 + lr;
            if (k >= p.K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (co >= p.Cout) continue;
                if (p.splits > 1) p.partial[((size_t)split * p.Cout + co) * p.K + k] = acc[i][j][r];
                else wgrad_store(p, co, k, acc[i][j][r]);
            }
        }
    if (do_bias && tid < BM) {
        const int co = m0 + tid;
        if (co < p.Cout) {
            if (p.splits > 1) p.partial[(size_t)p.splits * p.Cout * p.K + (size_t)split * p.Cout + co] = bsum;
            else p.dbias[co] = p.accumulate ? p.dbias[co] + bsum : bsum;
        }
    }
}
template <int BM, int BN, int NST = 2>
__global__ __launch_bounds__(256) void conv_wgrad_dma_f32(const WgradDesc p) {
    __shared__ __attribute__((aligned(128))) unsigned char smem[NST * 32 * (BM + BN) * 4];
    wgrad_dma_body<BM, BN, NST>(p, (int)blockIdx.x, (int)blockIdx.y, smem);
}

// Weight gradient of 3x3 / stride 1 / pad 1 convs with a spatial halo tile in LDS (exact-f32 MFMA).
// The generic kernels above re-fetch every input pixel once per filter tap and every dY row once per K tile: for a 64 -> 64 channel
// layer that is 9 x (dY + X) through the L2 fabric, which -- not the matrix pipe -- bounds them (~3.4 TB/s, 55-70 TFLOP/s).  Here a
// workgroup owns 64 output channels x (64 input channels x all 9 taps): per k tile of 32 output pixels (a TH x TW patch, TW = min(OW,
// 32)) it DMAs the dY rows (8 KB) and the (TH+2) x (TW+2) input halo of its 64-channel slice (<= 28 KB) ONCE, and the nine taps read
// shifted windows of that halo straight from LDS: 4.2x less DMA traffic per FLOP, 144 MFMAs per wave per tile.  Accumulators: 9 x 16
// VGPRs per lane.  Grid: (Cout/64 * Cin/64, splits); split slabs / bias partials / reduce kernel shared with the kernels above.
template <int TW>
__device__ __forceinline__ void wgrad_halo_body(const WgradDesc& p, const int tile_id, const int split, unsigned char* smem) {
    constexpr int TH = 32 / TW, HWD = TW + 2, HHT = TH + 2, HP = HWD * HHT;
    constexpr int NB = (HP + 15) / 16;                          // halo DMA passes (16 pixels x 64 channels per pass of 256 threads)
    constexpr int ABYTES = 32 * 64 * 4, BBYTES = NB * 16 * 64 * 4, BUF = ABYTES + BBYTES;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int Cin = p.C1;
    const int cin_blocks = Cin >> 6;
    const int m0 = (tile_id / cin_blocks) * 64, c0 = (tile_id % cin_blocks) * 64;
    const int nrt = p.M >> 5;
    const int rt_begin = split * p.rtiles_per_split;
    const int rt_end = min(nrt, rt_begin + p.rtiles_per_split);
    const int tiles_x = p.OW / TW, tiles_img = (p.OH / TH) * tiles_x;
    const float* zline = g_zero_line;
    const int piece = tid & 15, prow = tid >> 4;                // 16-B piece within a 64-float row; row / pixel within a pass

    auto issue = [&](int rt, int buf) {
        unsigned char* abase = smem + buf * BUF;
        unsigned char* bbase = abase + ABYTES;
        const size_t r0 = (size_t)rt * 32;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float* g = p.dy + (r0 + prow + j * 16) * p.Cout + m0 + piece * 4;
            __builtin_amdgcn_global_load_lds((gptr_w_t)g, (lptr_w_t)(abase + (j * 256 + wid * 64) * 16), 16, 0, 0);
        }
        const int img = rt / tiles_img, t = rt - img * tiles_img;
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int ih0 = ty * TH - 1, iw0 = tx * TW - 1;
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int pix = q * 16 + prow;
            const int hy = pix / HWD, hx = pix - hy * HWD;
            const int ih = ih0 + hy, iw = iw0 + hx;
            const bool ok = pix < HP && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const float* g = ok ? p.x + ((size_t)(img * p.H + ih) * p.W + iw) * Cin + c0 + piece * 4 : zline;
            __builtin_amdgcn_global_load_lds((gptr_w_t)g, (lptr_w_t)(bbase + (q * 256 + wid * 64) * 16), 16, 0, 0);
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int wm = (wid >> 1) * 32, wn = (wid & 1) * 32;
    const int lr = lane & 31, lk = lane >> 5;
    const bool do_bias = (p.dbias != nullptr) && (c0 == 0);
    float bsum = 0.f;

    if (rt_begin < rt_end) issue(rt_begin, 0);
    int buf = 0;
    for (int rt = rt_begin; rt < rt_end; ++rt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (rt + 1 < rt_end) issue(rt + 1, buf ^ 1);
        const float* As = reinterpret_cast<const float*>(smem + buf * BUF);
        const float* Hs = As + 32 * 64;
#pragma unroll 4
        for (int kk = 0; kk < 32; kk += 2) {
            const int r = kk + lk;
            const float a = As[r * 64 + wm + lr];
            const float* hp = Hs + ((r / TW) * HWD + (r % TW)) * 64 + wn + lr;      // tap (0,0) of this output pixel
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float b = hp[(kh * HWD + kw) * 64];
                    acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[kh * 3 + kw], 0, 0, 0);
                }
        }
        if (do_bias && tid < 64) {
#pragma unroll 8
            for (int r = 0; r < 32; ++r) bsum += As[r * 64 + tid];
        }
        buf ^= 1;
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int k = t * Cin + c0 + wn + lr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (p.splits > 1) p.partial[((size_t)split * p.Cout + co) * p.K + k] = acc[t][r];
            else wgrad_store(p, co, k, acc[t][r]);
        }
    }
    if (do_bias && tid < 64) {
        const int co = m0 + tid;
        if (p.splits > 1) p.partial[(size_t)p.splits * p.Cout * p.K + (size_t)split * p.Cout + co] = bsum;
        else p.dbias[co] = p.accumulate ? p.dbias[co] + bsum : bsum;
    }
}
template <int TW>
constexpr int wgrad_halo_lds() { return 2 * (32 * 64 * 4 + (((TW + 2) * (32 / TW + 2) + 15) / 16) * 16 * 64 * 4); }
template <int TW>
__global__ __launch_bounds__(256, 2) void conv_wgrad_halo_f32(const WgradDesc p) {
    __shared__ __attribute__((aligned(128))) unsigned char smem[wgrad_halo_lds<TW>()];
    wgrad_halo_body<TW>(p, (int)blockIdx.x, (int)blockIdx.y, smem);
}
static int g_precision = 0;   // 0: exact-f32 MFMA (parity configuration)  1: 16-bit MFMA inputs, fp32 storage / accumulate
int g_v2a_policy_f16 = 0;     // 16-bit format of the policy's MFMA mode: 0 bf16 (default), 1 IEEE fp16 (v2a_set_policy_half; the reference's GPU
                              // path is fp16 autocast + GradScaler: lb_online_trainer_v7.py:72-76,593,604-612)
static int g_wforce_bm = 0;   // experiments only (v2a_debug_force_wgrad_plan)
// split of the halo kernel: ~512 workgroups (two resident per CU), at least 8 k tiles per slice
static int wgrad_halo_split(int M, int Cout, int K) {
    const int Cin = K / 9;
    const int combos = (Cout / 64) * (Cin / 64);
    int s = 512 / (combos < 1 ? 1 : combos);
    const int deep = (M / 32) / 8;
    if (s > deep) s = deep;
    return s < 1 ? 1 : s;
}
// geometry-free part of the eligibility test (what the workspace query can know)
static bool wgrad_halo_shape_ok(int M, int Cout, int K) {
    // small problems lose: every workgroup writes a 64 x 576 slab, and with few (Cout, Cin) blocks all parallelism has to come from
    // the split (64 -> 64 channels at 32x32x128 images: 115 us against 88 us on the generic kernel); from ~8 GFLOP and >= 4 blocks up
    // the halo kernel wins by 20-40 % (tools/wgrad_sweep.py)
    if (!(g_precision == 0 && !g_wforce_bm && K % 9 == 0 && (K / 9) % 64 == 0 && Cout % 64 == 0 && M % 32 == 0)) return false;
    return (Cout / 64) * (K / 9 / 64) >= 4 && 2.0 * (double)M * (double)K * (double)Cout >= 8e9;
}

// bf16-MFMA weight gradient.  The MFMA wants 8 consecutive reduction rows per lane while HBM is contiguous along the OTHER axis
// (channels), so each loader thread owns a 8(rows) x 4(channels) register block: 8 coalesced float4 loads, then four b128 LDS
// stores of 8 bf16 along the reduction axis ([channel][32 rows + pad] tiles).  Threads [0,BM) stage dY^T, [BM,BM+BN) the im2col.
template <int BM, int BN, bool F16>
__global__ __launch_bounds__(256) void conv_wgrad_bf16(const WgradDesc p) {
    constexpr int BKT = 32, LDH = 40;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    __shared__ __attribute__((aligned(16))) uint16_t As[2][BM * LDH];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[2][BN * LDH];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = (p.K + BN - 1) / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
    const int split = blockIdx.y;
    const int Cin = p.C1 + p.C2;
    const int nrt = (p.M + BKT - 1) / BKT;
    const int rt_begin = split * p.rtiles_per_split;
    const int rt_end = min(nrt, rt_begin + p.rtiles_per_split);
    const bool isA = tid < BM;
    const bool isB = !isA && tid < BM + BN;
    const int t2 = isA ? tid : tid - BM;
    const int c4 = isA ? t2 % (BM / 4) : t2 % (BN / 4);
    const int rgrp = isA ? t2 / (BM / 4) : t2 / (BN / 4);
    // B column (k') decode is loop invariant
    int b_kh = 0, b_kw = 0, b_c = 0;
    bool b_kok = false;
    if (isB) {
        const int k = n0 + c4 * 4;
        b_kok = k < p.K;
        const int kk = b_kok ? k : 0;
        const int tap = kk / Cin;
        b_c = kk - tap * Cin;
        b_kh = tap / p.KW;
        b_kw = tap - b_kh * p.KW;
    }
    const bool do_bias = (p.dbias != nullptr) && (blockIdx.x % tiles_n == 0);
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    f32x4 rv[8];
    auto load_tile = [&](int rt) {
        const int r0 = rt * BKT + rgrp * 8;
        if (isA) {
            const int co = m0 + c4 * 4;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = r0 + e;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (r < p.M && co < p.Cout) v = *reinterpret_cast<const f32x4*>(p.dy + (size_t)r * p.Cout + co);
                rv[e] = v;
                if (do_bias) bsum += v;
            }
        } else if (isB) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = r0 + e;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (r < p.M && b_kok) {
                    const int t = (int)fdiv((uint32_t)r, p.fd_ow);
                    const int ow = r - t * p.OW;
                    const int img = (int)fdiv((uint32_t)t, p.fd_oh);
                    const int oh = t - img * p.OH;
                    int ih = oh * p.sh - p.ph + b_kh, iw = ow * p.sw - p.pw + b_kw;
                    bool ok = ih >= 0 && ih < p.HL && iw >= 0 && iw < p.WL;
                    if (p.idil > 1) {
                        ok = ok && (ih % p.idil == 0) && (iw % p.idil == 0);
                        ih /= p.idil;
                        iw /= p.idil;
                    }
                    if (p.ups) { ih >>= 1; iw >>= 1; }
                    if (ok) {
                        const size_t pix = ((size_t)img * p.H + ih) * p.W + iw;
                        const float* src = (b_c < p.C1) ? p.x + pix * p.C1 + b_c : p.x2 + pix * p.C2 + (b_c - p.C1);
                        v = *reinterpret_cast<const f32x4*>(src);
                    }
                }
                rv[e] = v;
            }
        }
    };
    auto store_tile = [&](int buf) {
        if (isA || isB) {
            uint16_t* dst = (isA ? &As[buf][0] : &Bs[buf][0]) + (c4 * 4) * LDH + rgrp * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint4 u = {v2a_pack_h2<F16>(rv[0][j], rv[1][j]), v2a_pack_h2<F16>(rv[2][j], rv[3][j]), v2a_pack_h2<F16>(rv[4][j], rv[5][j]),
                           v2a_pack_h2<F16>(rv[6][j], rv[7][j])};
                *reinterpret_cast<uint4*>(dst + j * LDH) = u;
            }
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
    if (rt_begin < rt_end) {
        load_tile(rt_begin);
        store_tile(0);
    }
    __syncthreads();
    int buf = 0;
    for (int rt = rt_begin; rt < rt_end; ++rt) {
        const bool more = (rt + 1) < rt_end;
        if (more) load_tile(rt + 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(&As[buf][(wm + i * 32 + lr) * LDH + 16 * h + 8 * lk]);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8*>(&Bs[buf][(wn + j * 32 + lr) * LDH + 16 * h + 8 * lk]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = v2a_mfma_h<F16>(a[i], b[j], acc[i][j]);
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int k = n0 + wn + j * 32 + lr;
            if (k >= p.K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (co >= p.Cout) continue;
                if (p.splits > 1) p.partial[((size_t)split * p.Cout + co) * p.K + k] = acc[i][j][r];
                else wgrad_store(p, co, k, acc[i][j][r]);
            }
        }
    if (do_bias) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(&As[0][0]);        // [4][BM] floats
        if (isA) {
#pragma unroll
            for (int j = 0; j < 4; ++j) red[rgrp * BM + c4 * 4 + j] = bsum[j];
        }
        __syncthreads();
        if (tid < BM) {
            const float t = red[tid] + red[BM + tid] + red[2 * BM + tid] + red[3 * BM + tid];
            const int co = m0 + tid;
            if (co < p.Cout) {
                if (p.splits > 1) p.partial[(size_t)p.splits * p.Cout * p.K + (size_t)split * p.Cout + co] = t;
                else p.dbias[co] = p.accumulate ? p.dbias[co] + t : t;
            }
        }
    }
}

// ---- fp32 weight gradient by three bf16 planes (round 4; the weight-gradient form of csrc/igemm_h.hip conv_igemm_f32x3).
// dW[co][k'] = sum_r dY[r][co] * X[r][k']: both operands are contiguous along the NON-reduction axis in HBM while the bf16 MFMA wants 8
// consecutive reduction rows r per lane.  The loader does no transposition at all: a thread takes one 16-B load (4 channels of one row),
// splits it into hi / mid / lo planes (x = hi + mid + lo exactly, see conv_igemm_f32x3) and writes three 8-B pieces into k-major LDS
// images [32 rows][BM | BN columns] (one image per plane, 16-B pieces XOR-swizzled by SWZ * (row & 3) like wgrad_tr_body below);
// `ds_read_b64_tr_b16` delivers the MFMA operand from them (two reads = 8 reduction rows of the lane's column).  A product block = the
// six plane products of weight >= 2^-16, smallest first.  Pipeline as in the forward kernel: global loads two reduction tiles ahead
// (two register sets), split + ds_write_b64 of tile t + 1 between the two MFMA groups of tile t, two LDS stages, one barrier per tile;
// every thread loads (no idle half as in the register-transposing first version).  fp32-equivalent accuracy (not bit-equal to the
// exact-f32 kernels: V2A_F32_CONV=exact keeps those).
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((address_space(1))) f32x4 gf32x4w_t;
typedef __attribute__((address_space(3))) bf16x4_t lbf16x4_t;
template <int BM, int BN, int WVM, int WVN, bool GEN>
__device__ __forceinline__ void wgrad_x3_body(const WgradDesc& p, const int tile_id, const int split, unsigned char* smem) {
    constexpr int NT = 64 * WVM * WVN, BKR = 32;
    constexpr int PITCH_A = BM * 2, PITCH_B = BN * 2;           // bytes per image row
    constexpr int PA = BKR * PITCH_A, PB = BKR * PITCH_B, STG = 3 * (PA + PB);
    constexpr int SWZ_A = BM == 128 ? 4 : 2, SWZ_B = BN == 128 ? 4 : 2;
    constexpr int TPR_A = BM / 4, RP_A = NT / TPR_A, AL = BKR / RP_A;      // threads per image row, rows per loader pass, loads per thread
    constexpr int TPR_B = BN / 4, RP_B = NT / TPR_B, BL = BKR / RP_B;
    constexpr int WM = BM / WVM, WN = BN / WVN, TM = WM / 32, TN = WN / 32;
    static_assert(AL >= 1 && BL >= 1 && TM >= 1 && TN >= 1 && (BM == 64 || BM == 128) && (BN == 64 || BN == 128), "tile shape");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = (p.K + BN - 1) / BN;
    const int m0 = (tile_id / tiles_n) * BM, n0 = (tile_id % tiles_n) * BN;
    const int Cin = p.C1 + p.C2;
    const int nrt = (p.M + BKR - 1) / BKR;
    const int rt_begin = split * p.rtiles_per_split;
    const int rt_end = min(nrt, rt_begin + p.rtiles_per_split);
    const float* zline = g_zero_line;

    const int a_c4 = tid % TPR_A, a_r = tid / TPR_A;
    const int a_co = m0 + a_c4 * 4;
    const bool a_cok = a_co < p.Cout;
    const float* a_src = p.dy + (a_cok ? a_co : 0);
    const int b_c4 = tid % TPR_B, b_r = tid / TPR_B;
    const int bk = n0 + b_c4 * 4;
    const bool b_kok = bk < p.K;
    const int btap = (b_kok ? bk : 0) / Cin;
    const int bci = (b_kok ? bk : 0) - btap * Cin;
    const int bkh = btap / p.KW, bkw = btap - bkh * p.KW;
    const bool bfirst = bci < p.C1;                             // column from x or from x2 (channel concat [x | x2]; C1 % 4 == 0)
    const float* b_src = bfirst ? p.x + bci : p.x2 + (bci - p.C1);
    const int bCs = bfirst ? p.C1 : p.C2;
    int it = rt_begin;
    const int shift = (p.ups || p.idil == 2) ? 1 : 0;
    const int pmask = (p.idil == 2) ? 1 : 0;
    // loop-invariant descriptor fields pinned in SGPRs (readfirstlane: not re-loaded from the kernel arguments inside the loop, where an
    // s_waitcnt lgkmcnt would also wait for the LDS reads); branch-free form of fdiv()
    auto sgpr = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    const int M_ = sgpr(p.M), Cout_ = sgpr(p.Cout), OW_ = sgpr(p.OW), OH_ = sgpr(p.OH), H_ = sgpr(p.H), W_ = sgpr(p.W), HL_ = sgpr(p.HL), WL_ = sgpr(p.WL);
    const int sh_ = sgpr(p.sh), sw_ = sgpr(p.sw), ph_ = sgpr(p.ph - bkh * 0), pw_ = sgpr(p.pw);
    const uint32_t ow_m = (uint32_t)sgpr((int)p.fd_ow.m), ow_s = (uint32_t)sgpr((int)((p.fd_ow.s - 1) & 31)), oh_m = (uint32_t)sgpr((int)p.fd_oh.m),
                   oh_s = (uint32_t)sgpr((int)((p.fd_oh.s - 1) & 31));
    const bool ow_one = OW_ <= 1, oh_one = OH_ <= 1;
    auto fdivb = [](uint32_t n, uint32_t m, uint32_t sm1, bool one) -> uint32_t {
        const uint32_t t = __umulhi(m, n);
        const uint32_t q = (t + ((n - t) >> 1)) >> sm1;
        return one ? n : q;
    };

    // request the next reduction tile (16-B loads into the given register set; past the slice's end: the zero line)
    auto load = [&](f32x4 (&ra)[AL], f32x4 (&rb)[BL]) {
        const bool live = it < rt_end;
        const int r0 = it * BKR;
#pragma unroll
        for (int j = 0; j < AL; ++j) {
            const int r = r0 + j * RP_A + a_r;
            uint32_t off = (uint32_t)r * (uint32_t)Cout_;
            asm volatile("" : "+v"(off));
            const float* g = a_src + off;
            g = (live & a_cok & (r < M_)) ? g : zline;
            ra[j] = *(const gf32x4w_t*)(uint64_t)g;             // (address space 1: a global_load, never a flat_load)
        }
#pragma unroll
        for (int j = 0; j < BL; ++j) {
            const int r = r0 + j * RP_B + b_r;
            const uint32_t rr = r < M_ ? (uint32_t)r : 0u;
            const uint32_t t = fdivb(rr, ow_m, ow_s, ow_one);
            const int ow = (int)(rr - t * (uint32_t)OW_);
            const uint32_t img = fdivb(t, oh_m, oh_s, oh_one);
            const int oh = (int)(t - img * (uint32_t)OH_);
            int ih = oh * sh_ - ph_ + bkh, iw = ow * sw_ - pw_ + bkw;
            bool ok = live & b_kok & (r < M_) & ((unsigned)ih < (unsigned)HL_) & ((unsigned)iw < (unsigned)WL_);
            if constexpr (GEN) {                                // folded nearest-x2 upsample / transposed conv (input dilation 2)
                ok = ok & (((ih | iw) & pmask) == 0);
                ih >>= shift;
                iw >>= shift;
            }
            uint32_t off = ((uint32_t)((int)img * H_ + ih) * (uint32_t)W_ + (uint32_t)iw) * (uint32_t)bCs;
            asm volatile("" : "+v"(off));
            const float* g = b_src + off;
            g = ok ? g : zline;
            rb[j] = *(const gf32x4w_t*)(uint64_t)g;
        }
        ++it;
    };
    // LDS position of this thread's 8-B piece (4 bf16) in a plane image: row rr, columns 4 c4 .. + 3
    int wa_off[AL], wb_off[BL];
#pragma unroll
    for (int j = 0; j < AL; ++j) {
        const int rr = j * RP_A + a_r;
        wa_off[j] = rr * PITCH_A + ((((a_c4 >> 1) ^ (SWZ_A * (rr & 3))) << 4) | ((a_c4 & 1) << 3));
    }
#pragma unroll
    for (int j = 0; j < BL; ++j) {
        const int rr = j * RP_B + b_r;
        wb_off[j] = 3 * PA + rr * PITCH_B + ((((b_c4 >> 1) ^ (SWZ_B * (rr & 3))) << 4) | ((b_c4 & 1) << 3));
    }
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};                         // column sums of dY over this thread's rows (bias gradient)
    auto split_store_a = [&](const f32x4 (&ra)[AL], unsigned char* stage) {
#pragma unroll
        for (int j = 0; j < AL; ++j) {
            uint32_t h0, m0_, l0, h1, m1, l1;
            bsum += ra[j];
            split3_pair(ra[j][0], ra[j][1], h0, m0_, l0);
            split3_pair(ra[j][2], ra[j][3], h1, m1, l1);
            *reinterpret_cast<uint2*>(stage + wa_off[j]) = uint2{h0, h1};
            *reinterpret_cast<uint2*>(stage + PA + wa_off[j]) = uint2{m0_, m1};
            *reinterpret_cast<uint2*>(stage + 2 * PA + wa_off[j]) = uint2{l0, l1};
        }
    };
    auto split_store_b = [&](const f32x4 (&rb)[BL], unsigned char* stage) {
#pragma unroll
        for (int j = 0; j < BL; ++j) {
            uint32_t h0, m0_, l0, h1, m1, l1;
            split3_pair(rb[j][0], rb[j][1], h0, m0_, l0);
            split3_pair(rb[j][2], rb[j][3], h1, m1, l1);
            *reinterpret_cast<uint2*>(stage + wb_off[j]) = uint2{h0, h1};
            *reinterpret_cast<uint2*>(stage + PB + wb_off[j]) = uint2{m0_, m1};
            *reinterpret_cast<uint2*>(stage + 2 * PB + wb_off[j]) = uint2{l0, l1};
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
    // transposing-read source address of this lane (see wgrad_tr_body): 16-lane group g = lane >> 4, segment s = lane & 15
    const int grp = lane >> 4, seg = lane & 15;
    const int trow = 8 * (grp >> 1) + (seg >> 2);               // + 16 for the second k step, + 4 for the second half of the operand
    const int tcol = 16 * (grp & 1) + 4 * (seg & 3);
    int a_tr[TM], b_tr[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int col = wm + 32 * i + tcol;
        a_tr[i] = trow * PITCH_A + (((col >> 3) ^ (SWZ_A * (trow & 3))) << 4) + ((col & 7) << 1);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = wn + 32 * j + tcol;
        b_tr[j] = 3 * PA + trow * PITCH_B + (((col >> 3) ^ (SWZ_B * (trow & 3))) << 4) + ((col & 7) << 1);
    }
    auto tr8 = [&](const unsigned char* q, int pitch) -> bf16x8 {
        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lbf16x4_t*)(q));
        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lbf16x4_t*)(q + 4 * pitch));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto mfma6 = [&](const unsigned char* base, int h) {
        bf16x8 a[3][TM], b[3][TN];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[q][i] = tr8(base + q * PA + a_tr[i] + h * 16 * PITCH_A, PITCH_A);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[q][j] = tr8(base + q * PB + b_tr[j] + h * 16 * PITCH_B, PITCH_B);
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

    // ---- prologue: tile rt_begin staged, tile rt_begin + 1 requested
    f32x4 ra0[AL], rb0[BL], ra1[AL], rb1[BL];
    load(ra0, rb0);
    load(ra1, rb1);
    split_store_a(ra0, smem);
    split_store_b(rb0, smem);
    __syncthreads();
    auto step = [&](f32x4 (&ra_c)[AL], f32x4 (&rb_c)[BL], f32x4 (&ra_n)[AL], f32x4 (&rb_n)[BL], int buf) {
        unsigned char* cur = smem + buf * STG;
        unsigned char* oth = smem + (buf ^ 1) * STG;
        load(ra_n, rb_n);                                       // tile rt + 2 -> the register set consumed one step ago
        mfma6(cur, 0);
        split_store_a(ra_c, oth);                               // (VALU of the split runs under the MFMAs around it)
        mfma6(cur, 1);
        split_store_b(rb_c, oth);
        __syncthreads();
    };
    // always in pairs (fixed register-set roles at the loop header); an odd slice multiplies one all-zero tile at the end
    for (int rt = rt_begin; rt < rt_end; rt += 2) {
        step(ra1, rb1, ra0, rb0, 0);
        step(ra0, rb0, ra1, rb1, 1);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int k = n0 + wn + j * 32 + lr;
            if (k >= p.K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (co >= p.Cout) continue;
                if (p.splits > 1) p.partial[((size_t)split * p.Cout + co) * p.K + k] = acc[i][j][r];
                else wgrad_store(p, co, k, acc[i][j][r]);
            }
        }
    if ((p.dbias != nullptr) && (tile_id % tiles_n == 0)) {     // (every wave is past the loop's last barrier: the stages are free)
        float* red = reinterpret_cast<float*>(smem);            // [RP_A][BM] floats
#pragma unroll
        for (int j = 0; j < 4; ++j) red[a_r * BM + a_c4 * 4 + j] = bsum[j];
        __syncthreads();
        if (tid < BM) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < RP_A; ++g) t += red[g * BM + tid];
            const int co = m0 + tid;
            if (co < p.Cout) {
                if (p.splits > 1) p.partial[(size_t)p.splits * p.Cout * p.K + (size_t)split * p.Cout + co] = t;
                else p.dbias[co] = p.accumulate ? p.dbias[co] + t : t;
            }
        }
    }
}

// ---- three-plane weight gradient with a spatial halo tile in LDS (round 4): the halo idea of wgrad_halo_body with the products of
// wgrad_x3_body, for the 3x3 / stride 1 / pad 1 layers (13 of a ResNet-18 encoder's 20 convs, 85 % of its weight-gradient FLOPs).
// wgrad_x3_body splits every gathered operand value once per filter tap and output tile -- (BM + BN) * 32 conversions per 6 * BM * BN * 32
// MACs, the conversion VALU and the LDS writes bound it at 75 TFLOP/s.  Here a workgroup owns 64 output channels x 64 input channels x
// all 9 taps: per reduction tile of 32 output pixels (a TH x TW patch) it loads the 32 dY rows and the (TH + 2) x (TW + 2) x 64-channel
// input halo ONCE, splits them ONCE into hi / mid / lo planes ([pixel][plane][64 channels] bf16, 384 B per pixel, 16-B pieces
// XOR-swizzled by 2 * (pixel & 3)), and the nine taps read shifted windows of the halo planes with `ds_read_b64_tr_b16` (the halo row
// pitch is padded to a multiple of 4 pixels, so a tap shift moves every lane's swizzle class alike and ONE address per tap serves
// the tile's four transposing reads through immediate offsets): 108 bf16 MFMAs per wave and tile against ~9 global loads, ~200
// conversion VALU and 27 ds_write_b64 per thread.  One LDS stage, two workgroups per CU (the other workgroup multiplies while this one
// converts); the next tile's global loads fly under the MFMAs.  fp32-equivalent accuracy as wgrad_x3_body (same six products,
// smallest first).
template <int TW>
struct X3H {
    static constexpr int TH = 32 / TW, HWD = TW + 2, HHT = TH + 2, HP = HWD * HHT;
    static constexpr int HWDP = (HWD + 3) & ~3;                  // halo row pitch in pixels (multiple of 4)
    static constexpr int NB = (HP + 15) / 16;                    // loader passes over the halo (16 pixels x 16 float4 per pass)
    static constexpr int PIX = 384;                              // bytes per pixel: 3 planes x 64 bf16
    static constexpr int ABYTES = 32 * PIX, HBYTES = HHT * HWDP * PIX, LDS = ABYTES + HBYTES;
    static constexpr int HOFF = (TW >= 16 ? (TW == 16 ? HWDP : 16) : 2 * HWDP) * PIX;      // byte step of 16 output pixels inside the halo image
};
template <int TW>
__device__ __forceinline__ void wgrad_x3h_body(const WgradDesc& p, const int tile_id, const int split, unsigned char* smem) {
    typedef X3H<TW> G;
    constexpr int TH = G::TH, HWD = G::HWD, HP = G::HP, HWDP = G::HWDP, NB = G::NB, PIX = G::PIX;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int Cin = p.C1;
    const int cin_blocks = Cin >> 6;
    const int m0 = (tile_id / cin_blocks) * 64, c0 = (tile_id % cin_blocks) * 64;
    const int nrt = p.M >> 5;
    const int rt_begin = split * p.rtiles_per_split;
    const int rt_end = min(nrt, rt_begin + p.rtiles_per_split);
    const int tiles_x = p.OW / TW, tiles_img = (p.OH / TH) * tiles_x;
    const float* zline = g_zero_line;
    const int c4 = tid & 15, prow = tid >> 4;                   // float4 within a 64-float row; row / pixel within a loader pass
    unsigned char* As = smem;
    unsigned char* Hs = smem + G::ABYTES;

    // loader: 2 float4 of dY (rows prow, prow + 16) and NB float4 of the halo (pixels 16 q + prow) per thread and tile
    int h_slot[NB];                                              // LDS pixel slot of the thread's halo pixels (-1: past the halo)
    int h_dy[NB], h_dx[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int hp = q * 16 + prow;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        h_slot[q] = hp < HP ? hy * HWDP + hx : -1;
        h_dy[q] = hy - 1;
        h_dx[q] = hx - 1;
    }
    auto load = [&](int rt, f32x4 (&ra)[2], f32x4 (&rh)[NB]) {
        const size_t r0 = (size_t)rt * 32;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            ra[j] = *(const gf32x4w_t*)(uint64_t)(p.dy + (r0 + prow + j * 16) * p.Cout + m0 + c4 * 4);
        const int img = rt / tiles_img, t = rt - img * tiles_img;
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int ih0 = ty * TH, iw0 = tx * TW;
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int ih = ih0 + h_dy[q], iw = iw0 + h_dx[q];
            const bool ok = h_slot[q] >= 0 && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const float* g = ok ? p.x + ((size_t)(img * p.H + ih) * p.W + iw) * Cin + c0 + c4 * 4 : zline;
            rh[q] = *(const gf32x4w_t*)(uint64_t)g;
        }
    };
    // LDS position of the thread's 8-B piece (4 bf16 of one plane) of pixel / row `ps`: + 128 per plane
    auto piece = [&](int ps) { return ps * PIX + ((((c4 >> 1) ^ ((ps & 3) << 1)) << 4) | ((c4 & 1) << 3)); };
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    auto split_store = [&](const f32x4 (&ra)[2], const f32x4 (&rh)[NB]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            uint32_t h0, m0_, l0, h1, m1, l1;
            bsum += ra[j];
            split3_pair(ra[j][0], ra[j][1], h0, m0_, l0);
            split3_pair(ra[j][2], ra[j][3], h1, m1, l1);
            unsigned char* d = As + piece(prow + j * 16);
            *reinterpret_cast<uint2*>(d) = uint2{h0, h1};
            *reinterpret_cast<uint2*>(d + 128) = uint2{m0_, m1};
            *reinterpret_cast<uint2*>(d + 256) = uint2{l0, l1};
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            if (h_slot[q] < 0) continue;
            uint32_t h0, m0_, l0, h1, m1, l1;
            split3_pair(rh[q][0], rh[q][1], h0, m0_, l0);
            split3_pair(rh[q][2], rh[q][3], h1, m1, l1);
            unsigned char* d = Hs + piece(h_slot[q]);
            *reinterpret_cast<uint2*>(d) = uint2{h0, h1};
            *reinterpret_cast<uint2*>(d + 128) = uint2{m0_, m1};
            *reinterpret_cast<uint2*>(d + 256) = uint2{l0, l1};
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int wm = (wid >> 1) * 32, wn = (wid & 1) * 32;
    const int lr = lane & 31, lk = lane >> 5;
    // transposing-read source of this lane (see wgrad_tr_body): 16-lane group grp, segment seg; the lane addresses reduction row
    // 8 (grp >> 1) + (seg >> 2) (+ 4: second half, + 16: second k step) and the 4 columns 16 (grp & 1) + 4 (seg & 3) .. + 3
    const int grp = lane >> 4, seg = lane & 15;
    const int trow = 8 * (grp >> 1) + (seg >> 2);
    const int tcol = 16 * (grp & 1) + 4 * (seg & 3);
    const int acol = wm + tcol, bcol = wn + tcol;
    const int a_addr = trow * PIX + ((((acol >> 3) ^ ((trow & 3) << 1)) << 4) + ((acol & 7) << 1));
    // halo pixel (tap 0, 0) of reduction row trow: patch row trow / TW, patch column trow % TW
    const int bpix0 = (trow / TW) * HWDP + (trow % TW);
    const int bcch = bcol >> 3, bsub = (bcol & 7) << 1;
    auto tr8 = [&](const unsigned char* q) -> bf16x8 {            // rows r .. r + 3 and r + 4 .. r + 7 of the lane's column
        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lbf16x4_t*)(q));
        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lbf16x4_t*)(q + 4 * PIX));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto compute = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8 a[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) a[q] = tr8(As + a_addr + h * 16 * PIX + q * 128);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int pix = bpix0 + kh * HWDP + kw;
                    const unsigned char* bq = Hs + pix * PIX + (((bcch ^ ((pix & 3) << 1)) << 4) + bsub) + h * G::HOFF;
                    bf16x8 b[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) b[q] = tr8(bq + q * 128);
                    f32x16 c = acc[kh * 3 + kw];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);     // lo  * hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);     // hi  * lo
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);     // mid * mid
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);     // mid * hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);     // hi  * mid
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);     // hi  * hi
                    acc[kh * 3 + kw] = c;
                }
        }
    };

    f32x4 ra[2], rh[NB];
    if (rt_begin < rt_end) {
        load(rt_begin, ra, rh);
        split_store(ra, rh);
    }
    __syncthreads();
    for (int rt = rt_begin; rt < rt_end; ++rt) {
        const bool more = rt + 1 < rt_end;
        if (more) load(rt + 1, ra, rh);                        // in flight under the MFMAs of tile rt
        compute();
        __syncthreads();                                        // every wave is done reading tile rt
        if (more) split_store(ra, rh);
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int k = t * Cin + c0 + wn + lr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (p.splits > 1) p.partial[((size_t)split * p.Cout + co) * p.K + k] = acc[t][r];
            else wgrad_store(p, co, k, acc[t][r]);
        }
    }
    if ((p.dbias != nullptr) && c0 == 0) {                      // (past the loop's last barrier: the stage is free)
        float* red = reinterpret_cast<float*>(smem);            // [16][64] floats
#pragma unroll
        for (int j = 0; j < 4; ++j) red[prow * 64 + c4 * 4 + j] = bsum[j];
        __syncthreads();
        if (tid < 64) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) t += red[g * 64 + tid];
            const int co = m0 + tid;
            if (p.splits > 1) p.partial[(size_t)p.splits * p.Cout * p.K + (size_t)split * p.Cout + co] = t;
            else p.dbias[co] = p.accumulate ? p.dbias[co] + t : t;
        }
    }
}

// bf16-MFMA weight gradient fed from bf16 TWINS of the operands (x_h / dy_h: the rounded copies the bf16 forward / data-gradient convs
// already made), instead of converting fp32 while staging: the kernel above is bound by operand traffic through the L2 fabric (32 KB
// per workgroup per 256 MFMA cycles), so half the bytes is the lever.  Loader thread = 4 reduction rows x 8 channels (one 16-B load per
// row), transposed into the [channel][32 rows + pad] LDS image with eight 8-B stores; MFMA loop, split slabs and epilogue as above.
template <int BM, int BN, bool F16>
__global__ __launch_bounds__(256) void conv_wgrad_bf16h(const WgradDesc p) {
    constexpr int BKT = 32, LDH = 40;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    __shared__ __attribute__((aligned(16))) uint16_t As[2][BM * LDH];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[2][BN * LDH];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = (p.K + BN - 1) / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
    const int split = blockIdx.y;
    const int Cin = p.C1;
    const int nrt = (p.M + BKT - 1) / BKT;
    const int rt_begin = split * p.rtiles_per_split;
    const int rt_end = min(nrt, rt_begin + p.rtiles_per_split);
    const bool isA = tid < BM;
    const bool isB = !isA && tid < BM + BN;
    const int t2 = isA ? tid : tid - BM;
    const int c8 = isA ? t2 % (BM / 8) : t2 % (BN / 8);
    const int rgrp = isA ? t2 / (BM / 8) : t2 / (BN / 8);          // 0..7: rows rgrp*4 .. +3 of the tile
    int b_kh = 0, b_kw = 0, b_c = 0;
    bool b_kok = false;
    if (isB) {
        const int k = n0 + c8 * 8;
        b_kok = k < p.K;
        const int kk = b_kok ? k : 0;
        const int tap = kk / Cin;
        b_c = kk - tap * Cin;
        b_kh = tap / p.KW;
        b_kw = tap - b_kh * p.KW;
    }
    const bool do_bias = (p.dbias != nullptr) && (blockIdx.x % tiles_n == 0);
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint4 rv[4];
    auto load_tile = [&](int rt) {
        const int r0 = rt * BKT + rgrp * 4;
        if (isA) {
            const int co = m0 + c8 * 8;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = r0 + e;
                uint4 v = {0u, 0u, 0u, 0u};
                if (r < p.M && co < p.Cout) v = *reinterpret_cast<const uint4*>(p.dyh + (size_t)r * p.Cout + co);
                rv[e] = v;
                if (do_bias) {
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        bsum[2 * q] += __uint_as_float(w[q] << 16);
                        bsum[2 * q + 1] += __uint_as_float(w[q] & 0xffff0000u);
                    }
                }
            }
        } else if (isB) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = r0 + e;
                uint4 v = {0u, 0u, 0u, 0u};
                if (r < p.M && b_kok) {
                    const int t = (int)fdiv((uint32_t)r, p.fd_ow);
                    const int ow = r - t * p.OW;
                    const int img = (int)fdiv((uint32_t)t, p.fd_oh);
                    const int oh = t - img * p.OH;
                    int ih = oh * p.sh - p.ph + b_kh, iw = ow * p.sw - p.pw + b_kw;
                    bool ok = ih >= 0 && ih < p.HL && iw >= 0 && iw < p.WL;
                    if (p.idil > 1) {
                        ok = ok && (ih % p.idil == 0) && (iw % p.idil == 0);
                        ih /= p.idil;
                        iw /= p.idil;
                    }
                    if (p.ups) { ih >>= 1; iw >>= 1; }
                    if (ok) v = *reinterpret_cast<const uint4*>(p.xh + (((size_t)img * p.H + ih) * p.W + iw) * Cin + b_c);
                }
                rv[e] = v;
            }
        }
    };
    auto store_tile = [&](int buf) {
        if (isA || isB) {
            uint16_t* dst = (isA ? &As[buf][0] : &Bs[buf][0]) + (c8 * 8) * LDH + rgrp * 4;
            const uint32_t w[4][4] = {{rv[0].x, rv[0].y, rv[0].z, rv[0].w}, {rv[1].x, rv[1].y, rv[1].z, rv[1].w},
                                      {rv[2].x, rv[2].y, rv[2].z, rv[2].w}, {rv[3].x, rv[3].y, rv[3].z, rv[3].w}};
#pragma unroll
            for (int q = 0; q < 4; ++q) {            // channel pair (2q, 2q+1): low / high halves of word q of every row
                uint2 lo = {(w[0][q] & 0xffffu) | (w[1][q] << 16), (w[2][q] & 0xffffu) | (w[3][q] << 16)};
                uint2 hi = {(w[0][q] >> 16) | (w[1][q] & 0xffff0000u), (w[2][q] >> 16) | (w[3][q] & 0xffff0000u)};
                *reinterpret_cast<uint2*>(dst + (2 * q) * LDH) = lo;
                *reinterpret_cast<uint2*>(dst + (2 * q + 1) * LDH) = hi;
            }
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
    if (rt_begin < rt_end) {
        load_tile(rt_begin);
        store_tile(0);
    }
    __syncthreads();
    int buf = 0;
    for (int rt = rt_begin; rt < rt_end; ++rt) {
        const bool more = (rt + 1) < rt_end;
        if (more) load_tile(rt + 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(&As[buf][(wm + i * 32 + lr) * LDH + 16 * h + 8 * lk]);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8*>(&Bs[buf][(wn + j * 32 + lr) * LDH + 16 * h + 8 * lk]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = v2a_mfma_h<F16>(a[i], b[j], acc[i][j]);
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int k = n0 + wn + j * 32 + lr;
            if (k >= p.K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (co >= p.Cout) continue;
                if (p.splits > 1) p.partial[((size_t)split * p.Cout + co) * p.K + k] = acc[i][j][r];
                else wgrad_store(p, co, k, acc[i][j][r]);
            }
        }
    if (do_bias) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(&As[0][0]);        // [8][BM] floats (4 KB of the 20 KB image)
        if (isA) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red[rgrp * BM + c8 * 8 + j] = bsum[j];
        }
        __syncthreads();
        if (tid < BM) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) t += red[g * BM + tid];
            const int co = m0 + tid;
            if (co < p.Cout) {
                if (p.splits > 1) p.partial[(size_t)p.splits * p.Cout * p.K + (size_t)split * p.Cout + co] = t;
                else p.dbias[co] = p.accumulate ? p.dbias[co] + t : t;
            }
        }
    }
}

// bf16 weight gradient with LDS-DMA staging and the gfx950 transposing LDS read.  Both operands are k-major in HBM (rows = reduction
// rows, 128 channels contiguous), which is what `global_load_lds` can deliver (lane-linear 16-B pieces) but not what the bf16 MFMA
// wants (8 consecutive reduction rows per lane).  `ds_read_b64_tr_b16` closes the gap: per 16-lane group it takes sixteen 8-B row
// segments and hands lane i the i-th 16-bit column of the 4 x 16 block they form (measured: tools/probes/tr_read_probe.hip --
// result(lane i, elem j) = segment[4j + (i >> 2)][i & 3]).  So source lane s of a group points at row (s >> 2), columns 4 (s & 3) ..
// + 3 of a [4 rows][16 columns] block and every lane receives 4 consecutive reduction rows of ITS column; two reads make the 8-row
// MFMA operand.  LDS image: [64 rows][128 columns] bf16 per operand (256-B rows), 16-B pieces XOR-swizzled by 4 * (row & 3) at the
// DMA source so that the four rows of a block sit in different 64-B bank quarters.  Single buffer (32 KB, 4 workgroups per CU) like
// the forward kernel; split slabs / bias partials / reduce kernel shared with the other weight-gradient kernels.
template <int BM, bool F16>
__device__ __forceinline__ void wgrad_tr_body(const WgradDesc& p, const int tile_id, const int split, unsigned char* smem) {
    // tile: BM (128 | 64) output channels x 128 k' columns, 64 reduction rows; BM = 64 serves the 64-channel layers (one wave row less
    // per workgroup column: waves 2 x 2 over 64 x 128, each 32 x 64)
    constexpr int BN = 128, BKR = 64, PITCH = 256, PITCH_A = BM * 2;
    constexpr int ACH = BM / 8, AROWS = 256 / ACH, APASS = BKR / AROWS;     // 16-B pieces per A row, rows per DMA pass, passes
    constexpr int ABYTES = BKR * PITCH_A;
    constexpr int WMT = BM / 2, TM = WMT / 32;                                // rows per wave, 32-row MFMA tiles per wave
    constexpr int ASWZ = BM == 128 ? 4 : 2;      // piece XOR per (row & 3): conflict-free for 256-B / 128-B rows (enumerated bank model; PMC = 0)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = (p.K + BN - 1) / BN;
    const int m0 = (tile_id / tiles_n) * BM, n0 = (tile_id % tiles_n) * BN;
    const int Cin = p.C1 + p.C2;
    const int nrt = (p.M + BKR - 1) / BKR;
    const int rt_begin = split * p.rtiles_per_split;
    const int rt_end = min(nrt, rt_begin + p.rtiles_per_split);
    const uint16_t* zline = reinterpret_cast<const uint16_t*>(g_zero_line);
    // DMA slot of this thread in pass j: row j*16 + tid/16, piece position tid%16 holding source piece (pos ^ 4*(row & 3))
    const int prow = tid >> 4;
    const int piece = (tid & 15) ^ (4 * (prow & 3));
    const int arow = tid / ACH;                                   // A image: ACH pieces per row
    const int apiece = (tid % ACH) ^ (ASWZ * (arow & 3));
    const int a_co = m0 + apiece * 8;
    const bool a_cok = a_co < p.Cout;
    const int bk = n0 + piece * 8;
    const bool b_kok = bk < p.K;
    const int btap = (b_kok ? bk : 0) / Cin;
    const int bci = (b_kok ? bk : 0) - btap * Cin;
    const int bkh = btap / p.KW, bkw = btap - bkh * p.KW;
    const bool bfirst = bci < p.C1;                               // piece from x or from x2 (channel concat [x | x2]; C1 % 8 == 0)
    const uint16_t* bsrc = bfirst ? p.xh : p.x2h;
    const int bCs = bfirst ? p.C1 : p.C2, bcc = bfirst ? bci : bci - p.C1;

    auto issue = [&](int rt) {
        const int r0 = rt * BKR;
#pragma unroll
        for (int j = 0; j < APASS; ++j) {
            const int r = r0 + j * AROWS + arow;
            const uint16_t* g = (a_cok && r < p.M) ? p.dyh + (size_t)r * p.Cout + a_co : zline;
            __builtin_amdgcn_global_load_lds((gptr_w_t)g, (lptr_w_t)(smem + (j * 256 + wid * 64) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + j * 16 + prow;
            const uint32_t rr = r < p.M ? (uint32_t)r : 0u;
            const uint32_t t = fdiv(rr, p.fd_ow);
            const int ow = (int)(rr - t * p.OW);
            const uint32_t img = fdiv(t, p.fd_oh);
            const int oh = (int)(t - img * p.OH);
            int ih = oh * p.sh - p.ph + bkh, iw = ow * p.sw - p.pw + bkw;
            bool ok = b_kok && r < p.M && (unsigned)ih < (unsigned)p.HL && (unsigned)iw < (unsigned)p.WL;
            if (p.idil > 1) {
                ok = ok && (ih % p.idil == 0) && (iw % p.idil == 0);
                ih /= p.idil;
                iw /= p.idil;
            }
            if (p.ups) { ih >>= 1; iw >>= 1; }
            const uint16_t* g = ok ? bsrc + ((size_t)((int)img * p.H + ih) * p.W + iw) * bCs + bcc : zline;
            __builtin_amdgcn_global_load_lds((gptr_w_t)g, (lptr_w_t)(smem + ABYTES + (j * 256 + wid * 64) * 16), 16, 0, 0);
        }
    };

    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wm = (wid >> 1) * WMT, wn = (wid & 1) * 64;
    const int lr = lane & 31, lk = lane >> 5;
    // transposing-read source address of this lane: group g = lane >> 4, segment s = lane & 15
    const int grp = lane >> 4, seg = lane & 15;
    const int trow = 8 * (grp >> 1) + (seg >> 2);                 // + kk (+ 4 for the second half of the operand)
    const int tcol = 16 * (grp & 1) + 4 * (seg & 3);              // + wm / wn + 32 * i
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto tr_addr = [&](int col, int pitch, int swz) -> uint32_t {       // byte offset inside an operand image for (row trow, column col)
        return (uint32_t)(trow * pitch + (((col >> 3) ^ (swz * (trow & 3))) << 4) + ((col & 7) << 1));
    };
    uint32_t aoff[TM], boff[2];
#pragma unroll
    for (int i = 0; i < TM; ++i) aoff[i] = lds0 + tr_addr(wm + 32 * i + tcol, PITCH_A, ASWZ);
#pragma unroll
    for (int i = 0; i < 2; ++i) boff[i] = lds0 + ABYTES + tr_addr(wn + 32 * i + tcol, PITCH, 4);
    const bool do_bias = (p.dbias != nullptr) && (tile_id % tiles_n == 0);
    float bsum = 0.f;                                             // thread tid < 128: column sum of dY over this block's rows

    for (int rt = rt_begin; rt < rt_end; ++rt) {
        issue(rt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BKR; kk += 16) {
            // all transposing reads of a k step and the wait for them live in ONE asm statement: the compiler does not know that the
            // outputs of a `ds_read` asm arrive asynchronously and is free to copy such a register before a separate s_waitcnt
            uint2 al[2], ah[2], bl[2], bh[2];
            const uint32_t a0 = aoff[0] + kk * PITCH_A, b0 = boff[0] + kk * PITCH, b1 = boff[1] + kk * PITCH;
            if constexpr (TM == 2) {
                const uint32_t a1 = aoff[TM - 1] + kk * PITCH_A;
                asm volatile(
                    "ds_read_b64_tr_b16 %0, %8\n ds_read_b64_tr_b16 %1, %8 offset:%12\n"
                    "ds_read_b64_tr_b16 %2, %9\n ds_read_b64_tr_b16 %3, %9 offset:%12\n"
                    "ds_read_b64_tr_b16 %4, %10\n ds_read_b64_tr_b16 %5, %10 offset:%13\n"
                    "ds_read_b64_tr_b16 %6, %11\n ds_read_b64_tr_b16 %7, %11 offset:%13\n"
                    "s_waitcnt lgkmcnt(0)"
                    : "=&v"(al[0]), "=&v"(ah[0]), "=&v"(al[1]), "=&v"(ah[1]), "=&v"(bl[0]), "=&v"(bh[0]), "=&v"(bl[1]), "=&v"(bh[1])
                    : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "n"(4 * PITCH_A), "n"(4 * PITCH)
                    : "memory");
            } else {
                asm volatile(
                    "ds_read_b64_tr_b16 %0, %6\n ds_read_b64_tr_b16 %1, %6 offset:%9\n"
                    "ds_read_b64_tr_b16 %2, %7\n ds_read_b64_tr_b16 %3, %7 offset:%10\n"
                    "ds_read_b64_tr_b16 %4, %8\n ds_read_b64_tr_b16 %5, %8 offset:%10\n"
                    "s_waitcnt lgkmcnt(0)"
                    : "=&v"(al[0]), "=&v"(ah[0]), "=&v"(bl[0]), "=&v"(bh[0]), "=&v"(bl[1]), "=&v"(bh[1])
                    : "v"(a0), "v"(b0), "v"(b1), "n"(4 * PITCH_A), "n"(4 * PITCH)
                    : "memory");
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const uint4 ua = {al[i].x, al[i].y, ah[i].x, ah[i].y}, ub = {bl[j].x, bl[j].y, bh[j].x, bh[j].y};
                    acc[i][j] = v2a_mfma_h<F16>(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), acc[i][j]);
                }
        }
        if (do_bias && tid < BM) {
            const int ch = tid >> 3, within = tid & 7;
#pragma unroll 8
            for (int r = 0; r < BKR; ++r) {
                const uint16_t h = *reinterpret_cast<const uint16_t*>(smem + r * PITCH_A + ((ch ^ (ASWZ * (r & 3))) << 4) + within * 2);
                bsum += __uint_as_float((uint32_t)h << 16);
            }
        }
        __syncthreads();                                          // everyone is done reading before the next tile overwrites the buffer
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = n0 + wn + j * 32 + lr;
            if (k >= p.K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (co >= p.Cout) continue;
                if (p.splits > 1) p.partial[((size_t)split * p.Cout + co) * p.K + k] = acc[i][j][r];
                else wgrad_store(p, co, k, acc[i][j][r]);
            }
        }
    if (do_bias && tid < BM) {
        const int co = m0 + tid;
        if (co < p.Cout) {
            if (p.splits > 1) p.partial[(size_t)p.splits * p.Cout * p.K + (size_t)split * p.Cout + co] = bsum;
            else p.dbias[co] = p.accumulate ? p.dbias[co] + bsum : bsum;
        }
    }
}
template <int BM, bool F16>
__global__ __launch_bounds__(256, 4) void conv_wgrad_tr_h(const WgradDesc p) {
    __shared__ __attribute__((aligned(128))) unsigned char smem[64 * BM * 2 + 64 * 256];
    wgrad_tr_body<BM, F16>(p, (int)blockIdx.x, (int)blockIdx.y, smem);
}

// ---- MANY weight gradients in ONE launch.  A 4.8-GFLOP policy-step gradient alone pays ~25-40 us of ramp-up / drain around ~48 us
// of steady-state work (tools/wgrad_sweep.py: 61-70 TFLOP/s at every tile / split plan, against 104 on the video-training shapes with
// the same kernels); grouped, the ramps of all problems but the first and the last overlap.  The descriptors travel in the kernel
// arguments (no device table: a captured hipGraph keeps them in its kernel node), block -> (problem, split, tile) by a prefix scan.
// variant 0: exact-f32 64x64 LDS-DMA body, 1 / 2: bf16 twin-fed 128- / 64-row body.
#define WGM_MAX 16
struct WgradMultiArgs {
    int n;
    int wg_end[WGM_MAX];      // exclusive prefix of workgroups per problem
    int variant[WGM_MAX];
    int tiles[WGM_MAX];       // output tiles per problem (workgroups = tiles * splits)
    WgradDesc d[WGM_MAX];
};
static_assert(sizeof(WgradMultiArgs) <= 4096, "kernel arguments exceed the 4 KB segment");

__global__ __launch_bounds__(256, 4) void conv_wgrad_multi_kernel(const WgradMultiArgs a) {
    __shared__ __attribute__((aligned(128))) unsigned char smem[32768];
    const int bid = (int)blockIdx.x;
    int i = 0;
    while (i + 1 < a.n && bid >= a.wg_end[i]) ++i;
    const int first = i ? a.wg_end[i - 1] : 0;
    const int nwg = a.wg_end[i] - first, tiles = a.tiles[i];
    // XCD-contiguous order inside a problem: consecutive slots of one XCD walk the tiles of one reduction slice (they share its dY
    // and input rows through that XCD's L2)
    const int lin = nwg >= 8 ? xcd_remap(bid - first, nwg) : bid - first;
    const int split = lin / tiles, tile = lin - split * tiles;
    const int v = a.variant[i];
    if (v == 0) wgrad_dma_body<64, 64, 2>(a.d[i], tile, split, smem);
    else if (a.d[i].f16) {                                      // twin format of this problem (uniform): IEEE fp16 | bf16
        if (v == 1) wgrad_tr_body<128, true>(a.d[i], tile, split, smem);
        else wgrad_tr_body<64, true>(a.d[i], tile, split, smem);
    } else if (v == 1) wgrad_tr_body<128, false>(a.d[i], tile, split, smem);
    else wgrad_tr_body<64, false>(a.d[i], tile, split, smem);
}

// The same for the halo-tile body (3x3 / stride 1 / pad 1 layers, exact f32): a workgroup owns 64 output channels x (64 input
// channels x 9 taps) and DMAs 2-4 x fewer bytes per FLOP than the 64x64 body -- with both camera encoders' chains on the chip the
// L2 -> LDS path (about 25 GB/s per CU), not the matrix pipe, bounds the 64x64 tiles.  variant 3 / 4 / 5: patch width 32 / 16 / 8.
__global__ __launch_bounds__(256, 2) void conv_wgrad_multi_halo_kernel(const WgradMultiArgs a) {
    __shared__ __attribute__((aligned(128))) unsigned char smem[wgrad_halo_lds<32>()];
    const int bid = (int)blockIdx.x;
    int i = 0;
    while (i + 1 < a.n && bid >= a.wg_end[i]) ++i;
    const int first = i ? a.wg_end[i - 1] : 0;
    const int nwg = a.wg_end[i] - first, tiles = a.tiles[i];
    const int lin = nwg >= 8 ? xcd_remap(bid - first, nwg) : bid - first;
    const int split = lin / tiles, tile = lin - split * tiles;
    const int v = a.variant[i];
    if (v == 3) wgrad_halo_body<32>(a.d[i], tile, split, smem);
    else if (v == 4) wgrad_halo_body<16>(a.d[i], tile, split, smem);
    else wgrad_halo_body<8>(a.d[i], tile, split, smem);
}

// The same for the three-bf16-plane body: variant 6 = 64 (output channels) x 128 (k') tiles, 4 waves of 64 x 32, two workgroups per
// CU; variant 7 = 128 x 128 tiles, 8 waves of 64 x 32, one workgroup per CU (one tile shape per launch).
template <int BT>
__global__ __launch_bounds__(BT == 64 ? 256 : 512, BT == 64 ? 2 : 1) void conv_wgrad_multi_x3_kernel(const WgradMultiArgs a) {
    __shared__ __attribute__((aligned(128))) unsigned char smem[2 * 3 * 32 * (BT + 128) * 2];
    const int bid = (int)blockIdx.x;
    int i = 0;
    while (i + 1 < a.n && bid >= a.wg_end[i]) ++i;
    const int first = i ? a.wg_end[i - 1] : 0;
    const int nwg = a.wg_end[i] - first, tiles = a.tiles[i];
    const int lin = nwg >= 8 ? xcd_remap(bid - first, nwg) : bid - first;
    const int split = lin / tiles, tile = lin - split * tiles;
    const bool gen = a.d[i].ups || a.d[i].idil > 1;
    if constexpr (BT == 64) {
        if (gen) wgrad_x3_body<64, 128, 1, 4, true>(a.d[i], tile, split, smem);
        else wgrad_x3_body<64, 128, 1, 4, false>(a.d[i], tile, split, smem);
    } else {
        if (gen) wgrad_x3_body<128, 128, 2, 4, true>(a.d[i], tile, split, smem);
        else wgrad_x3_body<128, 128, 2, 4, false>(a.d[i], tile, split, smem);
    }
}

// The same for the three-plane halo body (3x3 / stride 1 / pad 1 layers): variant 8 / 9 / 10 = patch width 32 / 16 / 8.
__global__ __launch_bounds__(256, 2) void conv_wgrad_multi_x3h_kernel(const WgradMultiArgs a) {
    __shared__ __attribute__((aligned(128))) unsigned char smem[X3H<32>::LDS];
    const int bid = (int)blockIdx.x;
    int i = 0;
    while (i + 1 < a.n && bid >= a.wg_end[i]) ++i;
    const int first = i ? a.wg_end[i - 1] : 0;
    const int nwg = a.wg_end[i] - first, tiles = a.tiles[i];
    const int lin = nwg >= 8 ? xcd_remap(bid - first, nwg) : bid - first;
    const int split = lin / tiles, tile = lin - split * tiles;
    const int v = a.variant[i];
    if (v == 8) wgrad_x3h_body<32>(a.d[i], tile, split, smem);
    else if (v == 9) wgrad_x3h_body<16>(a.d[i], tile, split, smem);
    else wgrad_x3h_body<8>(a.d[i], tile, split, smem);
}

__device__ __forceinline__ void wgrad_reduce_body(const WgradDesc& p, const unsigned bid, const unsigned nblk) {
    const size_t total = (size_t)p.Cout * p.K;
    if ((p.K & 3) == 0) {
        // four consecutive k' (same output channel, same tap when Cin % 4 == 0) per thread as one 16-B load per slab, four slabs in
        // flight: the slabs are read once at close to HBM speed instead of one dependent 4-B load at a time
        const size_t total4 = total >> 2;
        for (size_t i = (size_t)bid * 256 + threadIdx.x; i < total4; i += (size_t)nblk * 256) {
            const float* src = p.partial + i * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            int sl = 0;
            for (; sl + 4 <= p.splits; sl += 4) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(src + (size_t)sl * total);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(src + (size_t)(sl + 1) * total);
                const f32x4 a2 = *reinterpret_cast<const f32x4*>(src + (size_t)(sl + 2) * total);
                const f32x4 a3 = *reinterpret_cast<const f32x4*>(src + (size_t)(sl + 3) * total);
                v += (a0 + a1) + (a2 + a3);
            }
            for (; sl < p.splits; ++sl) v += *reinterpret_cast<const f32x4*>(src + (size_t)sl * total);
            const size_t idx = i * 4;
            const int co = (int)(idx / p.K), k = (int)(idx - (size_t)co * p.K);
#pragma unroll
            for (int e = 0; e < 4; ++e) wgrad_store(p, co, k + e, v[e]);
        }
    } else {
        for (size_t idx = (size_t)bid * 256 + threadIdx.x; idx < total; idx += (size_t)nblk * 256) {
            float v = 0.f;
            for (int s = 0; s < p.splits; ++s) v += p.partial[(size_t)s * total + idx];
            const int co = (int)(idx / p.K), k = (int)(idx - (size_t)co * p.K);
            wgrad_store(p, co, k, v);
        }
    }
    if (p.dbias) {
        const float* bp = p.partial + (size_t)p.splits * total;
        for (int co = (int)(bid * 256 + threadIdx.x); co < p.Cout; co += (int)(nblk * 256)) {
            float v = 0.f;
            for (int s = 0; s < p.splits; ++s) v += bp[(size_t)s * p.Cout + co];
            p.dbias[co] = p.accumulate ? p.dbias[co] + v : v;
        }
    }
}

// Split reduce for filters with several taps: the slabs are k'-major ([Cout][tap][Cin]), the gradient is torch-major ([Cout][Cin][tap]).
// One workgroup owns (output channel, 64 input channels): slab reads are coalesced along Cin, the sums are transposed through LDS and
// leave as ONE contiguous run of 64 * taps floats -- instead of 4-B stores 4 * taps bytes apart.  Cin % 64 == 0, taps <= 64.
__global__ __launch_bounds__(256) void wgrad_splitk_reduce(const WgradDesc p) { wgrad_reduce_body(p, blockIdx.x, gridDim.x); }
__device__ __forceinline__ void wgrad_reduce_taps_body(const WgradDesc& p, const unsigned bid, float* tile) {
    const int Cin = p.C1 + p.C2, taps = p.KH * p.KW;
    const int cblocks = Cin >> 6;
    const int co = bid / cblocks, ci0 = (bid % cblocks) << 6;
    const size_t total = (size_t)p.Cout * p.K;
    const int n = 64 * taps;
    const int ldt = taps + 1 > 65 ? 65 : taps + 1;
    if ((total & 3) == 0 && ((uintptr_t)p.partial & 15) == 0) {
        // 16-B loads: thread = (tap, four consecutive input channels); same per-element summation order as the scalar form below
        for (int t = threadIdx.x; t < 16 * taps; t += 256) {
            const int tap = t >> 4, c4 = (t & 15) << 2;
            const float* src = p.partial + (size_t)co * p.K + (size_t)tap * Cin + ci0 + c4;
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0, v2 = v0, v3 = v0;
            int sl = 0;
            for (; sl + 4 <= p.splits; sl += 4) {
                v0 += *reinterpret_cast<const f32x4*>(src + (size_t)sl * total);
                v1 += *reinterpret_cast<const f32x4*>(src + (size_t)(sl + 1) * total);
                v2 += *reinterpret_cast<const f32x4*>(src + (size_t)(sl + 2) * total);
                v3 += *reinterpret_cast<const f32x4*>(src + (size_t)(sl + 3) * total);
            }
            for (; sl < p.splits; ++sl) v0 += *reinterpret_cast<const f32x4*>(src + (size_t)sl * total);
            const f32x4 r = (v0 + v1) + (v2 + v3);
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[(c4 + e) * ldt + tap] = r[e];
        }
    } else {
        for (int t = threadIdx.x; t < n; t += 256) {
            const int tap = t >> 6, cl = t & 63;
            const float* src = p.partial + (size_t)co * p.K + (size_t)tap * Cin + ci0 + cl;
            float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
            int sl = 0;
            for (; sl + 4 <= p.splits; sl += 4) {
                v0 += src[(size_t)sl * total];
                v1 += src[(size_t)(sl + 1) * total];
                v2 += src[(size_t)(sl + 2) * total];
                v3 += src[(size_t)(sl + 3) * total];
            }
            for (; sl < p.splits; ++sl) v0 += src[(size_t)sl * total];
            tile[cl * ldt + tap] = (v0 + v1) + (v2 + v3);
        }
    }
    __syncthreads();
    float* dst = p.dw + ((size_t)co * Cin + ci0) * taps;
    const int ld = taps + 1 > 65 ? 65 : taps + 1;
    for (int t = threadIdx.x; t < n; t += 256) {
        const int cl = t / taps, tap = t - cl * taps;
        const float v = tile[cl * ld + tap];
        dst[t] = p.accumulate ? dst[t] + v : v;
    }
    if (p.dbias && (bid % cblocks) == 0 && threadIdx.x == 0) {
        const float* bp = p.partial + (size_t)p.splits * total;
        float v = 0.f;
        for (int s = 0; s < p.splits; ++s) v += bp[(size_t)s * p.Cout + co];
        p.dbias[co] = p.accumulate ? p.dbias[co] + v : v;
    }
}
__global__ __launch_bounds__(256) void wgrad_splitk_reduce_taps(const WgradDesc p) {
    __shared__ float tile[64 * 65];
    wgrad_reduce_taps_body(p, blockIdx.x, tile);
}
// The reduces of MANY weight gradients in one launch (deterministic: same per-element summation order as the single kernels).
// descs: the WgradDesc of every deferred gradient (its `partial` = that layer's own slab buffer); work[b] = {desc, block within it,
// blocks of that desc, 1 = multi-tap transposing form}.
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const WgradDesc* descs, const int4* work) {
    __shared__ float tile[64 * 65];
    const int4 w = work[blockIdx.x];
    const WgradDesc p = descs[w.x];
    if (w.w) wgrad_reduce_taps_body(p, (unsigned)w.y, tile);
    else wgrad_reduce_body(p, (unsigned)w.y, (unsigned)w.z);
}
static bool wgrad_reduce_is_taps(const WgradDesc& p) {
    const int Cin = p.C1 + p.C2, taps = p.KH * p.KW;
    return taps > 1 && taps <= 64 && (Cin & 63) == 0;
}
static int wgrad_reduce_blocks(const WgradDesc& p) {
    if (wgrad_reduce_is_taps(p)) return p.Cout * ((p.C1 + p.C2) >> 6);
    size_t total = (size_t)p.Cout * p.K;
    if ((p.K & 3) == 0) total >>= 2;
    int g = (int)((total + 255) / 256);
    return g > 4096 ? 4096 : g;
}
// v2a_conv2d_wgrad_deferred: the launchers below hand the finished descriptor to this thread's collector instead of launching
struct WgradDeferred {
    WgradDesc desc;
    int blocks, taps_form;
};
static thread_local WgradDeferred* tl_wgrad_defer = nullptr;
static void launch_wgrad_reduce(const WgradDesc& p, hipStream_t stream) {
    if (tl_wgrad_defer) {
        // the reduce reads only partial / dw / dbias and the shape: hand out a canonical descriptor (everything else zero, padding
        // included) so that equal pending reduces give byte-equal items -- the caller caches its device tables by item contents
        WgradDesc& q = tl_wgrad_defer->desc;
        __builtin_memset(&q, 0, sizeof(WgradDesc));
        q.dw = p.dw; q.partial = p.partial; q.dbias = p.dbias;
        q.C1 = p.C1; q.C2 = p.C2; q.Cout = p.Cout; q.KH = p.KH; q.KW = p.KW; q.K = p.K; q.splits = p.splits; q.accumulate = p.accumulate;
        tl_wgrad_defer->blocks = wgrad_reduce_blocks(p);
        tl_wgrad_defer->taps_form = wgrad_reduce_is_taps(p) ? 1 : 0;
        return;
    }
    if (wgrad_reduce_is_taps(p)) hipLaunchKernelGGL(wgrad_splitk_reduce_taps, dim3(wgrad_reduce_blocks(p)), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(wgrad_splitk_reduce, dim3(wgrad_reduce_blocks(p)), dim3(256), 0, stream, p);
}

// ------------------------------------------------------------------------------------------------ weight packs
// mode 0 (forward pack):        dst[co][kh][kw][ci]    = src[co][ci][kh][kw]
// mode 1 (data-gradient pack):  dst[ci][kh'][kw'][co]  = src[co][ci][KH-1-kh'][KW-1-kw']
// mode 2 (channel-window pack): dst[co][kh][kw][ci] with KW + 1 columns of Cin + 1 channels (v2a_conv2d_fwd_window_f32; dst pre-zeroed)
__global__ void pack_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int KH, int KW, int mode) {
    const size_t total = (size_t)Cout * Cin * KH * KW;
    const int taps = KH * KW;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        if (mode == 0 || mode == 2) {
            int ci = (int)(idx % Cin);
            size_t t = idx / Cin;
            int tap = (int)(t % taps);
            int co = (int)(t / taps);
            const float v = src[((size_t)co * Cin + ci) * taps + tap];
            if (mode == 0) dst[idx] = v;
            else {                                   // channel-window pack: [Cout][KH][KW + 1][Cin + 1], added column / channel left as they are (zero)
                const int kh = tap / KW, kw = tap - kh * KW;
                dst[(((size_t)co * KH + kh) * (KW + 1) + kw) * (Cin + 1) + ci] = v;
            }
        } else {
            int co = (int)(idx % Cout);
            size_t t = idx / Cout;
            int tap = (int)(t % taps);
            int ci = (int)(t / taps);
            dst[idx] = src[((size_t)co * Cin + ci) * taps + (taps - 1 - tap)];
        }
    }
}

// all packs of a model in TWO launches.  table (int64) per operand: {src, dst, Cout, Cin, taps, mode, dst_bf16 or 0}.
// mode 0 (forward pack [Cout][taps][Cin]) and mode 2 (its channel-window form): chunks (int32) {operand, start element}, PACK_CHUNK
// elements per workgroup.
// mode 1 (data-gradient pack [Cin][taps reversed][Cout] = a [Cout x Cin*taps] -> [Cin*taps x Cout] transpose with the taps of
// each input channel reversed): chunks {operand, 64x64 tile index}; the tile goes through LDS so that both the global reads
// (contiguous along Cin*taps) and the global writes (contiguous along Cout) are full lines.
// A non-zero 7th column also writes the bf16 twin of the operand (the LDS-DMA kernels' weight operand).
#define PACK_CHUNK 16384
__device__ __forceinline__ uint16_t f2h_pack(float f, int f16) {
    return f16 ? v2a_f2h<true>(f) : v2a_f2bf(f);
}
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const int64_t* table, const int* chunks, const int f16) {
    const int t = chunks[2 * blockIdx.x], start = chunks[2 * blockIdx.x + 1];
    const float* src = reinterpret_cast<const float*>(table[t * 7 + 0]);
    float* dst = reinterpret_cast<float*>(table[t * 7 + 1]);
    uint16_t* dsth = reinterpret_cast<uint16_t*>(table[t * 7 + 6]);
    const int Cout = (int)table[t * 7 + 2], Cin = (int)table[t * 7 + 3], taps = (int)table[t * 7 + 4];
    const long long total = (long long)Cout * Cin * taps;
    const int cnt = (int)min((long long)PACK_CHUNK, total - start);
    // (co, tap, ci) of the thread's first element by one 32-bit decomposition, then carried forward in steps of 256 elements (the
    // first version divided 64-bit indices three times per element: ~200 instructions per 4 bytes moved)
    const unsigned idx0 = (unsigned)start + threadIdx.x;             // every operand has fewer than 2^31 elements
    unsigned q = idx0 / (unsigned)Cin;
    int ci = (int)(idx0 - q * (unsigned)Cin);
    int co = (int)(q / (unsigned)taps);
    int tap = (int)(q - (unsigned)co * (unsigned)taps);
    // mode 2 ("channel window" pack of a square filter, v2a_conv2d_fwd_window_f32): dst[co][kh][kw'][ci'] with KW + 1 columns and
    // Cin + 1 channels per column -- the added column / channel stay zero (the caller zeroes dst once)
    const int mode = (int)table[t * 7 + 5] & 255;
    const bool planes = ((int)table[t * 7 + 5] & 256) != 0;      // dsth = the hi plane of three bf16 planes, `total` elements apart (conv_p3 operands)
    int kwn = 1;
    while (kwn * kwn < taps) ++kwn;
    for (int i = threadIdx.x; i < cnt; i += 256) {
        size_t idx = (size_t)start + i;
        const float v = src[((size_t)co * Cin + ci) * taps + tap];
        if (mode == 2) {
            const int kh = tap / kwn, kw = tap - kh * kwn;
            idx = (((size_t)co * kwn + kh) * (kwn + 1) + kw) * (Cin + 1) + ci;
        }
        if (dst) dst[idx] = v;
        if (dsth) {
            if (planes) { unsigned short h, m, l; v2a_split3x1(v, h, m, l); dsth[idx] = h; dsth[total + idx] = m; dsth[2 * total + idx] = l; }
            else dsth[idx] = f2h_pack(v, f16);
        }
        ci += 256;
        while (ci >= Cin) {
            ci -= Cin;
            if (++tap == taps) { tap = 0; ++co; }
        }
    }
}
__global__ __launch_bounds__(256) void pack_weights_multi_t_kernel(const int64_t* table, const int* chunks, const int f16) {
    __shared__ float tile[64][65];
    const int t = chunks[2 * blockIdx.x], tl = chunks[2 * blockIdx.x + 1];
    const float* src = reinterpret_cast<const float*>(table[t * 7 + 0]);
    float* dst = reinterpret_cast<float*>(table[t * 7 + 1]);
    uint16_t* dsth = reinterpret_cast<uint16_t*>(table[t * 7 + 6]);
    const int Cout = (int)table[t * 7 + 2], Cin = (int)table[t * 7 + 3], taps = (int)table[t * 7 + 4];
    const bool planes = ((int)table[t * 7 + 5] & 256) != 0;
    const size_t total = (size_t)Cout * Cin * taps;
    const int J = Cin * taps, tj = (J + 63) >> 6;
    const int c0 = (tl / tj) << 6, j0 = (tl % tj) << 6;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {                       // read: row = co, 64 contiguous j
        const int co = c0 + r, j = j0 + tx;
        tile[r][tx] = (co < Cout && j < J) ? src[(size_t)co * J + j] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {                       // write: row = j (tap reversed inside its channel), 64 contiguous co
        const int j = j0 + r, co = c0 + tx;
        if (j < J && co < Cout) {
            const int ci = j / taps, tap = j - ci * taps;
            const size_t o = ((size_t)ci * taps + (taps - 1 - tap)) * Cout + co;
            const float v = tile[tx][r];
            if (dst) dst[o] = v;
            if (dsth) {
                if (planes) { unsigned short h, m, l; v2a_split3x1(v, h, m, l); dsth[o] = h; dsth[total + o] = m; dsth[2 * total + o] = l; }
                else dsth[o] = f2h_pack(v, f16);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int pick_split(int tiles, int ktiles, int min_ktiles) {
    // fill the 512 workgroup slots (2 per CU) in ONE round: the largest split with tiles * s <= 512 that keeps every slice
    // >= min_ktiles deep (a power-of-two split overshoots into a second, mostly empty round: 9 tiles x 64 = 576 workgroups)
    if (tiles >= 512) return 1;
    int smax = ktiles / min_ktiles;
    if (smax > 64) smax = 64;
    int s = 512 / tiles;
    if (s > smax) s = smax;
    return s < 1 ? 1 : s;
}

// one tile/split plan shared by the workspace query and the launcher (they must agree)
static int g_force_bm = 0, g_force_bn = 0;     // experiments only (v2a_debug_force_tile)
static void conv_plan(int M, int Cout, int K, int* bm, int* bn, int* tiles, int* s) {
    *bm = M >= 4096 ? 128 : 64;
    *bn = Cout > 64 ? (*bm == 128 ? 128 : 64) : 64;
    if (*bm == 64) *bn = 64;
    // few-hundred-tile problems (ResNet layers at B=64) run better on 64x64 tiles: more workgroups per CU (tools/tile_sweep.py)
    if (*bm == 128 && cdiv(M, 128) * cdiv(Cout, *bn) < 600) { *bm = 64; *bn = 64; }
    if (g_force_bm) { *bm = g_force_bm; *bn = g_force_bn; }
    *tiles = cdiv(M, *bm) * cdiv(Cout, *bn);
    *s = pick_split(*tiles, cdiv(K, 32), 4);      // split granularity in 32-deep k tiles (valid for both BKT)
}
static int g_wforce_bn = 0, g_wforce_s = 0;     // experiments only (v2a_debug_force_wgrad_plan)
static void wgrad_plan(int M, int Cout, int K, int* bm, int* bn, int* tiles, int* s) {
    if (g_wforce_bm) {
        *bm = g_wforce_bm; *bn = g_wforce_bn;
        *tiles = cdiv(Cout, *bm) * cdiv(K, *bn);
        int smax = cdiv(M, BK) / 2;
        *s = g_wforce_s < 1 ? 1 : (g_wforce_s > smax ? (smax < 1 ? 1 : smax) : g_wforce_s);
        return;
    }
    if (g_precision == 0) {
        // fp32 (tools/wgrad_sweep.py over the policy-step and video-training shapes): what matters is ~1000 workgroups in flight, not a
        // single round of 512 -- 128x128 tiles for big contractions (>= 30 GFLOP), 64x64 otherwise; the reduction over M is split
        // to reach the target as long as every slice keeps >= 8 row tiles.  (The fp32 slabs cost <= 70 MB of traffic.)
        const double flops = 2.0 * (double)M * (double)K * (double)Cout;
        const int tiles128 = cdiv(Cout, 128) * cdiv(K, 128);
        const bool big = Cout > 64 && K > 64 && flops >= 30e9 && tiles128 >= 8;
        *bm = *bn = big ? 128 : 64;
        *tiles = cdiv(Cout, *bm) * cdiv(K, *bn);
        const int target = 1024;                    // workgroups aimed at per launch (tools/wgrad_sweep.py: flat between 512 and 1536)
        int sp = target / *tiles;
        const int cap = big ? 128 : 64, deep = cdiv(M, 32) / 8;
        if (sp > cap) sp = cap;
        if (sp > deep) sp = deep;
        *s = sp < 1 ? 1 : sp;
        return;
    }
    // bf16 MFMA (tools/wgrad_sweep.py --bf16): the 128x128 bf16 kernel beats every alternative by 1.3-3x as soon as the reduction is
    // split to ~768 workgroups, for the policy-step and the video-training shapes alike; narrow outputs keep the plan below
    if (Cout > 64 && K > 64) {
        *bm = *bn = 128;
        *tiles = cdiv(Cout, 128) * cdiv(K, 128);
        int sp = 768 / *tiles;
        int deep = cdiv(M, 32) / 8;
        if (deep > 256) deep = 256;
        if (sp > deep) sp = deep;
        *s = sp < 1 ? 1 : sp;
        return;
    }
    // largest tile that still yields >= 192 workgroups; split the reduction only when even 64x64 tiles cannot fill the chip
    *bm = Cout > 64 ? 128 : 64;
    *bn = (K > 64 && *bm == 128) ? 128 : 64;
    *tiles = cdiv(Cout, *bm) * cdiv(K, *bn);
    // very few big tiles mean a very deep split (9 tiles -> 56 slices): the fp32 slabs (write + re-read) then cost as much as the
    // contraction.  Quartering the tiles quarters the split depth at nearly the same MFMA efficiency
    if (*tiles < 96 && *bm == 128) {
        *bm = 64;
        *bn = 64;
        *tiles = cdiv(Cout, 64) * cdiv(K, 64);
    }
    if (M < 4096) {     // short reduction: get parallelism from smaller tiles; long reductions keep big tiles and split instead
        if (*tiles < 192 && *bm == 128 && *bn == 128) { *bn = 64; *tiles = cdiv(Cout, *bm) * cdiv(K, *bn); }
        if (*tiles < 192 && *bm == 128) { *bm = 64; *bn = 64; *tiles = cdiv(Cout, *bm) * cdiv(K, *bn); }
    }
    *s = (*tiles >= 192) ? 1 : pick_split(*tiles, cdiv(M, BK), 4);
}

extern "C" int v2a_get_f32_conv_mode(void);
static int g_wgrad_dma = -1;  // fp32 weight gradients with 128-row output tiles on the LDS-DMA kernel (V2A_WGRAD_DMA=0 / v2a_debug_wgrad_dma)

// Direct convolution for reductions of at most 64 values (ConditionalUnet1D's first layers over the 7 action channels: Conv1d k = 5 -> K = 35,
// the 1 x 1 residual conv -> K = 7, the data gradient of the final 256 -> 7 conv -> K = 7; model/conditional_unet1d.py:137-160,198-201).
// The tile kernels gather such operands scalar by scalar (33-47 us per launch for 9 MFLOP, on the critical path of the captured step);
// here a workgroup stages the windows of `rows` output rows in LDS, a thread owns one output channel with its K weights in registers:
// exact fp32 FMA chains in k order, bias / row vector / residual as the tile kernels' epilogue.
static int g_smallk_on = 1;
template <int KMAX>
__global__ __launch_bounds__(256) void conv_smallk_kernel(const ConvDesc p, int rows) {
    __shared__ float xs[8][KMAX];
    const int co = blockIdx.y * 256 + threadIdx.x;
    const int m0 = blockIdx.x * rows;
    for (int i = threadIdx.x; i < rows * KMAX; i += 256) {
        const int r = i / KMAX, k = i - r * KMAX, m = m0 + r;
        float v = 0.f;
        if (m < p.M && k < p.K) {
            const int ci = k % p.C1, t = k / p.C1;
            const int kw = t % p.KW, kh = t / p.KW;
            const int ow = m % p.OW, t2 = m / p.OW;
            const int oh = t2 % p.OH, n = t2 / p.OH;
            const int ih = oh * p.sh - p.ph + kh, iw = ow * p.sw - p.pw + kw;
            if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) v = p.x[((size_t)(n * p.H + ih) * p.W + iw) * p.C1 + ci];
        }
        xs[r][k] = v;
    }
    float w[KMAX];
    const bool live = co < p.Cout;
    const float* wr = p.w + (size_t)(live ? co : 0) * p.K;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) w[k] = (k < p.K) ? wr[k] : 0.f;
    __syncthreads();
    if (!live) return;
    const float b = p.bias ? p.bias[co] : 0.f;
    for (int r = 0; r < rows; ++r) {
        const int m = m0 + r;
        if (m >= p.M) break;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) acc = fmaf(xs[r][k], w[k], acc);     // (k >= K: both factors are zero)
        float v = acc + b;
        const size_t o = (size_t)m * p.Cout + co;
        if (p.rowvec) v += p.rowvec[(size_t)(m / p.rows_per_batch) * p.Cout + co];
        if (p.residual) v += p.residual[o];
        p.y[o] = v;
    }
}
extern "C" {

// process-wide MFMA precision of the contraction kernels (0 = f32 exact, 1 = bf16 inputs / f32 accumulate); returns the old value
int v2a_set_precision(int mode) {
    int old = g_precision;
    if (mode == 0 || mode == 1) g_precision = mode;
    return old;
}
int v2a_get_precision(void) { return g_precision; }
// 16-bit format of the policy's MFMA mode (v2a_set_precision(1)): 0 bf16 (default), 1 IEEE fp16 -- the register-staged kernels round
// fp32 operands to it, the twin-fed kernels read twins in it, GroupNorm / the weight packs emit twins in it.  Returns the old value.
int v2a_set_policy_half(int f16) {
    const int old = g_v2a_policy_f16;
    if (f16 == 0 || f16 == 1) g_v2a_policy_f16 = f16;
    return old;
}
int v2a_get_policy_half(void) { return g_v2a_policy_f16; }
static int wgrad_dma_on() {
    if (g_wgrad_dma < 0) {
        g_wgrad_dma = 1;
    }
    return g_wgrad_dma;
}
int v2a_debug_wgrad_dma(int on) { int old = wgrad_dma_on(); g_wgrad_dma = on ? 1 : 0; return old; }
// tuning aid: force the forward tile (128x128, 128x64, 64x64) or 0,0 to restore the heuristic
int v2a_debug_force_tile(int bm, int bn) {
    if (!((bm == 0 && bn == 0) || (bm == 128 && (bn == 128 || bn == 64)) || (bm == 64 && bn == 64))) return V2A_ERR_ARG;
    g_force_bm = bm;
    g_force_bn = bn;
    return V2A_OK;
}

// tuning aid (tools/wgrad_sweep.py): force the weight-gradient tile and split; 0,0,0 restores the heuristic
int v2a_debug_force_wgrad_plan(int bm, int bn, int split) {
    if (!((bm == 0 && bn == 0) || (bm == 128 && (bn == 128 || bn == 64)) || (bm == 64 && bn == 64))) return V2A_ERR_ARG;
    g_wforce_bm = bm; g_wforce_bn = bn; g_wforce_s = split;
    return V2A_OK;
}
// the tile / split plan the launchers will use (for benchmarks that label kernels: bench.py)
int v2a_conv2d_plan(int M, int Cout, int K, int* bm, int* bn, int* split) {
    int tiles;
    conv_plan(M, Cout, K, bm, bn, &tiles, split);
    return V2A_OK;
}
int v2a_conv2d_wgrad_plan(int M, int Cout, int K, int* bm, int* bn, int* split) {
    int tiles;
    wgrad_plan(M, Cout, K, bm, bn, &tiles, split);
    return V2A_OK;
}

// workspace (bytes) a conv forward may need for split-K slabs (same plan as the launcher)
size_t v2a_conv2d_workspace_bytes(int M, int Cout, int K) {
    int bm, bn, tiles, s;
    conv_plan(M, Cout, K, &bm, &bn, &tiles, &s);
    return s > 1 ? (size_t)s * M * Cout * sizeof(float) : 0;
}

// Generic NHWC conv forward (also data-gradient / transposed conv via idil, upsample via ups, concat via x2).
// replaces: torch Conv2d/Conv1d/Linear/ConvTranspose1d calls of the reference hot path (see file header).
int v2a_debug_set_smallk(int on) {       // returns the old value; 0: the tile kernels take these layers (round-5 form; A/B hook); < 0: query only
    const int old = g_smallk_on;
    if (on >= 0) g_smallk_on = on ? 1 : 0;
    return old;
}

int v2a_conv2d_fwd(const float* x, const float* x2, const float* w_packed, const float* bias, const float* rowvec,
                   const float* residual, float* y, float* y2, int csplit, int N, int H, int W, int C1, int C2, int OH,
                   int OW, int Cout, int KH, int KW, int sh, int sw, int ph, int pw, int idil, int ups,
                   int rows_per_batch, int bmode, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!x || !w_packed || !y || N <= 0 || Cout <= 0) return V2A_ERR_ARG;
    if (C2 > 0 && !x2) return V2A_ERR_ARG;
    ConvDesc p;
    p.x = x; p.x2 = x2; p.w = w_packed; p.bias = bias; p.rowvec = rowvec; p.residual = residual;
    p.y = y; p.y2 = y2; p.csplit = csplit; p.partial = (float*)workspace;
    p.N = N; p.H = H; p.W = W; p.C1 = C1; p.C2 = C2; p.OH = OH; p.OW = OW; p.Cout = Cout;
    p.KH = KH; p.KW = KW; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.idil = idil < 1 ? 1 : idil; p.ups = ups;
    p.HL = ups ? 2 * H : (p.idil > 1 ? (H - 1) * p.idil + 1 : H);
    p.WL = ups ? 2 * W : (p.idil > 1 ? (W - 1) * p.idil + 1 : W);
    p.M = N * OH * OW;
    const int Cin = C1 + C2;
    p.K = KH * KW * Cin;
    p.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
    p.bmode = bmode;
    const bool vec = (Cin % 16 == 0) && (C1 % 4 == 0) && (((uintptr_t)x & 15) == 0) && (!x2 || ((uintptr_t)x2 & 15) == 0) &&
                     (((uintptr_t)w_packed & 15) == 0);
    // the N-major (data-gradient) loader needs whole 16-wide k tiles inside one tap and float4 columns
    if (bmode && !(vec && Cout % 4 == 0)) return V2A_ERR_ARG;
    if (g_smallk_on && !vec && !bmode && !x2 && C2 == 0 && !y2 && p.idil == 1 && !ups && p.K <= 64 && p.M <= (1 << 20)) {
        const int rows = p.M >= 2048 ? 8 : 4;
        const dim3 grid(cdiv(p.M, rows), cdiv(Cout, 256));
        if (p.K <= 8) hipLaunchKernelGGL((conv_smallk_kernel<8>), grid, dim3(256), 0, stream, p, rows);
        else hipLaunchKernelGGL((conv_smallk_kernel<64>), grid, dim3(256), 0, stream, p, rows);
        V2A_CHECK_LAUNCH();
        return V2A_OK;
    }
    int bm, bn, tiles, s;
    conv_plan(p.M, Cout, p.K, &bm, &bn, &tiles, &s);
    // 32-deep tiles (full 128-B lines per row segment, half the barriers) pay off on the smaller tiles; the 128x128 tile keeps
    // 16-deep tiles for occupancy (measured: tools/conv_bench.py)
    const bool k32 = vec && (Cin % 32 == 0) && !(bm == 128 && bn == 128);
    const int bkt = k32 ? 32 : 16;
    const int nkt = cdiv(p.K, bkt);
    if (s > 1 && (size_t)s * p.M * Cout * sizeof(float) > workspace_bytes) return V2A_ERR_WORKSPACE;
    p.splitk = s;
    p.ktiles_per_split = cdiv(nkt, s);
    dim3 grid(tiles, s), block(256);
    if (g_precision == 1 && vec && Cin % 32 == 0 && !p.bmode) {     // bf16 MFMA (data gradients then use the flipped pack, bmode 0)
        p.splitk = s;
        p.ktiles_per_split = cdiv(cdiv(p.K, 32), s);
        if (g_v2a_policy_f16) {
            if (bm == 128 && bn == 128) hipLaunchKernelGGL((conv_igemm_bf16<128, 128, true>), grid, block, 0, stream, p);
            else if (bm == 128) hipLaunchKernelGGL((conv_igemm_bf16<128, 64, true>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((conv_igemm_bf16<64, 64, true>), grid, block, 0, stream, p);
        } else if (bm == 128 && bn == 128) hipLaunchKernelGGL((conv_igemm_bf16<128, 128, false>), grid, block, 0, stream, p);
        else if (bm == 128) hipLaunchKernelGGL((conv_igemm_bf16<128, 64, false>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((conv_igemm_bf16<64, 64, false>), grid, block, 0, stream, p);
        V2A_CHECK_LAUNCH();
        if (s > 1) {
            size_t total = (size_t)p.M * Cout;
            int g = (int)((total + 255) / 256);
            if (g > 4096) g = 4096;
            hipLaunchKernelGGL(conv_splitk_reduce, dim3(g), dim3(256), 0, stream, p);
            V2A_CHECK_LAUNCH();
        }
        return V2A_OK;
    }
#define LAUNCH(BM_, BN_, PF_)                                                                                              \
    do {                                                                                                                   \
        if (p.bmode && k32) hipLaunchKernelGGL((conv_igemm_f32<BM_, BN_, 32, true, true, PF_>), grid, block, 0, stream, p);  \
        else if (p.bmode) hipLaunchKernelGGL((conv_igemm_f32<BM_, BN_, 16, true, true, PF_>), grid, block, 0, stream, p);    \
        else if (k32) hipLaunchKernelGGL((conv_igemm_f32<BM_, BN_, 32, true, false, PF_>), grid, block, 0, stream, p);       \
        else if (vec) hipLaunchKernelGGL((conv_igemm_f32<BM_, BN_, 16, true, false, PF_>), grid, block, 0, stream, p);       \
        else hipLaunchKernelGGL((conv_igemm_f32<BM_, BN_, 16, false, false, 1>), grid, block, 0, stream, p);                 \
    } while (0)
    if (bm == 128 && bn == 128) LAUNCH(128, 128, 1);
    else if (bm == 128) LAUNCH(128, 64, 2);
    else LAUNCH(64, 64, 3);
#undef LAUNCH
    V2A_CHECK_LAUNCH();
    if (s > 1) {
        size_t total = (size_t)p.M * Cout;
        int g = (int)((total + 255) / 256);
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(conv_splitk_reduce, dim3(g), dim3(256), 0, stream, p);
        V2A_CHECK_LAUNCH();
    }
    return V2A_OK;
}

size_t v2a_conv2d_wgrad_workspace_bytes(int M, int Cout, int K) {
    int bm, bn, tiles, s;
    wgrad_plan(M, Cout, K, &bm, &bn, &tiles, &s);
    if (wgrad_halo_shape_ok(M, Cout, K)) {          // the halo kernel may take this problem (geometry permitting): cover its split too
        const int sh = wgrad_halo_split(M, Cout, K);
        if (sh > s) s = sh;
    }
    return s > 1 ? ((size_t)s * Cout * K + (size_t)s * Cout) * sizeof(float) : 0;
}

// Weight gradient from the bf16 twins of the operands (bf16 MFMA, fp32 accumulate, dw in the torch layout as v2a_conv2d_wgrad).
// For layers the 128-row bf16 tiles take: Cout > 64, K > 64, Cin % 8 == 0, Cout % 8 == 0, single input source.
static int wgrad_h_split(int M, int Cout, int K) {
    const int target = 768;                         // workgroups aimed at (swept 384 ... 1536: tools/wgrad_sweep.py --bf16)
    const int tiles = cdiv(Cout, Cout <= 64 ? 64 : 128) * cdiv(K, 128);
    int s = target / tiles, deep = cdiv(M, 32) / 8;
    if (deep > 256) deep = 256;
    if (s > deep) s = deep;
    return s < 1 ? 1 : s;
}
size_t v2a_conv2d_wgrad_h_workspace_bytes(int M, int Cout, int K) {
    const int s = wgrad_h_split(M, Cout, K);
    return s > 1 ? ((size_t)s * Cout * K + (size_t)s * Cout) * sizeof(float) : 0;
}
int v2a_conv2d_wgrad_h(const void* x_h, const void* x2_h, const void* dy_h, float* dw, float* dbias, int N, int H, int W, int C, int C2, int OH,
                       int OW, int Cout, int KH, int KW, int sh, int sw, int ph, int pw, int idil, int ups, int accumulate, void* workspace,
                       size_t workspace_bytes, hipStream_t stream) {
    if (!x_h || !dy_h || !dw || Cout < 64 || C % 8 != 0 || Cout % 8 != 0 || C2 < 0 || C2 % 8 != 0 || (C2 > 0 && !x2_h)) return V2A_ERR_ARG;
    WgradDesc p = {};
    p.xh = (const uint16_t*)x_h; p.x2h = (const uint16_t*)x2_h; p.dyh = (const uint16_t*)dy_h; p.dw = dw; p.dbias = dbias;
    p.partial = (float*)workspace;
    p.N = N; p.H = H; p.W = W; p.C1 = C; p.C2 = C2; p.OH = OH; p.OW = OW; p.Cout = Cout;
    p.KH = KH; p.KW = KW; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.idil = idil < 1 ? 1 : idil; p.ups = ups;
    p.HL = ups ? 2 * H : (p.idil > 1 ? (H - 1) * p.idil + 1 : H);
    p.WL = ups ? 2 * W : (p.idil > 1 ? (W - 1) * p.idil + 1 : W);
    p.M = N * OH * OW;
    p.K = KH * KW * (C + C2);
    if (p.K <= 64) return V2A_ERR_ARG;
    p.accumulate = accumulate;
    p.f16 = g_v2a_policy_f16;
    p.fd_ow = make_fastdiv((uint32_t)OW);
    p.fd_oh = make_fastdiv((uint32_t)OH);
    const bool bm64 = Cout <= 64;                   // 64-channel layers: the 64 x 128 instance
    const int tiles = cdiv(Cout, bm64 ? 64 : 128) * cdiv(p.K, 128);
    const int s = wgrad_h_split(p.M, Cout, p.K);
    if (s > 1 && ((size_t)s * Cout * p.K + (size_t)s * Cout) * sizeof(float) > workspace_bytes) return V2A_ERR_WORKSPACE;
    p.splits = s;
    if ((((uintptr_t)x_h | (uintptr_t)x2_h | (uintptr_t)dy_h) & 15) == 0 && (double)N * H * W * (C > C2 ? C : C2) < 2147483648.0) {
        p.rtiles_per_split = cdiv(cdiv(p.M, 64), s);
        if (g_v2a_policy_f16) {
            if (bm64) hipLaunchKernelGGL((conv_wgrad_tr_h<64, true>), dim3(tiles, s), dim3(256), 0, stream, p);
            else hipLaunchKernelGGL((conv_wgrad_tr_h<128, true>), dim3(tiles, s), dim3(256), 0, stream, p);
        } else if (bm64) hipLaunchKernelGGL((conv_wgrad_tr_h<64, false>), dim3(tiles, s), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((conv_wgrad_tr_h<128, false>), dim3(tiles, s), dim3(256), 0, stream, p);
    } else {
        p.rtiles_per_split = cdiv(cdiv(p.M, 32), s);
        if (g_v2a_policy_f16) hipLaunchKernelGGL((conv_wgrad_bf16h<128, 128, true>), dim3(tiles, s), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((conv_wgrad_bf16h<128, 128, false>), dim3(tiles, s), dim3(256), 0, stream, p);
    }
    V2A_CHECK_LAUNCH();
    if (s > 1) {
        launch_wgrad_reduce(p, stream);
        V2A_CHECK_LAUNCH();
    }
    return V2A_OK;
}

// Weight gradient of the conv described by the same geometry arguments; dw is written in the TORCH layout
// [Cout][Cin][KH][KW] so it can be handed to autograd / the optimiser unchanged.
int v2a_conv2d_wgrad(const float* x, const float* x2, const float* dy, float* dw, float* dbias, int N, int H, int W, int C1, int C2,
                     int OH, int OW, int Cout, int KH, int KW, int sh, int sw, int ph, int pw, int idil, int ups,
                     int accumulate, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!x || !dy || !dw) return V2A_ERR_ARG;
    WgradDesc p;
    p.x = x; p.x2 = x2; p.dy = dy; p.dw = dw; p.dbias = dbias; p.partial = (float*)workspace;
    p.N = N; p.H = H; p.W = W; p.C1 = C1; p.C2 = C2; p.OH = OH; p.OW = OW; p.Cout = Cout;
    p.KH = KH; p.KW = KW; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.idil = idil < 1 ? 1 : idil; p.ups = ups;
    p.HL = ups ? 2 * H : (p.idil > 1 ? (H - 1) * p.idil + 1 : H);
    p.WL = ups ? 2 * W : (p.idil > 1 ? (W - 1) * p.idil + 1 : W);
    p.M = N * OH * OW;
    const int Cin = C1 + C2;
    p.K = KH * KW * Cin;
    p.accumulate = accumulate;
    p.f16 = g_v2a_policy_f16;
    p.fd_ow = make_fastdiv((uint32_t)OW);
    p.fd_oh = make_fastdiv((uint32_t)OH);
    const bool veca = (Cout % 4 == 0) && (((uintptr_t)dy & 15) == 0);
    const bool vecb = (Cin % 4 == 0) && (C1 % 4 == 0) && (((uintptr_t)x & 15) == 0) && (!x2 || ((uintptr_t)x2 & 15) == 0);
    // 3x3 / stride 1 / pad 1 layers whose output rows tile into 32-pixel patches: the halo kernel
    if (KH == 3 && KW == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 && !ups && p.idil == 1 && !x2 && C2 == 0 && OH == H && OW == W &&
        veca && vecb && wgrad_halo_shape_ok(p.M, Cout, p.K) && (OW == 8 || OW == 16 || OW % 32 == 0) && OH % (OW >= 32 ? 1 : 32 / OW) == 0 &&
        (double)N * H * W * C1 < 2147483648.0) {
        const int hs = wgrad_halo_split(p.M, Cout, p.K);
        if (hs > 1 && ((size_t)hs * Cout * p.K + (size_t)hs * Cout) * sizeof(float) > workspace_bytes) return V2A_ERR_WORKSPACE;
        p.splits = hs;
        p.rtiles_per_split = cdiv(p.M / 32, hs);
        dim3 hgrid((Cout / 64) * (C1 / 64), hs);
        if (OW == 8) hipLaunchKernelGGL((conv_wgrad_halo_f32<8>), hgrid, dim3(256), 0, stream, p);
        else if (OW == 16) hipLaunchKernelGGL((conv_wgrad_halo_f32<16>), hgrid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((conv_wgrad_halo_f32<32>), hgrid, dim3(256), 0, stream, p);
        V2A_CHECK_LAUNCH();
        if (hs > 1) {
            launch_wgrad_reduce(p, stream);
            V2A_CHECK_LAUNCH();
        }
        return V2A_OK;
    }
    int bm, bn, tiles, s;
    wgrad_plan(p.M, Cout, p.K, &bm, &bn, &tiles, &s);
    const int nrt = cdiv(p.M, BK);
    if (s > 1 && ((size_t)s * Cout * p.K + (size_t)s * Cout) * sizeof(float) > workspace_bytes) return V2A_ERR_WORKSPACE;
    // few-channel inputs (the RGB stem padded to 4 channels: K = 196, a reduction over N * OH * OW = 262 144 rows) in the fp32
    // three-plane mode: the three-plane body as a one-problem launch of the grouped kernel, with the plan's split (the exact LDS-DMA
    // kernel it replaces: 173 us per encoder, at the tail of the step's weight-gradient branch)
    if (g_precision == 0 && v2a_get_f32_conv_mode() == 1 && veca && vecb && Cin <= 8 && !x2 && Cout % 4 == 0 &&
        p.K % 4 == 0 && p.idil == 1 && !ups && (double)N * H * W * C1 < 1073741824.0) {
        WgradMultiArgs a;
        __builtin_memset(&a, 0, sizeof(a));
        const int big = Cout >= 128 ? 1 : 0;
        const int xt = cdiv(Cout, big ? 128 : 64) * cdiv(p.K, 128);
        const int nrt32 = cdiv(p.M, 32);
        int sx = s > nrt32 ? nrt32 : s;
        const int per = cdiv(nrt32, sx);
        sx = cdiv(nrt32, per);                        // no empty slices (sx <= s: the workspace check above covers it)
        p.splits = sx;
        p.rtiles_per_split = per;
        a.n = 1;
        a.d[0] = p;
        a.variant[0] = big ? 7 : 6;
        a.tiles[0] = xt;
        for (int i = 0; i < WGM_MAX; ++i) a.wg_end[i] = xt * sx;
        if (big) hipLaunchKernelGGL(conv_wgrad_multi_x3_kernel<128>, dim3(xt * sx), dim3(512), 0, stream, a);
        else hipLaunchKernelGGL(conv_wgrad_multi_x3_kernel<64>, dim3(xt * sx), dim3(256), 0, stream, a);
        V2A_CHECK_LAUNCH();
        if (sx > 1) {
            launch_wgrad_reduce(p, stream);
            V2A_CHECK_LAUNCH();
        }
        return V2A_OK;
    }
    p.splits = s;
    p.rtiles_per_split = cdiv(nrt, s);
    dim3 grid(tiles, s), block(256);
    // bf16 mode: the 64x64 problems run faster on the exact-f32 LDS-DMA kernel below than on the register-staged bf16 one
    const bool dma_ok = wgrad_dma_on() && veca && vecb && Cout % 4 == 0 && p.K % 4 == 0 && (C1 + C2) % 4 == 0 && C1 % 4 == 0 &&
                        (double)N * H * W * (C1 > C2 ? C1 : C2) < 4294967296.0;
    if (g_precision == 1 && veca && vecb && !(bm == 64 && dma_ok)) {
        p.splits = s;
        p.rtiles_per_split = cdiv(cdiv(p.M, 32), s);
        if (g_v2a_policy_f16) {
            if (bm == 128 && bn == 128) hipLaunchKernelGGL((conv_wgrad_bf16<128, 128, true>), grid, block, 0, stream, p);
            else if (bm == 128) hipLaunchKernelGGL((conv_wgrad_bf16<128, 64, true>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((conv_wgrad_bf16<64, 64, true>), grid, block, 0, stream, p);
        } else if (bm == 128 && bn == 128) hipLaunchKernelGGL((conv_wgrad_bf16<128, 128, false>), grid, block, 0, stream, p);
        else if (bm == 128) hipLaunchKernelGGL((conv_wgrad_bf16<128, 64, false>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((conv_wgrad_bf16<64, 64, false>), grid, block, 0, stream, p);
        V2A_CHECK_LAUNCH();
        if (s > 1) {
            launch_wgrad_reduce(p, stream);
            V2A_CHECK_LAUNCH();
        }
        return V2A_OK;
    }
    // long reductions over whole 16-B pieces: the LDS-DMA kernel (32-row tiles; the split plan is in tiles of BK = 16 rows)
    if (dma_ok) {
        p.rtiles_per_split = cdiv(cdiv(p.M, 32), s);
        if (bm == 128 && bn == 128) hipLaunchKernelGGL((conv_wgrad_dma_f32<128, 128>), grid, block, 0, stream, p);
        else if (bm == 128) hipLaunchKernelGGL((conv_wgrad_dma_f32<128, 64>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((conv_wgrad_dma_f32<64, 64, 2>), grid, block, 0, stream, p);
        V2A_CHECK_LAUNCH();
        if (s > 1) {
            launch_wgrad_reduce(p, stream);
            V2A_CHECK_LAUNCH();
        }
        return V2A_OK;
    }
#define LAUNCHW(BM_, BN_)                                                                                   \
    do {                                                                                                    \
        if (veca && vecb) hipLaunchKernelGGL((conv_wgrad_f32<BM_, BN_, true, true>), grid, block, 0, stream, p);   \
        else if (veca) hipLaunchKernelGGL((conv_wgrad_f32<BM_, BN_, true, false>), grid, block, 0, stream, p);     \
        else if (vecb) hipLaunchKernelGGL((conv_wgrad_f32<BM_, BN_, false, true>), grid, block, 0, stream, p);     \
        else hipLaunchKernelGGL((conv_wgrad_f32<BM_, BN_, false, false>), grid, block, 0, stream, p);              \
    } while (0)
    if (bm == 128 && bn == 128) LAUNCHW(128, 128);
    else if (bm == 128) LAUNCHW(128, 64);
    else LAUNCHW(64, 64);
#undef LAUNCHW
    V2A_CHECK_LAUNCH();
    if (s > 1) {
        launch_wgrad_reduce(p, stream);
        V2A_CHECK_LAUNCH();
    }
    return V2A_OK;
}

// ---- deferred reduces: run the weight-gradient kernel(s) now, sum the split slabs of MANY layers later in one launch.
// `slabs` is the layer's OWN scratch buffer (>= v2a_conv2d_wgrad_workspace_bytes / _h_workspace_bytes; it must stay untouched until
// v2a_wgrad_reduce_multi ran).  item_out (HOST, v2a_wgrad_item_bytes() bytes) receives the reduce descriptor, *blocks_out the number
// of workgroups it needs (0: the kernel wrote dw itself, nothing to reduce) and *form_out the kernel form (work[].w).
int v2a_wgrad_item_bytes(void) { return (int)sizeof(WgradDesc); }
int v2a_conv2d_wgrad_deferred(const float* x, const float* x2, const float* dy, float* dw, float* dbias, int N, int H, int W, int C1, int C2,
                              int OH, int OW, int Cout, int KH, int KW, int sh, int sw, int ph, int pw, int idil, int ups, int accumulate,
                              void* slabs, size_t slab_bytes, void* item_out, int* blocks_out, int* form_out, hipStream_t stream) {
    if (!item_out || !blocks_out || !form_out) return V2A_ERR_ARG;
    WgradDeferred d;
    d.blocks = 0;
    d.taps_form = 0;
    tl_wgrad_defer = &d;
    const int rc = v2a_conv2d_wgrad(x, x2, dy, dw, dbias, N, H, W, C1, C2, OH, OW, Cout, KH, KW, sh, sw, ph, pw, idil, ups, accumulate, slabs,
                                    slab_bytes, stream);
    tl_wgrad_defer = nullptr;
    *blocks_out = d.blocks;
    *form_out = d.taps_form;
    if (d.blocks > 0) *reinterpret_cast<WgradDesc*>(item_out) = d.desc;
    return rc;
}
int v2a_conv2d_wgrad_h_deferred(const void* x_h, const void* x2_h, const void* dy_h, float* dw, float* dbias, int N, int H, int W, int C1,
                                int C2, int OH, int OW, int Cout, int KH, int KW, int sh, int sw, int ph, int pw, int idil, int ups,
                                int accumulate, void* slabs, size_t slab_bytes, void* item_out, int* blocks_out, int* form_out,
                                hipStream_t stream) {
    if (!item_out || !blocks_out || !form_out) return V2A_ERR_ARG;
    WgradDeferred d;
    d.blocks = 0;
    d.taps_form = 0;
    tl_wgrad_defer = &d;
    const int rc = v2a_conv2d_wgrad_h(x_h, x2_h, dy_h, dw, dbias, N, H, W, C1, C2, OH, OW, Cout, KH, KW, sh, sw, ph, pw, idil, ups, accumulate,
                                      slabs, slab_bytes, stream);
    tl_wgrad_defer = nullptr;
    *blocks_out = d.blocks;
    *form_out = d.taps_form;
    if (d.blocks > 0) *reinterpret_cast<WgradDesc*>(item_out) = d.desc;
    return rc;
}
// items_dev: the collected descriptors (device copy, v2a_wgrad_item_bytes() each); work_dev: [nwork][4] int32 = {item, block, blocks, form}
int v2a_wgrad_reduce_multi(const void* items_dev, const void* work_dev, int nwork, hipStream_t stream) {
    if (!items_dev || !work_dev || nwork < 0) return V2A_ERR_ARG;
    if (nwork == 0) return V2A_OK;
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(nwork), dim3(256), 0, stream, (const WgradDesc*)items_dev, (const int4*)work_dev);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

// ---- grouped weight gradients (conv_wgrad_multi_kernel).  Step 1, v2a_conv2d_wgrad_describe: fill the descriptor of ONE gradient
// for the grouped launch without launching anything.  want_splits = 0: planning call -- only *variant_out (0 exact-f32 64x64 LDS-DMA
// body on the fp32 operands, 1 / 2 bf16 twin-fed 128- / 64-row body on x_h / dy_h, -1 not eligible: use v2a_conv2d_wgrad), *tiles_out
// (output tiles) and *rtiles_out (reduction tiles: 32 rows for variant 0, 64 for the twin-fed bodies) are written.  want_splits >= 1:
// `slabs` (the layer's own scratch, >= (splits * Cout * K + splits * Cout) * 4 bytes when splits > 1) is entered, item_out (HOST,
// v2a_wgrad_item_bytes()) receives the main-kernel descriptor, *splits_out the split actually used (capped so that every slice keeps
// work), and ritem_out / *rblocks_out / *rform_out the reduce item for v2a_wgrad_reduce_multi (rblocks 0: nothing to reduce).
// kernel family of a grouped-launch variant: 0 = 64x64 exact / twin-fed bodies (0-2), 1 = halo body (3-5), 2 / 3 = three-plane bodies (6 / 7),
// 4 = three-plane halo body (8-10)
static int wgrad_family(int v) { return v <= 2 ? 0 : (v <= 5 ? 1 : (v == 6 ? 2 : (v == 7 ? 3 : 4))); }
int v2a_wgrad_family(int variant) { return wgrad_family(variant); }
int v2a_conv2d_wgrad_describe(const float* x, const float* x2, const float* dy, const void* x_h, const void* x2_h, const void* dy_h, float* dw,
                              float* dbias, int N, int H, int W, int C1, int C2, int OH, int OW, int Cout, int KH, int KW, int sh, int sw, int ph,
                              int pw, int idil, int ups, int accumulate, int want_splits, void* slabs, size_t slab_bytes, void* item_out,
                              int* variant_out, int* tiles_out, int* rtiles_out, int* splits_out, void* ritem_out, int* rblocks_out,
                              int* rform_out) {
    if (!variant_out || !tiles_out || !rtiles_out) return V2A_ERR_ARG;
    WgradDesc p;
    __builtin_memset(&p, 0, sizeof(WgradDesc));
    p.x = x; p.x2 = x2; p.dy = dy; p.xh = (const uint16_t*)x_h; p.x2h = (const uint16_t*)x2_h; p.dyh = (const uint16_t*)dy_h;
    p.dw = dw; p.dbias = dbias; p.partial = (float*)slabs;
    p.N = N; p.H = H; p.W = W; p.C1 = C1; p.C2 = C2; p.OH = OH; p.OW = OW; p.Cout = Cout;
    p.KH = KH; p.KW = KW; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.idil = idil < 1 ? 1 : idil; p.ups = ups;
    p.HL = ups ? 2 * H : (p.idil > 1 ? (H - 1) * p.idil + 1 : H);
    p.WL = ups ? 2 * W : (p.idil > 1 ? (W - 1) * p.idil + 1 : W);
    p.M = N * OH * OW;
    const int Cin = C1 + C2;
    p.K = KH * KW * Cin;
    p.accumulate = accumulate;
    p.f16 = g_v2a_policy_f16;
    p.fd_ow = make_fastdiv((uint32_t)OW);
    p.fd_oh = make_fastdiv((uint32_t)OH);
    int variant = -1;
    const double big = (double)N * H * W * (C1 > C2 ? C1 : C2);
    if (x_h && dy_h && (C2 == 0 || x2_h) && Cout >= 64 && p.K > 64 && C1 % 8 == 0 && C2 % 8 == 0 && Cout % 8 == 0 &&
        (((uintptr_t)x_h | (uintptr_t)x2_h | (uintptr_t)dy_h) & 15) == 0 && big < 2147483648.0) {
        variant = Cout <= 64 ? 2 : 1;
    } else if (x && dy && (C2 == 0 || x2) && Cout % 4 == 0 && p.K % 4 == 0 && Cin % 4 == 0 && C1 % 4 == 0 &&
               (((uintptr_t)x | (uintptr_t)x2 | (uintptr_t)dy) & 15) == 0 && big < 4294967296.0) {
        variant = 0;
        // 3x3 / stride 1 / pad 1 over whole 32-pixel patches: the halo-tile body (V2A_WGRAD_HALO=0 keeps the 64x64 body)
        if (KH == 3 && KW == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 && !ups && p.idil == 1 && C2 == 0 && OH == H &&
            OW == W && C1 % 64 == 0 && Cout % 64 == 0 && p.M % 32 == 0 && (OW == 8 || OW == 16 || OW % 32 == 0) &&
            OH % (OW >= 32 ? 1 : 32 / OW) == 0 && big < 2147483648.0)
            variant = OW == 8 ? 5 : (OW == 16 ? 4 : 3);
        if (v2a_get_f32_conv_mode() == 1 && p.idil <= 2 && big < 1073741824.0) {       // fp32 products from three bf16 planes (V2A_WGRAD_X3=0: the exact bodies)
            const bool halo_geo = variant >= 3;       // (the exact halo body's geometry test above: 3x3 / stride 1 / pad 1, whole 32-pixel patches)
            if (halo_geo && OW >= 8) variant = OW == 8 ? 10 : (OW == 16 ? 9 : 8);
            else variant = Cout >= 128 ? 7 : 6;
        }
    }
    *variant_out = variant;
    if (variant < 0) { *tiles_out = 0; *rtiles_out = 0; return V2A_OK; }
    const int rrows = (variant == 1 || variant == 2) ? 64 : 32;
    const int tiles = variant == 0 ? cdiv(Cout, 64) * cdiv(p.K, 64)
                      : variant == 6 ? cdiv(Cout, 64) * cdiv(p.K, 128)
                      : variant == 7 ? cdiv(Cout, 128) * cdiv(p.K, 128)
                      : (variant >= 8 ? (Cout / 64) * (C1 / 64)
                      : (variant >= 3 ? (Cout / 64) * (C1 / 64) : cdiv(Cout, variant == 1 ? 128 : 64) * cdiv(p.K, 128)));
    const int nrt = cdiv(p.M, rrows);
    *tiles_out = tiles;
    *rtiles_out = nrt;
    if (want_splits < 1) return V2A_OK;
    if (!dw || !item_out || !splits_out || !ritem_out || !rblocks_out || !rform_out) return V2A_ERR_ARG;
    int s = want_splits > nrt ? nrt : want_splits;
    int per = cdiv(nrt, s);
    s = cdiv(nrt, per);                               // no empty slices
    if (s > 1 && (!slabs || ((size_t)s * Cout * p.K + (size_t)s * Cout) * sizeof(float) > slab_bytes)) return V2A_ERR_WORKSPACE;
    p.splits = s;
    p.rtiles_per_split = per;
    *splits_out = s;
    *reinterpret_cast<WgradDesc*>(item_out) = p;
    *rblocks_out = 0;
    *rform_out = 0;
    if (s > 1) {
        WgradDeferred d;
        tl_wgrad_defer = &d;
        launch_wgrad_reduce(p, nullptr);
        tl_wgrad_defer = nullptr;
        *reinterpret_cast<WgradDesc*>(ritem_out) = d.desc;
        *rblocks_out = d.blocks;
        *rform_out = d.taps_form;
    }
    return V2A_OK;
}
int v2a_wgrad_multi_max(void) { return WGM_MAX; }
// Step 2: launch n <= v2a_wgrad_multi_max() described gradients as one kernel.  items / variants / tiles: HOST arrays (items =
// n * v2a_wgrad_item_bytes() bytes as written by v2a_conv2d_wgrad_describe); workgroups are issued in array order (put the deepest
// reductions first).
int v2a_conv2d_wgrad_multi(const void* items, const int* variants, const int* tiles, int n, hipStream_t stream) {
    if (!items || !variants || !tiles || n < 1 || n > WGM_MAX) return V2A_ERR_ARG;
    WgradMultiArgs a;
    __builtin_memset(&a, 0, sizeof(a));
    a.n = n;
    int tot = 0;
    for (int i = 0; i < n; ++i) {
        a.d[i] = reinterpret_cast<const WgradDesc*>(items)[i];
        if (variants[i] < 0 || variants[i] > 10 || tiles[i] < 1 || a.d[i].splits < 1) return V2A_ERR_ARG;
        if (wgrad_family(variants[i]) != wgrad_family(variants[0])) return V2A_ERR_ARG;      // one kernel family per launch
        a.variant[i] = variants[i];
        a.tiles[i] = tiles[i];
        tot += tiles[i] * a.d[i].splits;
        a.wg_end[i] = tot;
    }
    for (int i = n; i < WGM_MAX; ++i) a.wg_end[i] = tot;
    const int fam = wgrad_family(variants[0]);
    if (fam == 4) hipLaunchKernelGGL(conv_wgrad_multi_x3h_kernel, dim3(tot), dim3(256), 0, stream, a);
    else if (fam == 3) hipLaunchKernelGGL(conv_wgrad_multi_x3_kernel<128>, dim3(tot), dim3(512), 0, stream, a);
    else if (fam == 2) hipLaunchKernelGGL(conv_wgrad_multi_x3_kernel<64>, dim3(tot), dim3(256), 0, stream, a);
    else if (fam == 1) hipLaunchKernelGGL(conv_wgrad_multi_halo_kernel, dim3(tot), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(conv_wgrad_multi_kernel, dim3(tot), dim3(256), 0, stream, a);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

int v2a_pack_chunk_elems(void) { return PACK_CHUNK; }
// packs of many tensors per launch; see pack_weights_multi_kernel for the table layout ({src, dst, Cout, Cin, taps, mode, dst_bf16}).
// transposed = 0: the mode-0 rows (chunks = {operand, start}); transposed = 1: the mode-1 rows (chunks = {operand, 64x64 tile}).
int v2a_pack_weights_multi(const int64_t* table_dev, const int* chunks_dev, int nchunks, int transposed, hipStream_t stream) {
    if (!table_dev || !chunks_dev || nchunks <= 0) return V2A_ERR_ARG;
    if (transposed) hipLaunchKernelGGL(pack_weights_multi_t_kernel, dim3(nchunks), dim3(256), 0, stream, table_dev, chunks_dev, g_v2a_policy_f16);
    else hipLaunchKernelGGL(pack_weights_multi_kernel, dim3(nchunks), dim3(256), 0, stream, table_dev, chunks_dev, g_v2a_policy_f16);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

// torch-layout weight [Cout][Cin][KH][KW] -> packed operand (mode 0 forward, mode 1 data-gradient).
int v2a_pack_weight(const float* src, float* dst, int Cout, int Cin, int KH, int KW, int mode, hipStream_t stream) {
    if (!src || !dst) return V2A_ERR_ARG;
    size_t total = (size_t)Cout * Cin * KH * KW;
    int g = (int)((total + 255) / 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(g), dim3(256), 0, stream, src, dst, Cout, Cin, KH, KW, mode);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

}  // extern "C"
