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

#define BK 16
#define LDS_PAD 4
#define LDK 20

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

template <int BM, int BN, bool VEC, bool BNMAJ>
__global__ __launch_bounds__(256) void conv_igemm_f32(const ConvDesc p) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int AL = BM / 64, BL = BN / 64;   // float4 loads per thread per tile
    // row-major tiles [rows][16 k + 4 pad]: 80-B rows keep every float4 16-B aligned and make both the b128 store of a
    // loaded float4 and the b128 operand read (16 lanes x 4 words = all 64 banks) conflict-free.
    __shared__ __attribute__((aligned(16))) float As[2][BM][LDK];
    // BNMAJ (data gradient read from the forward pack): B rows are k, columns n contiguous -> k-major tile, b128 store, b32 reads
    __shared__ __attribute__((aligned(16))) float Bs[2][BNMAJ ? BK * (BN + LDS_PAD) : BN * LDK];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = (p.Cout + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int ntile = tiles_m * tiles_n;
    const int lin = xcd_remap(blockIdx.x, ntile);
    const int m0 = (lin / tiles_n) * BM, n0 = (lin % tiles_n) * BN;
    const int split = blockIdx.y;
    const int Cin = p.C1 + p.C2;
    const int nkt = (p.K + BK - 1) / BK;
    const int kt_begin = split * p.ktiles_per_split;
    const int kt_end = min(nkt, kt_begin + p.ktiles_per_split);

    // ---- per-thread loader state
    const int lrow = tid >> 2, chunk = tid & 3;
    int a_ihb[AL], a_iwb[AL], a_img[AL];
    bool a_ok[AL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        int m = m0 + lrow + i * 64;
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

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
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
                // rows kk = tid / (BN/4) + i * (1024/BN): k = k0 + kk = (tap, cred = c0 + kk); 4 consecutive n per thread
                const int taps = p.KH * p.KW;
                const int cbase = k0 - tap * Cin;
#pragma unroll
                for (int i = 0; i < BL; ++i) {
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
                    int n = n0 + lrow + i * 64;
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
                int n = n0 + lrow + i * 64;
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

    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AL; ++i) *reinterpret_cast<f32x4*>(&As[buf][lrow + i * 64][chunk * 4]) = ra[i];
        if (BNMAJ) {
#pragma unroll
            for (int i = 0; i < BL; ++i)
                *reinterpret_cast<f32x4*>(&Bs[buf][(tid / (BN / 4) + i * (1024 / BN)) * (BN + LDS_PAD) + (tid % (BN / 4)) * 4]) = rb[i];
        } else {
#pragma unroll
            for (int i = 0; i < BL; ++i) *reinterpret_cast<f32x4*>(&Bs[buf][(lrow + i * 64) * LDK + chunk * 4]) = rb[i];
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
        // one ds_read_b128 per operand tile feeds FOUR MFMA k-steps: in step (h, s) the lanes with lk = 0 supply k = 8h + s and
        // the lanes with lk = 1 supply k = 8h + 4 + s, for A and B alike (the k-order of an exact-f32 sum is free to choose).
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(&As[buf][wm + i * 32 + lr][8 * h + 4 * lk]);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (BNMAJ) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) b[j][q] = Bs[buf][(8 * h + 4 * lk + q) * (BN + LDS_PAD) + wn + j * 32 + lr];
                } else {
                    b[j] = *reinterpret_cast<const f32x4*>(&Bs[buf][(wn + j * 32 + lr) * LDK + 8 * h + 4 * lk]);
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
        if (more) store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
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
    float* dw;                         // torch layout [Cout][Cin][KH][KW]  (or [Cin][Cout][KH][KW]-free: see transposed)
    float* partial;                    // [splits][Cout][K']
    int N, H, W, C1, C2, OH, OW, Cout, KH, KW, sh, sw, ph, pw, idil, ups, HL, WL, M, K;
    int splits, rtiles_per_split;
    int accumulate;                    // 1: dw += result
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
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            const int r = r0 + b_r + i * BROWS;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < p.M) {
                const int ow = r % p.OW;
                const int t = r / p.OW;
                const int oh = t % p.OH;
                const int img = t / p.OH;
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
}

__global__ void wgrad_splitk_reduce(const WgradDesc p) {
    const size_t total = (size_t)p.Cout * p.K;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int s = 0; s < p.splits; ++s) v += p.partial[(size_t)s * total + idx];
        const int co = (int)(idx / p.K), k = (int)(idx - (size_t)co * p.K);
        wgrad_store(p, co, k, v);
    }
}

// ------------------------------------------------------------------------------------------------ weight packs
// mode 0 (forward pack):        dst[co][kh][kw][ci]    = src[co][ci][kh][kw]
// mode 1 (data-gradient pack):  dst[ci][kh'][kw'][co]  = src[co][ci][KH-1-kh'][KW-1-kw']
__global__ void pack_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int KH, int KW, int mode) {
    const size_t total = (size_t)Cout * Cin * KH * KW;
    const int taps = KH * KW;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        if (mode == 0) {
            int ci = (int)(idx % Cin);
            size_t t = idx / Cin;
            int tap = (int)(t % taps);
            int co = (int)(t / taps);
            dst[idx] = src[((size_t)co * Cin + ci) * taps + tap];
        } else {
            int co = (int)(idx % Cout);
            size_t t = idx / Cout;
            int tap = (int)(t % taps);
            int ci = (int)(t / taps);
            dst[idx] = src[((size_t)co * Cin + ci) * taps + (taps - 1 - tap)];
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int pick_split(int tiles, int ktiles, int min_ktiles) {
    int s = 1;
    // aim for >= 512 workgroups (2 per CU) while keeping every split >= min_ktiles deep
    while (tiles * s < 512 && ktiles / (s * 2) >= min_ktiles && s < 64) s *= 2;
    return s;
}

// one tile/split plan shared by the workspace query and the launcher (they must agree)
static void conv_plan(int M, int Cout, int K, int* bm, int* bn, int* tiles, int* s) {
    *bm = M >= 4096 ? 128 : 64;
    *bn = Cout > 64 ? (*bm == 128 ? 128 : 64) : 64;
    if (*bm == 64) *bn = 64;
    *tiles = cdiv(M, *bm) * cdiv(Cout, *bn);
    *s = pick_split(*tiles, cdiv(K, BK), 8);
}
static void wgrad_plan(int M, int Cout, int K, int* bm, int* bn, int* tiles, int* s) {
    *bm = Cout > 64 ? 128 : 64;
    *bn = (K > 64 && *bm == 128) ? 128 : 64;
    *tiles = cdiv(Cout, *bm) * cdiv(K, *bn);
    *s = pick_split(*tiles, cdiv(M, BK), 4);
}

extern "C" {

// workspace (bytes) a conv forward may need for split-K slabs (same plan as the launcher)
size_t v2a_conv2d_workspace_bytes(int M, int Cout, int K) {
    int bm, bn, tiles, s;
    conv_plan(M, Cout, K, &bm, &bn, &tiles, &s);
    return s > 1 ? (size_t)s * M * Cout * sizeof(float) : 0;
}

// Generic NHWC conv forward (also data-gradient / transposed conv via idil, upsample via ups, concat via x2).
// replaces: torch Conv2d/Conv1d/Linear/ConvTranspose1d calls of the reference hot path (see file header).
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
    int bm, bn, tiles, s;
    conv_plan(p.M, Cout, p.K, &bm, &bn, &tiles, &s);
    const int nkt = cdiv(p.K, BK);
    if (s > 1 && (size_t)s * p.M * Cout * sizeof(float) > workspace_bytes) return V2A_ERR_WORKSPACE;
    p.splitk = s;
    p.ktiles_per_split = cdiv(nkt, s);
    dim3 grid(tiles, s), block(256);
#define LAUNCH(BM_, BN_)                                                                                        \
    do {                                                                                                        \
        if (p.bmode) hipLaunchKernelGGL((conv_igemm_f32<BM_, BN_, true, true>), grid, block, 0, stream, p);      \
        else if (vec) hipLaunchKernelGGL((conv_igemm_f32<BM_, BN_, true, false>), grid, block, 0, stream, p);   \
        else hipLaunchKernelGGL((conv_igemm_f32<BM_, BN_, false, false>), grid, block, 0, stream, p);           \
    } while (0)
    if (bm == 128 && bn == 128) LAUNCH(128, 128);
    else if (bm == 128) LAUNCH(128, 64);
    else LAUNCH(64, 64);
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
    return s > 1 ? (size_t)s * Cout * K * sizeof(float) : 0;
}

// Weight gradient of the conv described by the same geometry arguments; dw is written in the TORCH layout
// [Cout][Cin][KH][KW] so it can be handed to autograd / the optimiser unchanged.
int v2a_conv2d_wgrad(const float* x, const float* x2, const float* dy, float* dw, int N, int H, int W, int C1, int C2,
                     int OH, int OW, int Cout, int KH, int KW, int sh, int sw, int ph, int pw, int idil, int ups,
                     int accumulate, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!x || !dy || !dw) return V2A_ERR_ARG;
    WgradDesc p;
    p.x = x; p.x2 = x2; p.dy = dy; p.dw = dw; p.partial = (float*)workspace;
    p.N = N; p.H = H; p.W = W; p.C1 = C1; p.C2 = C2; p.OH = OH; p.OW = OW; p.Cout = Cout;
    p.KH = KH; p.KW = KW; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.idil = idil < 1 ? 1 : idil; p.ups = ups;
    p.HL = ups ? 2 * H : (p.idil > 1 ? (H - 1) * p.idil + 1 : H);
    p.WL = ups ? 2 * W : (p.idil > 1 ? (W - 1) * p.idil + 1 : W);
    p.M = N * OH * OW;
    const int Cin = C1 + C2;
    p.K = KH * KW * Cin;
    p.accumulate = accumulate;
    const bool veca = (Cout % 4 == 0) && (((uintptr_t)dy & 15) == 0);
    const bool vecb = (Cin % 4 == 0) && (C1 % 4 == 0) && (((uintptr_t)x & 15) == 0) && (!x2 || ((uintptr_t)x2 & 15) == 0);
    int bm, bn, tiles, s;
    wgrad_plan(p.M, Cout, p.K, &bm, &bn, &tiles, &s);
    const int nrt = cdiv(p.M, BK);
    if (s > 1 && (size_t)s * Cout * p.K * sizeof(float) > workspace_bytes) return V2A_ERR_WORKSPACE;
    p.splits = s;
    p.rtiles_per_split = cdiv(nrt, s);
    dim3 grid(tiles, s), block(256);
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
        size_t total = (size_t)Cout * p.K;
        int g = (int)((total + 255) / 256);
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(wgrad_splitk_reduce, dim3(g), dim3(256), 0, stream, p);
        V2A_CHECK_LAUNCH();
    }
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
