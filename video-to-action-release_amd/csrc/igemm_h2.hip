// Multi-stage LDS-DMA convolution for the large layers of the bf16-storage video UNet.
//
// conv_igemm_h (igemm_h.hip) is bound by bytes in flight per CU: a k tile's DMA lands ~1.4 us after it is issued while the MFMA
// work on it takes ~0.2 us, and a 128x128 tile needs ~150 GB/s per CU to keep the matrix pipe busy -- more than 160 KB of LDS can
// keep in flight.  This kernel halves the bytes per FLOP and deepens the pipeline:
//   * 256 x BN output tile (BN = 256 or 128) per 512-thread workgroup (8 waves, one workgroup per CU), k tile = 32 bf16 = 64-B rows;
//   * S LDS stages (S x 32 KB / S x 24 KB <= 128 KB); the DMA of tile t+S-1 is issued while tile t is multiplied, so S-1 tiles are
//     always in flight.  Synchronisation is counted, not drained: `s_waitcnt vmcnt((S-2) * loads_per_stage)` retires exactly the
//     oldest stage of THIS wave, one raw `s_barrier` per k tile then makes it visible to all waves and doubles as the
//     write-after-read fence for the buffer that the next issue overwrites (it was multiplied one iteration ago);
//   * 64-B rows: LDS slot (row r, position p) holds the row's 16-B chunk p ^ ((r >> 2) & 3) (source-side swizzle of the lane-linear
//     DMA image) -> conflict-free ds_read_b128 operand fetches in all four 16-lane groups;
//   * same epilogue as conv_igemm_h: per-wave 32-row sub-tiles parked in LDS, 16-B bf16 stores, bias / embedding vector /
//     residual on 8-wide vectors, per-64-row GroupNorm statistics.
// No split-K: the launcher only takes this path when the tile count fills the chip.  Measured (tools/conv_h_bench.py): the 256x256
// instance is 3-5 % faster than conv_igemm_h's four-workgroups-per-CU form on the 256-channel layers (914 / 950 / 743 TFLOP/s); the
// 256x128 instance is slower (676 vs 791) and is not dispatched; staggering the two waves of a SIMD by half an iteration lost 2 %.
// What is still missing for the 60 % the structure allows is the fine-grained MFMA / ds_read / DMA interleave of an 8-phase schedule.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_2;
typedef const __attribute__((address_space(1))) void* gptr2_t;
typedef __attribute__((address_space(3))) void* lptr2_t;

struct FastDiv2 {
    uint32_t d, m, s;
};
static inline FastDiv2 make_fastdiv2(uint32_t d) {
    FastDiv2 f;
    f.d = d;
    if (d <= 1) { f.m = 0; f.s = 0; return f; }
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;
    f.s = s;
    f.m = (uint32_t)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
    return f;
}
__device__ __forceinline__ uint32_t fdiv2(uint32_t n, const FastDiv2& f) {
    if (f.d <= 1) return n;
    const uint32_t t = __umulhi(f.m, n);
    return (t + ((n - t) >> 1)) >> (f.s - 1);
}

struct ConvDescH2 {
    const uint16_t* x;
    const uint16_t* x2;
    const uint16_t* w;
    const float* bias;
    const float* rowvec;
    const uint16_t* residual;
    uint16_t* y;
    float* stats;
    const uint16_t* zeros;
    int N, H, W, C1, C2, OH, OW, Cout, KH, KW, sh, sw, ph, pw, ups, HL, WL, M, K, rows_per_batch;
    FastDiv2 fd_ow, fd_oh;
};

__device__ __forceinline__ int xcd_remap2(int bid, int nblk) {
    int q = nblk >> 3, r = nblk & 7;
    int xcd = bid & 7, slot = bid >> 3;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// WAVES_M x WAVES_N = 8 waves; each wave owns a (TM*32) x (TN*32) sub-tile.
template <int WAVES_M, int WAVES_N, int TM, int TN, int S, bool F16>
__global__ __launch_bounds__(512, 1) void conv_igemm_h2(const ConvDescH2 p) {
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr int ROWB = 64;                               // bytes per tile row: 32 bf16
    constexpr int AL = BM / 128, BL = BN / 128;            // DMA pieces per thread per stage (128 rows x 4 chunks per 512-thread pass)
    constexpr int NL = AL + BL;                            // DMA instructions per thread per stage
    constexpr int STAGE = (BM + BN) * ROWB;
    static_assert(WAVES_M * WAVES_N == 8 && BM % 128 == 0 && BN % 128 == 0, "tile shape");
    static_assert((S - 2) * NL <= 63, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(128))) unsigned char smem[S * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = (p.Cout + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int lin = xcd_remap2(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (lin / tiles_n) * BM, n0 = (lin % tiles_n) * BN;
    const int Cin = p.C1 + p.C2;
    const int nkt = p.K >> 5;

    // ---- DMA source state: pass j fills rows j*128 .. j*128+127; this thread's slot is (row j*128 + tid/4, position tid%4) and
    // carries chunk (tid%4) ^ ((row >> 2) & 3) = (tid%4) ^ ((tid >> 4) & 3)
    const int lrow = tid >> 2;
    const int chunk = (tid & 3) ^ ((tid >> 4) & 3);
    int a_ihb[AL], a_iwb[AL], a_img[AL];
#pragma unroll
    for (int j = 0; j < AL; ++j) {
        const int m = m0 + j * 128 + lrow;
        const bool ok = m < p.M;
        const uint32_t mm = ok ? (uint32_t)m : 0u;
        const uint32_t t = fdiv2(mm, p.fd_ow);
        const int ow = (int)(mm - t * p.OW);
        const uint32_t img = fdiv2(t, p.fd_oh);
        const int oh = (int)(t - img * p.OH);
        a_img[j] = (int)img;
        a_ihb[j] = ok ? oh * p.sh - p.ph : -(1 << 28);
        a_iwb[j] = ow * p.sw - p.pw;
    }
    const uint16_t* zsrc = p.zeros + chunk * 8;
    const uint16_t* b_src[BL];
    bool b_ok[BL];
#pragma unroll
    for (int j = 0; j < BL; ++j) {
        const int n = n0 + j * 128 + lrow;
        b_ok[j] = n < p.Cout;
        b_src[j] = b_ok[j] ? p.w + (size_t)n * p.K + chunk * 8 : zsrc;
    }
    int ik0 = 0, ic0 = 0, ikh = 0, ikw = 0;                // running (k, channel, tap) position of the next k tile to issue

    auto issue = [&](int buf) {
        unsigned char* abase = smem + buf * STAGE;
        unsigned char* bbase = abase + BM * ROWB;
        const bool first = ic0 < p.C1;
        const uint16_t* src = first ? p.x : p.x2;
        const uint32_t Cs = (uint32_t)(first ? p.C1 : p.C2);
        const uint32_t cc = (uint32_t)((first ? ic0 : ic0 - p.C1) + chunk * 8);
#pragma unroll
        for (int j = 0; j < AL; ++j) {
            int ih = a_ihb[j] + ikh, iw = a_iwb[j] + ikw;
            const bool ok = (unsigned)ih < (unsigned)p.HL && (unsigned)iw < (unsigned)p.WL;
            if (p.ups) { ih >>= 1; iw >>= 1; }
            const uint32_t off = ((uint32_t)(a_img[j] * p.H + ih) * (uint32_t)p.W + (uint32_t)iw) * Cs + cc;
            const uint16_t* g = ok ? src + off : zsrc;
            __builtin_amdgcn_global_load_lds((gptr2_t)g, (lptr2_t)(abase + (j * 512 + wid * 64) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < BL; ++j) {
            const uint16_t* g = b_src[j] + (b_ok[j] ? ik0 : 0);
            __builtin_amdgcn_global_load_lds((gptr2_t)g, (lptr2_t)(bbase + (j * 512 + wid * 64) * 16), 16, 0, 0);
        }
        ik0 += 32;
        ic0 += 32;
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

    const int wm = (wid / WAVES_N) * TM * 32, wn = (wid % WAVES_N) * TN * 32;
    const int lr = lane & 31, lk = lane >> 5;
    const int rswz = (lr >> 2) & 3;
    int a_off[TM], b_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_off[i] = (wm + i * 32 + lr) * ROWB;
#pragma unroll
    for (int j = 0; j < TN; ++j) b_off[j] = BM * ROWB + (wn + j * 32 + lr) * ROWB;

    // ---- prologue: S-1 stages in flight
#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < nkt) issue(s);
    int cbuf = 0, ibuf = S - 1;                            // buffer being multiplied / buffer the next issue fills
    for (int kt = 0; kt < nkt; ++kt) {
        // retire this wave's share of stage kt: everything issued after it may stay in flight
        const int later = min(nkt, kt + S - 1) - (kt + 1);
        if (later >= S - 2) wait_vmcnt<(S - 2) * NL>();
        else if (later == 3) wait_vmcnt<(S > 4 ? 3 : 0) * NL>();
        else if (later == 2) wait_vmcnt<(S > 3 ? 2 : 0) * NL>();
        else if (later == 1) wait_vmcnt<(S > 2 ? 1 : 0) * NL>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                      // stage kt visible everywhere; everyone finished multiplying stage kt-1
        if (kt + S - 1 < nkt) issue(ibuf);
        const unsigned char* base = smem + cbuf * STAGE;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pos = (((h << 1) | lk) ^ rswz) << 4;
            bf16x8_2 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8_2*>(base + a_off[i] + pos);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8_2*>(base + b_off[j] + pos);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = v2a_mfma_h<F16>(a[i], b[j], acc[i][j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's operand reads are done before it reaches the next barrier
        cbuf = (cbuf + 1 == S) ? 0 : cbuf + 1;
        ibuf = (ibuf + 1 == S) ? 0 : ibuf + 1;
    }

    // ---- epilogue
    wait_vmcnt<0>();
    __syncthreads();
    constexpr int WNC = TN * 32, LDC = WNC;
    static_assert(8 * 32 * LDC * 4 <= S * STAGE, "epilogue staging exceeds the stage buffers");
    float* cw = reinterpret_cast<float*>(smem) + wid * 32 * LDC;
    constexpr int V = WNC / 8;
    const int vrow = lane / V, vcol = (lane % V) * 8;
    const int n = n0 + wn + vcol;
    const bool vec_ok = (p.Cout % 8 == 0) && (n + 8 <= p.Cout);
    float bv[8], ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bv[e] = (p.bias && n + e < p.Cout) ? p.bias[n + e] : 0.f;
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
            const int m = m0 + wm + i * 32 + ml;
            if (m >= p.M || n >= p.Cout) continue;
            const int sx = (ml & 1) << 2;
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(&cw[ml * LDC + (vcol ^ sx)]);
            const f32x4 c1 = *reinterpret_cast<const f32x4*>(&cw[ml * LDC + ((vcol + 4) ^ sx)]);
            float v[8] = {c0[0] + bv[0], c0[1] + bv[1], c0[2] + bv[2], c0[3] + bv[3], c1[0] + bv[4], c1[1] + bv[5], c1[2] + bv[6], c1[3] + bv[7]};
            const size_t o = (size_t)m * p.Cout + n;
            if (vec_ok) {
                if (p.rowvec) {
                    const float* rv = p.rowvec + (size_t)(m / p.rows_per_batch) * p.Cout + n;
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>(rv), r1 = *reinterpret_cast<const f32x4*>(rv + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[e + 4] += r1[e]; }
                }
                if (p.residual) {
                    const uint4 u = *reinterpret_cast<const uint4*>(p.residual + o);
                    v[0] += v2a_lo_h2<F16>(u.x); v[1] += v2a_hi_h2<F16>(u.x);
                    v[2] += v2a_lo_h2<F16>(u.y); v[3] += v2a_hi_h2<F16>(u.y);
                    v[4] += v2a_lo_h2<F16>(u.z); v[5] += v2a_hi_h2<F16>(u.z);
                    v[6] += v2a_lo_h2<F16>(u.w); v[7] += v2a_hi_h2<F16>(u.w);
                }
                uint16_t h[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    h[e] = v2a_f2h<F16>(v[e]);
                    const float r = v2a_h2f<F16>(h[e]);
                    ssum[e] += r;
                    ssq[e] += r * r;
                }
                uint4 u;
                u.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
                u.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
                u.z = (uint32_t)h[4] | ((uint32_t)h[5] << 16);
                u.w = (uint32_t)h[6] | ((uint32_t)h[7] << 16);
                *reinterpret_cast<uint4*>(p.y + o) = u;
            } else {
                for (int e = 0; e < 8 && n + e < p.Cout; ++e) {
                    float t = v[e];
                    if (p.rowvec) t += p.rowvec[(size_t)(m / p.rows_per_batch) * p.Cout + n + e];
                    if (p.residual) t += v2a_h2f<F16>(p.residual[o + e]);
                    p.y[o + e] = v2a_f2h<F16>(t);
                }
            }
        }
        if ((i & 1) == 1 && p.stats && vec_ok) {            // two 32-row sub-tiles = one 64-row statistics block of this wave
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int o2 = V; o2 < 64; o2 <<= 1) {
                    ssum[e] += __shfl_xor(ssum[e], o2, 64);
                    ssq[e] += __shfl_xor(ssq[e], o2, 64);
                }
            }
            const int mb = m0 + wm + (i - 1) * 32;
            if (lane < V && mb < p.M) {
                float* dst = p.stats + (size_t)(mb >> 6) * 2 * p.Cout + n;
                f32x4 a0 = {ssum[0], ssum[1], ssum[2], ssum[3]}, a1 = {ssum[4], ssum[5], ssum[6], ssum[7]};
                f32x4 q0 = {ssq[0], ssq[1], ssq[2], ssq[3]}, q1 = {ssq[4], ssq[5], ssq[6], ssq[7]};
                *reinterpret_cast<f32x4*>(dst) = a0;
                *reinterpret_cast<f32x4*>(dst + 4) = a1;
                *reinterpret_cast<f32x4*>(dst + p.Cout) = q0;
                *reinterpret_cast<f32x4*>(dst + p.Cout + 4) = q1;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
        }
    }
}

extern "C" {

// 1 when v2a_conv2d_fwd_h2 takes this problem (bf16 storage, no split-K, enough tiles to fill the chip)
int v2a_conv2d_h2_eligible(int M, int Cout, int K, int C1, int C2) {
    if (C1 % 32 || C2 % 32 || K % 32 || Cout % 8 || Cout < 128) return 0;
    if (Cout % 256) return 0;          // 128-wide layers: both a 256x128 (5 stages) and a 512x128 (3 stages) instance measured slower
                                       // than conv_igemm_h's four-workgroups-per-CU form (676 / 744 vs 791 TFLOP/s): not dispatched
    const long tiles = (long)cdiv(M, 256) * (Cout / 256);
    return tiles >= 512 ? 1 : 0;
}

// Same contract as v2a_conv2d_fwd_h with bf16 output (y), optional bf16 residual, optional statistics; idil = 1 only.
int v2a_conv2d_fwd_h2(const void* x, const void* x2, const void* w_packed, const float* bias, const float* rowvec, const void* residual,
                      void* y, const void* zeros, int N, int H, int W, int C1, int C2, int Cout, int KH, int KW, int sh, int sw, int ph,
                      int pw, int ups, int OH, int OW, int rows_per_batch, float* stats, hipStream_t stream) {
    if (!x || !w_packed || !zeros || !y || N <= 0) return V2A_ERR_ARG;
    const int M = N * OH * OW, K = KH * KW * (C1 + C2);
    if (!v2a_conv2d_h2_eligible(M, Cout, K, C1, C2) || (C2 > 0 && !x2)) return V2A_ERR_ARG;
    if ((((uintptr_t)x | (uintptr_t)x2 | (uintptr_t)w_packed | (uintptr_t)zeros | (uintptr_t)y | (uintptr_t)residual) & 15) != 0) return V2A_ERR_ARG;
    if ((double)N * H * W * (C1 > C2 ? C1 : C2) >= 4294967296.0) return V2A_ERR_ARG;
    ConvDescH2 p;
    p.x = (const uint16_t*)x; p.x2 = (const uint16_t*)x2; p.w = (const uint16_t*)w_packed;
    p.bias = bias; p.rowvec = rowvec; p.residual = (const uint16_t*)residual; p.y = (uint16_t*)y; p.stats = stats;
    p.zeros = (const uint16_t*)zeros;
    p.N = N; p.H = H; p.W = W; p.C1 = C1; p.C2 = C2; p.OH = OH; p.OW = OW; p.Cout = Cout;
    p.KH = KH; p.KW = KW; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.ups = ups ? 1 : 0;
    p.HL = ups ? 2 * H : H; p.WL = ups ? 2 * W : W;
    p.M = M; p.K = K;
    p.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
    p.fd_ow = make_fastdiv2((uint32_t)OW);
    p.fd_oh = make_fastdiv2((uint32_t)OH);
    const int tiles = cdiv(M, 256) * (Cout / 256);
    if (g_v2a_half_f16) hipLaunchKernelGGL((conv_igemm_h2<2, 4, 4, 2, 4, true>), dim3(tiles), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL((conv_igemm_h2<2, 4, 4, 2, 4, false>), dim3(tiles), dim3(512), 0, stream, p);      // 256 x 256, 4 stages x 32 KB
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

}  // extern "C"
