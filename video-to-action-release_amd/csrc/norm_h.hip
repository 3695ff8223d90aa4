// GroupNorm (+ fused activation) forward over bf16 channels-last tensors [N, S, C]: the bf16-storage configuration of the video
// UNet's GroupNorm32 + SiLU (reference guided_diffusion/guided_diffusion/nn.py:26-28 computes the normalisation in fp32 on the
// autocast-half input and returns the input dtype: same contract here).  Optional second source = channel concat
// (decoder ResBlocks normalise cat[h, skip] without materialising it).
//
// HBM-bound, three launches:  column sums (16-B bf16x8 loads, fp32 registers -> LDS bins -> fp32 partial per row chunk)
//                             -> per (n, group) fp64 combine -> per (n, channel) scale / shift
//                             -> apply: y = act(x * a[n][c] + b[n][c]), 16-B loads and stores.
// 6 bytes per element of traffic (2 reads + 1 write of bf16) against 12 for the fp32 tensors.
#include "common.h"

struct GnDescH {
    const uint16_t* x;      // [N][S][C1]
    const uint16_t* x2;     // [N][S][C - C1] or null
    const float* gamma;     // [C]
    const float* beta;      // [C]
    uint16_t* y;            // [N][S][C]
    float* partial;         // [N][nchunk][2][C]
    float* ab;              // [N][2][C]  (scale, shift)
    const float* st1;       // optional precomputed statistics of x : [N * S/64][2][C1] (conv epilogue, csrc/igemm_h.hip)
    const float* st2;       // ... and of x2: [N * S/64][2][C - C1]
    float* mean;            // [N][G] or null
    float* rstd;            // [N][G] or null
    int N, S, C, C1, G, act, nchunk, rows_per_chunk;
    uint32_t l8_magic, l8_shift;   // idx / (C/8) == umulhi(idx, l8_magic) >> l8_shift for idx < 2^31
    float eps;
};

// activation of the bf16-storage path: the result is rounded to 8 mantissa bits, so SiLU runs on the approximate exp2 / rcp units
__device__ __forceinline__ float act_fwd_h(float x, int act) { return act == ACT_SILU ? v2a_silu_fast(x) : act_fwd(x, act); }

template <bool F16>
__global__ __launch_bounds__(256) void gn_stats_h(const GnDescH p) {
    extern __shared__ __attribute__((aligned(16))) float bins[];   // [rpi][2][C]: one slot per (row lane, column), combined in lane order
    const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int C = p.C, L8 = C >> 3;
    const int s0 = chunk * p.rows_per_chunk, s1 = min(p.S, s0 + p.rows_per_chunk);
    const int nrows = s1 - s0;
    const int C1 = p.x2 ? p.C1 : C, C2 = C - C1;
    const uint16_t* xa = p.x + ((size_t)n * p.S + s0) * C1;
    const uint16_t* xb = p.x2 ? p.x2 + ((size_t)n * p.S + s0) * C2 : nullptr;
    const int rpi = L8 >= 256 ? 1 : 256 / L8;                 // rows in flight per pass
    const int row0 = L8 >= 256 ? 0 : tid / L8;
    const bool active = L8 >= 256 ? true : (tid < rpi * L8);
    for (int c8 = active ? (L8 >= 256 ? tid : tid % L8) : L8; c8 < L8; c8 += 256) {
        const int c = c8 * 8;
        const bool first = c < C1;
        const uint16_t* src = first ? xa + c : xb + (c - C1);
        const int stride = first ? C1 : C2;
        float s[8], q[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { s[k] = 0.f; q[k] = 0.f; }
        int r = row0;
        for (; r + 3 * rpi < nrows; r += 4 * rpi) {           // four independent 16-B loads in flight
            uint4 u[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) u[k] = *reinterpret_cast<const uint4*>(src + (size_t)(r + k * rpi) * stride);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float f[8];
                v2a_unpack_h8<F16>(u[k], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
            }
        }
        for (; r < nrows; r += rpi) {
            float f[8];
            v2a_unpack_h8<F16>(*reinterpret_cast<const uint4*>(src + (size_t)r * stride), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
        }
        float* slot = bins + (size_t)row0 * 2 * C;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            slot[c + e] = s[e];
            slot[C + c + e] = q[e];
        }
    }
    __syncthreads();
    float* dst = p.partial + ((size_t)n * p.nchunk + chunk) * 2 * C;
    for (int i = tid; i < 2 * C; i += 256) {
        float a = 0.f;
        for (int l = 0; l < rpi; ++l) a += bins[(size_t)l * 2 * C + i];
        dst[i] = a;
    }
}

// one wave per (n, group): fp64 combine over chunks and the group's channels, then the per-channel affine of the apply pass
__global__ __launch_bounds__(64) void gn_finalize_h(const GnDescH p) {
    const int n = blockIdx.x / p.G, g = blockIdx.x % p.G, lane = threadIdx.x;
    const int cg = p.C / p.G;
    double s = 0.0, q = 0.0;
    for (int i = lane; i < p.nchunk * cg; i += 64) {
        const int ch = i / cg, c = g * cg + i % cg;
        const float* src = p.partial + ((size_t)n * p.nchunk + ch) * 2 * p.C;
        s += (double)src[c];
        q += (double)src[p.C + c];
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    const double cnt = (double)p.S * cg;
    const double mean = s / cnt;
    double var = q / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
    const float mu = (float)mean;
    if (lane == 0) {
        if (p.mean) p.mean[n * p.G + g] = mu;
        if (p.rstd) p.rstd[n * p.G + g] = rstd;
    }
    for (int i = lane; i < cg; i += 64) {
        const int c = g * cg + i;
        const float a = p.gamma[c] * rstd;
        p.ab[(size_t)n * 2 * p.C + c] = a;
        p.ab[(size_t)n * 2 * p.C + p.C + c] = p.beta[c] - mu * a;
    }
}

// gn_reduce_blocks_h + gn_finalize_h as ONE launch (round 5; VERDICT r4 next #3c: 130 such launch pairs per UNet forward): one 256-thread
// workgroup per (sample, group) sums the group's channels over all 64-row statistic blocks of the conv epilogues -- thread = (block, channel)
// pairs in a fixed stride order, fp32 per thread, fp64 across the workgroup (fixed tree) -- and writes mean / rstd and the group's
// rows of the scale / shift table.  The 32 groups of a sample read neighbouring 16 ... 160-B pieces of the same block rows: the
// statistics tensor comes from HBM once and from L2 after that.
__global__ __launch_bounds__(256) void gn_blocks_finalize_h(const GnDescH p) {
    __shared__ double sm[2][4];
    const int n = blockIdx.x / p.G, g = blockIdx.x % p.G, tid = threadIdx.x;
    const int C = p.C, cg = C / p.G, C1 = p.x2 ? p.C1 : C, C2 = C - C1;
    const int nb = p.S >> 6;
    const float* s1 = p.st1 + (size_t)n * nb * 2 * C1;
    const float* s2 = p.st2 ? p.st2 + (size_t)n * nb * 2 * C2 : nullptr;
    float s = 0.f, q = 0.f, sB = 0.f, qB = 0.f;
    const int total = nb * cg;
    int i = tid;
    for (; i + 256 < total; i += 512) {                            // two independent (block, channel) pairs in flight
        const int b0 = i / cg, c0 = g * cg + (i - b0 * cg);
        const int i1 = i + 256, b1 = i1 / cg, c1 = g * cg + (i1 - b1 * cg);
        const float* r0 = c0 < C1 ? s1 + (size_t)b0 * 2 * C1 + c0 : s2 + (size_t)b0 * 2 * C2 + (c0 - C1);
        const float* r1 = c1 < C1 ? s1 + (size_t)b1 * 2 * C1 + c1 : s2 + (size_t)b1 * 2 * C2 + (c1 - C1);
        const int h0 = c0 < C1 ? C1 : C2, h1 = c1 < C1 ? C1 : C2;
        s += r0[0]; q += r0[h0];
        sB += r1[0]; qB += r1[h1];
    }
    for (; i < total; i += 256) {
        const int b0 = i / cg, c0 = g * cg + (i - b0 * cg);
        const float* r0 = c0 < C1 ? s1 + (size_t)b0 * 2 * C1 + c0 : s2 + (size_t)b0 * 2 * C2 + (c0 - C1);
        s += r0[0]; q += r0[c0 < C1 ? C1 : C2];
    }
    double ds = wave_sum_d((double)s + (double)sB), dq = wave_sum_d((double)q + (double)qB);
    if ((tid & 63) == 0) { sm[0][tid >> 6] = ds; sm[1][tid >> 6] = dq; }
    __syncthreads();
    ds = (sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3]);
    dq = (sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3]);
    const double cnt = (double)p.S * cg;
    const double mean = ds / cnt;
    double var = dq / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
    const float mu = (float)mean;
    if (tid == 0) {
        if (p.mean) p.mean[n * p.G + g] = mu;
        if (p.rstd) p.rstd[n * p.G + g] = rstd;
    }
    for (int k = tid; k < cg; k += 256) {
        const int c = g * cg + k;
        const float a = p.gamma[c] * rstd;
        p.ab[(size_t)n * 2 * C + c] = a;
        p.ab[(size_t)n * 2 * C + C + c] = p.beta[c] - mu * a;
    }
}

// grid (chunks, N): 32-bit indexing inside one sample, row / column split by a launch-invariant reciprocal (no 64-bit divides),
// four independent 16-B loads in flight per thread.
template <bool F16>
__global__ __launch_bounds__(256) void gn_apply_h(const GnDescH p) {
    const int L8 = p.C >> 3;
    const uint32_t per_n = (uint32_t)p.S * (uint32_t)L8;
    const int n = blockIdx.y;
    const int C1 = p.x2 ? p.C1 : p.C, C2 = p.C - C1;
    const uint16_t* xa = p.x + (size_t)n * p.S * C1;
    const uint16_t* xb = p.x2 ? p.x2 + (size_t)n * p.S * C2 : nullptr;
    uint16_t* yo = p.y + (size_t)n * p.S * p.C;
    const float* ab = p.ab + (size_t)n * 2 * p.C;
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t i0 = blockIdx.x * 256u + threadIdx.x; i0 < per_n; i0 += 4 * stride) {
        uint4 u[4];
        uint32_t row[4];
        int c[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t idx = i0 + k * stride;
            ok[k] = idx < per_n;
            const uint32_t id = ok[k] ? idx : 0u;
            row[k] = (L8 == 1) ? id : (__umulhi(id, p.l8_magic) >> p.l8_shift);   // id / L8 (exact for id < 2^31: host-side magic)
            c[k] = (int)(id - row[k] * (uint32_t)L8) * 8;
            const uint16_t* src = (c[k] < C1) ? xa + (size_t)row[k] * C1 + c[k] : xb + (size_t)row[k] * C2 + (c[k] - C1);
            u[k] = ok[k] ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!ok[k]) continue;
            const float* a = ab + c[k];
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(a), a1 = *reinterpret_cast<const f32x4*>(a + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(a + p.C), b1 = *reinterpret_cast<const f32x4*>(a + p.C + 4);
            uint4 v;
            if (p.act == ACT_SILU || p.act == ACT_NONE) {
                const bool silu = p.act == ACT_SILU;
                v.x = v2a_gn_act2<F16>(u[k].x, v2a_f32x2{a0[0], a0[1]}, v2a_f32x2{b0[0], b0[1]}, silu);
                v.y = v2a_gn_act2<F16>(u[k].y, v2a_f32x2{a0[2], a0[3]}, v2a_f32x2{b0[2], b0[3]}, silu);
                v.z = v2a_gn_act2<F16>(u[k].z, v2a_f32x2{a1[0], a1[1]}, v2a_f32x2{b1[0], b1[1]}, silu);
                v.w = v2a_gn_act2<F16>(u[k].w, v2a_f32x2{a1[2], a1[3]}, v2a_f32x2{b1[2], b1[3]}, silu);
            } else {
                float f[8], o[8];
                v2a_unpack_h8<F16>(u[k], f);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = act_fwd_h(f[e] * a0[e] + b0[e], p.act);
                    o[e + 4] = act_fwd_h(f[e + 4] * a1[e] + b1[e], p.act);
                }
                v = uint4{v2a_pack_h2<F16>(o[0], o[1]), v2a_pack_h2<F16>(o[2], o[3]), v2a_pack_h2<F16>(o[4], o[5]), v2a_pack_h2<F16>(o[6], o[7])};
            }
            *reinterpret_cast<uint4*>(yo + (size_t)row[k] * p.C + c[k]) = v;
        }
    }
}

static void gn_chunks_h(int N, int S, int* nchunk, int* rows) {
    // ~2048 workgroups over the whole tensor, at least 64 rows per chunk
    int want = 2048 / (N > 0 ? N : 1);
    if (want < 1) want = 1;
    int r = cdiv(S, want);
    if (r < 64) r = 64;
    *rows = r;
    *nchunk = cdiv(S, r);
}

extern "C" {

size_t v2a_groupnorm_h_workspace_bytes(int N, int S, int C) {
    int nchunk, rows;
    gn_chunks_h(N, S, &nchunk, &rows);
    return ((size_t)N * nchunk * 2 * C + (size_t)N * 2 * C) * sizeof(float);
}

}  // extern "C"

// y = act(GroupNorm_G(cat[x, x2]) * gamma + beta), all activations bf16 [N][S][C]; mean / rstd ([N][G], fp32) optional outputs.
// C % 8 == 0, C1 % 8 == 0, C % G == 0.  stats1 / stats2 (optional): per-64-row [2][C_src] sum / sum-of-squares slabs written by
// v2a_conv2d_fwd_h for x / x2 -- the statistics pass is skipped (S % 64 == 0 required).
static int gn_prep_h(GnDescH& p, const void* x, const void* x2, int C1, const float* gamma, const float* beta, float* mean, float* rstd,
                     const float* stats1, const float* stats2, int N, int S, int C, int G, float eps, float* ab_out, void* workspace,
                     size_t workspace_bytes, hipStream_t stream) {
    if (!x || !gamma || !beta || N <= 0 || S <= 0 || C <= 0 || G <= 0 || C % G || C % 8) return V2A_ERR_ARG;
    if (x2 && (C1 <= 0 || C1 >= C || C1 % 8)) return V2A_ERR_ARG;
    if (2 * C * sizeof(float) > 64 * 1024) return V2A_ERR_ARG;
    p.x = (const uint16_t*)x; p.x2 = (const uint16_t*)x2; p.gamma = gamma; p.beta = beta; p.y = nullptr;
    p.mean = mean; p.rstd = rstd;
    p.st1 = stats1; p.st2 = stats2;
    if (stats1 && ((S & 63) || (x2 && !stats2))) return V2A_ERR_ARG;      // 64-row statistic blocks must tile every sample
    p.N = N; p.S = S; p.C = C; p.C1 = x2 ? C1 : C; p.G = G; p.act = 0; p.eps = eps;
    gn_chunks_h(N, S, &p.nchunk, &p.rows_per_chunk);
    const size_t need = ((size_t)N * p.nchunk * 2 * C + (size_t)N * 2 * C) * sizeof(float);
    if (!workspace || workspace_bytes < need) return V2A_ERR_WORKSPACE;
    p.partial = (float*)workspace;
    if (!stats1) {
        const size_t lds = (size_t)((C >> 3) >= 256 ? 1 : 256 / (C >> 3)) * 2 * C * sizeof(float);
        if (g_v2a_half_f16) hipLaunchKernelGGL(gn_stats_h<true>, dim3(p.nchunk, N), dim3(256), lds, stream, p);
        else hipLaunchKernelGGL(gn_stats_h<false>, dim3(p.nchunk, N), dim3(256), lds, stream, p);
        V2A_CHECK_LAUNCH();
    } else {
        // the conv epilogues' statistic blocks: summed and finalised by ONE launch (gn_blocks_finalize_h)
        p.ab = ab_out ? ab_out : p.partial + (size_t)N * p.nchunk * 2 * C;
        hipLaunchKernelGGL(gn_blocks_finalize_h, dim3(N * G), dim3(256), 0, stream, p);
        V2A_CHECK_LAUNCH();
        return V2A_OK;
    }
    p.ab = ab_out ? ab_out : p.partial + (size_t)N * p.nchunk * 2 * C;      // scale / shift table [N][2][C]
    hipLaunchKernelGGL(gn_finalize_h, dim3(N * G), dim3(64), 0, stream, p);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
static int gn_apply_launch_h(GnDescH& p, hipStream_t stream) {
    const int N = p.N, S = p.S, C = p.C;
    {   // round-up reciprocal of L8: exact quotient for every idx < 2^31 (Granlund-Montgomery with a 32+s bit magic, s = ceil(log2 d))
        const uint32_t d = (uint32_t)(C / 8);
        uint32_t sft = 0;
        while ((1u << sft) < d) ++sft;
        const uint64_t m = ((1ull << (31 + sft)) + d - 1) / d;      // < 2^32 because 2^sft < 2 d
        p.l8_magic = (uint32_t)m;
        p.l8_shift = sft - 1 + 0;                                    // umulhi gives >> 32; total shift 31 + sft
        if (d == 1) { p.l8_magic = 0; p.l8_shift = 0; }              // the kernel special-cases L8 == 1
    }
    if ((double)S * (C / 8) >= 2147483648.0) return V2A_ERR_ARG;
    const size_t vecs = (size_t)S * (C / 8);
    int g = (int)((vecs + 1023) / 1024);
    int cap = 8192 / N;
    if (cap < 32) cap = 32;
    if (g > cap) g = cap;
    if (g_v2a_half_f16) hipLaunchKernelGGL(gn_apply_h<true>, dim3(g, N), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(gn_apply_h<false>, dim3(g, N), dim3(256), 0, stream, p);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}


extern "C" {
int v2a_groupnorm_fwd_h(const void* x, const void* x2, int C1, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                        const float* stats1, const float* stats2, int N, int S, int C, int G, float eps, int act, void* workspace,
                        size_t workspace_bytes, hipStream_t stream) {
    if (!y) return V2A_ERR_ARG;
    GnDescH p;
    const int rc = gn_prep_h(p, x, x2, C1, gamma, beta, mean, rstd, stats1, stats2, N, S, C, G, eps, nullptr, workspace, workspace_bytes, stream);
    if (rc != V2A_OK) return rc;
    p.y = (uint16_t*)y; p.act = act;
    return gn_apply_launch_h(p, stream);
}
// The two halves of v2a_groupnorm_fwd_h on their own.  prep: statistics (or the conv epilogues' blocks) -> per-(sample, channel)
// scale / shift table ab_out [N][2][C] (fp32, 16-B aligned): y = act(x * ab[n][0][c] + ab[n][1][c]).  A consumer that applies the
// table itself (v2a_conv2d_fwd_h3_gn: the normalisation happens on the conv's input tile in LDS) makes the apply pass unnecessary.
int v2a_groupnorm_prep_h(const void* x, const void* x2, int C1, const float* gamma, const float* beta, float* mean, float* rstd,
                         const float* stats1, const float* stats2, int N, int S, int C, int G, float eps, float* ab_out, void* workspace,
                         size_t workspace_bytes, hipStream_t stream) {
    if (!ab_out || ((uintptr_t)ab_out & 15)) return V2A_ERR_ARG;
    GnDescH p;
    return gn_prep_h(p, x, x2, C1, gamma, beta, mean, rstd, stats1, stats2, N, S, C, G, eps, ab_out, workspace, workspace_bytes, stream);
}
int v2a_groupnorm_apply_h(const void* x, const void* x2, int C1, const float* ab, void* y, int N, int S, int C, int act, hipStream_t stream) {
    if (!x || !ab || !y || N <= 0 || S <= 0 || C <= 0 || C % 8 || (x2 && (C1 <= 0 || C1 >= C || C1 % 8))) return V2A_ERR_ARG;
    GnDescH p = {};
    p.x = (const uint16_t*)x; p.x2 = (const uint16_t*)x2; p.y = (uint16_t*)y; p.ab = (float*)ab;
    p.N = N; p.S = S; p.C = C; p.C1 = x2 ? C1 : C; p.G = 1; p.act = act;
    return gn_apply_launch_h(p, stream);
}
}  // extern "C"
