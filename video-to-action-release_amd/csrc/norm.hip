// GroupNorm (+ fused activation / residual / FiLM) forward and backward on channels-last tensors [N, S, C].
//
// replaces: GroupNorm32 + SiLU of the video UNet (reference nn.py:26-28, unet.py:187-190,211-216,289,628-631),
//           GroupNorm(C/16) + ReLU (+ residual) of the ResNet-18 encoders (multi_image_obs_encoder.py:66-74),
//           GroupNorm(8) + Mish + FiLM of Conv1dBlock / ConditionalResidualBlock1D (conv1d_components.py:23-40,
//           conditional_unet1d.py:46-66).
// order of the fused tail:  z = gn(x) [+ residual];  a = act(z);  out = film ? scale * a + shift : a
//
// Two paths.  HBM-bound tensors: column-reduce (wave64 + LDS bins, fp64 combine) -> finalize -> float4 apply.
// Small tensors (one (n, group) slab <= 8192 elements): one workgroup does stats + apply in one launch.
// Statistics are combined in fp64 (sum / sum-of-squares of fp32 partials) so that mean / rstd carry < 1e-6
// relative error -- the parity budget of the path is 1e-4.
#include "common.h"
#include <stdlib.h>

struct GnDesc {
    const float* x;         // [N][S][C]   (or channels [0, C1) of a virtual concat when x2 != null: row stride C1)
    const float* x2;        // channels [C1, C) of the virtual concat: [N][S][C - C1], forward large path only
    const float* gamma;     // [C]
    const float* beta;      // [C]
    const float* residual;  // [N][S][C] or null (added before the activation)
    const float* film;      // [N][2][C] (scale, shift) or null (applied after the activation)
    const float* dout;      // backward only
    float* y;               // forward output / backward dx
    float* dres;            // backward: gradient w.r.t. residual (= dz) or null
    float* dfilm;           // backward: [N][2][C] or null
    float* mean;            // [N][G]
    float* rstd;            // [N][G]
    float* colsum;          // [N][2][C] (stats: sum x, sum x^2; backward: sum dz, sum dz*xhat)
    double* partial;        // [N][nchunk][2][C]
    unsigned int l4_magic, l4_shift;   // idx / (C/4) == umulhi(idx, l4_magic) >> l4_shift for idx < 2^31 (gn_apply_fwd_rows)
    const float* st1;       // forward large path: per-64-row (sum, sum of squares) blocks of x from the producing conv's epilogue
    const float* st2;       // ... and of x2 ([N * S/64][2][C1] / [N * S/64][2][C - C1]); replaces the gn_colreduce pass over the tensor
    int N, S, C, G, act, nchunk, rows_per_chunk, C1;
    int yh_f16;             // format of the twin `yh`: 0 bf16, 1 IEEE fp16 (the policy's 16-bit mode: v2a_set_policy_half)
    unsigned short* yh;     // optional bf16 twin of the output (forward: y, backward: dx), same shape: feeds the bf16-MFMA convs
    size_t yh3;             // > 0 (float4 wave kernels only): yh is the hi plane of THREE bf16 planes yh3 elements apart (v2a_split3x2): conv_p3 operands
    float* gsum;            // large backward path: [N*G][2] = sum over the group's channels of gamma_c * colsum{0,1}[n][c]
    int film_ld;            // elements between the FiLM rows of consecutive samples (2*C when the [N][2][C] tensor is dense)
    float eps;
    // wave path only -- the tensor this GroupNorm normalises (forward: x; backward: dout) may still be the split-K slabs of the conv that
    // produced it: element = sum_s slab[s][idx] (+ cbias[c]) (+ sresid[idx]), summed here in the reduce kernel's order instead of in a
    // launch of its own; `sout` (optional) receives the finished tensor for other readers (the tape / weight gradients).
    const float* slabs;     // [nslab][N*S*C] or null
    const float* cbias;     // [C] conv bias or null
    const float* sresid;    // [N][S][C] residual the conv epilogue would have added, or null
    float* sout;            // [N][S][C] or null
    int nslab;
    size_t slab_stride;
    // float4 wave forward only -- added to the OUTPUT (after activation / FiLM): the residual branch of a ConditionalResidualBlock1D
    // (conditional_unet1d.py:46-66: out = blocks[1](out) + residual_conv(x)), either dense (the identity branch, or a finished 1x1 conv) or
    // still the split-K slabs of the 1x1 residual conv (+ its bias): out = y + ((sum_s post_slabs[s] + post_bias) | post)
    const float* post;
    const float* post_slabs;
    const float* post_bias;
    int post_nslab;
    size_t post_stride;
};

__device__ __forceinline__ unsigned short gn_f2h(float f, int f16) {      // round to nearest even, as v2a_cast_f32_h (bf16 | IEEE fp16)
    return f16 ? v2a_f2h<true>(f) : v2a_f2bf(f);
}
__device__ __forceinline__ void gn_store_twin4(unsigned short* yh, size_t i4, const f32x4& o, int f16) {
    uint2 u;
    u.x = f16 ? v2a_pack_h2<true>(o[0], o[1]) : v2a_pack_h2<false>(o[0], o[1]);
    u.y = f16 ? v2a_pack_h2<true>(o[2], o[3]) : v2a_pack_h2<false>(o[2], o[3]);
    reinterpret_cast<uint2*>(yh)[i4] = u;
}
// the float4 wave kernels' 16-bit side output: one twin (16-bit MFMA modes), or the three bf16 planes of the fp32 value (conv_p3 operands)
__device__ __forceinline__ void gn_store_twin_or_planes4(const GnDesc& p, size_t off, const f32x4& o) {
    if (p.yh3 == 0) { gn_store_twin4(p.yh, off >> 2, o, p.yh_f16); return; }
    unsigned int h0, m0, l0, h1, m1, l1;
    v2a_split3x2(o[0], o[1], h0, m0, l0);
    v2a_split3x2(o[2], o[3], h1, m1, l1);
    *reinterpret_cast<uint2*>(p.yh + off) = uint2{h0, h1};
    *reinterpret_cast<uint2*>(p.yh + p.yh3 + off) = uint2{m0, m1};
    *reinterpret_cast<uint2*>(p.yh + 2 * p.yh3 + off) = uint2{l0, l1};
}

// -------------------------------------------------------------------------------------------- large path
// MODE 0: (x, x^2).  MODE 1: (dz, dz * xhat) with dz = dout * act'(z), z = gn(x) [+ residual].
// Thread -> fixed float4 column(s): with L4 = C/4 columns, rpi = max(1, 256 / L4) rows are processed in parallel and a
// thread keeps its column's partial sums in registers for the whole slab (4 independent loads in flight per thread).
// Deterministic: every (row lane, column) owns one LDS slot ([rpi][2][C] floats); the rpi lanes of a column are then added in lane
// order by one thread -- no float atomics anywhere, so two runs give bitwise equal statistics.
template <int MODE>
__global__ __launch_bounds__(256) void gn_colreduce(const GnDesc p) {
    extern __shared__ __attribute__((aligned(16))) float bins[];   // [rpi][2][C]
    const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int C = p.C, L4 = C >> 2, cg = C / p.G;
    const int s0 = chunk * p.rows_per_chunk, s1 = min(p.S, s0 + p.rows_per_chunk);
    const int nrows = s1 - s0;
    const bool two = (MODE == 0) && p.x2 != nullptr;
    const int C1 = two ? p.C1 : C, C2 = C - C1;
    const float* xa = p.x + ((size_t)n * p.S + s0) * C1;
    const float* xb = two ? p.x2 + ((size_t)n * p.S + s0) * C2 : xa;
    const float* dptr = MODE ? p.dout + ((size_t)n * p.S + s0) * C : nullptr;
    const float* rptr = (MODE && p.residual) ? p.residual + ((size_t)n * p.S + s0) * C : nullptr;
    const int rpi = L4 >= 256 ? 1 : 256 / L4;                 // rows in flight per pass
    const int row0 = L4 >= 256 ? 0 : tid / L4;
    const bool active = L4 >= 256 ? true : (tid < rpi * L4);
    const int c4_first = active ? (L4 >= 256 ? tid : tid % L4) : L4;
    for (int c4 = c4_first; c4 < L4; c4 += 256) {
        const int cc = c4 * 4;
        const bool first = cc < C1;
        const float* src = first ? xa : xb;
        src += first ? cc : (cc - C1);
        const int ld = first ? C1 : C2;
        float a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
        float mu[4] = {0, 0, 0, 0}, rs[4] = {0, 0, 0, 0}, gm[4] = {0, 0, 0, 0}, bt[4] = {0, 0, 0, 0};
        if (MODE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int g = (cc + j) / cg;          // per element: a float4 may straddle two groups when cg % 4 != 0
                mu[j] = p.mean[n * p.G + g];
                rs[j] = p.rstd[n * p.G + g];
                gm[j] = p.gamma[cc + j];
                bt[j] = p.beta[cc + j];
            }
        }
// (manual prefetch below)
        for (int r = row0; r < nrows; r += rpi) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)r * ld);
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { a0[j] += v[j]; a1[j] += v[j] * v[j]; }
            } else {
                const f32x4 d = *reinterpret_cast<const f32x4*>(dptr + (size_t)r * C + cc);
                f32x4 rr = {0.f, 0.f, 0.f, 0.f};
                if (rptr) rr = *reinterpret_cast<const f32x4*>(rptr + (size_t)r * C + cc);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (v[j] - mu[j]) * rs[j];
                    const float z = xh * gm[j] + bt[j] + rr[j];
                    const float dz = d[j] * act_bwd(z, p.act);
                    a0[j] += dz;
                    a1[j] += dz * xh;
                }
            }
        }
        float* slot = bins + (size_t)row0 * 2 * C;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            slot[cc + j] = a0[j];
            slot[C + cc + j] = a1[j];
        }
    }
    __syncthreads();
    double* out = p.partial + ((size_t)n * p.nchunk + chunk) * 2 * C;
    for (int i = tid; i < 2 * C; i += 256) {
        double a = 0.0;
        for (int l = 0; l < rpi; ++l) a += (double)bins[(size_t)l * 2 * C + i];
        out[i] = a;
    }
}

// one block per (n, group): sums the chunk partials of the group's channels in fp64.
// MODE 0 -> mean / rstd.  MODE 1 -> colsum[n][2][C] (fp32) for the group's channels.
// The statistics pass without reading the tensor: the fp32 conv epilogue (csrc/igemm_h.hip, conv_igemm_h<.., float, ..>) left
// the sums of every 64-row block; this kernel adds a chunk's blocks in double, in block order, into the same [N][nchunk][2][C] partials
// gn_colreduce<0> writes (rows_per_chunk counts 64-row blocks here).  Thread = one column of the virtual [2][C] slab.
__global__ __launch_bounds__(256) void gn_reduce_blocks_f32(const GnDesc p) {
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int nb = p.S >> 6;
    const int b0 = chunk * p.rows_per_chunk, b1 = min(nb, b0 + p.rows_per_chunk);
    const int C = p.C, C1 = p.st2 ? p.C1 : C, C2 = C - C1;
    double* dst = p.partial + ((size_t)n * p.nchunk + chunk) * 2 * C;
    for (int col = threadIdx.x; col < 2 * C; col += 256) {
        const int half = col >= C, c = half ? col - C : col;
        const bool first = c < C1;
        const float* src = first ? p.st1 + (size_t)half * C1 + c : p.st2 + (size_t)half * C2 + (c - C1);
        const size_t stride = first ? 2 * (size_t)C1 : 2 * (size_t)C2;
        const float* q = src + ((size_t)n * nb + b0) * stride;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int b = b0;
        for (; b + 3 < b1; b += 4, q += 4 * stride) {
            a0 += (double)q[0];
            a1 += (double)q[stride];
            a2 += (double)q[2 * stride];
            a3 += (double)q[3 * stride];
        }
        for (; b < b1; ++b, q += stride) a0 += (double)q[0];
        dst[col] = (a0 + a1) + (a2 + a3);
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void gn_finalize(const GnDesc p) {
    __shared__ double red[2][4];
    const int n = blockIdx.x / p.G, g = blockIdx.x % p.G, tid = threadIdx.x, C = p.C, cg = C / p.G;
    if (MODE == 0) {
        double s = 0.0, q = 0.0;
        for (int i = tid; i < p.nchunk * cg; i += 256) {
            const int k = i / cg, c = g * cg + i % cg;
            const double* src = p.partial + ((size_t)n * p.nchunk + k) * 2 * C;
            s += src[c];
            q += src[C + c];
        }
        s = wave_sum_d(s);
        q = wave_sum_d(q);
        if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = q; }
        __syncthreads();
        if (tid == 0) {
            s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
            q = red[1][0] + red[1][1] + red[1][2] + red[1][3];
            const double cnt = (double)p.S * cg;
            const double mu = s / cnt;
            double var = q / cnt - mu * mu;
            if (var < 0.0) var = 0.0;
            p.mean[n * p.G + g] = (float)mu;
            p.rstd[n * p.G + g] = (float)(1.0 / sqrt(var + (double)p.eps));
        }
    } else {
        // one wave per output (which, channel): lanes split the chunk partials, then a wave reduction; the group's two gamma-weighted
        // sums (what gn_apply_bwd needs per element) are formed here once instead of per float4 there
        __shared__ float cs_s[2][64];
        const int lane = tid & 63, wid = tid >> 6;
        for (int o = wid; o < 2 * cg; o += 4) {
            const int which = o / cg, c = g * cg + o % cg;
            double s = 0.0;
            for (int k = lane; k < p.nchunk; k += 64) s += p.partial[((size_t)n * p.nchunk + k) * 2 * C + which * C + c];
            s = wave_sum_d(s);
            if (lane == 0) {
                p.colsum[(size_t)n * 2 * C + which * C + c] = (float)s;
                if (cg <= 64) cs_s[which][o % cg] = (float)s;
            }
        }
        __syncthreads();
        if (tid < 2 && p.gsum) {
            float a = 0.f;
            for (int j = 0; j < cg; ++j) {
                const float v = cg <= 64 ? cs_s[tid][j] : p.colsum[(size_t)n * 2 * C + tid * C + g * cg + j];
                a += p.gamma[g * cg + j] * v;
            }
            p.gsum[(size_t)(n * p.G + g) * 2 + tid] = a;
        }
    }
}

// The apply pass of the large path, sample-major: grid (chunks, N), 32-bit indexing inside one sample with the row / column split by a
// launch-invariant reciprocal, four independent 16-B loads in flight per thread, one (mean, rstd) pair per float4 (needs cg % 4 == 0).
// Same arithmetic, in the same order, as gn_apply_fwd below -- which divides 64-bit indices three times per float4 and ran at 3.2 TB/s.
__global__ __launch_bounds__(256) void gn_apply_fwd_rows(const GnDesc p) {
    const int C = p.C, L4 = C >> 2, cg = C / p.G;
    const uint32_t per_n = (uint32_t)p.S * (uint32_t)L4;
    const int n = blockIdx.y;
    const int C1 = p.x2 ? p.C1 : C, C2 = C - C1;
    const float* xa = p.x + (size_t)n * p.S * C1;
    const float* xb = p.x2 ? p.x2 + (size_t)n * p.S * C2 : nullptr;
    const float* rs = p.residual ? p.residual + (size_t)n * p.S * C : nullptr;
    float* yo = p.y + (size_t)n * p.S * C;
    const float* mean = p.mean + (size_t)n * p.G;
    const float* rstd = p.rstd + (size_t)n * p.G;
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t i0 = blockIdx.x * 256u + threadIdx.x; i0 < per_n; i0 += 4 * stride) {
        f32x4 v[4], r[4];
        uint32_t row[4];
        int cc[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t idx = i0 + k * stride;
            ok[k] = idx < per_n;
            const uint32_t id = ok[k] ? idx : 0u;
            row[k] = (L4 == 1) ? id : (__umulhi(id, p.l4_magic) >> p.l4_shift);
            cc[k] = (int)(id - row[k] * (uint32_t)L4) * 4;
            const float* src = (cc[k] < C1) ? xa + (size_t)row[k] * C1 + cc[k] : xb + (size_t)row[k] * C2 + (cc[k] - C1);
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            v[k] = ok[k] ? *reinterpret_cast<const f32x4*>(src) : z4;
            r[k] = (ok[k] && rs) ? *reinterpret_cast<const f32x4*>(rs + (size_t)row[k] * C + cc[k]) : z4;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!ok[k]) continue;
            const int c = cc[k], g = c / cg;
            const float mu = mean[g], rsd = rstd[g];
            const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + c), bt = *reinterpret_cast<const f32x4*>(p.beta + c);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = (v[k][j] - mu) * rsd * gm[j] + bt[j] + r[k][j];
                float a = act_fwd(z, p.act);
                if (p.film) a = p.film[(size_t)n * p.film_ld + c + j] * a + p.film[(size_t)n * p.film_ld + C + c + j];
                o[j] = a;
            }
            const size_t off4 = ((size_t)row[k] * C + c) >> 2;
            reinterpret_cast<f32x4*>(yo)[off4] = o;
            if (p.yh) gn_store_twin4(p.yh + (size_t)n * p.S * C, off4, o, p.yh_f16);
        }
    }
}

__global__ __launch_bounds__(256) void gn_apply_fwd(const GnDesc p) {
    const int C = p.C, L4 = C >> 2, cg = C / p.G;
    const size_t per_n = (size_t)p.S * L4, total = per_n * p.N;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(p.x);
    const f32x4* r4 = reinterpret_cast<const f32x4*>(p.residual);
    f32x4* y4 = reinterpret_cast<f32x4*>(p.y);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int n = (int)(i / per_n);
        const int c4 = (int)(i % L4);
        f32x4 v;
        if (p.x2) {
            const size_t row = i / L4;
            const int cc = c4 * 4;
            v = (cc < p.C1) ? *reinterpret_cast<const f32x4*>(p.x + row * p.C1 + cc)
                            : *reinterpret_cast<const f32x4*>(p.x2 + row * (C - p.C1) + (cc - p.C1));
        } else {
            v = x4[i];
        }
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (r4) r = r4[i];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c4 * 4 + j;
            const int g = c / cg;
            float z = (v[j] - p.mean[n * p.G + g]) * p.rstd[n * p.G + g] * p.gamma[c] + p.beta[c] + r[j];
            float a = act_fwd(z, p.act);
            if (p.film) a = p.film[(size_t)n * p.film_ld + c] * a + p.film[(size_t)n * p.film_ld + C + c];
            o[j] = a;
        }
        y4[i] = o;
        if (p.yh) gn_store_twin4(p.yh, i, o, p.yh_f16);
    }
}

// dx = rstd * (gamma * dz - (A1 + xhat * A2) / cnt),  A1 = sum_{c in g} gamma_c * colsum0[n][c], A2 likewise with colsum1
__global__ __launch_bounds__(256) void gn_apply_bwd(const GnDesc p) {
    const int C = p.C, L4 = C >> 2, cg = C / p.G;
    const size_t per_n = (size_t)p.S * L4, total = per_n * p.N;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(p.x);
    const f32x4* d4 = reinterpret_cast<const f32x4*>(p.dout);
    const f32x4* r4 = reinterpret_cast<const f32x4*>(p.residual);
    f32x4* y4 = reinterpret_cast<f32x4*>(p.y);
    f32x4* dr4 = reinterpret_cast<f32x4*>(p.dres);
    const float inv_cnt = 1.0f / ((float)p.S * cg);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int n = (int)(i / per_n);
        const int c4 = (int)(i % L4);
        f32x4 v = x4[i], d = d4[i];
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (r4) r = r4[i];
        f32x4 o, dzv;
        int gprev = -1;
        float A1 = 0.f, A2 = 0.f, mu = 0.f, rs = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c4 * 4 + j;
            const int g = c / cg;
            if (g != gprev) {            // group constants change only when the float4 crosses into another group
                gprev = g;
                mu = p.mean[n * p.G + g];
                rs = p.rstd[n * p.G + g];
                A1 = p.gsum[(size_t)(n * p.G + g) * 2];
                A2 = p.gsum[(size_t)(n * p.G + g) * 2 + 1];
            }
            const float xh = (v[j] - mu) * rs;
            const float z = xh * p.gamma[c] + p.beta[c] + r[j];
            const float dz = d[j] * act_bwd(z, p.act);
            dzv[j] = dz;
            o[j] = rs * (p.gamma[c] * dz - (A1 + xh * A2) * inv_cnt);
        }
        y4[i] = o;
        if (p.yh) gn_store_twin4(p.yh, i, o, p.yh_f16);
        if (dr4) dr4[i] = dzv;
    }
}

// element(s) of the tensor a small-path launch normalises: a plain read, or the sum of the producing conv's split-K slabs in slab
// order (+ bias / + residual), written back to `sout` for the other readers (see GnDesc::slabs)
__device__ __forceinline__ float gn_src1(const GnDesc& p, const float* dense, size_t off, int c) {
    if (p.nslab <= 0) return dense[off];
    float t = 0.f;
    for (int sI = 0; sI < p.nslab; ++sI) t += p.slabs[(size_t)sI * p.slab_stride + off];
    if (p.cbias) t += p.cbias[c];
    if (p.sresid) t += p.sresid[off];
    if (p.sout) p.sout[off] = t;
    return t;
}
__device__ __forceinline__ f32x4 gn_src4(const GnDesc& p, const float* dense, size_t off, int c) {
    if (p.nslab <= 0) return *reinterpret_cast<const f32x4*>(dense + off);
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    int sI = 0;
    for (; sI + 2 <= p.nslab; sI += 2) {             // two independent 16-B loads in flight, added in slab order
        const f32x4 a = *reinterpret_cast<const f32x4*>(p.slabs + (size_t)sI * p.slab_stride + off);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.slabs + (size_t)(sI + 1) * p.slab_stride + off);
        t += a;
        t += b;
    }
    if (sI < p.nslab) t += *reinterpret_cast<const f32x4*>(p.slabs + (size_t)sI * p.slab_stride + off);
    if (p.cbias) t += *reinterpret_cast<const f32x4*>(p.cbias + c);
    if (p.sresid) t += *reinterpret_cast<const f32x4*>(p.sresid + off);
    if (p.sout) *reinterpret_cast<f32x4*>(p.sout + off) = t;
    return t;
}

// -------------------------------------------------------------------------------------------- small path
// one workgroup per (n, g): E = S * cg elements staged in LDS.  Forward: two-pass (centred) variance.
// sum of the per-wave partials of a workgroup, in wave order (deterministic); pairs first for the four-wave case's historical order
template <int NW>
__device__ __forceinline__ float gn_wave_partials(const float* red) {
    if (NW == 4) return red[0] + red[1] + red[2] + red[3];
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += red[w];
    return t;
}
template <bool VEC, int NT>
__global__ __launch_bounds__(NT) void gn_small_fwd(const GnDesc p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [E] + 8 scratch
    const int n = blockIdx.x / p.G, g = blockIdx.x % p.G, tid = threadIdx.x;
    const int C = p.C, cg = C / p.G, E = p.S * cg;
    float* red = sm + E;
    const size_t base = (size_t)n * p.S * C + (size_t)g * cg;
    float s = 0.f;
    if (VEC) {               // cg % 4 == 0: 16-B loads (a group's channels of one row are contiguous)
        const int cg4 = cg >> 2, E4 = E >> 2;
#pragma unroll 4
        for (int i = tid; i < E4; i += NT) {
            const int row = i / cg4, c4 = i - row * cg4;
            const f32x4 v = gn_src4(p, p.x, base + (size_t)row * C + c4 * 4, g * cg + c4 * 4);
            *reinterpret_cast<f32x4*>(sm + i * 4) = v;
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
    } else {
#pragma unroll 4
        for (int i = tid; i < E; i += NT) {
            const int row = i / cg, cc = i - row * cg;
            const float v = gn_src1(p, p.x, base + (size_t)row * C + cc, g * cg + cc);
            sm[i] = v;
            s += v;
        }
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float mu = gn_wave_partials<NT / 64>(red) / (float)E;
    __syncthreads();
    float q = 0.f;
#pragma unroll 4
    for (int i = tid; i < E; i += NT) { const float d = sm[i] - mu; q += d * d; }
    q = wave_sum(q);
    if ((tid & 63) == 0) red[tid >> 6] = q;
    __syncthreads();
    const float var = gn_wave_partials<NT / 64>(red) / (float)E;
    const float rs = 1.0f / sqrtf(var + p.eps);
    if (tid == 0) { p.mean[n * p.G + g] = mu; p.rstd[n * p.G + g] = rs; }
    if (VEC) {
        const int cg4 = cg >> 2, E4 = E >> 2;
#pragma unroll 2
        for (int i = tid; i < E4; i += NT) {
            const int row = i / cg4, c4 = i - row * cg4, c = g * cg + c4 * 4;
            const size_t off = base + (size_t)row * C + c4 * 4;
            const f32x4 xv = *reinterpret_cast<const f32x4*>(sm + i * 4);
            const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + c), bt = *reinterpret_cast<const f32x4*>(p.beta + c);
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            if (p.residual) r = *reinterpret_cast<const f32x4*>(p.residual + off);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = act_fwd((xv[j] - mu) * rs * gm[j] + bt[j] + r[j], p.act);
                if (p.film) a = p.film[(size_t)n * p.film_ld + c + j] * a + p.film[(size_t)n * p.film_ld + C + c + j];
                o[j] = a;
            }
            *reinterpret_cast<f32x4*>(p.y + off) = o;
            if (p.yh) gn_store_twin4(p.yh, off >> 2, o, p.yh_f16);
        }
        return;
    }
#pragma unroll 4
    for (int i = tid; i < E; i += NT) {
        const int row = i / cg, cc = i - row * cg, c = g * cg + cc;
        const size_t off = base + (size_t)row * C + cc;
        float z = (sm[i] - mu) * rs * p.gamma[c] + p.beta[c];
        if (p.residual) z += p.residual[off];
        float a = act_fwd(z, p.act);
        if (p.film) a = p.film[(size_t)n * p.film_ld + c] * a + p.film[(size_t)n * p.film_ld + C + c];
        p.y[off] = a;
        if (p.yh) p.yh[off] = gn_f2h(a, p.yh_f16);
    }
}

// backward of the small path.  Also emits per-(n, c) sums: colsum[n][0][c] = sum_s dz, colsum[n][1][c] = sum_s dz*xhat
// (dgamma / dbeta = their sum over n: gn_param_grads / gn_param_grads_multi) and dfilm[n][0][c] = sum_s dout * a,
// dfilm[n][1][c] = sum_s dout.  Deterministic: the per-element terms are staged in LDS and every column is summed over its rows in
// a fixed order (thread = (row slice, column), then the slices in order) -- no float atomics, in LDS or in HBM.
template <bool VEC, int NT>
__global__ __launch_bounds__(NT) void gn_small_bwd(const GnDesc p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // xhat[E], dz[E], (film: dout*a[E], dout[E]), part[nsl][4][cg], red[8]
    const int n = blockIdx.x / p.G, g = blockIdx.x % p.G, tid = threadIdx.x;
    const int C = p.C, cg = C / p.G, E = p.S * cg;
    const bool film = p.film != nullptr;
    float* xh = sm;
    float* dzs = sm + E;
    float* fa = sm + 2 * E;                 // film only
    float* fd = sm + 3 * E;
    const int nsl = cg >= NT ? 1 : NT / cg;                 // row slices summed in parallel per column
    float* part = sm + (film ? 4 : 2) * E;  // [nsl][4][cg]
    float* red = part + (size_t)nsl * 4 * cg;
    const size_t base = (size_t)n * p.S * C + (size_t)g * cg;
    const float mu = p.mean[n * p.G + g], rs = p.rstd[n * p.G + g];
    float A1 = 0.f, A2 = 0.f;
    if (VEC && !film) {
        const int cg4 = cg >> 2, E4 = E >> 2;
#pragma unroll 2
        for (int i = tid; i < E4; i += NT) {
            const int row = i / cg4, c4 = i - row * cg4, c = g * cg + c4 * 4;
            const size_t off = base + (size_t)row * C + c4 * 4;
            const f32x4 xv = *reinterpret_cast<const f32x4*>(p.x + off), dv = gn_src4(p, p.dout, off, c);
            const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + c), bt = *reinterpret_cast<const f32x4*>(p.beta + c);
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            if (p.residual) r = *reinterpret_cast<const f32x4*>(p.residual + off);
            f32x4 hv, zv;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float h = (xv[j] - mu) * rs;
                const float z = h * gm[j] + bt[j] + r[j];
                const float dz = dv[j] * act_bwd(z, p.act);
                hv[j] = h;
                zv[j] = dz;
                A1 += dz * gm[j];
                A2 += dz * gm[j] * h;
            }
            *reinterpret_cast<f32x4*>(xh + i * 4) = hv;
            *reinterpret_cast<f32x4*>(dzs + i * 4) = zv;
        }
    } else {
#pragma unroll 4
    for (int i = tid; i < E; i += NT) {
        const int row = i / cg, cc = i - row * cg, c = g * cg + cc;
        const size_t off = base + (size_t)row * C + cc;
        const float h = (p.x[off] - mu) * rs;
        float z = h * p.gamma[c] + p.beta[c];
        if (p.residual) z += p.residual[off];
        const float dout = gn_src1(p, p.dout, off, c);
        float da = dout;
        if (film) {
            const float a = act_fwd(z, p.act);
            da = dout * p.film[(size_t)n * p.film_ld + c];
            fa[i] = dout * a;
            fd[i] = dout;
        }
        const float dz = da * act_bwd(z, p.act);
        xh[i] = h;
        dzs[i] = dz;
        A1 += dz * p.gamma[c];
        A2 += dz * p.gamma[c] * h;
    }
    }
    A1 = wave_sum(A1);
    A2 = wave_sum(A2);
    if ((tid & 63) == 0) { red[tid >> 6] = A1; red[NT / 64 + (tid >> 6)] = A2; }
    __syncthreads();
    A1 = gn_wave_partials<NT / 64>(red);
    A2 = gn_wave_partials<NT / 64>(red + NT / 64);
    const float inv = 1.0f / (float)E;
    if (VEC) {
        const int cg4 = cg >> 2, E4 = E >> 2;
#pragma unroll 2
        for (int i = tid; i < E4; i += NT) {
            const int row = i / cg4, c4 = i - row * cg4, c = g * cg + c4 * 4;
            const size_t off = base + (size_t)row * C + c4 * 4;
            const f32x4 hv = *reinterpret_cast<const f32x4*>(xh + i * 4), zv = *reinterpret_cast<const f32x4*>(dzs + i * 4);
            const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + c);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = rs * (gm[j] * zv[j] - (A1 + hv[j] * A2) * inv);
            *reinterpret_cast<f32x4*>(p.y + off) = o;
            if (p.yh) gn_store_twin4(p.yh, off >> 2, o, p.yh_f16);
            if (p.dres) *reinterpret_cast<f32x4*>(p.dres + off) = zv;
        }
    } else {
#pragma unroll 4
    for (int i = tid; i < E; i += NT) {
        const int row = i / cg, cc = i - row * cg, c = g * cg + cc;
        const size_t off = base + (size_t)row * C + cc;
        const float dxv = rs * (p.gamma[c] * dzs[i] - (A1 + xh[i] * A2) * inv);
        p.y[off] = dxv;
        if (p.yh) p.yh[off] = gn_f2h(dxv, p.yh_f16);
        if (p.dres) p.dres[off] = dzs[i];
    }
    }
    // column sums: thread (slice, cc) adds rows slice, slice + nsl, ... in order; then the slices in order
    for (int t = tid; t < nsl * cg; t += NT) {
        const int sl = t / cg, cc = t - sl * cg;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int row = sl; row < p.S; row += nsl) {
            const int i = row * cg + cc;
            const float dz = dzs[i];
            s0 += dz;
            s1 += dz * xh[i];
            if (film) { s2 += fa[i]; s3 += fd[i]; }
        }
        float* dst = part + (size_t)sl * 4 * cg;
        dst[cc] = s0; dst[cg + cc] = s1; dst[2 * cg + cc] = s2; dst[3 * cg + cc] = s3;
    }
    __syncthreads();
    for (int cc = tid; cc < cg; cc += NT) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int sl = 0; sl < nsl; ++sl) {
            const float* src = part + (size_t)sl * 4 * cg;
            s0 += src[cc]; s1 += src[cg + cc]; s2 += src[2 * cg + cc]; s3 += src[3 * cg + cc];
        }
        const int c = g * cg + cc;
        p.colsum[(size_t)n * 2 * C + c] = s0;
        p.colsum[(size_t)n * 2 * C + C + c] = s1;
        if (p.dfilm) {
            p.dfilm[(size_t)n * p.film_ld + c] = s2;
            p.dfilm[(size_t)n * p.film_ld + C + c] = s3;
        }
    }
}

// -------------------------------------------------------------------------------------------- wave path
// One WAVE per (n, group) for slabs of E = S * CG <= 1024 elements with CG in {16, 32, 64, 128}: the ConditionalUnet1D blocks
// (E = 512: 16 x 32, 8 x 64, 4 x 128) and the deep ResNet stages.  Everything lives in registers (E / 64 values per lane), the only
// cross-lane traffic is wave shuffles -- no LDS, no __syncthreads -- so a launch costs what its loads cost (3-4 us instead of the
// 9 / 15 us of the workgroup-per-slab kernels: these launches sit on the serial chain of the train step).  Element i of the slab
// (row i / CG, channel i % CG) belongs to lane i % 64, so a lane sees a fixed set of channels: per-channel sums over the rows
// (dgamma / dbeta / dFiLM contributions) are lane-local for CG >= 64 and need log2(64 / CG) shuffles otherwise.  Deterministic.
// the lane's `epl` elements of the tensor the wave normalises (forward: x, backward: dout): a plain read, or -- when the producing
// conv left its split-K slabs -- the slab sum in slab order (+ bias, + residual), loaded in batches of four slabs so that up to
// 4 * epl independent loads are in flight per lane (the launch is latency-bound: two waves per CU).
template <int CG, int MAXE>
__device__ __forceinline__ void gn_wave_gather(const GnDesc& p, const float* dense, size_t base, int g, int lane, int epl, float (&v)[MAXE]) {
    const int C = p.C;
    size_t off[MAXE];
#pragma unroll
    for (int j = 0; j < MAXE; ++j) {
        const int i = lane + 64 * j;
        off[j] = base + (size_t)(i / CG) * C + (i % CG);
        v[j] = 0.f;
    }
    if (p.nslab <= 0) {
#pragma unroll
        for (int j = 0; j < MAXE; ++j)
            if (j < epl) v[j] = dense[off[j]];
        return;
    }
    for (int s0 = 0; s0 < p.nslab; s0 += 4) {
        float u[4][MAXE];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < MAXE; ++j)
                u[q][j] = (s0 + q < p.nslab && j < epl) ? p.slabs[(size_t)(s0 + q) * p.slab_stride + off[j]] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < MAXE; ++j) v[j] += u[q][j];
    }
#pragma unroll
    for (int j = 0; j < MAXE; ++j)
        if (j < epl) {
            if (p.cbias) v[j] += p.cbias[g * CG + (lane + 64 * j) % CG];
            if (p.sresid) v[j] += p.sresid[off[j]];
            if (p.sout) p.sout[off[j]] = v[j];
        }
}

template <int CG, int MAXE>
__global__ __launch_bounds__(64) void gn_wave_fwd(const GnDesc p) {
    const int lane = threadIdx.x;
    const int wv = blockIdx.x;
    const int n = wv / p.G, g = wv - n * p.G;
    const int C = p.C, epl = (p.S * CG) >> 6;
    const size_t base = (size_t)n * p.S * C + (size_t)g * CG;
    // every load that does not depend on the statistics goes out BEFORE the gather (whose slab loop ends in a wait): the launch is a
    // chain of memory round trips on one wave per SIMD, and this way the affine / residual / FiLM operands travel with the first of them
    float gmv[MAXE], btv[MAXE], rsd[MAXE], f0[MAXE], f1[MAXE];
#pragma unroll
    for (int j = 0; j < MAXE; ++j) {
        gmv[j] = btv[j] = rsd[j] = f0[j] = f1[j] = 0.f;
        if (j < epl) {
            const int i = lane + 64 * j, row = i / CG, cc = i % CG, c = g * CG + cc;
            gmv[j] = p.gamma[c];
            btv[j] = p.beta[c];
            if (p.residual) rsd[j] = p.residual[base + (size_t)row * C + cc];
            if (p.film) { f0[j] = p.film[(size_t)n * p.film_ld + c]; f1[j] = p.film[(size_t)n * p.film_ld + C + c]; }
        }
    }
    float v[MAXE];
    gn_wave_gather<CG, MAXE>(p, p.x, base, g, lane, epl, v);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXE; ++j) s += v[j];
    const float inv = 1.0f / (float)(p.S * CG);
    const float mu = wave_sum(s) * inv;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAXE; ++j)
        if (j < epl) { const float d = v[j] - mu; q += d * d; }
    const float rs = 1.0f / sqrtf(wave_sum(q) * inv + p.eps);
    if (lane == 0) { p.mean[wv] = mu; p.rstd[wv] = rs; }
#pragma unroll
    for (int j = 0; j < MAXE; ++j)
        if (j < epl) {
            const int i = lane + 64 * j, row = i / CG, cc = i % CG;
            const size_t off = base + (size_t)row * C + cc;
            float z = (v[j] - mu) * rs * gmv[j] + btv[j];
            if (p.residual) z += rsd[j];
            float a = act_fwd(z, p.act);
            if (p.film) a = f0[j] * a + f1[j];
            p.y[off] = a;
            if (p.yh) p.yh[off] = gn_f2h(a, p.yh_f16);
        }
}

template <int CG, int MAXE>
__global__ __launch_bounds__(64) void gn_wave_bwd(const GnDesc p) {
    constexpr int NCOL = CG >= 64 ? CG / 64 : 1;          // channels a lane owns
    const int lane = threadIdx.x;
    const int wv = blockIdx.x;
    const int n = wv / p.G, g = wv - n * p.G;
    const int C = p.C, epl = (p.S * CG) >> 6;
    const size_t base = (size_t)n * p.S * C + (size_t)g * CG;
    const float mu = p.mean[wv], rs = p.rstd[wv];
    const bool film = p.film != nullptr;
    float xh[MAXE], dz[MAXE], dov[MAXE];
    // the saved input, the affine / residual / FiLM operands: requested ahead of the gather of dout (see gn_wave_fwd)
    float xv[MAXE], gmv[MAXE], btv[MAXE], rsd[MAXE], f0[MAXE];
#pragma unroll
    for (int j = 0; j < MAXE; ++j) {
        xv[j] = gmv[j] = btv[j] = rsd[j] = f0[j] = 0.f;
        if (j < epl) {
            const int i = lane + 64 * j, row = i / CG, cc = i % CG, c = g * CG + cc;
            const size_t off = base + (size_t)row * C + cc;
            xv[j] = p.x[off];
            gmv[j] = p.gamma[c];
            btv[j] = p.beta[c];
            if (p.residual) rsd[j] = p.residual[off];
            if (film) f0[j] = p.film[(size_t)n * p.film_ld + c];
        }
    }
    gn_wave_gather<CG, MAXE>(p, p.dout, base, g, lane, epl, dov);
    float c0[NCOL], c1[NCOL], c2[NCOL], c3[NCOL];
#pragma unroll
    for (int k = 0; k < NCOL; ++k) c0[k] = c1[k] = c2[k] = c3[k] = 0.f;
    float A1 = 0.f, A2 = 0.f;
#pragma unroll
    for (int j = 0; j < MAXE; ++j) {
        xh[j] = dz[j] = 0.f;
        if (j < epl) {
            const float h = (xv[j] - mu) * rs;
            const float gm = gmv[j];
            float z = h * gm + btv[j];
            if (p.residual) z += rsd[j];
            const float dout = dov[j];
            float da = dout;
            const int k = CG >= 64 ? (j % NCOL) : 0;
            if (film) {
                const float a = act_fwd(z, p.act);
                da = dout * f0[j];
                c2[k] += dout * a;
                c3[k] += dout;
            }
            const float d = da * act_bwd(z, p.act);
            xh[j] = h;
            dz[j] = d;
            c0[k] += d;
            c1[k] += d * h;
            A1 += d * gm;
            A2 += d * gm * h;
        }
    }
    A1 = wave_sum(A1);
    A2 = wave_sum(A2);
    const float inv = 1.0f / (float)(p.S * CG);
#pragma unroll
    for (int j = 0; j < MAXE; ++j)
        if (j < epl) {
            const int i = lane + 64 * j, row = i / CG, cc = i % CG;
            const size_t off = base + (size_t)row * C + cc;
            const float dxv = rs * (gmv[j] * dz[j] - (A1 + xh[j] * A2) * inv);
            p.y[off] = dxv;
            if (p.yh) p.yh[off] = gn_f2h(dxv, p.yh_f16);
            if (p.dres) p.dres[off] = dz[j];
        }
    if (CG < 64) {                       // lanes l, l + CG, l + 2 CG ... hold the same channel: fixed xor tree
#pragma unroll
        for (int o = 32; o >= CG; o >>= 1) {
            c0[0] += __shfl_xor(c0[0], o, 64);
            c1[0] += __shfl_xor(c1[0], o, 64);
            if (film) { c2[0] += __shfl_xor(c2[0], o, 64); c3[0] += __shfl_xor(c3[0], o, 64); }
        }
    }
    if (lane < (CG < 64 ? CG : 64)) {
#pragma unroll
        for (int k = 0; k < NCOL; ++k) {
            const int c = g * CG + lane + 64 * k;
            p.colsum[(size_t)n * 2 * C + c] = c0[k];
            p.colsum[(size_t)n * 2 * C + C + c] = c1[k];
            if (p.dfilm) {
                p.dfilm[(size_t)n * p.film_ld + c] = c2[k];
                p.dfilm[(size_t)n * p.film_ld + C + c] = c3[k];
            }
        }
    }
}

// ---- float4 form of the wave path (round 4): slabs of E = S * CG = 256 * NJ elements, CG in {16, 32, 64, 128}, NJ in {1, 2, 4} -- the
// deep ResNet stages (CG = 16: 8 x 8 x 256 channels, 4 x 4 x 512) and every Conv1dBlock of the ConditionalUnet1D (E = 512 / 1024).
// gn_wave_*<CG, .> above reads them with E / 64 scalar loads per tensor and lane and keeps eight E/64-element arrays per lane.  Here a
// row of the slab (CG floats) is CG / 4 lanes x one b128 load, a pass of the wave covers 256 / CG rows, NJ passes the slab: every
// instruction moves whole 64 ... 512-B row segments, a lane owns four FIXED channels (gamma / beta / FiLM operands once per lane), and
// the per-channel sums (dgamma / dbeta / dFiLM contributions) are xor shuffles over the lanes of a 16-B column.
template <int NJ>
__device__ __forceinline__ void gn_wv_gather(const GnDesc& p, const float* dense, const size_t (&off)[NJ], int c0, f32x4 (&v)[NJ]) {
    if (p.nslab <= 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) v[j] = *reinterpret_cast<const f32x4*>(dense + off[j]);
        return;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // slab order; eight slabs in flight (<= 128 registers at NJ = 4): a launch is one wave per SIMD, every round of this loop is a full
    // memory round trip, and the deep ResNet stages arrive in 8-13 slabs
    constexpr int INFL = 8;
    for (int s0 = 0; s0 < p.nslab; s0 += INFL) {
        f32x4 u[INFL][NJ];
#pragma unroll
        for (int q = 0; q < INFL; ++q)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                u[q][j] = (s0 + q < p.nslab) ? *reinterpret_cast<const f32x4*>(p.slabs + (size_t)(s0 + q) * p.slab_stride + off[j])
                                             : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < INFL; ++q)
#pragma unroll
            for (int j = 0; j < NJ; ++j) v[j] += u[q][j];
    }
    if (p.cbias) {
        const f32x4 cb = *reinterpret_cast<const f32x4*>(p.cbias + c0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) v[j] += cb;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (p.sresid) v[j] += *reinterpret_cast<const f32x4*>(p.sresid + off[j]);
        if (p.sout) *reinterpret_cast<f32x4*>(p.sout + off[j]) = v[j];
    }
}

template <int CG, int NJ>
__global__ __launch_bounds__(64) void gn_wavev_fwd(const GnDesc p) {
    constexpr int LPR = CG / 4, RPP = 64 / LPR;                 // lanes per row, rows per pass
    const int lane = threadIdx.x, wv = blockIdx.x;
    const int n = wv / p.G, g = wv - n * p.G;
    const int C = p.C, c0 = g * CG + (lane % LPR) * 4;
    size_t off[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) off[j] = ((size_t)n * p.S + (lane / LPR) + RPP * j) * C + c0;
    const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + c0), bt = *reinterpret_cast<const f32x4*>(p.beta + c0);
    const bool film = p.film != nullptr;
    f32x4 f0 = {1.f, 1.f, 1.f, 1.f}, f1 = {0.f, 0.f, 0.f, 0.f};
    if (film) {
        f0 = *reinterpret_cast<const f32x4*>(p.film + (size_t)n * p.film_ld + c0);
        f1 = *reinterpret_cast<const f32x4*>(p.film + (size_t)n * p.film_ld + C + c0);
    }
    f32x4 rsd[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) rsd[j] = p.residual ? *reinterpret_cast<const f32x4*>(p.residual + off[j]) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 v[NJ];
    gn_wv_gather<NJ>(p, p.x, off, c0, v);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    const float inv = 1.0f / (float)(p.S * CG);
    const float mu = wave_sum(s) * inv;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mu; q += d * d; }
    const float rs = 1.0f / sqrtf(wave_sum(q) * inv + p.eps);
    if (lane == 0) { p.mean[wv] = mu; p.rstd[wv] = rs; }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float z = (v[j][e] - mu) * rs * gm[e] + bt[e];
            if (p.residual) z += rsd[j][e];
            float a = act_fwd(z, p.act);
            if (film) a = f0[e] * a + f1[e];
            o[e] = a;
        }
        if (p.post_nslab > 0) {                       // (the reduce kernel's order: slabs ascending from 0, then the bias, then the other addend)
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            for (int sl = 0; sl < p.post_nslab; ++sl) t += *reinterpret_cast<const f32x4*>(p.post_slabs + (size_t)sl * p.post_stride + off[j]);
            if (p.post_bias) t += *reinterpret_cast<const f32x4*>(p.post_bias + c0);
            o = t + o;
        } else if (p.post) {
            o = o + *reinterpret_cast<const f32x4*>(p.post + off[j]);
        }
        *reinterpret_cast<f32x4*>(p.y + off[j]) = o;
        if (p.yh) gn_store_twin_or_planes4(p, off[j], o);
    }
}

template <int CG, int NJ>
__global__ __launch_bounds__(64) void gn_wavev_bwd(const GnDesc p) {
    constexpr int LPR = CG / 4, RPP = 64 / LPR;
    const int lane = threadIdx.x, wv = blockIdx.x;
    const int n = wv / p.G, g = wv - n * p.G;
    const int C = p.C, c0 = g * CG + (lane % LPR) * 4;
    size_t off[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) off[j] = ((size_t)n * p.S + (lane / LPR) + RPP * j) * C + c0;
    const float mu = p.mean[wv], rs = p.rstd[wv];
    const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + c0), bt = *reinterpret_cast<const f32x4*>(p.beta + c0);
    const bool film = p.film != nullptr;
    f32x4 f0 = {1.f, 1.f, 1.f, 1.f};
    if (film) f0 = *reinterpret_cast<const f32x4*>(p.film + (size_t)n * p.film_ld + c0);
    f32x4 xv[NJ], rsd[NJ], dz[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        xv[j] = *reinterpret_cast<const f32x4*>(p.x + off[j]);
        rsd[j] = p.residual ? *reinterpret_cast<const f32x4*>(p.residual + off[j]) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    gn_wv_gather<NJ>(p, p.dout, off, c0, dz);
    f32x4 cs0 = {0.f, 0.f, 0.f, 0.f}, cs1 = {0.f, 0.f, 0.f, 0.f}, cs2 = {0.f, 0.f, 0.f, 0.f}, cs3 = {0.f, 0.f, 0.f, 0.f};
    float A1 = 0.f, A2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float h = (xv[j][e] - mu) * rs;
            float z = h * gm[e] + bt[e];
            if (p.residual) z += rsd[j][e];
            const float dout = dz[j][e];
            float da = dout;
            if (film) {
                da = dout * f0[e];
                cs2[e] += dout * act_fwd(z, p.act);
                cs3[e] += dout;
            }
            const float d = da * act_bwd(z, p.act);
            xv[j][e] = h;
            dz[j][e] = d;
            cs0[e] += d;
            cs1[e] += d * h;
            A1 += d * gm[e];
            A2 += d * gm[e] * h;
        }
    A1 = wave_sum(A1);
    A2 = wave_sum(A2);
    const float inv = 1.0f / (float)(p.S * CG);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (gm[e] * dz[j][e] - (A1 + xv[j][e] * A2) * inv);
        *reinterpret_cast<f32x4*>(p.y + off[j]) = o;
        if (p.yh) gn_store_twin_or_planes4(p, off[j], o);
        if (p.dres) *reinterpret_cast<f32x4*>(p.dres + off[j]) = dz[j];
    }
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1)                // lanes with equal lane % LPR hold the same four channels
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cs0[e] += __shfl_xor(cs0[e], o, 64);
            cs1[e] += __shfl_xor(cs1[e], o, 64);
            if (film) { cs2[e] += __shfl_xor(cs2[e], o, 64); cs3[e] += __shfl_xor(cs3[e], o, 64); }
        }
    if (lane < LPR) {
        *reinterpret_cast<f32x4*>(p.colsum + (size_t)n * 2 * C + c0) = cs0;
        *reinterpret_cast<f32x4*>(p.colsum + (size_t)n * 2 * C + C + c0) = cs1;
        if (p.dfilm) {
            *reinterpret_cast<f32x4*>(p.dfilm + (size_t)n * p.film_ld + c0) = cs2;
            *reinterpret_cast<f32x4*>(p.dfilm + (size_t)n * p.film_ld + C + c0) = cs3;
        }
    }
}
// 1 / 2 / 4 passes of the float4 wave kernels (0: not eligible)
static int gn_wavev_passes(const GnDesc& p, int cg) {
    if (!(cg == 16 || cg == 32 || cg == 64 || cg == 128) || (p.C & 3) || (p.film_ld & 3)) return 0;
    const long E = (long)p.S * cg;
    if (E != 256 && E != 512 && E != 1024) return 0;
    const uintptr_t a = (uintptr_t)p.x | (uintptr_t)p.y | (uintptr_t)p.residual | (uintptr_t)p.dout | (uintptr_t)p.dres | (uintptr_t)p.gamma |
                        (uintptr_t)p.beta | (uintptr_t)p.slabs | (uintptr_t)p.cbias |
                        (uintptr_t)p.sresid | (uintptr_t)p.sout | (uintptr_t)p.colsum | (uintptr_t)p.yh | (uintptr_t)p.film | (uintptr_t)p.dfilm |
                        (uintptr_t)p.post | (uintptr_t)p.post_slabs | (uintptr_t)p.post_bias;
    if ((a & 15) != 0 || (p.slab_stride & 3) != 0 || (p.post_stride & 3) != 0) return 0;
    return (int)(E / 256);
}
#define V2A_GNWV_LAUNCH(KERN, CGV, NJV) hipLaunchKernelGGL((KERN<CGV, NJV>), grid, block, 0, stream, p)
#define V2A_GNWV_NJ(KERN, CGV, NJ_)                                                                             \
    do {                                                                                                          \
        if ((NJ_) == 1) V2A_GNWV_LAUNCH(KERN, CGV, 1);                                                            \
        else if ((NJ_) == 2) V2A_GNWV_LAUNCH(KERN, CGV, 2);                                                       \
        else V2A_GNWV_LAUNCH(KERN, CGV, 4);                                                                       \
    } while (0)
#define V2A_GNWV(KERN, CG_, NJ_)                                                                                \
    do {                                                                                                          \
        if ((CG_) == 16) V2A_GNWV_NJ(KERN, 16, NJ_);                                                              \
        else if ((CG_) == 32) V2A_GNWV_NJ(KERN, 32, NJ_);                                                         \
        else if ((CG_) == 64) V2A_GNWV_NJ(KERN, 64, NJ_);                                                         \
        else V2A_GNWV_NJ(KERN, 128, NJ_);                                                                         \
    } while (0)

static bool gn_wave_ok(int S, int cg) {
    if (!(cg == 16 || cg == 32 || cg == 64 || cg == 128)) return false;
    const long E = (long)S * cg;
    return E % 64 == 0 && E / 64 <= 16;
}

// dgamma[c] = sum_n colsum[n][1][c], dbeta[c] = sum_n colsum[n][0][c]
__global__ __launch_bounds__(256) void gn_param_grads(const float* colsum, float* dgamma, float* dbeta, int N, int C, int accumulate) {
    __shared__ double sm[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int n = w; n < N; n += 4) { b += colsum[(size_t)n * 2 * C + c]; a += colsum[(size_t)n * 2 * C + C + c]; }
    if (accumulate && w == 0 && c < C) { a += dgamma[c]; b += dbeta[c]; }
    sm[0][w][threadIdx.x & 63] = a;
    sm[1][w][threadIdx.x & 63] = b;
    __syncthreads();
    if (w == 0 && c < C) {
        const int l = threadIdx.x;
        dgamma[c] = (float)(sm[0][0][l] + sm[0][1][l] + sm[0][2][l] + sm[0][3][l]);
        dbeta[c] = (float)(sm[1][0][l] + sm[1][1][l] + sm[1][2][l] + sm[1][3][l]);
    }
}

// the same for many GroupNorm layers in one launch: table rows = {colsum ptr, dgamma ptr, dbeta ptr, N, C}, work = (row, 64-channel block)
__global__ __launch_bounds__(256) void gn_param_grads_multi(const long long* table, const int* work) {
    __shared__ double sm[2][4][64];
    const long long* row = table + (size_t)work[2 * blockIdx.x] * 5;
    const float* colsum = reinterpret_cast<const float*>(row[0]);
    float* dgamma = reinterpret_cast<float*>(row[1]);
    float* dbeta = reinterpret_cast<float*>(row[2]);
    const int N = (int)row[3], C = (int)row[4];
    const int c = work[2 * blockIdx.x + 1] * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int n = w; n < N; n += 4) { b += colsum[(size_t)n * 2 * C + c]; a += colsum[(size_t)n * 2 * C + C + c]; }
    sm[0][w][threadIdx.x & 63] = a;
    sm[1][w][threadIdx.x & 63] = b;
    __syncthreads();
    if (w == 0 && c < C) {
        const int l = threadIdx.x;
        dgamma[c] = (float)(sm[0][0][l] + sm[0][1][l] + sm[0][2][l] + sm[0][3][l]);
        dbeta[c] = (float)(sm[1][0][l] + sm[1][1][l] + sm[1][2][l] + sm[1][3][l]);
    }
}

#define GN_SMALL_MAX 16384      // 64 KB of LDS forward, 128 KB backward: the ResNet layer-1 slabs (32 x 32 x 16) still take one launch

static size_t gn_colreduce_lds(int C) {           // [rpi][2][C] floats, rpi = row lanes of gn_colreduce
    const int L4 = C >> 2;
    const int rpi = L4 >= 256 ? 1 : 256 / L4;
    return (size_t)rpi * 2 * C * sizeof(float);
}

static void gn_chunks(int N, int S, int C, int* nchunk, int* rows) {
    // >= ~1024 workgroups overall, slabs of at least 16 KB
    int min_rows = (16384 / (C * 4));
    if (min_rows < 1) min_rows = 1;
    int want = (1024 + N - 1) / N;
    int r = (S + want - 1) / want;
    if (r < min_rows) r = min_rows;
    *rows = r;
    *nchunk = (S + r - 1) / r;
}

// post-activation addend of a GroupNorm forward launch (explicit operands of v2a_groupnorm_fwd_s; all null / 0: none)
struct GnPost { const float* dense; const float* slabs; const float* bias; int nslab; size_t stride; size_t yh3; };
extern "C" {

size_t v2a_groupnorm_workspace_bytes(int N, int S, int C, int G) {
    if ((long)S * (C / G) <= GN_SMALL_MAX) return 0;
    int nchunk, rows;
    gn_chunks(N, S, C, &nchunk, &rows);
    return (size_t)N * nchunk * 2 * C * sizeof(double) + (size_t)N * G * 2 * sizeof(float);     // chunk partials + per-group sums
}

int v2a_groupnorm_fwd_s(const float* x, const float* x2, int C1, const float* gamma, const float* beta, const float* residual,
                        const float* film, int film_ld, float* y, void* y_h, size_t yh_plane_stride, float* mean, float* rstd, int N, int S, int C,
                        int G, float eps, int act, const float* slabs, int nslab, size_t slab_stride, const float* cbias, const float* post,
                        const float* post_slabs, int post_nslab, size_t post_stride, const float* post_bias,
                        void* workspace, size_t workspace_bytes, hipStream_t stream);
int v2a_groupnorm_bwd_s(const float* x, const float* gamma, const float* beta, const float* residual, const float* film, int film_ld,
                        const float* dout, const float* mean, const float* rstd, float* dx, void* dx_h, size_t yh_plane_stride, float* dres,
                        float* dfilm, float* colsum, float* dgamma, float* dbeta, int accumulate_params, int N, int S, int C, int G, int act,
                        const float* slabs, int nslab, size_t slab_stride, const float* sresid, float* dout_sum,
                        void* workspace, size_t workspace_bytes, hipStream_t stream);
// y = film(act(gn(x) + residual)); mean/rstd [N*G] are saved for the backward.
// x2 != null: the input is the channel concat [x | x2] (decoder skip, reference unet.py:681) read from both sources in place.
int v2a_groupnorm_fwd_t(const float* x, const float* x2, int C1, const float* gamma, const float* beta, const float* residual,
                        const float* film, int film_ld, float* y, void* y_h, float* mean, float* rstd, int N, int S, int C, int G, float eps,
                        int act, void* workspace, size_t workspace_bytes, hipStream_t stream);
int v2a_groupnorm_fwd(const float* x, const float* x2, int C1, const float* gamma, const float* beta, const float* residual,
                      const float* film, int film_ld, float* y, float* mean, float* rstd, int N, int S, int C, int G, float eps, int act,
                      void* workspace, size_t workspace_bytes, hipStream_t stream) {
    return v2a_groupnorm_fwd_t(x, x2, C1, gamma, beta, residual, film, film_ld, y, nullptr, mean, rstd, N, S, C, G, eps, act, workspace,
                               workspace_bytes, stream);
}
// same, additionally writing the bf16 twin of y (y_h, may be null): the operand of the bf16-MFMA conv that consumes y
int v2a_groupnorm_fwd_t(const float* x, const float* x2, int C1, const float* gamma, const float* beta, const float* residual,
                        const float* film, int film_ld, float* y, void* y_h, float* mean, float* rstd, int N, int S, int C, int G, float eps,
                        int act, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    return v2a_groupnorm_fwd_s(x, x2, C1, gamma, beta, residual, film, film_ld, y, y_h, 0, mean, rstd, N, S, C, G, eps, act, nullptr, 0, 0,
                               nullptr, nullptr, nullptr, 0, 0, nullptr, workspace, workspace_bytes, stream);
}
// 1 when a GroupNorm over [N, S, C] with G groups runs on the wave path, i.e. accepts its input as split-K slabs (v2a_groupnorm_*_s)
int v2a_groupnorm_takes_slabs(int S, int C, int G) {
    return (G > 0 && C % G == 0 && (gn_wave_ok(S, C / G) || (long)S * (C / G) <= GN_SMALL_MAX)) ? 1 : 0;
}
// same as v2a_groupnorm_fwd_t; with nslab > 0 the normalised tensor is sum_s slabs[s][.] + cbias[c] (the split-K partial sums and bias
// of the conv that produces it: its reduce launch is folded into this one) and `x` receives that sum (kept for the backward).
static int gn_fwd_impl(const float* x, const float* x2, int C1, const float* gamma, const float* beta, const float* residual,
                       const float* film, int film_ld, float* y, void* y_h, float* mean, float* rstd, int N, int S, int C, int G, float eps,
                       int act, const float* slabs, int nslab, size_t slab_stride, const float* cbias, const float* stats1,
                       const float* stats2, const GnPost& post, void* workspace, size_t workspace_bytes, hipStream_t stream);
int v2a_groupnorm_fwd_s(const float* x, const float* x2, int C1, const float* gamma, const float* beta, const float* residual,
                        const float* film, int film_ld, float* y, void* y_h, size_t yh_plane_stride, float* mean, float* rstd, int N, int S, int C,
                        int G, float eps, int act, const float* slabs, int nslab, size_t slab_stride, const float* cbias, const float* post,
                        const float* post_slabs, int post_nslab, size_t post_stride, const float* post_bias,
                        void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if ((post && post_slabs) || (post_slabs && post_nslab < 1) || (!post_slabs && post_nslab > 0)) return V2A_ERR_ARG;
    const GnPost po = {post, post_slabs, post_slabs ? post_bias : nullptr, post_slabs ? post_nslab : 0, post_stride, y_h ? yh_plane_stride : 0};
    return gn_fwd_impl(x, x2, C1, gamma, beta, residual, film, film_ld, y, y_h, mean, rstd, N, S, C, G, eps, act, slabs, nslab, slab_stride,
                       cbias, nullptr, nullptr, po, workspace, workspace_bytes, stream);
}
// v2a_groupnorm_fwd with the statistics pass replaced by the producing convs' epilogue sums: stats1 [N * S/64][2][C1] (x) and, for a
// virtual concat, stats2 [N * S/64][2][C - C1] (x2), as v2a_conv2d_fwd_dma_f32(..., stats, ...) writes them.  Needs S % 64 == 0 and the
// large path (S * C/G above the one-workgroup limit); otherwise the arguments are ignored and the statistics are computed from x.
int v2a_groupnorm_fwd_st(const float* x, const float* x2, int C1, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                         int N, int S, int C, int G, float eps, int act, const float* stats1, const float* stats2, void* workspace,
                         size_t workspace_bytes, hipStream_t stream) {
    return gn_fwd_impl(x, x2, C1, gamma, beta, nullptr, nullptr, 0, y, nullptr, mean, rstd, N, S, C, G, eps, act, nullptr, 0, 0, nullptr,
                       stats1, stats2, GnPost{}, workspace, workspace_bytes, stream);
}
}  // extern "C"
// Statistics of a GroupNorm whose apply pass is folded into the consuming conv (v2a_conv2d_fwd_x3p_gn): mean / rstd [N][G] from the
// producing conv's per-64-row (sum, sum of squares) blocks `stats` [N * S/64][2][C] -- the reduce + finalise launches of
// v2a_groupnorm_fwd_st (same arithmetic, in double), without the pass over the tensor.  S % 64 == 0; workspace >= N * 64 * 2 * C * 8 B.
extern "C" int v2a_groupnorm_stats_f32(const float* stats, float* mean, float* rstd, int N, int S, int C, int G, float eps, void* workspace,
                                       size_t workspace_bytes, hipStream_t stream) {
    if (!stats || !mean || !rstd || N <= 0 || G <= 0 || C % G || S % 64) return V2A_ERR_ARG;
    GnDesc p = {};
    p.N = N; p.S = S; p.C = C; p.G = G; p.eps = eps; p.mean = mean; p.rstd = rstd; p.C1 = C;
    const int nb = S >> 6;
    p.nchunk = nb < 64 ? nb : 64;
    p.rows_per_chunk = (nb + p.nchunk - 1) / p.nchunk;
    if ((size_t)N * p.nchunk * 2 * C * sizeof(double) > workspace_bytes) return V2A_ERR_WORKSPACE;
    p.partial = (double*)workspace;
    p.st1 = stats; p.st2 = nullptr;
    hipLaunchKernelGGL(gn_reduce_blocks_f32, dim3(p.nchunk, N), dim3(256), 0, stream, p);
    V2A_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_finalize<0>, dim3(N * G), dim3(256), 0, stream, p);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
extern "C" int v2a_groupnorm_takes_post(int S, int C, int G) {
    if (G <= 0 || C % G != 0) return 0;
    GnDesc t = {};
    t.S = S; t.C = C;
    return gn_wavev_passes(t, C / G) > 0 ? 1 : 0;
}
extern int g_v2a_policy_f16;
static int gn_fwd_impl(const float* x, const float* x2, int C1, const float* gamma, const float* beta, const float* residual,
                       const float* film, int film_ld, float* y, void* y_h, float* mean, float* rstd, int N, int S, int C, int G, float eps,
                       int act, const float* slabs, int nslab, size_t slab_stride, const float* cbias, const float* stats1,
                       const float* stats2, const GnPost& post, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || C % G != 0) return V2A_ERR_ARG;
    if (nslab > 0 && (!slabs || x2 || !v2a_groupnorm_takes_slabs(S, C, G))) return V2A_ERR_ARG;
    if (x2 && (C1 <= 0 || C1 >= C || C1 % 4 != 0 || (long)S * (C / G) <= GN_SMALL_MAX)) return V2A_ERR_ARG;
    GnDesc p = {};
    p.yh_f16 = g_v2a_policy_f16;
    p.post = post.dense; p.post_slabs = post.slabs; p.post_bias = post.bias; p.post_nslab = post.nslab; p.post_stride = post.stride;
    const bool has_post = post.dense || post.slabs;
    p.yh3 = post.yh3;
    if (p.yh3 && ((p.yh3 & 3) || p.yh3 < (size_t)N * S * C)) return V2A_ERR_ARG;
    p.x2 = x2; p.C1 = x2 ? C1 : C;
    p.x = x; p.gamma = gamma; p.beta = beta; p.residual = residual; p.film = film; p.y = y; p.mean = mean; p.rstd = rstd;
    p.yh = (unsigned short*)y_h;
    p.film_ld = film_ld > 0 ? film_ld : 2 * C;
    p.N = N; p.S = S; p.C = C; p.G = G; p.act = act; p.eps = eps;
    const int cg = C / G;
    const long E = (long)S * cg;
    if (nslab > 0) { p.slabs = slabs; p.nslab = nslab; p.slab_stride = slab_stride; p.cbias = cbias; p.sout = (float*)x; }
    if ((has_post || p.yh3) && (x2 || !gn_wave_ok(S, cg) || gn_wavev_passes(p, cg) == 0)) return V2A_ERR_ARG;      // only the float4 wave kernels add it / write planes
    if (!x2 && gn_wave_ok(S, cg)) {
        const dim3 grid(N * G), block(64);
        const bool small = E <= 512;             // 8 values per lane (the ConditionalUnet1D slabs) or 16
        if (const int nj = gn_wavev_passes(p, cg)) {
            V2A_GNWV(gn_wavev_fwd, cg, nj);
            V2A_CHECK_LAUNCH();
            return V2A_OK;
        }
#define V2A_GNW_F(CGV) do { if (small) hipLaunchKernelGGL((gn_wave_fwd<CGV, 8>), grid, block, 0, stream, p); \
                            else hipLaunchKernelGGL((gn_wave_fwd<CGV, 16>), grid, block, 0, stream, p); } while (0)
        if (cg == 16) V2A_GNW_F(16);
        else if (cg == 32) V2A_GNW_F(32);
        else if (cg == 64) V2A_GNW_F(64);
        else V2A_GNW_F(128);
#undef V2A_GNW_F
        V2A_CHECK_LAUNCH();
        return V2A_OK;
    }
    if (E <= GN_SMALL_MAX) {
        size_t lds = (E + 32) * sizeof(float);
        const bool vec = cg % 4 == 0 && C % 4 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0;
        // 1024 threads per slab from 8 K elements up: with one 256-thread workgroup per (sample, group) a 64-KB slab was fetched in four
        // dependent rounds of loads by four waves per CU (19 us for the first ResNet stage); sixteen waves fetch it in one
        const bool wide = vec && E >= 8192;
        if (lds > 64 * 1024) {
            (void)hipFuncSetAttribute((const void*)gn_small_fwd<true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void*)gn_small_fwd<true, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void*)gn_small_fwd<false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        if (wide) hipLaunchKernelGGL((gn_small_fwd<true, 1024>), dim3(N * G), dim3(1024), lds, stream, p);
        else if (vec) hipLaunchKernelGGL((gn_small_fwd<true, 256>), dim3(N * G), dim3(256), lds, stream, p);
        else hipLaunchKernelGGL((gn_small_fwd<false, 256>), dim3(N * G), dim3(256), lds, stream, p);
        V2A_CHECK_LAUNCH();
        return V2A_OK;
    }
    if (C % 4 != 0) return V2A_ERR_ARG;
    gn_chunks(N, S, C, &p.nchunk, &p.rows_per_chunk);
    if ((size_t)N * p.nchunk * 2 * C * sizeof(double) > workspace_bytes) return V2A_ERR_WORKSPACE;
    p.partial = (double*)workspace;
    if (stats1 && S % 64 == 0 && (!x2 || stats2)) {
        const int nb = S >> 6;
        int nch = nb < 64 ? nb : 64;                 // the workspace was sized for gn_chunks' count, which is never below this
        if (nch > p.nchunk) nch = p.nchunk;
        p.nchunk = nch;
        p.rows_per_chunk = (nb + nch - 1) / nch;     // 64-row blocks per chunk
        p.st1 = stats1; p.st2 = x2 ? stats2 : nullptr;
        hipLaunchKernelGGL(gn_reduce_blocks_f32, dim3(p.nchunk, N), dim3(256), 0, stream, p);
    } else {
        hipLaunchKernelGGL(gn_colreduce<0>, dim3(p.nchunk, N), dim3(256), gn_colreduce_lds(C), stream, p);
    }
    V2A_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_finalize<0>, dim3(N * G), dim3(256), 0, stream, p);
    V2A_CHECK_LAUNCH();
    const bool aligned = (((uintptr_t)x | (uintptr_t)x2 | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0;
    if (cg % 4 == 0 && aligned && (!x2 || C1 % 4 == 0) && (double)S * (C / 4) < 2147483648.0) {
        const uint32_t d = (uint32_t)(C / 4);            // round-up reciprocal of L4: exact quotient for every idx < 2^31
        uint32_t sft = 0;
        while ((1u << sft) < d) ++sft;
        p.l4_magic = (uint32_t)(((1ull << (31 + sft)) + d - 1) / d);
        p.l4_shift = sft - 1;
        if (d == 1) { p.l4_magic = 0; p.l4_shift = 0; }
        const size_t vecs = (size_t)S * (C / 4);
        int g = (int)((vecs + 1023) / 1024);
        const int cap = (2048 + N - 1) / N;               // ~2048 workgroups over the whole tensor
        if (g > cap) g = cap;
        if (g < 1) g = 1;
        hipLaunchKernelGGL(gn_apply_fwd_rows, dim3(g, N), dim3(256), 0, stream, p);
        V2A_CHECK_LAUNCH();
        return V2A_OK;
    }
    size_t total4 = (size_t)N * S * (C / 4);
    int grid = (int)((total4 + 255) / 256);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(gn_apply_fwd, dim3(grid), dim3(256), 0, stream, p);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
extern "C" {

// Backward of v2a_groupnorm_fwd.  dx [N,S,C]; dres (optional) = gradient of the residual input;
// dfilm (optional) [N][2][C]; colsum [N][2][C] output (per-sample sums of dz and dz*xhat); dgamma/dbeta [C]: overwritten (added to
// when accumulate_params = 1) by a reduction launch over n, or null when the caller batches that reduction for many layers
// (v2a_gn_param_grads_multi).  No atomics anywhere: results are bitwise reproducible.
int v2a_groupnorm_bwd_t(const float* x, const float* gamma, const float* beta, const float* residual, const float* film, int film_ld,
                        const float* dout, const float* mean, const float* rstd, float* dx, void* dx_h, float* dres, float* dfilm,
                        float* colsum, float* dgamma, float* dbeta, int accumulate_params, int N, int S, int C, int G, int act,
                        void* workspace, size_t workspace_bytes, hipStream_t stream);
int v2a_groupnorm_bwd(const float* x, const float* gamma, const float* beta, const float* residual, const float* film,
                      int film_ld, const float* dout, const float* mean, const float* rstd, float* dx, float* dres, float* dfilm,
                      float* colsum, float* dgamma, float* dbeta, int accumulate_params, int N, int S, int C, int G, int act,
                      void* workspace, size_t workspace_bytes, hipStream_t stream) {
    return v2a_groupnorm_bwd_t(x, gamma, beta, residual, film, film_ld, dout, mean, rstd, dx, nullptr, dres, dfilm, colsum, dgamma, dbeta,
                               accumulate_params, N, S, C, G, act, workspace, workspace_bytes, stream);
}
// same, additionally writing the bf16 twin of dx (dx_h, may be null)
int v2a_groupnorm_bwd_t(const float* x, const float* gamma, const float* beta, const float* residual, const float* film, int film_ld,
                        const float* dout, const float* mean, const float* rstd, float* dx, void* dx_h, float* dres, float* dfilm,
                        float* colsum, float* dgamma, float* dbeta, int accumulate_params, int N, int S, int C, int G, int act,
                        void* workspace, size_t workspace_bytes, hipStream_t stream) {
    return v2a_groupnorm_bwd_s(x, gamma, beta, residual, film, film_ld, dout, mean, rstd, dx, dx_h, 0, dres, dfilm, colsum, dgamma, dbeta,
                               accumulate_params, N, S, C, G, act, nullptr, 0, 0, nullptr, nullptr, workspace, workspace_bytes, stream);
}
// same; with nslab > 0 the incoming gradient is sum_s slabs[s][.] + sresid[.] (split-K partial sums and epilogue residual of the data-
// gradient conv that produces it) and `dout_sum` (optional) receives that sum for its other readers; `dout` is then ignored.
int v2a_groupnorm_bwd_s(const float* x, const float* gamma, const float* beta, const float* residual, const float* film, int film_ld,
                        const float* dout, const float* mean, const float* rstd, float* dx, void* dx_h, size_t yh_plane_stride, float* dres,
                        float* dfilm, float* colsum, float* dgamma, float* dbeta, int accumulate_params, int N, int S, int C, int G, int act,
                        const float* slabs, int nslab, size_t slab_stride, const float* sresid, float* dout_sum,
                        void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!x || !gamma || !beta || (!dout && nslab <= 0) || !mean || !rstd || !dx || !colsum || C % G != 0) return V2A_ERR_ARG;
    if (nslab > 0 && (!slabs || !v2a_groupnorm_takes_slabs(S, C, G))) return V2A_ERR_ARG;
    GnDesc p = {};
    p.yh_f16 = g_v2a_policy_f16;
    p.x = x; p.gamma = gamma; p.beta = beta; p.residual = residual; p.film = film; p.dout = dout;
    p.film_ld = film_ld > 0 ? film_ld : 2 * C;
    p.mean = (float*)mean; p.rstd = (float*)rstd; p.y = dx; p.yh = (unsigned short*)dx_h; p.dres = dres; p.dfilm = dfilm; p.colsum = colsum;
    p.yh3 = dx_h ? yh_plane_stride : 0;
    if (p.yh3 && ((p.yh3 & 3) || p.yh3 < (size_t)N * S * C)) return V2A_ERR_ARG;
    p.N = N; p.S = S; p.C = C; p.G = G; p.act = act;
    const int cg = C / G;
    const long E = (long)S * cg;
    if (nslab > 0) { p.slabs = slabs; p.nslab = nslab; p.slab_stride = slab_stride; p.sresid = sresid; p.sout = dout_sum; }
    if (p.yh3 && !(gn_wave_ok(S, cg) && gn_wavev_passes(p, cg))) return V2A_ERR_ARG;       // planes: the float4 wave kernels only
    if (gn_wave_ok(S, cg)) {
        const dim3 grid(N * G), block(64);
        const bool small = E <= 512;
        if (const int nj = gn_wavev_passes(p, cg)) {
            V2A_GNWV(gn_wavev_bwd, cg, nj);
        } else
#define V2A_GNW_B(CGV) do { if (small) hipLaunchKernelGGL((gn_wave_bwd<CGV, 8>), grid, block, 0, stream, p); \
                            else hipLaunchKernelGGL((gn_wave_bwd<CGV, 16>), grid, block, 0, stream, p); } while (0)
        if (cg == 16) V2A_GNW_B(16);
        else if (cg == 32) V2A_GNW_B(32);
        else if (cg == 64) V2A_GNW_B(64);
        else V2A_GNW_B(128);
#undef V2A_GNW_B
        V2A_CHECK_LAUNCH();
    } else if (E <= GN_SMALL_MAX) {
        const bool vec = cg % 4 == 0 && C % 4 == 0 &&
                         (((uintptr_t)x | (uintptr_t)dx | (uintptr_t)dout | (uintptr_t)residual | (uintptr_t)dres | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0;
        bool wide = vec && E >= 8192;
        int nt = wide ? 1024 : 256;
        int nsl = cg >= nt ? 1 : nt / cg;
        size_t lds = ((film ? 4 : 2) * E + (size_t)nsl * 4 * cg + 32) * sizeof(float);
        if (lds > 160 * 1024 && wide) {                 // the wider column-sum scratch does not fit: fall back to 256 threads
            wide = false; nt = 256;
            nsl = cg >= nt ? 1 : nt / cg;
            lds = ((film ? 4 : 2) * E + (size_t)nsl * 4 * cg + 32) * sizeof(float);
        }
        if (lds > 160 * 1024) return V2A_ERR_ARG;
        if (lds > 64 * 1024) {
            (void)hipFuncSetAttribute((const void*)gn_small_bwd<true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void*)gn_small_bwd<true, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void*)gn_small_bwd<false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        if (wide) hipLaunchKernelGGL((gn_small_bwd<true, 1024>), dim3(N * G), dim3(1024), lds, stream, p);
        else if (vec) hipLaunchKernelGGL((gn_small_bwd<true, 256>), dim3(N * G), dim3(256), lds, stream, p);
        else hipLaunchKernelGGL((gn_small_bwd<false, 256>), dim3(N * G), dim3(256), lds, stream, p);
        V2A_CHECK_LAUNCH();
    } else {
        if (film || dfilm) return V2A_ERR_ARG;   // FiLM only occurs on the small (Conv1d) path
        if (C % 4 != 0) return V2A_ERR_ARG;
        gn_chunks(N, S, C, &p.nchunk, &p.rows_per_chunk);
        if ((size_t)N * p.nchunk * 2 * C * sizeof(double) + (size_t)N * G * 2 * sizeof(float) > workspace_bytes) return V2A_ERR_WORKSPACE;
        p.partial = (double*)workspace;
        p.gsum = (float*)((char*)workspace + (size_t)N * p.nchunk * 2 * C * sizeof(double));
        hipLaunchKernelGGL(gn_colreduce<1>, dim3(p.nchunk, N), dim3(256), gn_colreduce_lds(C), stream, p);
        V2A_CHECK_LAUNCH();
        hipLaunchKernelGGL(gn_finalize<1>, dim3(N * G), dim3(256), 0, stream, p);
        V2A_CHECK_LAUNCH();
        size_t total4 = (size_t)N * S * (C / 4);
        int grid = (int)((total4 + 255) / 256);
        if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL(gn_apply_bwd, dim3(grid), dim3(256), 0, stream, p);
        V2A_CHECK_LAUNCH();
    }
    if (dgamma && dbeta) {        // null: the caller sums colsum over n later (v2a_gn_param_grads_multi, one launch for many layers)
        hipLaunchKernelGGL(gn_param_grads, dim3((C + 63) / 64), dim3(256), 0, stream, colsum, dgamma, dbeta, N, C, accumulate_params);
        V2A_CHECK_LAUNCH();
    }
    return V2A_OK;
}

// dgamma / dbeta of many GroupNorm layers in one launch (fixed summation order over n, fp64): table [nrows][5] int64 rows
// {colsum ptr ([N][2][C], as written by v2a_groupnorm_bwd), dgamma ptr, dbeta ptr, N, C}; work [nwork][2] int32 = (row, 64-channel block).
int v2a_gn_param_grads_multi(const void* table, const void* work, int nwork, hipStream_t stream) {
    if (!table || !work || nwork < 0) return V2A_ERR_ARG;
    if (nwork == 0) return V2A_OK;
    hipLaunchKernelGGL(gn_param_grads_multi, dim3(nwork), dim3(256), 0, stream, (const long long*)table, (const int*)work);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

}  // extern "C"
