// Per-frame spatial self-attention of the video UNet ("legacy" head order), fp32, flash-style online softmax.
// replaces QKVAttentionLegacy.forward (reference guided_diffusion/unet.py:341-358):
//   qkv rows = (frame, position), columns = [head][q(ch) | k(ch) | v(ch)]  (reshape(bs*heads, 3*ch, L).split(ch))
//   w = softmax_f32((q*s)^T (k*s)), s = ch^-1/4;  a = v w^T;  output columns = [head][ch]
// One workgroup per (frame, head); K and V tiles of up to 256 keys live in LDS (2 x 256 x ch x 4 B <= 64 KB for
// ch = 32), every lane owns one query row: scores come from broadcast LDS reads, softmax state stays in registers.
// 0.25 % of the sampler FLOPs -> a VALU kernel; the heavy projections around it run on MFMA.
#include "common.h"

// T = float (fp32 storage) or uint16_t (bf16 storage: loads widen to fp32, the output is rounded to nearest even)
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const uint16_t* p) { return __uint_as_float((uint32_t)*p << 16); }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(const uint16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    f32x4 v = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
    return v;
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4(uint16_t* p, f32x4 v) {
    uint32_t u[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { u[i] = __float_as_uint(v[i]); u[i] += 0x7fffu + ((u[i] >> 16) & 1u); }
    uint2 o = {(u[0] >> 16) | (u[1] & 0xffff0000u), (u[2] >> 16) | (u[3] & 0xffff0000u)};
    *reinterpret_cast<uint2*>(p) = o;
}

template <int CH, typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out, int L, int heads) {
    constexpr int KT = 256;
    extern __shared__ __attribute__((aligned(16))) float sm[];   // K[KT][CH], V[KT][CH]
    float* Ks = sm;
    float* Vs = sm + KT * CH;
    const int n = blockIdx.x / heads, h = blockIdx.x % heads;
    const int C3 = heads * 3 * CH, C = heads * CH;
    const T* base = qkv + (size_t)n * L * C3 + (size_t)h * 3 * CH;
    const float s = 1.0f / sqrtf(sqrtf((float)CH));
    for (int q0 = 0; q0 < L; q0 += blockDim.x) {
        const int qi = q0 + threadIdx.x;
        const bool active = qi < L;
        float q[CH], o[CH];
        float m = -INFINITY, l = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) { q[c] = active ? ld1(base + (size_t)qi * C3 + c) * s : 0.f; o[c] = 0.f; }
        for (int k0 = 0; k0 < L; k0 += KT) {
            const int kn = min(KT, L - k0);
            __syncthreads();
            for (int i = threadIdx.x; i < kn * (CH / 4); i += blockDim.x) {
                const int r = i / (CH / 4), c4 = i % (CH / 4);
                const T* src = base + (size_t)(k0 + r) * C3 + CH + c4 * 4;
                f32x4 kv = ld4(src);
                f32x4 vv = ld4(src + CH);
                kv *= s;
                *reinterpret_cast<f32x4*>(&Ks[r * CH + c4 * 4]) = kv;
                *reinterpret_cast<f32x4*>(&Vs[r * CH + c4 * 4]) = vv;
            }
            __syncthreads();
            for (int j0 = 0; j0 < kn; j0 += 8) {
                float sc[8];
                float cm = -INFINITY;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float d = -INFINITY;
                    if (j0 + j < kn) {
                        d = 0.f;
                        const float* kr = &Ks[(j0 + j) * CH];
#pragma unroll
                        for (int c = 0; c < CH; ++c) d += q[c] * kr[c];
                    }
                    sc[j] = d;
                    cm = fmaxf(cm, d);
                }
                const float mn = fmaxf(m, cm);
                const float alpha = expf(m - mn);   // exp(-inf) = 0 on the first chunk
                l *= alpha;
#pragma unroll
                for (int c = 0; c < CH; ++c) o[c] *= alpha;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j0 + j < kn) {
                        const float pj = expf(sc[j] - mn);
                        l += pj;
                        const float* vr = &Vs[(j0 + j) * CH];
#pragma unroll
                        for (int c = 0; c < CH; ++c) o[c] += pj * vr[c];
                    }
                }
                m = mn;
            }
        }
        if (active) {
            const float inv = 1.0f / l;
            T* dst = out + ((size_t)n * L + qi) * C + (size_t)h * CH;
#pragma unroll
            for (int c = 0; c < CH; c += 4) {
                f32x4 v = {o[c] * inv, o[c + 1] * inv, o[c + 2] * inv, o[c + 3] * inv};
                st4(dst + c, v);
            }
        }
    }
}

// Generic small attention for the PerceiverResampler (reference imagen.py:283-319): q,k l2-normalised per head then
// scaled elementwise by q_scale / k_scale, sim * 8, softmax, @ v.  q [B, Lq, H*D], kv [B, Lk, 2*H*D] (k | v),
// out [B, Lq, H*D].  One workgroup per (b, head), one lane per query; runs once per sample() call (t-independent).
template <int D>
__global__ __launch_bounds__(128) void perceiver_attn_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                             const float* __restrict__ q_scale, const float* __restrict__ k_scale,
                                                             float* __restrict__ out, int Lq, int Lk, int H, float sim_scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // K[Lk][D], V[Lk][D]
    float* Ks = sm;
    float* Vs = sm + (size_t)Lk * D;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int HD = H * D;
    for (int r = threadIdx.x; r < Lk; r += blockDim.x) {
        const float* kr = kv + ((size_t)b * Lk + r) * 2 * HD + h * D;
        float nrm = 0.f;
        for (int c = 0; c < D; ++c) nrm += kr[c] * kr[c];
        nrm = fmaxf(sqrtf(nrm), 1e-12f);   // F.normalize eps
        for (int c = 0; c < D; ++c) {
            Ks[r * D + c] = kr[c] / nrm * k_scale[c];
            Vs[r * D + c] = kr[HD + c];
        }
    }
    __syncthreads();
    for (int qi = threadIdx.x; qi < Lq; qi += blockDim.x) {
        const float* qr = q + ((size_t)b * Lq + qi) * HD + h * D;
        float qq[D], o[D];
        float nrm = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) { qq[c] = qr[c]; nrm += qq[c] * qq[c]; o[c] = 0.f; }
        nrm = fmaxf(sqrtf(nrm), 1e-12f);
#pragma unroll
        for (int c = 0; c < D; ++c) qq[c] = qq[c] / nrm * q_scale[c];
        float m = -INFINITY;
        for (int j = 0; j < Lk; ++j) {
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) d += qq[c] * Ks[j * D + c];
            m = fmaxf(m, d * sim_scale);
        }
        float l = 0.f;
        for (int j = 0; j < Lk; ++j) {
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) d += qq[c] * Ks[j * D + c];
            const float pj = expf(d * sim_scale - m);
            l += pj;
#pragma unroll
            for (int c = 0; c < D; ++c) o[c] += pj * Vs[j * D + c];
        }
        float* dst = out + ((size_t)b * Lq + qi) * HD + h * D;
#pragma unroll
        for (int c = 0; c < D; ++c) dst[c] = o[c] / l;
    }
}

// LayerNorm over the last dim (biased variance, eps inside rsqrt): y = (x-mean)*rsqrt(var+eps)*g (+ b)
// covers nn.LayerNorm (with bias) and imagen's LayerNorm (gain only) -- reference imagen.py:198-211.
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, const float* g, const float* b, float* y, int rows, int D, float eps) {
    __shared__ float red[8];
    const int r = blockIdx.x;
    const float* xr = x + (size_t)r * D;
    float s = 0.f;
    for (int i = threadIdx.x; i < D; i += 256) s += xr[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float mu = (red[0] + red[1] + red[2] + red[3]) / (float)D;
    float q = 0.f;
    for (int i = threadIdx.x; i < D; i += 256) { const float d = xr[i] - mu; q += d * d; }
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = q;
    __syncthreads();
    const float var = (red[4] + red[5] + red[6] + red[7]) / (float)D;
    const float rs = 1.0f / sqrtf(var + eps);
    for (int i = threadIdx.x; i < D; i += 256) {
        float v = (xr[i] - mu) * rs * g[i];
        if (b) v += b[i];
        y[(size_t)r * D + i] = v;
    }
}

// out[b][d] = mean_r x[b][r][d]
__global__ void mean_rows_kernel(const float* x, float* out, int B, int R, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    const int b = i / D, d = i % D;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += x[((size_t)b * R + r) * D + d];
    out[i] = s / (float)R;
}

extern "C" {

int v2a_attention_fwd(const float* qkv, float* out, int n_frames, int L, int heads, int head_ch, hipStream_t s) {
    if (!qkv || !out) return V2A_ERR_ARG;
    const int threads = L >= 256 ? 256 : ((L + 63) / 64) * 64;
    const size_t lds = (size_t)2 * 256 * head_ch * sizeof(float);
    dim3 grid(n_frames * heads);
    if (lds > 64 * 1024) {
        if (head_ch == 64) (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<64, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    switch (head_ch) {
        case 16: hipLaunchKernelGGL((attn_fwd_kernel<16, float>), grid, dim3(threads), lds, s, qkv, out, L, heads); break;
        case 32: hipLaunchKernelGGL((attn_fwd_kernel<32, float>), grid, dim3(threads), lds, s, qkv, out, L, heads); break;
        case 64: hipLaunchKernelGGL((attn_fwd_kernel<64, float>), grid, dim3(threads), lds, s, qkv, out, L, heads); break;
        default: return V2A_ERR_ARG;
    }
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

// bf16-storage variant: qkv / out are bf16 ([n_frames*L][3*C] -> [n_frames*L][C]); softmax state and accumulation stay fp32
int v2a_attention_fwd_h(const void* qkv, void* out, int n_frames, int L, int heads, int head_ch, hipStream_t s) {
    if (!qkv || !out) return V2A_ERR_ARG;
    const int threads = L >= 256 ? 256 : ((L + 63) / 64) * 64;
    const size_t lds = (size_t)2 * 256 * head_ch * sizeof(float);
    dim3 grid(n_frames * heads);
    const uint16_t* q = (const uint16_t*)qkv;
    uint16_t* o = (uint16_t*)out;
    if (lds > 64 * 1024) {
        if (head_ch == 64) (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<64, uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    switch (head_ch) {
        case 16: hipLaunchKernelGGL((attn_fwd_kernel<16, uint16_t>), grid, dim3(threads), lds, s, q, o, L, heads); break;
        case 32: hipLaunchKernelGGL((attn_fwd_kernel<32, uint16_t>), grid, dim3(threads), lds, s, q, o, L, heads); break;
        case 64: hipLaunchKernelGGL((attn_fwd_kernel<64, uint16_t>), grid, dim3(threads), lds, s, q, o, L, heads); break;
        default: return V2A_ERR_ARG;
    }
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

int v2a_perceiver_attention(const float* q, const float* kv, const float* q_scale, const float* k_scale, float* out, int B, int Lq,
                            int Lk, int H, int D, float sim_scale, hipStream_t s) {
    if (D != 64) return V2A_ERR_ARG;
    const size_t lds = (size_t)2 * Lk * D * sizeof(float);
    if (lds > 160 * 1024) return V2A_ERR_ARG;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)perceiver_attn_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((perceiver_attn_kernel<64>), dim3(B * H), dim3(128), lds, s, q, kv, q_scale, k_scale, out, Lq, Lk, H, sim_scale);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

int v2a_layernorm(const float* x, const float* g, const float* b, float* y, int rows, int D, float eps, hipStream_t s) {
    hipLaunchKernelGGL(layernorm_kernel, dim3(rows), dim3(256), 0, s, x, g, b, y, rows, D, eps);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

int v2a_mean_rows(const float* x, float* out, int B, int R, int D, hipStream_t s) {
    hipLaunchKernelGGL(mean_rows_kernel, dim3((B * D + 255) / 256), dim3(256), 0, s, x, out, B, R, D);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

}  // extern "C"
