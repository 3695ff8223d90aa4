// Shared helpers for the gfx950 (MI355X / CDNA4) kernels of libv2a_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define V2A_OK 0
#define V2A_ERR_ARG -1
#define V2A_ERR_LAUNCH -2
#define V2A_ERR_WORKSPACE -3

#define V2A_CHECK_LAUNCH()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return V2A_ERR_LAUNCH;       \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// activation ids shared by norm/elementwise kernels and the C ABI
enum { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2, ACT_MISH = 3, ACT_GELU = 4 };

// fp32 -> bf16, round to nearest even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, one instruction where the integer
// add-and-shift idiom takes four); finite values round identically, NaNs stay NaNs.
__device__ __forceinline__ unsigned short v2a_f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
typedef __attribute__((ext_vector_type(2))) float v2a_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 v2a_bf16x2;
__device__ __forceinline__ unsigned int v2a_pack_bf16x2(float lo, float hi) {
    v2a_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, v2a_bf16x2));
}
// fp32 -> three bf16 planes (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid); both differences are exact in fp32, the planes carry 24
// significant bits): the operand format of the three-plane fp32 convs (csrc/igemm_h.hip conv_igemm_f32x3 / conv_halo_x3 / conv_p3)
__device__ __forceinline__ void v2a_split3x2(float x0, float x1, unsigned int& h, unsigned int& m, unsigned int& l) {
    h = v2a_pack_bf16x2(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = v2a_pack_bf16x2(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = v2a_pack_bf16x2(s0, s1);
}
__device__ __forceinline__ void v2a_split3x1(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    h = v2a_f2bf(x);
    const float r = x - __uint_as_float((unsigned int)h << 16);
    m = v2a_f2bf(r);
    l = v2a_f2bf(r - __uint_as_float((unsigned int)m << 16));
}
// ---- the two 16-bit storage formats of the video-storage kernels: bf16 (default) and IEEE fp16 (the reference's GPU path is fp16
// autocast: lb_online_trainer_v7.py:72-76,889).  Kernels carry the format as a template flag F16; tensors are uint16_t either way.
// Conversions round to nearest even (v_cvt_pk_bf16_f32 / v_cvt_f16_f32); the 32x32x16 MFMA exists for both at the same rate.
typedef __attribute__((ext_vector_type(2))) _Float16 v2a_f16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 v2a_f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 v2a_bf16x8;
extern thread_local int g_v2a_half_f16;   // per host thread: 0 bf16, 1 fp16 (v2a_set_half_format; the host wrappers set it from the tensor dtype)
template <bool F16> __device__ __forceinline__ unsigned short v2a_f2h(float f) {
    if constexpr (F16) return __builtin_bit_cast(unsigned short, (_Float16)f);
    else return v2a_f2bf(f);
}
template <bool F16> __device__ __forceinline__ float v2a_h2f(unsigned short h) {
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, h);
    else return __uint_as_float((unsigned int)h << 16);
}
template <bool F16> __device__ __forceinline__ unsigned int v2a_pack_h2(float lo, float hi) {
    if constexpr (F16) {
        v2a_f32x2 v = {lo, hi};
        return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, v2a_f16x2));
    } else return v2a_pack_bf16x2(lo, hi);
}
template <bool F16> __device__ __forceinline__ float v2a_lo_h2(unsigned int u) {      // element 0 of a packed pair
    if constexpr (F16) return (float)__builtin_bit_cast(v2a_f16x2, u)[0];
    else return __uint_as_float(u << 16);
}
template <bool F16> __device__ __forceinline__ float v2a_hi_h2(unsigned int u) {      // element 1
    if constexpr (F16) return (float)__builtin_bit_cast(v2a_f16x2, u)[1];
    else return __uint_as_float(u & 0xffff0000u);
}
template <bool F16> __device__ __forceinline__ void v2a_unpack_h8(const uint4 u, float* f) {
    f[0] = v2a_lo_h2<F16>(u.x); f[1] = v2a_hi_h2<F16>(u.x); f[2] = v2a_lo_h2<F16>(u.y); f[3] = v2a_hi_h2<F16>(u.y);
    f[4] = v2a_lo_h2<F16>(u.z); f[5] = v2a_hi_h2<F16>(u.z); f[6] = v2a_lo_h2<F16>(u.w); f[7] = v2a_hi_h2<F16>(u.w);
}
// D = A (32 x 16) * B (16 x 32) + C with 8 16-bit elements per lane and operand, fp32 accumulate
template <bool F16, typename V8> __device__ __forceinline__ f32x16 v2a_mfma_h(const V8 a, const V8 b, const f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v2a_f16x8, a), __builtin_bit_cast(v2a_f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v2a_bf16x8, a), __builtin_bit_cast(v2a_bf16x8, b), c, 0, 0, 0);
}

// SiLU for outputs that are rounded to bf16 anyway: exp2 + reciprocal approximations (~2 ulp of fp32) instead of expf + IEEE division
__device__ __forceinline__ float v2a_silu_fast(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}

// GroupNorm affine + activation on one packed pair of 16-bit elements, two-wide fp32 VALU (v_pk_fma / v_pk_mul / v_pk_add):
// z = x * a + b; SiLU: z * rcp(1 + exp2(-log2(e) * z)) -- the arithmetic of v2a_silu_fast.  Shared by gn_apply_h and the halo conv's
// loader, which must agree bit for bit.
template <bool F16> __device__ __forceinline__ unsigned int v2a_gn_act2(unsigned int u, v2a_f32x2 a, v2a_f32x2 b, bool silu) {
    const v2a_f32x2 x = {v2a_lo_h2<F16>(u), v2a_hi_h2<F16>(u)};
    v2a_f32x2 z = __builtin_elementwise_fma(x, a, b);
    if (silu) {
        const v2a_f32x2 t = z * -1.44269504088896340736f;
        const v2a_f32x2 d = v2a_f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + 1.0f;
        z = z * v2a_f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    }
    return v2a_pack_h2<F16>(z[0], z[1]);
}

__device__ __forceinline__ float act_fwd(float x, int act) {
    switch (act) {
        case ACT_SILU: return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));      // v_exp_f32 + v_rcp_f32, ~1 ulp each
        case ACT_RELU: return x > 0.f ? x : 0.f;
        case ACT_MISH: {
            // x * tanh(softplus(x)) with torch's softplus threshold 20.  tanh(log(1 + n)) = w / (w + 2) for n = e^x, w = n (n + 2): one
            // v_exp_f32 and one v_rcp_f32 (~1 ulp each) where expf + log1pf + tanhf cost ~150 instructions -- 40 % of a ConditionalUnet1D
            // GroupNorm launch (eight elements per lane on one wave per SIMD)
            if (x > 20.f) return x;
            const float n = __builtin_amdgcn_exp2f(x * 1.44269504088896340736f), w = n * (n + 2.f);
            return x * (w * __builtin_amdgcn_rcpf(w + 2.f));
        }
        case ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
        default: return x;
    }
}

// d act(x) / dx
__device__ __forceinline__ float act_bwd(float x, int act) {
    switch (act) {
        case ACT_SILU: {
            const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
            return s * (1.0f + x * (1.0f - s));
        }
        case ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case ACT_MISH: {
            // t + x (1 - t^2) sigmoid(x) with t = tanh(softplus(x)) = w / (w + 2): 1 - t^2 = 4 (w + 1) / (w + 2)^2 without the cancellation,
            // sigmoid = n / (1 + n)
            if (x > 20.f) return 1.f;
            const float n = __builtin_amdgcn_exp2f(x * 1.44269504088896340736f), w = n * (n + 2.f);
            const float r = __builtin_amdgcn_rcpf(w + 2.f), t = w * r;
            const float sg = n * __builtin_amdgcn_rcpf(1.f + n);
            return t + x * (4.f * (w + 1.f) * r * r) * sg;
        }
        case ACT_GELU: {   // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
            const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
            return cdf + x * 0.39894228040143267794f * expf(-0.5f * x * x);
        }
        default: return 1.f;
    }
}

// Sum / max over the 64 lanes, result in every lane (all lanes must be active).  Butterflies inside the 16-lane rows are DPP permutes, the
// four row totals meet through v_readlane: 47 ns per dependent sum against 180 ns for six ds_bpermute shuffles (tools/probes/r5/
// wave_sum_bench.hip) -- a GroupNorm launch chains two to four of these.
template <int CTRL> __device__ __forceinline__ float v2a_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += v2a_dpp<0xB1>(v);          // quad_perm [1,0,3,2]
    v += v2a_dpp<0x4E>(v);          // quad_perm [2,3,0,1]
    v += v2a_dpp<0x141>(v);         // row_half_mirror
    v += v2a_dpp<0x140>(v);         // row_mirror
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, v2a_dpp<0xB1>(v));
    v = fmaxf(v, v2a_dpp<0x4E>(v));
    v = fmaxf(v, v2a_dpp<0x141>(v));
    v = fmaxf(v, v2a_dpp<0x140>(v));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// Stateless dropout: element `idx` of random stream `stream` under `seed` is kept with probability 1 - p.  A splitmix64 finaliser of
// (seed, stream, idx); forward and backward recompute the same decision, nothing is stored.  Returns the multiplier 0 or 1/(1-p).
__device__ __forceinline__ float dropout_scale(unsigned long long seed, unsigned long long stream, unsigned long long idx, float p, float inv_keep) {
    unsigned long long z = seed ^ (stream * 0x9E3779B97F4A7C15ull) ^ (idx + 0xD1B54A32D192ED03ull) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);          // 24 uniform bits in [0, 1)
    return u < p ? 0.0f : inv_keep;
}

