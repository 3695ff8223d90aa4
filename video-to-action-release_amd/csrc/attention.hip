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
    uint2 o = {v2a_pack_bf16x2(v[0], v[1]), v2a_pack_bf16x2(v[2], v[3])};
    *reinterpret_cast<uint2*>(p) = o;
}
struct f16a { uint16_t v; };       // storage tag of the fp16 instances
__device__ __forceinline__ float ld1(const f16a* p) { return v2a_h2f<true>(p->v); }
__device__ __forceinline__ f32x4 ld4(const f16a* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    f32x4 v = {v2a_lo_h2<true>(u.x), v2a_hi_h2<true>(u.x), v2a_lo_h2<true>(u.y), v2a_hi_h2<true>(u.y)};
    return v;
}
__device__ __forceinline__ void st4(f16a* p, f32x4 v) {
    uint2 o = {v2a_pack_h2<true>(v[0], v[1]), v2a_pack_h2<true>(v[2], v[3])};
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

// Backward of QKVAttentionLegacy (video-model training, reference unet.py:341-358 under autograd).  One workgroup per (frame, head);
// Q, K, V and dO of the head live in LDS (fp32).  Phase A: thread = query row i recomputes its softmax statistics (m_i, l_i),
// D_i = dO_i . O_i, and dQ_i = s^2 sum_j dS_ij K_j with dS_ij = P_ij (dO_i . V_j - D_i).  Phase B: thread = key row j accumulates
// dV_j = sum_i P_ij dO_i and dK_j = s^2 sum_i dS_ij Q_i from the LDS-resident rows -- deterministic, no atomics.
template <int CH>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ o, const float* __restrict__ dout,
                                                       float* __restrict__ dqkv, int L, int heads) {
    extern __shared__ __attribute__((aligned(16))) float smb[];
    float* Qs = smb;                 // [L][CH]
    float* Ks = Qs + (size_t)L * CH;
    float* Vs = Ks + (size_t)L * CH;
    float* Gs = Vs + (size_t)L * CH;  // dO
    float* Ms = Gs + (size_t)L * CH;  // [L] row max
    float* Ls = Ms + L;               // [L] row sum
    float* Ds = Ls + L;               // [L] dO_i . O_i
    const int n = blockIdx.x / heads, h = blockIdx.x % heads;
    const int C3 = heads * 3 * CH, C = heads * CH;
    const float* base = qkv + (size_t)n * L * C3 + (size_t)h * 3 * CH;
    float* dbase = dqkv + (size_t)n * L * C3 + (size_t)h * 3 * CH;
    const float* ob = o + (size_t)n * L * C + (size_t)h * CH;
    const float* gb = dout + (size_t)n * L * C + (size_t)h * CH;
    const float s2 = 1.0f / sqrtf((float)CH);
    for (int i = threadIdx.x; i < L * (CH / 4); i += blockDim.x) {
        const int r = i / (CH / 4), c4 = (i % (CH / 4)) * 4;
        const float* src = base + (size_t)r * C3 + c4;
        *reinterpret_cast<f32x4*>(&Qs[r * CH + c4]) = *reinterpret_cast<const f32x4*>(src);
        *reinterpret_cast<f32x4*>(&Ks[r * CH + c4]) = *reinterpret_cast<const f32x4*>(src + CH);
        *reinterpret_cast<f32x4*>(&Vs[r * CH + c4]) = *reinterpret_cast<const f32x4*>(src + 2 * CH);
        *reinterpret_cast<f32x4*>(&Gs[r * CH + c4]) = *reinterpret_cast<const f32x4*>(gb + (size_t)r * C + c4);
    }
    __syncthreads();
    // ---- phase A: per query row
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        float q[CH], g[CH], dq[CH];
        float D = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            q[c] = Qs[i * CH + c];
            g[c] = Gs[i * CH + c];
            D += g[c] * ob[(size_t)i * C + c];
            dq[c] = 0.f;
        }
        float m = -INFINITY;
        for (int j = 0; j < L; ++j) {
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) d += q[c] * Ks[j * CH + c];
            m = fmaxf(m, d * s2);
        }
        float l = 0.f;
        for (int j = 0; j < L; ++j) {
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) d += q[c] * Ks[j * CH + c];
            l += expf(d * s2 - m);
        }
        const float inv = 1.0f / l;
        for (int j = 0; j < L; ++j) {
            float d = 0.f, dp = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                d += q[c] * Ks[j * CH + c];
                dp += g[c] * Vs[j * CH + c];
            }
            const float pij = expf(d * s2 - m) * inv;
            const float ds = pij * (dp - D) * s2;
#pragma unroll
            for (int c = 0; c < CH; ++c) dq[c] += ds * Ks[j * CH + c];
        }
        Ms[i] = m;
        Ls[i] = inv;
        Ds[i] = D;
#pragma unroll
        for (int c = 0; c < CH; c += 4) {
            f32x4 v = {dq[c], dq[c + 1], dq[c + 2], dq[c + 3]};
            *reinterpret_cast<f32x4*>(dbase + (size_t)i * C3 + c) = v;
        }
    }
    __syncthreads();
    // ---- phase B: per key row
    for (int j = threadIdx.x; j < L; j += blockDim.x) {
        float k[CH], v[CH], dk[CH], dv[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            k[c] = Ks[j * CH + c];
            v[c] = Vs[j * CH + c];
            dk[c] = 0.f;
            dv[c] = 0.f;
        }
        for (int i = 0; i < L; ++i) {
            float d = 0.f, dp = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                d += Qs[i * CH + c] * k[c];
                dp += Gs[i * CH + c] * v[c];
            }
            const float pij = expf(d * s2 - Ms[i]) * Ls[i];
            const float ds = pij * (dp - Ds[i]) * s2;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                dv[c] += pij * Gs[i * CH + c];
                dk[c] += ds * Qs[i * CH + c];
            }
        }
#pragma unroll
        for (int c = 0; c < CH; c += 4) {
            f32x4 a = {dk[c], dk[c + 1], dk[c + 2], dk[c + 3]}, b = {dv[c], dv[c + 1], dv[c + 2], dv[c + 3]};
            *reinterpret_cast<f32x4*>(dbase + (size_t)j * C3 + CH + c) = a;
            *reinterpret_cast<f32x4*>(dbase + (size_t)j * C3 + 2 * CH + c) = b;
        }
    }
}

// MFMA attention for the bf16-storage configuration (head_ch = 32).  One workgroup per (frame, head); K and V^T of the head live in
// LDS, every wave owns 64 queries.  The scores are computed SWAPPED, S^T = K Q^T (32x32x16 bf16 MFMA, keys = rows, queries =
// columns): the accumulator then holds, per lane, 16 keys of ONE query -- exactly the k-slot order in which the same registers,
// rounded to bf16, are the B operand of O^T += V^T P^T.  No LDS round trip for P; the A operand V^T is read as two 8-B pieces per
// k-step ({k0..k0+3}, {k0+8..k0+11}: the accumulator's row order).  Online softmax per query: the two lanes that share a query
// (lane, lane ^ 32) exchange their partial max / sum with one shuffle per key tile.  Bank-conflict-free layouts: K rows padded to
// 80 B (ds_read_b128), V^T rows to 2L + 8 B (ds_read_b64).
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_a;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_a;


template <bool F16>
__global__ __launch_bounds__(256) void attn_mfma_h_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out, int L, int heads) {
    constexpr int CH = 32, LDK = 40;                          // K row stride in bf16 (80 B)
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    uint16_t* Ks = reinterpret_cast<uint16_t*>(sm_raw);        // [L][LDK]
    const int LDV = L + 4;                                     // V^T row stride in bf16 (2L + 8 B)
    uint16_t* Vt = Ks + (size_t)L * LDK;                       // [CH][LDV]
    const int n = blockIdx.x / heads, h = blockIdx.x % heads;
    const int C3 = heads * 3 * CH, C = heads * CH;
    const uint16_t* base = qkv + (size_t)n * L * C3 + (size_t)h * 3 * CH;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lr = lane & 31, lk = lane >> 5;

    // ---- stage K (row-major, padded) and V^T
    for (int i = tid; i < L * 4; i += blockDim.x) {           // 4 x 16-B pieces per key row for K and for V
        const int r = i >> 2, c = i & 3;
        const uint16_t* src = base + (size_t)r * C3 + CH + c * 8;
        const uint4 kv = *reinterpret_cast<const uint4*>(src);
        const uint4 vv = *reinterpret_cast<const uint4*>(src + CH);
        *reinterpret_cast<uint4*>(Ks + r * LDK + c * 8) = kv;
        const uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Vt[(c * 8 + 2 * e) * LDV + r] = (uint16_t)(w[e] & 0xffffu);
            Vt[(c * 8 + 2 * e + 1) * LDV + r] = (uint16_t)(w[e] >> 16);
        }
    }
    __syncthreads();

    const float s2 = 1.0f / sqrtf((float)CH);                 // (ch^-1/4)^2: q and k are both scaled by ch^-1/4 in the reference
    for (int q0 = wid * 64; q0 < L; q0 += (blockDim.x >> 6) * 64) {
        // Q fragments straight from HBM: B operand of S^T = K Q^T, lane -> (query q0 + t*32 + lr, d = kstep*16 + lk*8 .. +8)
        bf16x8_a qf[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int q = q0 + t * 32 + lr;
                uint4 u = make_uint4(0, 0, 0, 0);
                if (q < L) u = *reinterpret_cast<const uint4*>(base + (size_t)q * C3 + ks * 16 + lk * 8);
                qf[t][ks] = *reinterpret_cast<bf16x8_a*>(&u);
            }
        f32x16 o[2];
        float m[2], l[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            m[t] = -INFINITY;
            l[t] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
        }
        for (int k0 = 0; k0 < L; k0 += 32) {
            bf16x8_a kf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) kf[ks] = *reinterpret_cast<const bf16x8_a*>(Ks + (k0 + lr) * LDK + ks * 16 + lk * 8);
            // V^T fragments: A operand of O^T += V^T P^T, lane -> (d = lr, keys in the accumulator's row order)
            bf16x8_a vf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint16_t* vp = Vt + lr * LDV + k0 + ks * 16 + lk * 4;
                const uint2 a = *reinterpret_cast<const uint2*>(vp);          // keys k0 + 16ks + 4lk + {0..3}
                const uint2 b = *reinterpret_cast<const uint2*>(vp + 8);      // keys k0 + 16ks + 4lk + 8 + {0..3}
                uint4 u = make_uint4(a.x, a.y, b.x, b.y);
                vf[ks] = *reinterpret_cast<bf16x8_a*>(&u);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 sacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
                sacc = v2a_mfma_h<F16>(kf[0], qf[t][0], sacc);
                sacc = v2a_mfma_h<F16>(kf[1], qf[t][1], sacc);
                float cm = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sacc[r] *= s2;
                    cm = fmaxf(cm, sacc[r]);
                }
                cm = fmaxf(cm, __shfl_xor(cm, 32, 64));                      // the partner lane holds the other 16 keys of this query
                const float mn = fmaxf(m[t], cm);
                const float alpha = __expf(m[t] - mn);
                float ps = 0.f;
                float pv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    pv[r] = __expf(sacc[r] - mn);
                    ps += pv[r];
                }
                ps += __shfl_xor(ps, 32, 64);
                l[t] = l[t] * alpha + ps;
                m[t] = mn;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
                uint4 p0 = make_uint4(v2a_pack_h2<F16>(pv[0], pv[1]), v2a_pack_h2<F16>(pv[2], pv[3]), v2a_pack_h2<F16>(pv[4], pv[5]), v2a_pack_h2<F16>(pv[6], pv[7]));
                uint4 p1 = make_uint4(v2a_pack_h2<F16>(pv[8], pv[9]), v2a_pack_h2<F16>(pv[10], pv[11]), v2a_pack_h2<F16>(pv[12], pv[13]), v2a_pack_h2<F16>(pv[14], pv[15]));
                o[t] = v2a_mfma_h<F16>(vf[0], *reinterpret_cast<bf16x8_a*>(&p0), o[t]);
                o[t] = v2a_mfma_h<F16>(vf[1], *reinterpret_cast<bf16x8_a*>(&p1), o[t]);
            }
        }
        // O^T accumulator: lane -> query lr of tile t, channels d = (r & 3) + 8 (r >> 2) + 4 lk : four runs of 4 channels (8 B each)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q = q0 + t * 32 + lr;
            if (q >= L) continue;
            const float inv = 1.0f / l[t];
            uint16_t* dst = out + ((size_t)n * L + q) * C + (size_t)h * CH + 4 * lk;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 u = {v2a_pack_h2<F16>(o[t][4 * g] * inv, o[t][4 * g + 1] * inv), v2a_pack_h2<F16>(o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv)};
                *reinterpret_cast<uint2*>(dst + 8 * g) = u;
            }
        }
    }
}

// MFMA attention for the fp32 (parity) configuration, head_ch = 32, L <= 256 keys (the video UNet attends at 16 x 16 and 8 x 8):
// attn_mfma_h_kernel's structure with every fp32 operand as three bf16 planes (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid))
// and every product block as the six plane products of weight >= 2^-16, smallest first -- the arithmetic of the three-plane convs
// (csrc/igemm_h.hip conv_igemm_f32x3; fp32-equivalent: each bf16 x bf16 product is exact in the MFMA's fp32 accumulation, the dropped
// products are <= 2^-24 |a||b|).  K and V^T planes of the head live in LDS (split once while staged); Q is split in registers; the
// probabilities P (fp32, softmax state in fp32 as before) are split in registers on their way into O^T += V^T P^T.  Scores are scaled by
// ch^-1/2 after the product (the reference scales q and k by ch^-1/4 each, unet.py:349-353: one rounding apart).
__global__ __launch_bounds__(256) void attn_mfma_x3_kernel(const float* __restrict__ qkv, float* __restrict__ out, int L, int heads) {
    constexpr int CH = 32, LDK = 40;                          // K row stride in bf16 (80 B: conflict-free ds_read_b128)
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    const int LDV = L + 4;                                     // V^T row stride in bf16 (2L + 8 B)
    const int KP = L * LDK, VP = CH * LDV;                     // elements per plane
    uint16_t* Ks = reinterpret_cast<uint16_t*>(sm_raw);        // [3][L][LDK]
    uint16_t* Vt = Ks + 3 * (size_t)KP;                        // [3][CH][LDV]
    const int n = blockIdx.x / heads, h = blockIdx.x % heads;
    const int C3 = heads * 3 * CH, C = heads * CH;
    const float* base = qkv + (size_t)n * L * C3 + (size_t)h * 3 * CH;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lr = lane & 31, lk = lane >> 5;

    // ---- stage K (row-major, padded) and V^T, split into planes
    for (int i = tid; i < L * 4; i += blockDim.x) {           // 4 x 8-channel pieces per key row for K and for V
        const int r = i >> 2, c = i & 3;
        const float* src = base + (size_t)r * C3 + CH + c * 8;
        const f32x4 k0 = *reinterpret_cast<const f32x4*>(src), k1 = *reinterpret_cast<const f32x4*>(src + 4);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + CH), v1 = *reinterpret_cast<const f32x4*>(src + CH + 4);
        uint32_t kh[4], km[4], kl[4], vh[4], vm[4], vl[4];
        v2a_split3x2(k0[0], k0[1], kh[0], km[0], kl[0]); v2a_split3x2(k0[2], k0[3], kh[1], km[1], kl[1]);
        v2a_split3x2(k1[0], k1[1], kh[2], km[2], kl[2]); v2a_split3x2(k1[2], k1[3], kh[3], km[3], kl[3]);
        v2a_split3x2(v0[0], v0[1], vh[0], vm[0], vl[0]); v2a_split3x2(v0[2], v0[3], vh[1], vm[1], vl[1]);
        v2a_split3x2(v1[0], v1[1], vh[2], vm[2], vl[2]); v2a_split3x2(v1[2], v1[3], vh[3], vm[3], vl[3]);
        uint16_t* kd = Ks + r * LDK + c * 8;
        *reinterpret_cast<uint4*>(kd) = make_uint4(kh[0], kh[1], kh[2], kh[3]);
        *reinterpret_cast<uint4*>(kd + KP) = make_uint4(km[0], km[1], km[2], km[3]);
        *reinterpret_cast<uint4*>(kd + 2 * KP) = make_uint4(kl[0], kl[1], kl[2], kl[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint16_t* vd = Vt + (c * 8 + 2 * e) * LDV + r;
            vd[0] = (uint16_t)(vh[e] & 0xffffu); vd[LDV] = (uint16_t)(vh[e] >> 16);
            vd[VP] = (uint16_t)(vm[e] & 0xffffu); vd[VP + LDV] = (uint16_t)(vm[e] >> 16);
            vd[2 * VP] = (uint16_t)(vl[e] & 0xffffu); vd[2 * VP + LDV] = (uint16_t)(vl[e] >> 16);
        }
    }
    __syncthreads();

    const float s2 = 1.0f / sqrtf((float)CH);                 // (ch^-1/4)^2
#define V2A_X3A_SIX(ACC, A, B)                                                                  \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], B[0], ACC, 0, 0, 0);   /* lo  * hi  */ \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[2], ACC, 0, 0, 0);   /* hi  * lo  */ \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[1], ACC, 0, 0, 0);   /* mid * mid */ \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[0], ACC, 0, 0, 0);   /* mid * hi  */ \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[1], ACC, 0, 0, 0);   /* hi  * mid */ \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[0], ACC, 0, 0, 0);   /* hi  * hi  */
    for (int q0 = wid * 64; q0 < L; q0 += (blockDim.x >> 6) * 64) {
        // Q planes straight from HBM: B operand of S^T = K Q^T, lane -> (query q0 + t*32 + lr, d = kstep*16 + lk*8 .. +8)
        bf16x8_a qf[2][2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int q = q0 + t * 32 + lr;
                f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
                if (q < L) {
                    const float* qp = base + (size_t)q * C3 + ks * 16 + lk * 8;
                    a = *reinterpret_cast<const f32x4*>(qp);
                    b = *reinterpret_cast<const f32x4*>(qp + 4);
                }
                uint32_t hh[4], mm[4], ll[4];
                v2a_split3x2(a[0], a[1], hh[0], mm[0], ll[0]); v2a_split3x2(a[2], a[3], hh[1], mm[1], ll[1]);
                v2a_split3x2(b[0], b[1], hh[2], mm[2], ll[2]); v2a_split3x2(b[2], b[3], hh[3], mm[3], ll[3]);
                uint4 uh = make_uint4(hh[0], hh[1], hh[2], hh[3]), um = make_uint4(mm[0], mm[1], mm[2], mm[3]), ul = make_uint4(ll[0], ll[1], ll[2], ll[3]);
                qf[t][ks][0] = *reinterpret_cast<bf16x8_a*>(&uh);
                qf[t][ks][1] = *reinterpret_cast<bf16x8_a*>(&um);
                qf[t][ks][2] = *reinterpret_cast<bf16x8_a*>(&ul);
            }
        f32x16 o[2];
        float m[2], l[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            m[t] = -INFINITY;
            l[t] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
        }
        for (int k0 = 0; k0 < L; k0 += 32) {
            bf16x8_a kf[2][3], vf[2][3];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    kf[ks][p] = *reinterpret_cast<const bf16x8_a*>(Ks + p * KP + (k0 + lr) * LDK + ks * 16 + lk * 8);
                    // V^T: A operand of O^T += V^T P^T, lane -> (d = lr, keys in the accumulator's row order)
                    const uint16_t* vp = Vt + p * VP + lr * LDV + k0 + ks * 16 + lk * 4;
                    const uint2 a = *reinterpret_cast<const uint2*>(vp);          // keys k0 + 16ks + 4lk + {0..3}
                    const uint2 b = *reinterpret_cast<const uint2*>(vp + 8);      // keys k0 + 16ks + 4lk + 8 + {0..3}
                    uint4 u = make_uint4(a.x, a.y, b.x, b.y);
                    vf[ks][p] = *reinterpret_cast<bf16x8_a*>(&u);
                }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 sacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
                V2A_X3A_SIX(sacc, kf[0], qf[t][0])
                V2A_X3A_SIX(sacc, kf[1], qf[t][1])
                float cm = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sacc[r] *= s2;
                    cm = fmaxf(cm, sacc[r]);
                }
                cm = fmaxf(cm, __shfl_xor(cm, 32, 64));                      // the partner lane holds the other 16 keys of this query
                const float mn = fmaxf(m[t], cm);
                const float alpha = expf(m[t] - mn);
                float ps = 0.f;
                float pv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    pv[r] = expf(sacc[r] - mn);
                    ps += pv[r];
                }
                ps += __shfl_xor(ps, 32, 64);
                l[t] = l[t] * alpha + ps;
                m[t] = mn;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
                bf16x8_a pf[2][3];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    uint32_t hh[4], mm[4], ll[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v2a_split3x2(pv[ks * 8 + 2 * e], pv[ks * 8 + 2 * e + 1], hh[e], mm[e], ll[e]);
                    uint4 uh = make_uint4(hh[0], hh[1], hh[2], hh[3]), um = make_uint4(mm[0], mm[1], mm[2], mm[3]), ul = make_uint4(ll[0], ll[1], ll[2], ll[3]);
                    pf[ks][0] = *reinterpret_cast<bf16x8_a*>(&uh);
                    pf[ks][1] = *reinterpret_cast<bf16x8_a*>(&um);
                    pf[ks][2] = *reinterpret_cast<bf16x8_a*>(&ul);
                }
                V2A_X3A_SIX(o[t], vf[0], pf[0])
                V2A_X3A_SIX(o[t], vf[1], pf[1])
            }
        }
        // O^T accumulator: lane -> query lr of tile t, channels d = (r & 3) + 8 (r >> 2) + 4 lk : four runs of 4 channels (16 B each)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q = q0 + t * 32 + lr;
            if (q >= L) continue;
            const float inv = 1.0f / l[t];
            float* dst = out + ((size_t)n * L + q) * C + (size_t)h * CH + 4 * lk;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv};
                *reinterpret_cast<f32x4*>(dst + 8 * g) = v;
            }
        }
    }
#undef V2A_X3A_SIX
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

// Backward of perceiver_attn_kernel (PerceiverResampler training).  One workgroup per (b, head); K-hat, V, Q-hat, dO and the
// per-query softmax statistics live in LDS.  Phase A (thread = query): m_i, l_i, D_i = dO_i . O_i, dQhat_i = s sum_j dS_ij Khat_j ->
// dq_i through the l2-normalisation and the learned per-channel scale; phase B (thread = key): dV_j, dKhat_j -> dk_j.  The scale
// gradients are per-row contributions reduced over rows inside the workgroup and written per (b, head): dscale [B*H][2][D].
template <int D>
__global__ __launch_bounds__(128) void perceiver_attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                                 const float* __restrict__ q_scale, const float* __restrict__ k_scale,
                                                                 const float* __restrict__ out, const float* __restrict__ dout,
                                                                 float* __restrict__ dq, float* __restrict__ dkv, float* __restrict__ dscale,
                                                                 int Lq, int Lk, int H, float sim_scale) {
    extern __shared__ __attribute__((aligned(16))) float smp[];
    float* Kh = smp;                              // [Lk][D] normalised, scaled keys
    float* Vs = Kh + (size_t)Lk * D;              // [Lk][D]
    float* Qh = Vs + (size_t)Lk * D;              // [Lq][D] normalised, scaled queries
    float* Gs = Qh + (size_t)Lq * D;              // [Lq][D] dO
    float* Cs = Gs + (size_t)Lq * D;              // [max(Lq, Lk)][D] per-row scale-gradient contributions
    float* Ms = Cs + (size_t)max(Lq, Lk) * D;     // [Lq] max, [Lq] 1/sum, [Lq] D_i, then [Lk] 1/|k|
    float* Ls = Ms + Lq;
    float* Ds = Ls + Lq;
    float* Kn = Ds + Lq;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int HD = H * D;
    for (int r = threadIdx.x; r < Lk; r += blockDim.x) {
        const float* kr = kv + ((size_t)b * Lk + r) * 2 * HD + h * D;
        float nrm = 0.f;
        for (int c = 0; c < D; ++c) nrm += kr[c] * kr[c];
        nrm = fmaxf(sqrtf(nrm), 1e-12f);
        Kn[r] = 1.0f / nrm;
        for (int c = 0; c < D; ++c) {
            Kh[r * D + c] = kr[c] / nrm * k_scale[c];
            Vs[r * D + c] = kr[HD + c];
        }
    }
    __syncthreads();
    // ---- phase A
    for (int i = threadIdx.x; i < Lq; i += blockDim.x) {
        const float* qr = q + ((size_t)b * Lq + i) * HD + h * D;
        const float* gr = dout + ((size_t)b * Lq + i) * HD + h * D;
        const float* orow = out + ((size_t)b * Lq + i) * HD + h * D;
        float qn[D], g[D], dqh[D];
        float nrm = 0.f, Dd = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) { qn[c] = qr[c]; nrm += qn[c] * qn[c]; g[c] = gr[c]; Dd += g[c] * orow[c]; dqh[c] = 0.f; }
        nrm = fmaxf(sqrtf(nrm), 1e-12f);
        const float inrm = 1.0f / nrm;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            qn[c] *= inrm;                                     // unit query
            Qh[i * D + c] = qn[c] * q_scale[c];
            Gs[i * D + c] = g[c];
        }
        float m = -INFINITY;
        for (int j = 0; j < Lk; ++j) {
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) d += Qh[i * D + c] * Kh[j * D + c];
            m = fmaxf(m, d * sim_scale);
        }
        float l = 0.f;
        for (int j = 0; j < Lk; ++j) {
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) d += Qh[i * D + c] * Kh[j * D + c];
            l += expf(d * sim_scale - m);
        }
        const float il = 1.0f / l;
        for (int j = 0; j < Lk; ++j) {
            float d = 0.f, dp = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) { d += Qh[i * D + c] * Kh[j * D + c]; dp += g[c] * Vs[j * D + c]; }
            const float ds = expf(d * sim_scale - m) * il * (dp - Dd) * sim_scale;
#pragma unroll
            for (int c = 0; c < D; ++c) dqh[c] += ds * Kh[j * D + c];
        }
        Ms[i] = m; Ls[i] = il; Ds[i] = Dd;
        // through qhat = unit(q) * q_scale
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            Cs[i * D + c] = dqh[c] * qn[c];                    // contribution to d q_scale
            dqh[c] *= q_scale[c];                              // d unit(q)
            dot += dqh[c] * qn[c];
        }
        float* dst = dq + ((size_t)b * Lq + i) * HD + h * D;
#pragma unroll
        for (int c = 0; c < D; ++c) dst[c] = (dqh[c] - qn[c] * dot) * inrm;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float t = 0.f;
        for (int i = 0; i < Lq; ++i) t += Cs[i * D + c];
        dscale[((size_t)blockIdx.x * 2 + 0) * D + c] = t;
    }
    __syncthreads();
    // ---- phase B
    for (int j = threadIdx.x; j < Lk; j += blockDim.x) {
        float kh[D], v[D], dkh[D], dv[D];
#pragma unroll
        for (int c = 0; c < D; ++c) { kh[c] = Kh[j * D + c]; v[c] = Vs[j * D + c]; dkh[c] = 0.f; dv[c] = 0.f; }
        for (int i = 0; i < Lq; ++i) {
            float d = 0.f, dp = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) { d += Qh[i * D + c] * kh[c]; dp += Gs[i * D + c] * v[c]; }
            const float pij = expf(d * sim_scale - Ms[i]) * Ls[i];
            const float ds = pij * (dp - Ds[i]) * sim_scale;
#pragma unroll
            for (int c = 0; c < D; ++c) { dv[c] += pij * Gs[i * D + c]; dkh[c] += ds * Qh[i * D + c]; }
        }
        const float inrm = Kn[j];
        float dot = 0.f;
        float kn[D];
#pragma unroll
        for (int c = 0; c < D; ++c) {
            kn[c] = (k_scale[c] != 0.f) ? kh[c] / k_scale[c] : 0.f;     // unit key
            Cs[j * D + c] = dkh[c] * kn[c];
            dkh[c] *= k_scale[c];
            dot += dkh[c] * kn[c];
        }
        float* dst = dkv + ((size_t)b * Lk + j) * 2 * HD + h * D;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            dst[c] = (dkh[c] - kn[c] * dot) * inrm;
            dst[HD + c] = dv[c];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float t = 0.f;
        for (int j = 0; j < Lk; ++j) t += Cs[j * D + c];
        dscale[((size_t)blockIdx.x * 2 + 1) * D + c] = t;
    }
}

// LayerNorm backward, one workgroup per row: dx, and the row's contributions to dg (dy * xhat) and db (dy) in contrib [rows][2][D]
// (summed over rows by a column-sum launch).
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* x, const float* g, const float* dy, float* dx, float* contrib,
                                                            int rows, int D, float eps) {
    __shared__ float red[12];
    const int r = blockIdx.x;
    const float* xr = x + (size_t)r * D;
    const float* dr = dy + (size_t)r * D;
    float s = 0.f;
    for (int i = threadIdx.x; i < D; i += 256) s += xr[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float mu = (red[0] + red[1] + red[2] + red[3]) / (float)D;
    float qv = 0.f;
    for (int i = threadIdx.x; i < D; i += 256) { const float d = xr[i] - mu; qv += d * d; }
    qv = wave_sum(qv);
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = qv;
    __syncthreads();
    const float rs = 1.0f / sqrtf((red[4] + red[5] + red[6] + red[7]) / (float)D + eps);
    float a = 0.f, bsum = 0.f;                    // sum dxhat, sum dxhat * xhat
    for (int i = threadIdx.x; i < D; i += 256) {
        const float xh = (xr[i] - mu) * rs, dxh = dr[i] * g[i];
        a += dxh;
        bsum += dxh * xh;
    }
    a = wave_sum(a);
    bsum = wave_sum(bsum);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[4 + (threadIdx.x >> 6)] = bsum; }
    __syncthreads();
    const float ma = (red[0] + red[1] + red[2] + red[3]) / (float)D, mb = (red[4] + red[5] + red[6] + red[7]) / (float)D;
    for (int i = threadIdx.x; i < D; i += 256) {
        const float xh = (xr[i] - mu) * rs, dxh = dr[i] * g[i];
        dx[(size_t)r * D + i] = rs * (dxh - ma - xh * mb);
        contrib[((size_t)r * 2 + 0) * D + i] = dr[i] * xh;
        contrib[((size_t)r * 2 + 1) * D + i] = dr[i];
    }
}

// dx[b][r][:] = scale * dout[b][:]  (backward of a mean over rows with scale = 1/R)
__global__ void bcast_rows_kernel(const float* dout, float* dx, int B, int R, int D, float scale) {
    const size_t total = (size_t)B * R * D;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(idx % D);
        const int b = (int)(idx / ((size_t)R * D));
        dx[idx] = dout[(size_t)b * D + d] * scale;
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

// ------------------------------------------------------------------------------------------------------------------------------
// Multi-head attention of the Transformer policy backbone (torch.nn.MultiheadAttention inside nn.TransformerEncoder/DecoderLayer,
// reference transformer_for_diffusion.py:75-110): softmax(q k^T / sqrt(D) + mask) v per (batch, head), additive float mask [Tq][Tk]
// (-inf = blocked) or none.  q / k / v are column blocks of packed projection outputs: row r of batch b lives at
// ptr + (b*T + r)*ld + head*D.  Sequences are short (horizon 10-16, 2-17 condition tokens): one workgroup per (b, head), K and V
// resident in LDS, one thread per query row.  The backward recomputes the probabilities (nothing but q, k, v is kept from the forward).
__global__ __launch_bounds__(64) void mha_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                     const float* __restrict__ mask, float* __restrict__ out, int Tq, int Tk, int H, int D,
                                                     int ldq, int ldk, int ldv, float scale, float pdrop, unsigned long long seed,
                                                     unsigned long long stream) {
    extern __shared__ __attribute__((aligned(16))) float sm_mha[];
    float* Ks = sm_mha;                       // [Tk][D]
    float* Vs = Ks + (size_t)Tk * D;          // [Tk][D]
    float* Ps = Vs + (size_t)Tk * D;          // [blockDim][Tk] scores of the row each thread owns
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    for (int e = threadIdx.x; e < Tk * D; e += blockDim.x) {
        const int r = e / D, c = e - r * D;
        Ks[e] = k[((size_t)b * Tk + r) * ldk + h * D + c];
        Vs[e] = v[((size_t)b * Tk + r) * ldv + h * D + c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Tq; i += blockDim.x) {
        const float* qr = q + ((size_t)b * Tq + i) * ldq + h * D;
        float* pr = Ps + (size_t)threadIdx.x * Tk;
        float m = -INFINITY;
        for (int j = 0; j < Tk; ++j) {
            float d = 0.f;
            for (int c = 0; c < D; ++c) d += qr[c] * Ks[j * D + c];
            d = d * scale + (mask ? mask[i * Tk + j] : 0.f);
            pr[j] = d;
            m = fmaxf(m, d);
        }
        float l = 0.f;
        for (int j = 0; j < Tk; ++j) { const float e = expf(pr[j] - m); pr[j] = e; l += e; }
        const float il = 1.0f / l;
        if (pdrop > 0.f) {                       // dropout on the normalised probabilities (element index: ((b*H + h)*Tq + i)*Tk + j)
            const float ik = 1.0f / (1.0f - pdrop);
            for (int j = 0; j < Tk; ++j) pr[j] *= dropout_scale(seed, stream, ((size_t)blockIdx.x * Tq + i) * Tk + j, pdrop, ik);
        }
        float* orow = out + ((size_t)b * Tq + i) * (size_t)(H * D) + h * D;
        for (int c = 0; c < D; ++c) {
            float a = 0.f;
            for (int j = 0; j < Tk; ++j) a += pr[j] * Vs[j * D + c];
            orow[c] = a * il;
        }
    }
}

// dq / dk / dv land in buffers with the same packed layout (ld*) as q / k / v.  Phase A (thread = query row): P, dS rows into LDS and
// dq; phase B (thread = key row): dv_j = sum_i P_ij dO_i, dk_j = sum_i dS_ij q_i.  Fixed summation order: deterministic.
__global__ __launch_bounds__(64) void mha_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                     const float* __restrict__ mask, const float* __restrict__ dout, float* __restrict__ dq,
                                                     float* __restrict__ dk, float* __restrict__ dv, int Tq, int Tk, int H, int D, int ldq,
                                                     int ldk, int ldv, float scale, float pdrop, unsigned long long seed,
                                                     unsigned long long stream) {
    extern __shared__ __attribute__((aligned(16))) float sm_mha[];
    float* Ks = sm_mha;                       // [Tk][D]
    float* Vs = Ks + (size_t)Tk * D;          // [Tk][D]
    float* Qs = Vs + (size_t)Tk * D;          // [Tq][D]
    float* Gs = Qs + (size_t)Tq * D;          // [Tq][D]  dO
    float* Ps = Gs + (size_t)Tq * D;          // [Tq][Tk] probabilities
    float* Ss = Ps + (size_t)Tq * Tk;         // [Tq][Tk] dS (already times scale)
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int E = H * D;
    for (int e = threadIdx.x; e < Tk * D; e += blockDim.x) {
        const int r = e / D, c = e - r * D;
        Ks[e] = k[((size_t)b * Tk + r) * ldk + h * D + c];
        Vs[e] = v[((size_t)b * Tk + r) * ldv + h * D + c];
    }
    for (int e = threadIdx.x; e < Tq * D; e += blockDim.x) {
        const int r = e / D, c = e - r * D;
        Qs[e] = q[((size_t)b * Tq + r) * ldq + h * D + c];
        Gs[e] = dout[((size_t)b * Tq + r) * E + h * D + c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Tq; i += blockDim.x) {
        float* pr = Ps + (size_t)i * Tk;
        float* sr = Ss + (size_t)i * Tk;
        float m = -INFINITY;
        for (int j = 0; j < Tk; ++j) {
            float d = 0.f;
            for (int c = 0; c < D; ++c) d += Qs[i * D + c] * Ks[j * D + c];
            d = d * scale + (mask ? mask[i * Tk + j] : 0.f);
            pr[j] = d;
            m = fmaxf(m, d);
        }
        float l = 0.f;
        for (int j = 0; j < Tk; ++j) { const float e = expf(pr[j] - m); pr[j] = e; l += e; }
        const float il = 1.0f / l;
        float Dd = 0.f;
        const float ik = pdrop > 0.f ? 1.0f / (1.0f - pdrop) : 1.0f;
        for (int j = 0; j < Tk; ++j) {
            float dp = 0.f;
            for (int c = 0; c < D; ++c) dp += Gs[i * D + c] * Vs[j * D + c];
            const float keep = pdrop > 0.f ? dropout_scale(seed, stream, ((size_t)blockIdx.x * Tq + i) * Tk + j, pdrop, ik) : 1.0f;
            pr[j] *= il;
            dp *= keep;                          // gradient w.r.t. the probability before dropout
            sr[j] = dp;
            Dd += pr[j] * dp;
        }
        for (int j = 0; j < Tk; ++j) {
            sr[j] = pr[j] * (sr[j] - Dd) * scale;
            if (pdrop > 0.f) pr[j] *= dropout_scale(seed, stream, ((size_t)blockIdx.x * Tq + i) * Tk + j, pdrop, ik);   // P' for dV
        }
        float* dst = dq + ((size_t)b * Tq + i) * ldq + h * D;
        for (int c = 0; c < D; ++c) {
            float a = 0.f;
            for (int j = 0; j < Tk; ++j) a += sr[j] * Ks[j * D + c];
            dst[c] = a;
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < Tk; j += blockDim.x) {
        float* dkr = dk + ((size_t)b * Tk + j) * ldk + h * D;
        float* dvr = dv + ((size_t)b * Tk + j) * ldv + h * D;
        for (int c = 0; c < D; ++c) {
            float a = 0.f, g = 0.f;
            for (int i = 0; i < Tq; ++i) {
                a += Ss[i * Tk + j] * Qs[i * D + c];
                g += Ps[i * Tk + j] * Gs[i * D + c];
            }
            dkr[c] = a;
            dvr[c] = g;
        }
    }
}

extern "C" {

int v2a_get_f32_conv_mode(void);
int v2a_attention_fwd(const float* qkv, float* out, int n_frames, int L, int heads, int head_ch, hipStream_t s) {
    if (!qkv || !out) return V2A_ERR_ARG;
    if (head_ch == 32 && L % 32 == 0 && L <= 256 && v2a_get_f32_conv_mode() == 1 && ((((uintptr_t)qkv | (uintptr_t)out) & 15) == 0)) {
        // three-plane mode (the default fp32 arithmetic of the convs around it): QK^T and PV on the matrix pipe
        const size_t ldsx = (size_t)3 * L * 40 * 2 + (size_t)3 * 32 * (L + 4) * 2;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)attn_mfma_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 256 * 80 + 3 * 32 * 260 * 2);
            attr_set = true;
        }
        hipLaunchKernelGGL(attn_mfma_x3_kernel, dim3(n_frames * heads), dim3(256), ldsx, s, qkv, out, L, heads);
        V2A_CHECK_LAUNCH();
        return V2A_OK;
    }
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

// d(qkv) of the per-frame attention given qkv, its output `out` and d(out); fp32; L * head_ch limited by LDS (4 head matrices resident)
int v2a_attention_bwd(const float* qkv, const float* out, const float* dout, float* dqkv, int n_frames, int L, int heads, int head_ch,
                      hipStream_t s) {
    if (!qkv || !out || !dout || !dqkv) return V2A_ERR_ARG;
    const size_t lds = ((size_t)4 * L * head_ch + 3 * (size_t)L) * sizeof(float);
    if (lds > 160 * 1024) return V2A_ERR_ARG;
    const int threads = L >= 256 ? 256 : ((L + 63) / 64) * 64;
    dim3 grid(n_frames * heads);
#define ATTN_BWD(CH_)                                                                                                         \
    do {                                                                                                                      \
        if (lds > 64 * 1024)                                                                                                  \
            (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<CH_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((attn_bwd_kernel<CH_>), grid, dim3(threads), lds, s, qkv, out, dout, dqkv, L, heads);              \
    } while (0)
    switch (head_ch) {
        case 16: ATTN_BWD(16); break;
        case 32: ATTN_BWD(32); break;
        default: return V2A_ERR_ARG;
    }
#undef ATTN_BWD
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

// bf16-storage variant: qkv / out are bf16 ([n_frames*L][3*C] -> [n_frames*L][C]); softmax state and accumulation stay fp32
int v2a_attention_fwd_h(const void* qkv, void* out, int n_frames, int L, int heads, int head_ch, hipStream_t s) {
    if (!qkv || !out) return V2A_ERR_ARG;
    if (head_ch == 32 && L % 32 == 0 && L >= 32 && L <= 1024 && (heads * 32) % 8 == 0) {      // MFMA path: K and V^T of a head fit in LDS
        const size_t lds_m = (size_t)L * 40 * 2 + (size_t)32 * (L + 4) * 2;
        if (lds_m > 64 * 1024) {
            (void)hipFuncSetAttribute((const void*)attn_mfma_h_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m);
            (void)hipFuncSetAttribute((const void*)attn_mfma_h_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m);
        }
        const int waves = L >= 256 ? 4 : (L + 63) / 64;
        if (g_v2a_half_f16) hipLaunchKernelGGL(attn_mfma_h_kernel<true>, dim3(n_frames * heads), dim3(waves * 64), lds_m, s, (const uint16_t*)qkv, (uint16_t*)out, L, heads);
        else hipLaunchKernelGGL(attn_mfma_h_kernel<false>, dim3(n_frames * heads), dim3(waves * 64), lds_m, s, (const uint16_t*)qkv, (uint16_t*)out, L, heads);
        V2A_CHECK_LAUNCH();
        return V2A_OK;
    }
    const int threads = L >= 256 ? 256 : ((L + 63) / 64) * 64;
    const size_t lds = (size_t)2 * 256 * head_ch * sizeof(float);
    dim3 grid(n_frames * heads);
    const uint16_t* q = (const uint16_t*)qkv;
    uint16_t* o = (uint16_t*)out;
    if (lds > 64 * 1024) {
        if (head_ch == 64) {
            (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<64, uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<64, f16a>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
    }
    if (g_v2a_half_f16) {
        const f16a* qf = (const f16a*)qkv;
        f16a* of = (f16a*)out;
        switch (head_ch) {
            case 16: hipLaunchKernelGGL((attn_fwd_kernel<16, f16a>), grid, dim3(threads), lds, s, qf, of, L, heads); break;
            case 32: hipLaunchKernelGGL((attn_fwd_kernel<32, f16a>), grid, dim3(threads), lds, s, qf, of, L, heads); break;
            case 64: hipLaunchKernelGGL((attn_fwd_kernel<64, f16a>), grid, dim3(threads), lds, s, qf, of, L, heads); break;
            default: return V2A_ERR_ARG;
        }
        V2A_CHECK_LAUNCH();
        return V2A_OK;
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

int v2a_perceiver_attention_bwd(const float* q, const float* kv, const float* q_scale, const float* k_scale, const float* out,
                                const float* dout, float* dq, float* dkv, float* dscale, int B, int Lq, int Lk, int H, int D,
                                float sim_scale, hipStream_t s) {
    if (D != 64 || !q || !kv || !out || !dout || !dq || !dkv || !dscale) return V2A_ERR_ARG;
    const int Lm = Lq > Lk ? Lq : Lk;
    const size_t lds = ((size_t)2 * Lk * D + (size_t)2 * Lq * D + (size_t)Lm * D + 3 * (size_t)Lq + Lk) * sizeof(float);
    if (lds > 160 * 1024) return V2A_ERR_ARG;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)perceiver_attn_bwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((perceiver_attn_bwd_kernel<64>), dim3(B * H), dim3(128), lds, s, q, kv, q_scale, k_scale, out, dout, dq, dkv, dscale,
                       Lq, Lk, H, sim_scale);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

int v2a_layernorm_bwd(const float* x, const float* g, const float* dy, float* dx, float* contrib, int rows, int D, float eps, hipStream_t s) {
    if (!x || !g || !dy || !dx || !contrib || rows <= 0) return V2A_ERR_ARG;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(rows), dim3(256), 0, s, x, g, dy, dx, contrib, rows, D, eps);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

int v2a_bcast_rows(const float* dout, float* dx, int B, int R, int D, float scale, hipStream_t s) {
    if (!dout || !dx) return V2A_ERR_ARG;
    const size_t total = (size_t)B * R * D;
    int g = (int)((total + 255) / 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(bcast_rows_kernel, dim3(g), dim3(256), 0, s, dout, dx, B, R, D, scale);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

int v2a_mha_fwd(const float* q, const float* k, const float* v, const float* mask, float* out, int B, int Tq, int Tk, int H, int D, int ldq,
                int ldk, int ldv, float p_drop, uint64_t seed, uint64_t stream_id, hipStream_t s) {
    if (!q || !k || !v || !out || B <= 0 || Tq <= 0 || Tk <= 0 || H <= 0 || D <= 0 || !(p_drop >= 0.f && p_drop < 1.f)) return V2A_ERR_ARG;
    const size_t lds = ((size_t)2 * Tk * D + (size_t)64 * Tk) * sizeof(float);
    if (lds > 160 * 1024) return V2A_ERR_ARG;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)mha_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mha_fwd_kernel, dim3(B * H), dim3(64), lds, s, q, k, v, mask, out, Tq, Tk, H, D, ldq, ldk, ldv, 1.0f / sqrtf((float)D),
                       p_drop, (unsigned long long)seed, (unsigned long long)stream_id);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_mha_bwd(const float* q, const float* k, const float* v, const float* mask, const float* dout, float* dq, float* dk, float* dv, int B,
                int Tq, int Tk, int H, int D, int ldq, int ldk, int ldv, float p_drop, uint64_t seed, uint64_t stream_id, hipStream_t s) {
    if (!q || !k || !v || !dout || !dq || !dk || !dv || B <= 0 || Tq <= 0 || Tk <= 0 || H <= 0 || D <= 0 || !(p_drop >= 0.f && p_drop < 1.f))
        return V2A_ERR_ARG;
    const size_t lds = ((size_t)2 * Tk * D + (size_t)2 * Tq * D + (size_t)2 * Tq * Tk) * sizeof(float);
    if (lds > 160 * 1024) return V2A_ERR_ARG;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)mha_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mha_bwd_kernel, dim3(B * H), dim3(64), lds, s, q, k, v, mask, dout, dq, dk, dv, Tq, Tk, H, D, ldq, ldk, ldv,
                       1.0f / sqrtf((float)D), p_drop, (unsigned long long)seed, (unsigned long long)stream_id);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

}  // extern "C"
