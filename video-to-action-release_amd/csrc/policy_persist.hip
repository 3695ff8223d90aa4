// predict_action's denoising loop as ONE launch: every scheduler step of the ConditionalUnet1D, the scheduler update and the action
// un-normalisation inside a persistent kernel (reference: DiffusionUnetImagePolicy.conditional_sample / predict_action,
// diffusion_policy/diffusion_unet_image_policy.py:88-133,139-201; the network: model/conditional_unet1d.py:14-246,
// model/conv1d_components.py:7-40).  The rollout loop calls predict_action ~35 times per sub-goal at batch 1
// (diffuser/libero/lb_online_trainer_v7.py:1060-1122): a latency path.
//
// Why a persistent kernel.  At batch 1-2 a layer is a [<= 32 rows] x [5 * Cin] x [Cout] product: 16 FMAs per weight element, no reuse to
// speak of -- the step is bound by streaming the 65 M fp32 parameters (260 MB, ~50 us at HBM speed) and by the ~100 dependent launches it
// used to take (5-6 us each in a replayed graph: 0.6 ms per scheduler step, 5 ms per call).  Here a "launch" is a grid barrier (a few us)
// and a layer's GroupNorm + Mish + FiLM + residual add live in the NEXT layer's loader, so a scheduler step is 35 ops and 30 barriers.
// Measured (profiles/r05_predict_persistent.txt): 3.9 ms for the eight steps against 5.0 ms layer by layer; a phase costs ~14 us, most of
// it dependent trips through the memory side, which is why this does not reach the weight-streaming floor.
//
// Execution model.  `nwg` workgroups of 512 threads (default 128: half the CUs), all resident, walk the same op list; a grid barrier
// separates phases.
// An op is a 1-D convolution over [B][T][C] (channels last) tensors:
//   loader   every workgroup builds the op's WHOLE input in its LDS: up to two channel-concatenated sources, each
//            plain | mish(x) | mish(groupnorm(x)) [* film_scale + film_shift] [+ addend] | sin/cos step embedding;
//            the composite can also be stored to HBM as a plain tensor (a slice per workgroup) for later consumers (identity residuals, skips)
//   product  one wave per output channel (or 2 / 4 waves splitting the input channels), lanes over input channels, the weights read from the
//            engine's fp32 forward pack [Cout][k][Cin] (the operand the training step's convs use: a tap's channels are contiguous; torch's
//            [Cout][Cin][k] puts a lane's k taps 20 bytes apart and was measured at 1.4 TB/s on the 21 MB layers), the time window of the
//            input held in registers (each LDS value feeds k FMAs), 4 / 8 / 16 output times per pass
//   epilogue bias; the last op of a step applies the DDPM / DDIM update to the trajectory in place.
// ConvTranspose1d (Upsample1d) reads torch's [Cin][Cout][4] directly: same wave-per-output-channel scheme, a lane's four taps are 16 bytes.
// Arithmetic is plain fp32 FMA (exact products, fp32 sums) -- nothing is rounded to 16 bits on this path.
#include "common.h"
#include <string.h>

#define PP_THREADS 512
#define PP_WAVES 8
#define PP_MAX_LDS (156 * 1024)
#define PP_MAX_WG 1024
#define PP_BARRIER_BUDGET 200000000ull      // 100 MHz ticks a workgroup waits at one barrier before it gives up (2 s)
#define PP_U 8               // input-channel iterations (64 channels each) whose weights are loaded ahead

enum { PP_SRC_NONE = 0, PP_SRC_PLAIN = 1, PP_SRC_MISH = 2, PP_SRC_GN_MISH = 3, PP_SRC_SINCOS = 4 };
enum { PP_CONV = 0, PP_CONVT = 1 };

// Host and device share these layouts (v2a_hip/policy_persist.py mirrors them with ctypes; v2a_policy_persist_op_bytes guards the size).
struct PPSrc {
    const float* a;        // [B][T][C] (GN kinds: the raw conv output)        SINCOS: unused
    const float* gamma;    // GN affine [C]
    const float* beta;
    const float* film;     // [.][2][C] (scale | shift) of this residual block; sample b of scheduler step s at film + (s * B + b) * 2C
    const float* addend;   // plain [B][T][C] added behind the activation (the residual branch), or NULL
    float* store;          // when set: the composite is also written here as a plain [B][T][C] tensor
    const int* tsteps;     // SINCOS: timestep of row r is tsteps[(row0 + r) / rows_per_step]
    int kind, C, groups, tmod;        // tmod > 0: time index t reads row t % tmod of `a` (a per-sample vector broadcast over scheduler steps)
    int rows_per_step, row0, pad1, pad2;
};
struct PPOp {
    PPSrc src[2];
    const float* w;
    const float* bias;
    float* out;            // [B][Tout][Cout]
    int type, B, Tin, Tout, Cin, Cout, K, stride, pad, ksplit, barrier_after, sched;     // sched: 1 = scheduler update of `traj` with out as eps
    int pad0, pad1;
};
struct PPArgs {
    const PPOp* prologue;  // run once (step encoder + FiLM rows of every scheduler step)
    const PPOp* step_ops;  // run per scheduler step
    const float* coef;     // [nsteps][5]: {sqrt(1-a_t), sqrt(a_t), c0, c1, sigma}  (policy_sched.py ddpm_coeffs / ddim_coeffs)
    const float* noise;    // [nsteps][B][T][Da] (DDPM) or NULL
    float* traj;           // [B][T][Da]: the sample after every scheduler step (normalised)
    const float* amin;     // action limits [Da] or NULL (= -1 / +1)
    const float* amax;
    float* action;         // [B][T][Da] un-normalised
    unsigned* barrier;     // PP_MAX_WG flags (zeroed by the launch entry point)
    const float* init;     // [B][T][Da]: the initial noise (read by the first scheduler step instead of `traj`; never written)
    int* err;              // device-visible error word (pinned host memory): set when a barrier wait exceeds its budget; or NULL
    unsigned long long* trace;   // NULL, or [ops executed][4] 100 MHz ticks of workgroup 0: op start, loader done, product done, barrier passed
    int n_prologue, n_step, nsteps, mode, B, T, Da, pad0;
};

// Grid barrier: workgroup w publishes flags[w] = phase, wave 0 of every workgroup polls all flags (lane l reads flags l, l + 64, ...) until
// none is behind.  No fences: see the note on the data path below.  Phases only grow within a launch; the host zeroes the flags before it.
__device__ __forceinline__ void pp_grid_barrier(unsigned* flags, unsigned& phase, int* err) {
    __syncthreads();                                                // (the caller has waited for its write-through stores: see the main loop)
    ++phase;
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int n = gridDim.x;
        const unsigned long long t0 = wall_clock64();
        for (unsigned spins = 1;; ++spins) {
            unsigned behind = 0;                               // (no short circuit: a lane's loads are independent and in flight together)
            for (int i = threadIdx.x; i < n; i += 64) behind |= (unsigned)(__hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase);
            if (__all(behind == 0)) break;
            // A workgroup that never arrives (not resident: the launch asked for more workgroups than the device can hold next to what else
            // is running) must not hang the GPU: after PP_BARRIER_BUDGET of wall clock the wait is abandoned, the error word raised and every
            // later barrier of the launch falls through (the numbers are garbage; the host raises on its next look at the word).
            if ((spins & 1023u) == 0 && err &&
                (wall_clock64() - t0 > PP_BARRIER_BUDGET || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)) {
                if (threadIdx.x == 0) __hip_atomic_store(err, (int)phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
}

// ---- data that crosses workgroups within the launch travels PAST the per-XCD L2s: stores write through (sc1), loads re-fetch (sc1), and the
// barrier carries no cache maintenance.  An agent-scope release / acquire pair per barrier (L2 write-back + invalidate by 256 workgroups) was
// measured at 15 us, the barrier without it at 3.7 us (tools/probes/r5/barrier_bench.hip, profiles/r05_predict_persistent.txt).  Weights,
// GroupNorm parameters and the FiLM rows (written once, before their first read) use ordinary cached loads.
typedef unsigned int pp_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pp_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 pp_ld4c(__amdgpu_buffer_rsrc_t r, size_t elem) {          // 16 bytes at float index `elem`, coherent
    const pp_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(elem * 4), 0, 16);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float pp_ldc(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pp_stc(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pp_st4c(float* p, float4 v) {
    pp_stc(p, v.x); pp_stc(p + 1, v.y); pp_stc(p + 2, v.z); pp_stc(p + 3, v.w);
}

// sum over the 64 lanes, in every lane: butterflies inside the 16-lane rows with DPP permutes, the four row totals through readlane (a
// ds_bpermute shuffle costs ~0.2 us here, and the GroupNorm statistics need four of these sums in a row)
template <int CTRL> __device__ __forceinline__ float pp_dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float pp_wave_sum(float v) {
    v += pp_dpp_f<0xB1>(v);         // quad_perm [1,0,3,2]
    v += pp_dpp_f<0x4E>(v);         // quad_perm [2,3,0,1]
    v += pp_dpp_f<0x141>(v);        // row_half_mirror
    v += pp_dpp_f<0x140>(v);        // row_mirror
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// ---- loader: the op's input [B][Tp][Ct] (Tp = Tin + 2 * pad rows per sample, zero halo) into LDS.
// One source at a time.  Vector form (C % 4 == 0, every float4 column owned by one thread for all its rows): the global loads of a thread's
// rows are issued together (a workgroup has one wave per SIMD -- nobody else hides the round trip), per-channel parameters live in registers.
#define PP_NB 4                // rows a thread loads ahead (512 threads: the 4096-8192 values of a residual block's tensor are <= 4 float4 each)

// mish(x) = x tanh(log(1 + e^x)) = x w / (w + 2) with w = e^x (e^x + 2): one expf and one division where the literal form takes expf, log1pf and
// tanhf -- every workgroup evaluates the whole layer input, and with one wave per SIMD the three libm calls were a third of a loader's time
__device__ __forceinline__ float pp_mish(float x) {
    if (x > 20.f) return x;                    // (torch's softplus threshold; beyond it tanh is 1 in fp32 anyway)
    const float n = __builtin_amdgcn_exp2f(x * 1.44269504088896340736f), w = n * (n + 2.f);      // v_exp_f32 / v_rcp_f32: ~1 ulp each
    return x * (w * __builtin_amdgcn_rcpf(w + 2.f));
}
__device__ __forceinline__ float4 pp_mish4(float4 v) { return make_float4(pp_mish(v.x), pp_mish(v.y), pp_mish(v.z), pp_mish(v.w)); }

__device__ __forceinline__ void pp_load_src_v4(const PPArgs& A, const PPOp& op, const PPSrc& s, int step, float* xs, float* stats, int Tp, int Ct,
                                               int c_off) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = op.B, Tin = op.Tin, pad = op.pad, C = s.C, Q = C / 4;
    const int tpr = Q < PP_THREADS ? Q : PP_THREADS;            // threads per row (Q is a power of two or a multiple of 256: checked by the caller)
    const int tsh = 31 - __clz(tpr);                            // (a power of two: the caller checked)
    const int nrg = PP_THREADS >> tsh, rg = tid >> tsh;         // row groups
    const int rows = B * Tin;
    const int Tsrc = s.tmod > 0 ? s.tmod : Tin;
    const float* src_a = (step == 0 && s.a == A.traj) ? A.init : s.a;
    if (s.kind == PP_SRC_GN_MISH && Q <= PP_THREADS && rows <= nrg * PP_NB) {
        // ---- the usual case (a residual block's 4096-8192 values): one column and <= PP_NB rows per thread, EVERYTHING it needs from HBM
        // requested in one go (raw values, residual addend, GroupNorm affine, FiLM rows), the values stay in registers across the statistics
        const int col = tid & (tpr - 1);                             // (tpr is a power of two here)
        const int G = s.groups, cg = C / G, qg = cg / 4, g = (4 * col) / cg;
        const __amdgpu_buffer_rsrc_t ra = pp_rsrc(src_a), rd = pp_rsrc(s.addend ? s.addend : src_a);
        float4 v[PP_NB], ad[PP_NB];
#pragma unroll
        for (int j = 0; j < PP_NB; ++j) {
            const int r = rg + j * nrg;
            v[j] = ad[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < rows) {
                v[j] = pp_ld4c(ra, (size_t)r * C + 4 * col);
                if (s.addend) ad[j] = pp_ld4c(rd, (size_t)r * C + 4 * col);
            }
        }
        const float4 ga = *reinterpret_cast<const float4*>(s.gamma + 4 * col), be = *reinterpret_cast<const float4*>(s.beta + 4 * col);
        float4 fs[2], fb[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            fs[b] = make_float4(1.f, 1.f, 1.f, 1.f);
            fb[b] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (s.film && b < B) {
                const float* f = s.film + ((size_t)step * B + b) * 2 * C;
                fs[b] = *reinterpret_cast<const float4*>(f + 4 * col);
                fb[b] = *reinterpret_cast<const float4*>(f + C + 4 * col);
            }
        }
#pragma unroll
        for (int j = 0; j < PP_NB; ++j) {
            const int r = rg + j * nrg;
            if (r < rows) {
                const int b = r >= Tin ? 1 : 0, t = r - b * Tin;
                *reinterpret_cast<float4*>(xs + ((size_t)b * Tp + pad + t) * Ct + c_off + 4 * col) = v[j];
            }
        }
        __syncthreads();
        const float inv_n = 1.0f / (float)(Tin * cg);
        for (int p = wave; p < B * G; p += PP_WAVES) {
            const int b = p >= G ? 1 : 0, gg = p - b * G;
            const float* base = xs + ((size_t)b * Tp + pad) * Ct + c_off + gg * cg;
            float sum = 0.f;
            for (int i = lane; i < Tin * qg; i += 64) {
                const int t = i / qg, q = i - t * qg;
                const float4 x = *reinterpret_cast<const float4*>(base + (size_t)t * Ct + 4 * q);
                sum += (x.x + x.y) + (x.z + x.w);
            }
            const float mean = pp_wave_sum(sum) * inv_n;
            float sq = 0.f;
            for (int i = lane; i < Tin * qg; i += 64) {
                const int t = i / qg, q = i - t * qg;
                const float4 x = *reinterpret_cast<const float4*>(base + (size_t)t * Ct + 4 * q);
                const float d0 = x.x - mean, d1 = x.y - mean, d2 = x.z - mean, d3 = x.w - mean;
                sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
            const float var = pp_wave_sum(sq) * inv_n;
            if (lane == 0) {
                stats[2 * p] = mean;
                stats[2 * p + 1] = 1.0f / sqrtf(var + 1e-5f);
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PP_NB; ++j) {
            const int r = rg + j * nrg;
            if (r < rows) {
                const int b = r >= Tin ? 1 : 0, t = r - b * Tin;
                const float mean = stats[2 * (b * G + g)], rstd = stats[2 * (b * G + g) + 1];
                float4 x = v[j];
                x = make_float4((x.x - mean) * rstd * ga.x + be.x, (x.y - mean) * rstd * ga.y + be.y, (x.z - mean) * rstd * ga.z + be.z,
                                (x.w - mean) * rstd * ga.w + be.w);
                x = pp_mish4(x);
                const float4 sc = b == 0 ? fs[0] : fs[1], sh = b == 0 ? fb[0] : fb[1];
                x = make_float4(sc.x * x.x + sh.x + ad[j].x, sc.y * x.y + sh.y + ad[j].y, sc.z * x.z + sh.z + ad[j].z, sc.w * x.w + sh.w + ad[j].w);
                *reinterpret_cast<float4*>(xs + ((size_t)b * Tp + pad + t) * Ct + c_off + 4 * col) = x;
                if (s.store && (r % (int)gridDim.x) == (int)blockIdx.x) pp_st4c(s.store + (size_t)r * C + 4 * col, x);
            }
        }
        return;
    }
    const __amdgpu_buffer_rsrc_t ra = pp_rsrc(src_a), rd = pp_rsrc(s.addend ? s.addend : src_a);
    for (int col = tid % tpr; col < Q; col += tpr) {
        for (int r0 = rg; r0 < rows; r0 += nrg * PP_NB) {
            float4 v[PP_NB];
#pragma unroll
            for (int j = 0; j < PP_NB; ++j) {
                const int r = r0 + j * nrg;
                if (r < rows) {
                    const int b = r / Tin, t = r - b * Tin;
                    v[j] = pp_ld4c(ra, ((size_t)b * Tsrc + (s.tmod > 0 ? t % s.tmod : t)) * C + 4 * col);
                }
            }
#pragma unroll
            for (int j = 0; j < PP_NB; ++j) {
                const int r = r0 + j * nrg;
                if (r < rows) {
                    const int b = r / Tin, t = r - b * Tin;
                    if (s.kind == PP_SRC_MISH) v[j] = pp_mish4(v[j]);
                    *reinterpret_cast<float4*>(xs + ((size_t)b * Tp + pad + t) * Ct + c_off + 4 * col) = v[j];
                }
            }
        }
    }
    if (s.kind != PP_SRC_GN_MISH) return;
    __syncthreads();
    const int G = s.groups, cg = C / G, qg = cg / 4;            // float4 columns per group
    const float inv_n = 1.0f / (float)(Tin * cg);
    // statistics: (sample, group) pairs over the four waves, two passes over LDS (mean, then the centred second moment: nn.GroupNorm's biased
    // variance)
    for (int p = wave; p < B * G; p += PP_WAVES) {
        const int b = p / G, g = p - b * G;
        const float* base = xs + ((size_t)b * Tp + pad) * Ct + c_off + g * cg;
        float sum = 0.f;
        for (int i = lane; i < Tin * qg; i += 64) {
            const int t = i / qg, q = i - t * qg;
            const float4 x = *reinterpret_cast<const float4*>(base + (size_t)t * Ct + 4 * q);
            sum += (x.x + x.y) + (x.z + x.w);
        }
        const float mean = pp_wave_sum(sum) * inv_n;
        float sq = 0.f;
        for (int i = lane; i < Tin * qg; i += 64) {
            const int t = i / qg, q = i - t * qg;
            const float4 x = *reinterpret_cast<const float4*>(base + (size_t)t * Ct + 4 * q);
            const float d0 = x.x - mean, d1 = x.y - mean, d2 = x.z - mean, d3 = x.w - mean;
            sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
        const float var = pp_wave_sum(sq) * inv_n;
        if (lane == 0) {
            stats[2 * p] = mean;
            stats[2 * p + 1] = 1.0f / sqrtf(var + 1e-5f);
        }
    }
    __syncthreads();
    for (int col = tid % tpr; col < Q; col += tpr) {
        const float4 ga = *reinterpret_cast<const float4*>(s.gamma + 4 * col), be = *reinterpret_cast<const float4*>(s.beta + 4 * col);
        float4 fs[2], fb[2];                                         // FiLM (scale | shift) of up to two samples
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            fs[b] = make_float4(1.f, 1.f, 1.f, 1.f);
            fb[b] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (s.film && b < B) {
                const float* f = s.film + ((size_t)step * B + b) * 2 * C;
                fs[b] = *reinterpret_cast<const float4*>(f + 4 * col);
                fb[b] = *reinterpret_cast<const float4*>(f + C + 4 * col);
            }
        }
        const int g = (4 * col) / cg;
        for (int r0 = rg; r0 < rows; r0 += nrg * PP_NB) {
            float4 ad[PP_NB];
#pragma unroll
            for (int j = 0; j < PP_NB; ++j) {
                const int r = r0 + j * nrg;
                ad[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (s.addend && r < rows) ad[j] = pp_ld4c(rd, (size_t)r * C + 4 * col);
            }
#pragma unroll
            for (int j = 0; j < PP_NB; ++j) {
                const int r = r0 + j * nrg;
                if (r < rows) {
                    const int b = r / Tin, t = r - b * Tin;
                    float4* px = reinterpret_cast<float4*>(xs + ((size_t)b * Tp + pad + t) * Ct + c_off + 4 * col);
                    const float mean = stats[2 * (b * G + g)], rstd = stats[2 * (b * G + g) + 1];
                    float4 x = *px;
                    x = make_float4((x.x - mean) * rstd * ga.x + be.x, (x.y - mean) * rstd * ga.y + be.y, (x.z - mean) * rstd * ga.z + be.z,
                                    (x.w - mean) * rstd * ga.w + be.w);
                    x = pp_mish4(x);
                    const float4 sc = b == 0 ? fs[0] : fs[1], sh = b == 0 ? fb[0] : fb[1];
                    x = make_float4(sc.x * x.x + sh.x + ad[j].x, sc.y * x.y + sh.y + ad[j].y, sc.z * x.z + sh.z + ad[j].z, sc.w * x.w + sh.w + ad[j].w);
                    *px = x;
                    if (s.store && (r % (int)gridDim.x) == (int)blockIdx.x) pp_st4c(s.store + (size_t)r * C + 4 * col, x);
                }
            }
        }
    }
}

// scalar form: any channel count (the 7-channel trajectory, the sin / cos embedding)
__device__ __forceinline__ void pp_load_src_scalar(const PPArgs& A, const PPOp& op, const PPSrc& s, int step, float* xs, float* stats, int Tp, int Ct,
                                                   int c_off) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = op.B, Tin = op.Tin, pad = op.pad, C = s.C;
    if (s.kind == PP_SRC_SINCOS) {
        const int half = C / 2;
        for (int i = tid; i < B * Tin * C; i += PP_THREADS) {
            const int c = i % C, t = (i / C) % Tin, b = i / (C * Tin);
            const float tv = (float)s.tsteps[(s.row0 + t) / s.rows_per_step];
            const int k = c < half ? c : c - half;
            const float a = tv * expf((float)k * -(logf(10000.f) / (float)(half - 1)));
            xs[((size_t)b * Tp + pad + t) * Ct + c_off + c] = c < half ? sinf(a) : cosf(a);
        }
        return;
    }
    const float* src_a = (step == 0 && s.a == A.traj) ? A.init : s.a;
    for (int i = tid; i < B * Tin * C; i += PP_THREADS) {
        const int c = i % C, t = (i / C) % Tin, b = i / (C * Tin);
        const int tr = s.tmod > 0 ? t % s.tmod : t;
        float v = pp_ldc(src_a + ((size_t)b * (s.tmod > 0 ? s.tmod : Tin) + tr) * C + c);
        if (s.kind == PP_SRC_MISH) v = pp_mish(v);
        xs[((size_t)b * Tp + pad + t) * Ct + c_off + c] = v;
    }
    if (s.kind != PP_SRC_GN_MISH) return;
    __syncthreads();
    const int G = s.groups, cg = C / G, n = Tin * cg;
    for (int p = wave; p < B * G; p += PP_WAVES) {
        const int b = p / G, g = p % G;
        float sum = 0.f;
        for (int i = lane; i < n; i += 64) sum += xs[((size_t)b * Tp + pad + i / cg) * Ct + c_off + g * cg + i % cg];
        const float mean = pp_wave_sum(sum) / (float)n;
        float sq = 0.f;
        for (int i = lane; i < n; i += 64) {
            const float d = xs[((size_t)b * Tp + pad + i / cg) * Ct + c_off + g * cg + i % cg] - mean;
            sq += d * d;
        }
        const float var = pp_wave_sum(sq) / (float)n;
        if (lane == 0) {
            stats[2 * p] = mean;
            stats[2 * p + 1] = 1.0f / sqrtf(var + 1e-5f);
        }
    }
    __syncthreads();
    for (int i = tid; i < B * Tin * C; i += PP_THREADS) {
        const int c = i % C, t = (i / C) % Tin, b = i / (C * Tin);
        const int p = b * G + c / cg;
        float* px = &xs[((size_t)b * Tp + pad + t) * Ct + c_off + c];
        float v = (*px - stats[2 * p]) * stats[2 * p + 1] * s.gamma[c] + s.beta[c];
        v = pp_mish(v);
        if (s.film) {
            const float* f = s.film + ((size_t)step * B + b) * 2 * C;
            v = f[c] * v + f[C + c];
        }
        if (s.addend) v += pp_ldc(s.addend + i);
        *px = v;
        if (s.store && ((i / C) % (int)gridDim.x) == (int)blockIdx.x) pp_stc(s.store + i, v);
    }
}

__device__ __forceinline__ void pp_load(const PPArgs& A, const PPOp& op, int step, float* xs, float* stats, int Tp, int Ct) {
    const int tid = threadIdx.x;
    const int B = op.B, Tin = op.Tin, pad = op.pad;
    // zero every row of the image that no source writes: the halos and the rows behind them that a register window may reach
    for (int b = 0; b < B; ++b)
        for (int r = 0; r < Tp; ++r) {
            if (r >= pad && r < pad + Tin) continue;
            float* row = xs + ((size_t)b * Tp + r) * Ct;
            for (int c = tid; c < Ct; c += PP_THREADS) row[c] = 0.f;
        }
    int c_off = 0;
    for (int si = 0; si < 2; ++si) {
        const PPSrc& s = op.src[si];
        if (s.kind == PP_SRC_NONE) break;
        const int Q = s.C / 4;
        const bool pow2 = Q > 0 && (Q & (Q - 1)) == 0;
        const bool v4 = s.kind != PP_SRC_SINCOS && (s.C % 4) == 0 && (c_off % 4) == 0 && (Ct % 4) == 0 && (pow2 || Q % PP_THREADS == 0) &&
                        (s.kind != PP_SRC_GN_MISH || ((s.C / s.groups) % 4 == 0 && B <= 2));
        if (v4) pp_load_src_v4(A, op, s, step, xs, stats, Tp, Ct, c_off);
        else pp_load_src_scalar(A, op, s, step, xs, stats, Tp, Ct, c_off);
        c_off += s.C;
    }
    __syncthreads();
}

// acc[MT] over the 64 lanes: afterwards lane l holds the total of row l % MT.  Recursive halving inside the 16-lane rows with DPP lane
// permutes (row mirror, half mirror, quad reverse, quad swap: the partner differs in the bit that decides which half a lane keeps), then the
// lanes that hold the same row are summed (row rotates, two cross-row exchanges).  The same thing through ds_bpermute shuffles took 3.5 us for
// 16 rows -- half of a small layer's product (profiles/r05_predict_persistent.txt).
template <int CTRL> __device__ __forceinline__ float pp_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int MT, int BIT, int CTRL> __device__ __forceinline__ void pp_halve(float (&a)[MT], int lane, int n) {
    const bool up = (lane & BIT) != 0;
#pragma unroll
    for (int j = 0; j < MT / 2; ++j) {
        if (j < n) {
            const float keep = up ? a[j + n] : a[j], send = up ? a[j] : a[j + n];
            a[j] = keep + pp_dpp<CTRL>(send);
        }
    }
}
template <int MT> __device__ __forceinline__ float pp_reduce_rows(float (&a)[MT], int lane) {
    static_assert(MT == 4 || MT == 8 || MT == 16, "rows per pass");
    if constexpr (MT == 16) pp_halve<MT, 8, 0x140>(a, lane, 8);                 // row_mirror: partner 15 - l
    if constexpr (MT >= 8) pp_halve<MT, 4, 0x141>(a, lane, 4);                  // row_half_mirror: partner 7 - l within its half
    pp_halve<MT, 2, 0x1B>(a, lane, 2);                                          // quad_perm [3,2,1,0]
    pp_halve<MT, 1, 0xB1>(a, lane, 1);                                          // quad_perm [1,0,3,2]
    float v = a[0];
    if constexpr (MT <= 8) v += pp_dpp<0x128>(v);                               // row_ror:8
    if constexpr (MT <= 4) v += pp_dpp<0x124>(v);                               // row_ror:4
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
template <int MT> __device__ __forceinline__ int pp_row_of_lane(int lane) { return lane & (MT - 1); }

__device__ __forceinline__ void pp_sched(const PPArgs& A, int step, size_t idx, float eps) {
    const float* cf = A.coef + 5 * step;
    const float x = step == 0 ? A.init[idx] : pp_ldc(A.traj + idx);
    float x0 = (x - cf[0] * eps) / cf[1];
    x0 = fminf(fmaxf(x0, -1.f), 1.f);
    float v;
    if (A.mode == 0) {
        v = cf[2] * x0 + cf[3] * x;
        if (A.noise && cf[4] != 0.f) v += cf[4] * A.noise[(size_t)step * A.B * A.T * A.Da + idx];
    } else {
        v = cf[2] * x0 + cf[3] * eps;
    }
    pp_stc(A.traj + idx, v);
}

// ---- weights of an op's FIRST round (first output channel of this wave, first PP_U x 64 input channels of its part), requested long before
// they are used: right after the previous op's product, so that the barrier and the loader hide the HBM round trip.  For most layers this is
// all the wave needs (ksplit keeps a part at <= 512 channels).
#define PP_KMAX 5
__device__ __forceinline__ void pp_prefetch(const PPOp& op, float (&wpre)[PP_U * PP_KMAX]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ks = op.ksplit, ksh = ks >> 1, per_wg = PP_WAVES >> ksh, sub = wave & (ks - 1), Ct = op.Cin, K = op.K;      // ks in {1, 2, 4}
    const int ci_per = (Ct + ks - 1) >> ksh, ci_lo = sub * ci_per, ci_hi = min(Ct, ci_lo + ci_per);
    const int co = blockIdx.x * per_wg + (wave >> ksh);
    const bool live = co < op.Cout;
#pragma unroll
    for (int u = 0; u < PP_U; ++u) {
        const int ci = ci_lo + lane + 64 * u;
        const bool ok = live && ci < ci_hi;
        // Conv1d: the engine's forward pack [Cout][K][Cin] (a tap's 64 channels are 256 contiguous bytes per wave); ConvTranspose1d: torch's
        // [Cin][Cout][4]
        const float* p = op.type == PP_CONV ? op.w + (size_t)(live ? co : 0) * K * Ct + (ok ? ci : 0)
                                            : op.w + ((size_t)(ok ? ci : 0) * op.Cout + (live ? co : 0)) * 4;
        const size_t ts = op.type == PP_CONV ? (size_t)Ct : 1;
#pragma unroll
        for (int k = 0; k < PP_KMAX; ++k) wpre[u * PP_KMAX + k] = (ok && k < K) ? p[k * ts] : 0.f;
    }
}

// ---- product, Conv1d: out[b][t][co] = bias[co] + sum_{ci, tap} W[co][tap][ci] * x[b][t * S + tap - pad][ci] (W: the forward pack).  Up to two tiles of MT output
// times (the two samples of a batch, or the two halves of a 32-row prologue layer) share one pass over the weights.
template <int K, int S, int MT>
__device__ __forceinline__ void pp_conv(const PPArgs& A, const PPOp& op, int step, const float* xs, float* red, int Tp, int Ct,
                                        const float (&wpre)[PP_U * PP_KMAX]) {
    constexpr int NX = (MT - 1) * S + K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ks = op.ksplit, ksh = ks >> 1;                    // waves per output channel (1, 2, 4) and its log2
    const int per_wg = PP_WAVES >> ksh;                         // output channels a workgroup works on at a time
    const int sub = wave & (ks - 1);                            // which part of the input channels this wave takes
    const int ci_per = (Ct + ks - 1) >> ksh;
    const int ci_lo = sub * ci_per, ci_hi = min(Ct, ci_lo + ci_per);
    const int rl = pp_row_of_lane<MT>(lane);
    const bool split_t = op.Tout > MT;                          // two passes over the times of one sample (32-row prologue layers)
    const bool two = split_t || op.B > 1;                       // (the host refuses more than two tiles)
    const int b1 = split_t ? 0 : 1, t1 = split_t ? MT : 0;     // tile 1: (sample, first time)
    const float* xb0 = xs;
    const float* xb1 = xs + ((size_t)b1 * Tp + (size_t)t1 * S) * Ct;
    for (int co0 = blockIdx.x * per_wg; co0 < op.Cout; co0 += gridDim.x * per_wg) {          // (workgroup-uniform trip count)
        const int co = co0 + (wave >> ksh);
        const bool live = co < op.Cout;
        const float* wrow = op.w + (size_t)(live ? co : 0) * Ct * K;
        float acc0[MT], acc1[MT];
#pragma unroll
        for (int r = 0; r < MT; ++r) acc0[r] = acc1[r] = 0.f;
        if (live) {
            for (int base = ci_lo + lane; base < ci_hi; base += 64 * PP_U) {
                float w[PP_U][K];
                if (co0 == (int)blockIdx.x * per_wg && base == ci_lo + lane) {
#pragma unroll
                    for (int u = 0; u < PP_U; ++u)
#pragma unroll
                        for (int k = 0; k < K; ++k) w[u][k] = wpre[u * PP_KMAX + k];
                } else {
#pragma unroll
                    for (int u = 0; u < PP_U; ++u) {
                        const int ci = base + 64 * u;
#pragma unroll
                        for (int k = 0; k < K; ++k) w[u][k] = ci < ci_hi ? wrow[(size_t)k * Ct + ci] : 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < PP_U; ++u) {
                    const int ci = base + 64 * u;
                    if (ci < ci_hi) {
                        float x[NX];
#pragma unroll
                        for (int j = 0; j < NX; ++j) x[j] = xb0[(size_t)j * Ct + ci];
#pragma unroll
                        for (int r = 0; r < MT; ++r)
#pragma unroll
                            for (int k = 0; k < K; ++k) acc0[r] = fmaf(w[u][k], x[r * S + k], acc0[r]);
                        if (two) {
#pragma unroll
                            for (int j = 0; j < NX; ++j) x[j] = xb1[(size_t)j * Ct + ci];
#pragma unroll
                            for (int r = 0; r < MT; ++r)
#pragma unroll
                                for (int k = 0; k < K; ++k) acc1[r] = fmaf(w[u][k], x[r * S + k], acc1[r]);
                        }
                    }
                }
            }
        }
        float v0 = pp_reduce_rows<MT>(acc0, lane), v1 = 0.f;
        if (two) v1 = pp_reduce_rows<MT>(acc1, lane);
        if (ks > 1) {                                    // the parts of one output channel meet in LDS, summed in part order
            __syncthreads();
            if (lane < MT) {
                red[(wave * 2) * MT + rl] = v0;
                red[(wave * 2 + 1) * MT + rl] = v1;
            }
            __syncthreads();
            if (sub == 0) {
                v0 = red[(wave * 2) * MT + rl];
                v1 = red[(wave * 2 + 1) * MT + rl];
                for (int j = 1; j < ks; ++j) {
                    v0 += red[((wave + j) * 2) * MT + rl];
                    v1 += red[((wave + j) * 2 + 1) * MT + rl];
                }
            }
        }
        if (live && sub == 0 && lane < MT) {
            const float bias = op.bias ? op.bias[co] : 0.f;
            if (rl < op.Tout) {
                const size_t idx = ((size_t)rl) * op.Cout + co;
                if (op.sched) pp_sched(A, step, idx, v0 + bias);
                else pp_stc(op.out + idx, v0 + bias);
            }
            if (two && t1 + rl < op.Tout) {
                const size_t idx = ((size_t)b1 * op.Tout + t1 + rl) * op.Cout + co;
                if (op.sched) pp_sched(A, step, idx, v1 + bias);
                else pp_stc(op.out + idx, v1 + bias);
            }
        }
    }
}

// ---- product, ConvTranspose1d(k = 4, stride 2, pad 1), TI input times: out[2m] = x[m] w1 + x[m-1] w3, out[2m+1] = x[m] w2 + x[m+1] w0 with
// W [Cin][Cout][4].  Same wave-per-output-channel scheme as pp_conv; a lane's four taps are 16 consecutive bytes, the lanes of a wave stride
// Cout * 16 bytes apart (the four waves of a workgroup take neighbouring output channels, i.e. the other three quarters of the same 64-byte lines).
template <int TI>
__device__ __forceinline__ void pp_convt(const PPOp& op, const float* xs, float* red, int Tp, int Ct, const float (&wpre)[PP_U * PP_KMAX]) {
    constexpr int MT = 2 * TI;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ks = op.ksplit, ksh = ks >> 1, per_wg = PP_WAVES >> ksh, sub = wave & (ks - 1);
    const int ci_per = (Ct + ks - 1) >> ksh;
    const int ci_lo = sub * ci_per, ci_hi = min(Ct, ci_lo + ci_per);
    const int rl = pp_row_of_lane<MT>(lane);
    const bool two = op.B > 1;
    for (int co0 = blockIdx.x * per_wg; co0 < op.Cout; co0 += gridDim.x * per_wg) {
        const int co = co0 + (wave >> ksh);
        const bool live = co < op.Cout;
        float acc0[MT], acc1[MT];
#pragma unroll
        for (int r = 0; r < MT; ++r) acc0[r] = acc1[r] = 0.f;
        if (live) {
            const float* xb1 = xs + (size_t)Tp * Ct;                  // sample 1; row j = input time j - 1 (one halo row in front)
            for (int base = ci_lo + lane; base < ci_hi; base += 64 * PP_U) {
                float w[PP_U][4];
                if (co0 == (int)blockIdx.x * per_wg && base == ci_lo + lane) {
#pragma unroll
                    for (int u = 0; u < PP_U; ++u)
#pragma unroll
                        for (int k = 0; k < 4; ++k) w[u][k] = wpre[u * PP_KMAX + k];
                } else {
#pragma unroll
                    for (int u = 0; u < PP_U; ++u) {
                        const int ci = base + 64 * u;
                        const f32x4 t = ci < ci_hi ? *reinterpret_cast<const f32x4*>(op.w + ((size_t)ci * op.Cout + co) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                        w[u][0] = t[0]; w[u][1] = t[1]; w[u][2] = t[2]; w[u][3] = t[3];
                    }
                }
#pragma unroll
                for (int u = 0; u < PP_U; ++u) {
                    const int ci = base + 64 * u;
                    if (ci < ci_hi) {
                        float x[TI + 2];
#pragma unroll
                        for (int j = 0; j < TI + 2; ++j) x[j] = xs[(size_t)j * Ct + ci];
#pragma unroll
                        for (int m = 0; m < TI; ++m) {
                            acc0[2 * m] = fmaf(w[u][1], x[m + 1], fmaf(w[u][3], x[m], acc0[2 * m]));
                            acc0[2 * m + 1] = fmaf(w[u][2], x[m + 1], fmaf(w[u][0], x[m + 2], acc0[2 * m + 1]));
                        }
                        if (two) {
#pragma unroll
                            for (int j = 0; j < TI + 2; ++j) x[j] = xb1[(size_t)j * Ct + ci];
#pragma unroll
                            for (int m = 0; m < TI; ++m) {
                                acc1[2 * m] = fmaf(w[u][1], x[m + 1], fmaf(w[u][3], x[m], acc1[2 * m]));
                                acc1[2 * m + 1] = fmaf(w[u][2], x[m + 1], fmaf(w[u][0], x[m + 2], acc1[2 * m + 1]));
                            }
                        }
                    }
                }
            }
        }
        float v0 = pp_reduce_rows<MT>(acc0, lane), v1 = 0.f;
        if (two) v1 = pp_reduce_rows<MT>(acc1, lane);
        if (ks > 1) {
            __syncthreads();
            if (lane < MT) {
                red[(wave * 2) * MT + rl] = v0;
                red[(wave * 2 + 1) * MT + rl] = v1;
            }
            __syncthreads();
            if (sub == 0) {
                v0 = red[(wave * 2) * MT + rl];
                v1 = red[(wave * 2 + 1) * MT + rl];
                for (int j = 1; j < ks; ++j) {
                    v0 += red[((wave + j) * 2) * MT + rl];
                    v1 += red[((wave + j) * 2 + 1) * MT + rl];
                }
            }
        }
        if (live && sub == 0 && lane < MT) {
            const float bias = op.bias ? op.bias[co] : 0.f;
            pp_stc(op.out + ((size_t)rl) * op.Cout + co, v0 + bias);
            if (two) pp_stc(op.out + ((size_t)op.Tout + rl) * op.Cout + co, v1 + bias);
        }
    }
}

#define PP_TRACE(slot)                                                                     \
    if (A.trace && blockIdx.x == 0 && threadIdx.x == 0) A.trace[4 * (size_t)opn + slot] = wall_clock64();

__device__ __forceinline__ void pp_run_op(const PPArgs& A, const PPOp& op, int step, float* lds, int opn, const float (&wpre)[PP_U * PP_KMAX]) {
    __syncthreads();                          // the LDS image of the previous op is not read any more
    PP_TRACE(0)
    const int Ct = op.Cin;
    int Tp = op.Tin + 2 * op.pad;
    // the register window of the last pass may reach behind the last real row: those rows exist (zero or stale, their products are discarded)
    int MT = op.Tout <= 4 ? 4 : (op.Tout <= 8 ? 8 : 16);
    if (op.type == PP_CONV) {
        const int passes = (op.Tout + MT - 1) / MT;
        const int need = (passes * MT - 1) * op.stride + op.K;
        if (need > Tp) Tp = need;
    }
    float* stats = lds;                       // [B * groups][2]
    float* xs = lds + 64;
    float* red = xs + (size_t)op.B * Tp * Ct;     // split reduction scratch behind the input image: PP_WAVES * 2 * 16 floats
    pp_load(A, op, step, xs, stats, Tp, Ct);
    PP_TRACE(1)
    if (op.type == PP_CONVT) {
        if (op.Tin == 4) pp_convt<4>(op, xs, red, Tp, Ct, wpre);
        else pp_convt<8>(op, xs, red, Tp, Ct, wpre);
        return;
    }
#define PP_CASE(KK, SS)                                                              \
    if (op.K == KK && op.stride == SS) {                                             \
        if (MT == 4) pp_conv<KK, SS, 4>(A, op, step, xs, red, Tp, Ct, wpre);               \
        else if (MT == 8) pp_conv<KK, SS, 8>(A, op, step, xs, red, Tp, Ct, wpre);          \
        else pp_conv<KK, SS, 16>(A, op, step, xs, red, Tp, Ct, wpre);                      \
        return;                                                                      \
    }
    PP_CASE(5, 1)
    PP_CASE(3, 1)
    PP_CASE(1, 1)
    PP_CASE(3, 2)
#undef PP_CASE
}

__global__ __launch_bounds__(PP_THREADS) void policy_persist_kernel(PPArgs A) {
    extern __shared__ float lds[];
    unsigned target = 0;                       // (barrier phase)
    int opn = 0;
    float wpre[PP_U * PP_KMAX];                // the next op's first weights, in flight across the barrier and the loader
    const int total = A.n_prologue + A.nsteps * A.n_step;
    pp_prefetch(A.n_prologue > 0 ? A.prologue[0] : A.step_ops[0], wpre);
    for (int s = 0, i = 0; opn < total; ++opn) {
        const bool pro = opn < A.n_prologue;
        const PPOp& op = pro ? A.prologue[opn] : A.step_ops[i];
        pp_run_op(A, op, s, lds, opn, wpre);
        PP_TRACE(2)
        // which op comes next (the step ops repeat)
        int ni = i, ns = s;
        if (!pro) {
            if (++ni == A.n_step) { ni = 0; ++ns; }
        }
        __builtin_amdgcn_s_waitcnt(0);           // this wave's write-through stores have arrived -- BEFORE the prefetch below is issued, so
        asm volatile("" ::: "memory");            // that nothing at the barrier has to wait for the prefetched weights
        if (opn + 1 < total) pp_prefetch(opn + 1 < A.n_prologue ? A.prologue[opn + 1] : A.step_ops[ni], wpre);
        if (op.barrier_after) pp_grid_barrier(A.barrier, target, A.err);
        PP_TRACE(3)
        i = ni;
        s = ns;
    }
    // un-normalise (normalizer.py:152-157: clamp to [-1, 1] only if any element is outside, then map to the action limits); the step ops end
    // with a barrier, so the trajectory is final here
    if (blockIdx.x == 0) {
        __shared__ int any;
        if (threadIdx.x == 0) any = 0;
        __syncthreads();
        const int n = A.B * A.T * A.Da;
        int f = 0;
        for (int i = threadIdx.x; i < n; i += PP_THREADS) {
            const float v = pp_ldc(A.traj + i);
            f |= (v > 1.f || v < -1.f);
        }
        if (f) atomicOr(&any, 1);
        __syncthreads();
        const int clampit = any;
        for (int i = threadIdx.x; i < n; i += PP_THREADS) {
            float v = pp_ldc(A.traj + i);
            if (clampit) v = fminf(fmaxf(v, -1.f), 1.f);
            v = (v + 1.f) / 2.0f;
            const float lo = A.amin ? A.amin[i % A.Da] : -1.0f, hi = A.amax ? A.amax[i % A.Da] : 1.0f;
            A.action[i] = v * (hi - lo) + lo;
        }
    }
}

extern "C" {

size_t v2a_policy_persist_op_bytes(void) { return sizeof(PPOp); }
int v2a_policy_persist_waves_per_wg(void) { return PP_WAVES; }
size_t v2a_policy_persist_args_bytes(void) { return sizeof(PPArgs); }

// LDS floats an op needs (host side: validate a program before the first launch).  0 = the op cannot run (too large / unsupported shape).
size_t v2a_policy_persist_lds_bytes(int B, int Tin, int Tout, int Cin, int K, int stride, int pad, int type) {
    int Tp = Tin + 2 * pad;
    if (type == PP_CONV) {
        const int MT = Tout <= 4 ? 4 : (Tout <= 8 ? 8 : 16);
        const int passes = (Tout + MT - 1) / MT;
        const int need = (passes * MT - 1) * stride + K;
        if (need > Tp) Tp = need;
        if (B * passes > 2) return 0;                 // two accumulator tiles per pass over the weights
        if (!((K == 5 && stride == 1) || (K == 3 && stride == 1) || (K == 1 && stride == 1) || (K == 3 && stride == 2))) return 0;
    } else {
        if (K != 4 || stride != 2 || pad != 1 || (Tin != 4 && Tin != 8) || Tout != 2 * Tin || B > 2) return 0;
    }
    const size_t bytes = (64 + (size_t)B * Tp * Cin + PP_WAVES * 2 * 16) * sizeof(float);
    return bytes <= PP_MAX_LDS ? bytes : 0;
}

// args_host: a PPArgs filled by the caller (device pointers inside).  nwg workgroups must all be resident: nwg <= number of CUs.
int v2a_policy_persist_launch(const void* args_host, int nwg, size_t lds_bytes, hipStream_t stream) {
    if (!args_host || nwg < 1 || nwg > PP_MAX_WG || lds_bytes > PP_MAX_LDS) return V2A_ERR_ARG;
    PPArgs A;
    ::memcpy(&A, args_host, sizeof(A));
    if (!A.step_ops || !A.traj || !A.init || !A.action || !A.barrier || !A.coef || A.nsteps < 1 || A.B < 1) return V2A_ERR_ARG;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(policy_persist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PP_MAX_LDS) !=
            hipSuccess)
            return V2A_ERR_LAUNCH;
        attr_set = true;
    }
    if (A.err) {                                  // (a pinned host word from v2a_dp_errword_alloc: the kernel needs its device address)
        int* dev = nullptr;
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&dev), A.err, 0) != hipSuccess) return V2A_ERR_ARG;
        A.err = dev;
    }
    if (hipMemsetAsync(A.barrier, 0, PP_MAX_WG * sizeof(unsigned), stream) != hipSuccess) return V2A_ERR_LAUNCH;
    hipLaunchKernelGGL(policy_persist_kernel, dim3(nwg), dim3(PP_THREADS), lds_bytes, stream, A);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

}  // extern "C"
