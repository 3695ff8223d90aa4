// Fused multi-tensor optimiser tail of the policy train step (HBM-bound, fp32 master weights):
//   global grad L2 norm -> clip to max_norm -> AdamW -> zero grads -> EMA
// replaces lb_online_trainer_v7.py:604-624 (accelerator.clip_grad_norm_, AdamW.step, zero_grad, ema.update) with
// the arithmetic of torch.nn.utils.clip_grad_norm_, torch.optim.AdamW and ema_pytorch 0.2.3 (restated; see
// oracle/optim.py).  All step-dependent scalars live in a device-side state block advanced by a 1-thread kernel,
// so the whole tail is capturable in a hipGraph and replayed with no host involvement.
//
// Tensor table (device, int64): per tensor {p, g, m, v, ema, numel}; chunk map (int32): {tensor, start}.
#include "common.h"

#define MT_CHUNK 16384

struct OptState {           // device resident
    double lr, b1, b2, eps, wd, max_norm;
    double ema_inv_gamma, ema_power, ema_min, ema_beta;
    long long step;          // AdamW step (after increment)
    long long ema_step;      // ema_pytorch `step` buffer
    int ema_initted;
    int ema_update_after, ema_update_every;
    // derived per step
    float clip_coef, grad_norm;
    float step_size, inv_sqrt_bc2, decay_mul;   // decay_mul = 1 - lr*wd
    float ema_decay;
    int ema_mode;            // bit0 copy, bit1 lerp
    // dynamic loss scaling of the fp16 policy mode = torch.cuda.amp.GradScaler's contract (the reference trains under accelerate's fp16
    // mixed precision: lb_online_trainer_v7.py:72-76; backward of the scaled loss :604, clip_grad_norm_ on UNSCALED gradients :608,
    // optimiser step skipped on inf / nan :612, ema.update() regardless :623).  scaler_on = 0: everything below is inert.
    int scaler_on;
    float loss_scale;        // S: the loss-gradient kernel multiplies by it (v2a_mse_loss reads it from here), the gradients carry it
    int growth_tracker, growth_interval;
    float growth_factor, backoff_factor;
    int skip;                // this step found a non-finite gradient: parameters / moments / Adam step untouched
    long long skipped_steps;
};

__global__ __launch_bounds__(256) void mt_sumsq_kernel(const int64_t* table, const int* chunks_all, double* partial_all, int chunk0) {
    __shared__ double sm[4];
    const int* chunks = chunks_all + 2 * chunk0;       // (a launch may cover a sub-range of the chunk list: v2a_opt_presum)
    double* partial = partial_all + chunk0;
    const int t = chunks[2 * blockIdx.x], start = chunks[2 * blockIdx.x + 1];
    const float* g = reinterpret_cast<const float*>(table[t * 6 + 1]);
    const long long n = table[t * 6 + 5];
    const int cnt = (int)min((long long)MT_CHUNK, n - start);
    float s = 0.f;
    const float* gp = g + start;
    if ((((uintptr_t)gp) & 15) == 0) {                  // 16-B pieces (the 4-B loop ran at 3.3 TB/s)
        const int c4 = cnt >> 2;
        for (int i = threadIdx.x; i < c4; i += 256) {
            const f32x4 v = reinterpret_cast<const f32x4*>(gp)[i];
            s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
        for (int i = (c4 << 2) + threadIdx.x; i < cnt; i += 256) { const float v = gp[i]; s += v * v; }
    } else {
        for (int i = threadIdx.x; i < cnt; i += 256) { const float v = gp[i]; s += v * v; }
    }
    double d = wave_sum_d((double)s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// one workgroup: finish the norm, advance step counters, derive this step's scalars
__global__ __launch_bounds__(256) void opt_advance_kernel(OptState* st, const double* partial, int nchunks) {
    __shared__ double sm[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nchunks; i += 256) s += partial[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x != 0) return;
    double norm = sqrt(sm[0] + sm[1] + sm[2] + sm[3]);
    double unscale = 1.0;
    st->skip = 0;
    if (st->scaler_on) {
        // GradScaler.unscale_ + step + update: the sum of squares of the SCALED gradients overflows / is nan exactly when one of them is
        const bool bad = !(norm == norm) || norm > 3.0e38;
        unscale = 1.0 / (double)st->loss_scale;
        if (bad) {
            st->skip = 1;
            st->skipped_steps += 1;
            st->loss_scale *= st->backoff_factor;
            st->growth_tracker = 0;
        } else if (++st->growth_tracker >= st->growth_interval) {
            st->loss_scale *= st->growth_factor;
            st->growth_tracker = 0;
        }
        norm *= unscale;
    }
    st->grad_norm = (float)norm;
    double coef = st->max_norm / ((double)(float)norm + 1e-6);
    st->clip_coef = (float)((coef > 1.0 ? 1.0 : coef) * unscale);      // the update kernel multiplies every gradient by this: unscale is free
    if (!st->skip) st->step += 1;
    const double bc1 = 1.0 - pow(st->b1, (double)st->step), bc2 = 1.0 - pow(st->b2, (double)st->step);
    st->step_size = (float)(st->lr / bc1);
    st->inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    st->decay_mul = (float)(1.0 - st->lr * st->wd);
    // ema_pytorch.EMA.update()
    const long long es = st->ema_step;
    st->ema_step = es + 1;
    int mode = 0;
    if (es % st->ema_update_every == 0) {
        if (es <= st->ema_update_after) mode = 1;
        else {
            if (!st->ema_initted) { mode |= 1; st->ema_initted = 1; }
            mode |= 2;
        }
    }
    double epoch = (double)(st->ema_step - st->ema_update_after - 1);
    if (epoch < 0) epoch = 0;
    double dec = 0.0;
    if (epoch > 0) {
        dec = 1.0 - pow(1.0 + epoch / st->ema_inv_gamma, -st->ema_power);
        if (dec < st->ema_min) dec = st->ema_min;
        if (dec > st->ema_beta) dec = st->ema_beta;
    }
    st->ema_decay = (float)dec;
    st->ema_mode = mode;
}

// 16-bit side copy of a packed operand element: format 0 bf16, 1 IEEE fp16 (the twins of the 16-bit MFMA modes), 2 = the three bf16 planes
// hi / mid / lo of the fp32 value, `n` (the tensor's element count) apart: conv_p3's pre-split weight operand
__device__ __forceinline__ void opt_store_twin(unsigned short* pkh, size_t i, float x, int fmt, size_t n) {
    if (fmt == 2) {
        unsigned short h, m, l;
        v2a_split3x1(x, h, m, l);
        pkh[i] = h; pkh[n + i] = m; pkh[2 * n + i] = l;
    } else pkh[i] = fmt ? v2a_f2h<true>(x) : v2a_f2bf(x);
}
// `packs` (optional, int64 [tensor][6] = {dst, Cin, taps, dst_window, dst_16bit_twin, twin is fp16}): the updated parameter also goes, re-laid, into the conv operand
// packs the next forward reads -- dst[co][tap][ci] (the forward pack of csrc/igemm.hip pack_weights_multi_kernel, mode 0; plain copies
// are Cin = numel, taps = 1) and, for the RGB stem, dst_window[co][kh][kw'][ci'] (mode 2) -- instead of a pack launch that reads every
// parameter again (0.36 ms of launches in the policy step's serial tail).  (co, ci, tap) of the thread's first element by one
// decomposition, then carried forward in steps of 256 elements.
__global__ __launch_bounds__(256) void mt_adamw_ema_kernel(const int64_t* table, const int* chunks, const OptState* st, int zero_grad,
                                                           const int64_t* packs) {
    const int t = chunks[2 * blockIdx.x], start = chunks[2 * blockIdx.x + 1];
    float* pk = packs ? reinterpret_cast<float*>(packs[t * 6 + 0]) : nullptr;
    float* pw = packs ? reinterpret_cast<float*>(packs[t * 6 + 3]) : nullptr;
    unsigned short* pkh = packs ? reinterpret_cast<unsigned short*>(packs[t * 6 + 4]) : nullptr;      // 16-bit twin of the forward pack (16-bit MFMA modes)
    const int pf16 = packs ? (int)packs[t * 6 + 5] : 0;
    const int pCin = packs ? (int)packs[t * 6 + 1] : 1, ptaps = packs ? (int)packs[t * 6 + 2] : 1;
    int kco = 0, kci = 0, ktap = 0, kwn = 1;
    const int step_tap = 256 % ptaps, step_ci = 256 / ptaps;
    if (pk || pw || pkh) {
        const unsigned idx0 = (unsigned)start + threadIdx.x;    // torch layout [co][ci][tap]
        const unsigned q = idx0 / (unsigned)ptaps;
        ktap = (int)(idx0 - q * (unsigned)ptaps);
        kco = (int)(q / (unsigned)pCin);
        kci = (int)(q - (unsigned)kco * (unsigned)pCin);
        while (kwn * kwn < ptaps) ++kwn;
    }
    float* p = reinterpret_cast<float*>(table[t * 6 + 0]) + start;
    float* g = reinterpret_cast<float*>(table[t * 6 + 1]) + start;
    float* m = reinterpret_cast<float*>(table[t * 6 + 2]) + start;
    float* v = reinterpret_cast<float*>(table[t * 6 + 3]) + start;
    float* e = table[t * 6 + 4] ? reinterpret_cast<float*>(table[t * 6 + 4]) + start : nullptr;
    const long long n = table[t * 6 + 5];
    const int cnt = (int)min((long long)MT_CHUNK, n - start);
    const float clip = st->clip_coef, b1 = (float)st->b1, b2 = (float)st->b2, eps = (float)st->eps;
    const float step_size = st->step_size, isb2 = st->inv_sqrt_bc2, dmul = st->decay_mul, dec = st->ema_decay;
    const int mode = st->ema_mode;
    if (st->skip) {                                     // non-finite gradients: no parameter / moment update; EMA and zero_grad still run
        for (int i = threadIdx.x; i < cnt; i += 256) {
            if (zero_grad) g[i] = 0.f;
            if (e && mode) {
                const float pv = p[i];
                float ev = (mode & 1) ? pv : e[i];
                if (mode & 2) { const float d = (ev - pv) * (1.0f - dec); ev = ev - d; }
                e[i] = ev;
            }
        }
        return;
    }
    // filters with more than one tap and a forward pack (the k = 5 Conv1d and 3 x 3 Conv2d weights: 85 M of the 87 M parameters): the
    // updated values of a quarter chunk are parked in LDS and leave for the pack in DESTINATION order -- thread = filter row (co, ci),
    // tap t of 64 consecutive rows = 64 consecutive floats of the pack.  (The element-order stores of the loop further down put a
    // wave's 64 values on `taps` lines in 52-B / 28-B pieces and cost the kernel 0.12 ms; a row-per-thread loop over all five arrays
    // fixed the stores and lost more on its strided loads.)  LDS reads at stride `taps` floats: conflict-free for odd tap counts.
    if ((pk || pkh) && !pw && ptaps > 1) {
        __shared__ float park[MT_CHUNK / 4];
        constexpr int Q = MT_CHUNK / 4;
        for (int base = 0; base < cnt; base += Q) {
            const int qn = min(Q, cnt - base);
            for (int i = base + threadIdx.x; i < base + qn; i += 256) {
                const float gr = g[i] * clip;
                float pv = p[i] * dmul;
                float mv = m[i];
                mv = mv + (gr - mv) * (1.0f - b1);
                const float vv = v[i] * b2 + (1.0f - b2) * gr * gr;
                const float denom = sqrtf(vv) * isb2 + eps;
                pv = pv - step_size * (mv / denom);
                p[i] = pv;
                m[i] = mv;
                v[i] = vv;
                if (zero_grad) g[i] = 0.f;
                if (e && mode) {
                    float ev = (mode & 1) ? pv : e[i];
                    if (mode & 2) { const float d = (ev - pv) * (1.0f - dec); ev = ev - d; }
                    e[i] = ev;
                }
                park[i - base] = pv;
            }
            __syncthreads();
            const int qs = start + base, qe = qs + qn;              // element range of this quarter in the tensor
            const int r_lo = qs / ptaps, r_hi = (qe - 1) / ptaps;
            for (int r = r_lo + threadIdx.x; r <= r_hi; r += 256) {
                const int co = r / pCin, ci = r - co * pCin;
                const int e0 = r * ptaps;
                const size_t d0 = (size_t)co * ptaps * pCin + ci;
                for (int tp = 0; tp < ptaps; ++tp) {
                    const int el = e0 + tp;
                    if (el >= qs && el < qe) {
                        const float x = park[el - qs];
                        if (pk) pk[d0 + (size_t)tp * pCin] = x;
                        if (pkh) opt_store_twin(pkh, d0 + (size_t)tp * pCin, x, pf16, (size_t)n);
                    }
                }
            }
            __syncthreads();
        }
        return;
    }
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const float gr = g[i] * clip;
        float pv = p[i] * dmul;
        float mv = m[i];
        mv = mv + (gr - mv) * (1.0f - b1);              // lerp_(g, 1-b1)
        const float vv = v[i] * b2 + (1.0f - b2) * gr * gr;
        const float denom = sqrtf(vv) * isb2 + eps;
        pv = pv - step_size * (mv / denom);
        p[i] = pv;
        m[i] = mv;
        v[i] = vv;
        if (zero_grad) g[i] = 0.f;
        if (e && mode) {
            float ev = (mode & 1) ? pv : e[i];
            if (mode & 2) { const float d = (ev - pv) * (1.0f - dec); ev = ev - d; }
            e[i] = ev;
        }
        if (pk || pw || pkh) {
            if (pk) pk[((size_t)kco * ptaps + ktap) * pCin + kci] = pv;
            if (pkh) opt_store_twin(pkh, ((size_t)kco * ptaps + ktap) * pCin + kci, pv, pf16, (size_t)n);
            if (pw) {
                const int kh = ktap / kwn, kw = ktap - kh * kwn;
                pw[(((size_t)kco * kwn + kh) * (kwn + 1) + kw) * (pCin + 1) + kci] = pv;
            }
            ktap += step_tap;
            kci += step_ci;
            if (ktap >= ptaps) { ktap -= ptaps; ++kci; }
            while (kci >= pCin) { kci -= pCin; ++kco; }
        }
    }
}

// grads *= 1/world (after an RCCL sum all-reduce) -- folded into the clip by scaling the table's grads in place
__global__ void mt_scale_grads_kernel(const int64_t* table, const int* chunks, float scale) {
    const int t = chunks[2 * blockIdx.x], start = chunks[2 * blockIdx.x + 1];
    float* g = reinterpret_cast<float*>(table[t * 6 + 1]) + start;
    const long long n = table[t * 6 + 5];
    const int cnt = (int)min((long long)MT_CHUNK, n - start);
    for (int i = threadIdx.x; i < cnt; i += 256) g[i] *= scale;
}

extern "C" {

// Gradient-norm partial sums of chunks [first, first + count) ahead of the optimiser step: the ConditionalUnet1D slice (75 % of the
// parameters) is final long before the encoders' -- PolicyTrainer sums it on the weight-gradient stream while the encoder backward runs;
// the caller then passes the SAME (first, count) as presum_first / presum_count to v2a_opt_step_packed (same table, same `partial`, a
// stream ordered after this one), which sums only the remaining chunks.  Stateless: nothing is remembered between the two calls.
int v2a_opt_presum(const int64_t* table_dev, const int* chunks_dev, int first, int count, double* partial_dev, hipStream_t s) {
    if (!table_dev || !chunks_dev || !partial_dev || first < 0 || count <= 0) return V2A_ERR_ARG;
    hipLaunchKernelGGL(mt_sumsq_kernel, dim3(count), dim3(256), 0, s, table_dev, chunks_dev, partial_dev, first);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

int v2a_opt_chunk_elems(void) { return MT_CHUNK; }
size_t v2a_opt_state_bytes(void) { return sizeof(OptState); }

// Fill a HOST OptState image (caller copies it to the device once).
int v2a_opt_state_init(void* host_state, double lr, double b1, double b2, double eps, double wd, double max_norm,
                       double ema_inv_gamma, double ema_power, double ema_min, double ema_beta, int ema_update_after,
                       int ema_update_every) {
    if (!host_state) return V2A_ERR_ARG;
    OptState s = {};
    s.lr = lr; s.b1 = b1; s.b2 = b2; s.eps = eps; s.wd = wd; s.max_norm = max_norm;
    s.ema_inv_gamma = ema_inv_gamma; s.ema_power = ema_power; s.ema_min = ema_min; s.ema_beta = ema_beta;
    s.ema_update_after = ema_update_after; s.ema_update_every = ema_update_every < 1 ? 1 : ema_update_every;
    *reinterpret_cast<OptState*>(host_state) = s;
    return V2A_OK;
}
// read back {grad_norm, clip_coef, step, ema_decay} from a HOST copy of the state
int v2a_opt_state_peek(const void* host_state, float* grad_norm, float* clip_coef, long long* step, float* ema_decay) {
    const OptState* s = reinterpret_cast<const OptState*>(host_state);
    if (grad_norm) *grad_norm = s->grad_norm;
    if (clip_coef) *clip_coef = s->clip_coef;
    if (step) *step = s->step;
    if (ema_decay) *ema_decay = s->ema_decay;
    return V2A_OK;
}

// checkpoint support (lb_online_trainer_v7.py:367-408 saves opt / ema state): read / patch the step counters of a HOST copy
int v2a_opt_state_counters(const void* host_state, long long* step, long long* ema_step, int* ema_initted) {
    const OptState* s = reinterpret_cast<const OptState*>(host_state);
    if (!s) return V2A_ERR_ARG;
    if (step) *step = s->step;
    if (ema_step) *ema_step = s->ema_step;
    if (ema_initted) *ema_initted = s->ema_initted;
    return V2A_OK;
}
// dynamic loss scaling (fp16 policy mode): patch a HOST copy of the state.  init_scale <= 0 switches the scaler off.  growth_tracker:
// steps since the last growth / back-off (0 for a fresh scaler; a resumed checkpoint restores GradScaler's `_growth_tracker` here).
int v2a_opt_state_set_scaler(void* host_state, double init_scale, double growth_factor, double backoff_factor, int growth_interval,
                             int growth_tracker) {
    OptState* s = reinterpret_cast<OptState*>(host_state);
    if (!s) return V2A_ERR_ARG;
    s->scaler_on = init_scale > 0 ? 1 : 0;
    s->loss_scale = init_scale > 0 ? (float)init_scale : 1.0f;
    s->growth_factor = (float)growth_factor; s->backoff_factor = (float)backoff_factor;
    s->growth_interval = growth_interval < 1 ? 1 : growth_interval;
    s->growth_tracker = growth_tracker > 0 ? growth_tracker : 0; s->skip = 0; s->skipped_steps = 0;
    return V2A_OK;
}
// {loss_scale, growth_tracker, skip flag of the last step, skipped steps so far, growth factor, back-off factor, growth interval, on} of a
// HOST copy (every out pointer optional)
int v2a_opt_state_scaler(const void* host_state, float* loss_scale, int* growth_tracker, int* skipped_last, long long* skipped_steps,
                         float* growth_factor, float* backoff_factor, int* growth_interval, int* scaler_on) {
    const OptState* s = reinterpret_cast<const OptState*>(host_state);
    if (!s) return V2A_ERR_ARG;
    if (loss_scale) *loss_scale = s->loss_scale;
    if (growth_tracker) *growth_tracker = s->growth_tracker;
    if (skipped_last) *skipped_last = s->skip;
    if (skipped_steps) *skipped_steps = s->skipped_steps;
    if (growth_factor) *growth_factor = s->growth_factor;
    if (backoff_factor) *backoff_factor = s->backoff_factor;
    if (growth_interval) *growth_interval = s->growth_interval;
    if (scaler_on) *scaler_on = s->scaler_on;
    return V2A_OK;
}
// byte offset of loss_scale inside the device state block (v2a_mse_loss reads the scale from there)
size_t v2a_opt_state_loss_scale_offset(void) { return offsetof(OptState, loss_scale); }
int v2a_opt_state_set_counters(void* host_state, long long step, long long ema_step, int ema_initted, double lr) {
    OptState* s = reinterpret_cast<OptState*>(host_state);
    if (!s || step < 0 || ema_step < 0) return V2A_ERR_ARG;
    s->step = step;
    s->ema_step = ema_step;
    s->ema_initted = ema_initted ? 1 : 0;
    if (lr > 0) s->lr = lr;
    return V2A_OK;
}

// One optimiser step over all tensors.  partial: nchunks doubles of scratch.
// packs_dev: optional [tensors][6] int64 table {dst forward pack or 0, Cin, taps, dst channel-window pack or 0, dst 16-bit twin of the
// forward pack or 0, 1 when that twin is IEEE fp16 (0: bf16)} -- the update kernel then
// also writes the re-laid conv operands (see mt_adamw_ema_kernel); null: parameters only.
// presum_first / presum_count: chunks [first, first + count) of `partial` already hold this step's sums of squares (v2a_opt_presum with
// the same range, ordered before this call); 0 / 0: everything is summed here.
int v2a_opt_step_packed(const int64_t* table_dev, const int* chunks_dev, int nchunks, void* state_dev, double* partial_dev,
                        int zero_grad, const int64_t* packs_dev, int presum_first, int presum_count, hipStream_t s) {
    if (!table_dev || !chunks_dev || !state_dev || !partial_dev || nchunks <= 0) return V2A_ERR_ARG;
    if (presum_first < 0 || presum_count < 0 || presum_first + presum_count > nchunks) return V2A_ERR_ARG;
    const int pf = presum_count > 0 ? presum_first : 0, pc = presum_count;
    if (pf > 0) {
        hipLaunchKernelGGL(mt_sumsq_kernel, dim3(pf), dim3(256), 0, s, table_dev, chunks_dev, partial_dev, 0);
        V2A_CHECK_LAUNCH();
    }
    if (pf + pc < nchunks) {
        hipLaunchKernelGGL(mt_sumsq_kernel, dim3(nchunks - pf - pc), dim3(256), 0, s, table_dev, chunks_dev, partial_dev, pf + pc);
        V2A_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(opt_advance_kernel, dim3(1), dim3(256), 0, s, (OptState*)state_dev, partial_dev, nchunks);
    V2A_CHECK_LAUNCH();
    hipLaunchKernelGGL(mt_adamw_ema_kernel, dim3(nchunks), dim3(256), 0, s, table_dev, chunks_dev, (const OptState*)state_dev, zero_grad,
                       packs_dev);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_opt_step(const int64_t* table_dev, const int* chunks_dev, int nchunks, void* state_dev, double* partial_dev,
                 int zero_grad, hipStream_t s) {
    return v2a_opt_step_packed(table_dev, chunks_dev, nchunks, state_dev, partial_dev, zero_grad, nullptr, 0, 0, s);
}
int v2a_opt_scale_grads(const int64_t* table_dev, const int* chunks_dev, int nchunks, float scale, hipStream_t s) {
    hipLaunchKernelGGL(mt_scale_grads_kernel, dim3(nchunks), dim3(256), 0, s, table_dev, chunks_dev, scale);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

}  // extern "C"
