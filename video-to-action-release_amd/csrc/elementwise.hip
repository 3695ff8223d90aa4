// HBM-bound elementwise / small-reduction kernels of the hot path (gfx950).  All fp32, channels-last.
#include "common.h"

#define GRID_FOR(total) dim3((unsigned)((((size_t)(total) + 255) / 256) > 8192 ? 8192 : (((size_t)(total) + 255) / 256)))

// ----------------------------------------------------------------------------------------- activations / axpy
__global__ void act_fwd_kernel(const float* x, float* y, size_t n, int act) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = act_fwd(x[i], act);
}
__global__ void act_bwd_kernel(const float* x, const float* dy, float* dx, size_t n, int act) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dx[i] = dy[i] * act_bwd(x[i], act);
}
// out = a + alpha * b
__global__ void axpy_kernel(const float* a, const float* b, float* out, float alpha, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = a[i] + alpha * b[i];
}
// x *= *scalar (device scalar: the upstream gradient of the loss, e.g. a GradScaler factor)
__global__ void scale_by_device_scalar_kernel(float* x, size_t n, const float* scalar) {
    const float s = *scalar;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] *= s;
}
// strided 2-D copy: dst[r][dc0 + c] = src[r][sc0 + c]
__global__ void copy2d_kernel(const float* src, float* dst, int rows, int cols, int lds_, int ldd, int accumulate) {
    const size_t total = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        const float v = src[(size_t)r * lds_ + c];
        float* d = dst + (size_t)r * ldd + c;
        *d = accumulate ? (*d + v) : v;
    }
}
// out[c] = sum_r x[r][c]   (bias gradients); one wave per 64 columns, fp64 accumulate across row-chunks
// 16 columns per workgroup, 16 row lanes per column (rows r, r + 16, ... summed in order, then the lanes in order: deterministic);
// fp64 accumulation.  (The 64-column form ran 8 workgroups for a 512-channel bias gradient: 47 us of one-load-at-a-time latency.)
__global__ __launch_bounds__(256) void colsum_kernel(const float* x, float* out, int rows, int cols, int accumulate) {
    __shared__ double sm[16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s0 = 0.0, s1 = 0.0;
    if (c < cols) {
        int r = rl;
        for (; r + 16 < rows; r += 32) {             // two independent loads in flight
            s0 += (double)x[(size_t)r * cols + c];
            s1 += (double)x[(size_t)(r + 16) * cols + c];
        }
        if (r < rows) s0 += (double)x[(size_t)r * cols + c];
    }
    sm[rl][cl] = s0 + s1;
    __syncthreads();
    if (rl == 0 && c < cols) {
        double t = 0.0;
#pragma unroll
        for (int l = 0; l < 16; ++l) t += sm[l][cl];
        out[c] = accumulate ? out[c] + (float)t : (float)t;
    }
}

// ----------------------------------------------------------------------------------------- embeddings
// kind 0: policy SinusoidalPosEmb (reference positional_embedding.py:10-17): [sin | cos], freq_i = exp(-ln(1e4)/(half-1) * i)
// kind 1: video timestep_embedding (reference nn.py:171-189):               [cos | sin], freq_i = exp(-ln(1e4) * i / half)
__global__ void sincos_embed_kernel(const int64_t* t, float* out, int B, int dim, int kind) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, k = i % half;
    const float tv = (float)t[b];
    float f;
    if (kind == 0) f = expf((float)k * -(logf(10000.f) / (float)(half - 1)));
    else f = expf(-logf(10000.f) * (float)k / (float)half);
    const float a = tv * f;
    if (kind == 0) { out[(size_t)b * dim + k] = sinf(a); out[(size_t)b * dim + half + k] = cosf(a); }
    else { out[(size_t)b * dim + k] = cosf(a); out[(size_t)b * dim + half + k] = sinf(a); }
}

// ----------------------------------------------------------------------------------------- policy DDPM glue
// noisy = sqrt(ac[t_b]) * normalise(act) + sqrt(1 - ac[t_b]) * noise   (reference diffusion_unet_image_policy.py:255).  normalise is
// LimitsConstNormalizer.normalize (normalizer.py:139-146), 2*((a-min)/(max-min))-1 evaluated literally, with the per-channel action
// limits of the policy's shape_meta (amin / amax [act_dim]; NULL = the Libero limits -1 / +1).
__global__ void add_noise_kernel(const float* act, const float* noise, const int64_t* t, const float* ac, float* out, int B, int per,
                                 const float* amin, const float* amax, int act_dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * per) return;
    const int b = i / per;
    const float a = ac[t[b]];
    const float lo = amin ? amin[i % act_dim] : -1.0f, hi = amax ? amax[i % act_dim] : 1.0f;
    const float na = 2.0f * ((act[i] - lo) / (hi - lo)) - 1.0f;
    out[i] = sqrtf(a) * na + sqrtf(1.0f - a) * noise[i];
}
// loss = mean((pred - target)^2) ; dpred = 2 (pred - target) / n      (single workgroup: n = B*16*7 is tiny)
// grad_scale (device, optional): the loss gradient is multiplied by *grad_scale -- the dynamic loss scale of the fp16 mode (csrc/optim.hip
// OptState.loss_scale; the reported loss stays unscaled)
__global__ __launch_bounds__(256) void mse_loss_kernel(const float* pred, const float* target, float* loss, float* dpred, int n,
                                                       const float* grad_scale) {
    __shared__ double sm[4];
    double s = 0.0;
    const float gs = grad_scale ? *grad_scale : 1.0f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float d = pred[i] - target[i];
        s += (double)d * d;
        if (dpred) dpred[i] = (2.0f * d / (float)n) * gs;
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)((sm[0] + sm[1] + sm[2] + sm[3]) / (double)n);
}
// ---- video-model training (GoalGaussianDiffusion.forward / p_losses, goal_diffusion.py:674-724) -------------------------------------
// q_sample on the 'b (f c) h w' layout: out = sqrt_acp[t_b] * x0 + sqrt(1-acp)[t_b] * noise, x0 = 2 img - 1 when `normalize`
__global__ void video_qsample_kernel(const float* __restrict__ img, const float* __restrict__ noise, const int64_t* __restrict__ t,
                                     const float* __restrict__ sa_tab, const float* __restrict__ s1_tab, float* __restrict__ out, int B,
                                     size_t per, int normalize) {
    const size_t total = (size_t)B * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        const float x0 = normalize ? img[i] * 2.0f - 1.0f : img[i];
        out[i] = sa_tab[t[b]] * x0 + s1_tab[t[b]] * noise[i];
    }
}
// difference between the model output (channels-last [B,f,HW,ci]) and the objective's target built from img / noise ('b (f c) h w')
__device__ __forceinline__ float video_loss_diff(const float* out_cl, const float* img, const float* noise, size_t base, int e, int HW, int ci,
                                                 float sa, float s1, int objective, int normalize) {
    const int c = e % ci;
    const int hw = (e / ci) % HW;
    const int fr = e / (ci * HW);
    const size_t j = base + ((size_t)(fr * ci + c)) * HW + hw;
    const float x0 = normalize ? img[j] * 2.0f - 1.0f : img[j];
    const float nz = noise[j];
    const float tgt = objective == 0 ? nz : (objective == 1 ? x0 : sa * nz - s1 * x0);
    return out_cl[base + e] - tgt;
}
__global__ __launch_bounds__(256) void video_loss_partial_kernel(const float* __restrict__ out_cl, const float* __restrict__ img,
                                                                 const float* __restrict__ noise, const int64_t* __restrict__ t,
                                                                 const float* __restrict__ sa_tab, const float* __restrict__ s1_tab,
                                                                 double* __restrict__ partial, int f, int HW, int ci, int objective, int l1,
                                                                 int normalize) {
    __shared__ double sm[4];
    const int b = blockIdx.y;
    const int per = f * HW * ci;
    const float sa = sa_tab[t[b]], s1 = s1_tab[t[b]];
    double acc = 0.0;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < per; e += gridDim.x * 256) {
        const float d = video_loss_diff(out_cl, img, noise, (size_t)b * per, e, HW, ci, sa, s1, objective, normalize);
        acc += l1 ? (double)fabsf(d) : (double)d * d;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
// loss = mean_b( loss_weight[t_b] * mean_e(l) )       (fixed summation order: deterministic)
__global__ void video_loss_final_kernel(const double* __restrict__ partial, const int64_t* __restrict__ t, const float* __restrict__ w_tab,
                                        float* __restrict__ loss, int B, int G, double per) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double tot = 0.0;
    for (int b = 0; b < B; ++b) {
        double sb = 0.0;
        for (int g = 0; g < G; ++g) sb += partial[(size_t)b * G + g];
        tot += (double)((float)(sb / per) * w_tab[t[b]]);
    }
    loss[0] = (float)(tot / B);
}
// d loss / d out (channels-last), scaled by the upstream gradient `gscale` (a device scalar: no host synchronisation)
__global__ void video_loss_bwd_kernel(const float* __restrict__ out_cl, const float* __restrict__ img, const float* __restrict__ noise,
                                      const int64_t* __restrict__ t, const float* __restrict__ sa_tab, const float* __restrict__ s1_tab,
                                      const float* __restrict__ w_tab, const float* __restrict__ gscale, float* __restrict__ dout, int B, int f,
                                      int HW, int ci, int objective, int l1, int normalize) {
    const int per = f * HW * ci;
    const size_t total = (size_t)B * per;
    const float g = gscale ? gscale[0] : 1.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        const int e = (int)(i - (size_t)b * per);
        const int64_t tb = t[b];
        const float d = video_loss_diff(out_cl, img, noise, (size_t)b * per, e, HW, ci, sa_tab[tb], s1_tab[tb], objective, normalize);
        const float k = g * w_tab[tb] / ((float)B * (float)per);
        dout[i] = l1 ? k * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : k * 2.0f * d;
    }
}
// y = x * keep/(1-p) with the stateless mask of common.h (the same call on dy is the backward)
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, float p, float inv_keep, unsigned long long seed,
                               unsigned long long stream) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = x[i] * dropout_scale(seed, stream, i, p, inv_keep);
}
// DDPM / DDIM scheduler step on the action trajectory (third-party diffusers algorithm, restated: see oracle/schedulers.py)
// coef = {sqrt_b_t, sqrt_a_t, c0, ct, sigma} (DDPM) or {sqrt_b_t, sqrt_a_t, sqrt_a_prev, dir, 0} (DDIM, mode 1)
__global__ void policy_sched_step_kernel(const float* eps, const float* sample, const float* noise, float* out, int n, float c_sb,
                                         float c_sa, float c0, float c1, float sigma, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x0 = (sample[i] - c_sb * eps[i]) / c_sa;
    x0 = fminf(fmaxf(x0, -1.f), 1.f);
    float v;
    if (mode == 0) {
        v = c0 * x0 + c1 * sample[i];
        if (noise) v += sigma * noise[i];
    } else {
        v = c0 * x0 + c1 * eps[i];
    }
    out[i] = v;
}
// unnormalize actions: clamp to [-1,1] only if ANY element is out of range (normalizer.py:152-157), then (x+1)/2*(max-min)+min with
// the per-channel action limits (NULL = -1 / +1).
__global__ __launch_bounds__(256) void unnormalize_action_kernel(const float* x, float* out, int n, const float* amin, const float* amax,
                                                                 int act_dim) {
    __shared__ int any;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    int f = 0;
    for (int i = threadIdx.x; i < n; i += 256) f |= (x[i] > 1.f || x[i] < -1.f);
    if (f) atomicOr(&any, 1);
    __syncthreads();
    const int clampit = any;
    for (int i = threadIdx.x; i < n; i += 256) {
        float v = x[i];
        if (clampit) v = fminf(fmaxf(v, -1.f), 1.f);
        v = (v + 1.f) / 2.0f;
        const float lo = amin ? amin[i % act_dim] : -1.0f, hi = amax ? amax[i % act_dim] : 1.0f;
        out[i] = v * (hi - lo) + lo;
    }
}

// ----------------------------------------------------------------------------------------- layout converters
// NCHW float (or uint8) image batch -> NHWC float, with the policy image normalisation 2*x-1 when `normalize`
// (reference normalizer.py:139-146 with min 0 / max 1).  u8 sources are divided by 255 first (img_utils.py:27-37).
template <typename T>
__global__ void nchw_to_nhwc_kernel(const T* src, float* dst, int N, int C, int HW, int normalize, float denom) {
    const size_t total = (size_t)N * HW * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const size_t t = i / C;
        const int hw = (int)(t % HW);
        const int n = (int)(t / HW);
        float v = (float)src[((size_t)n * C + c) * HW + hw] / denom;
        if (normalize) v = 2.0f * ((v - 0.0f) / (1.0f - 0.0f)) - 1.0f;
        dst[i] = v;
    }
}
// The same conversion into a zero-bordered 4-channel image [N][H + 2 pad][W + 2 pad][4] (channel 3 = 0): the layout the channel-window
// stem conv (v2a_conv2d_fwd_window_f32) and its weight gradient read.  Only the interior is written; the caller zeroes the buffer once.
template <typename T>
__global__ void nchw_to_nhwc4p_kernel(const T* src, float* dst, int N, int H, int W, int pad, int normalize, float denom) {
    const size_t total = (size_t)N * H * W;
    const int HW = H * W, Wp = W + 2 * pad, Hp = H + 2 * pad;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int hw = (int)(i % HW);
        const int n = (int)(i / HW);
        const int h = hw / W, w = hw - h * W;
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = (float)src[((size_t)n * 3 + c) * HW + hw] / denom;
            if (normalize) v = 2.0f * ((v - 0.0f) / (1.0f - 0.0f)) - 1.0f;
            o[c] = v;
        }
        o[3] = 0.f;
        *reinterpret_cast<f32x4*>(dst + (((size_t)n * Hp + h + pad) * Wp + w + pad) * 4) = o;
    }
}
__global__ void nhwc_to_nchw_kernel(const float* src, float* dst, int N, int C, int HW) {
    const size_t total = (size_t)N * HW * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int hw = (int)(i % HW);
        const size_t t = i / HW;
        const int c = (int)(t % C);
        const int n = (int)(t / C);
        dst[i] = src[((size_t)n * HW + hw) * C + c];
    }
}

// ----------------------------------------------------------------------------------------- video sampler glue
// Unet_Libero input pack (reference unet.py:217-220): img [B, 3f, H, W] ('b (f c) h w'), x_cond [B,3,H,W]
//   -> xin [B, f, H, W, 6] channels-last (3 noisy + 3 cond, cond repeated over frames)
__global__ void video_pack_kernel(const float* img, const float* cond, float* xin, int B, int f, int HW, size_t img_bs, size_t cond_bs, int ci) {
    const int CT = ci + 3;                                  // ci channels per generated frame (3 RGB, 2 flow) + the RGB conditioning image
    const size_t total = (size_t)B * f * HW * CT;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % CT);
        size_t t = i / CT;
        const int hw = (int)(t % HW);
        t /= HW;
        const int fr = (int)(t % f);
        const int b = (int)(t / f);
        float v;
        if (c < ci) v = img[(size_t)b * img_bs + ((size_t)fr * ci + c) * HW + hw];
        else v = cond[(size_t)b * cond_bs + (size_t)(c - ci) * HW + hw];
        xin[i] = v;
    }
}
// One fused denoise step on the sampler state (reference goal_diffusion.py:484-497,549-580,617-634), v-prediction.
//   v: UNet output [B, f, H, W, 3] channels-last (for CFG: v_u = unconditional half, gw > 0)
//   img/out: [B, 3f, H, W]   noise: same layout or null
// mode 0 (ancestral): x0 = clamp(sa*x - s1*v); out = c1*x0 + c2*x + sigma*noise
// mode 1 (DDIM):      x0 = sa*x - s1*v; eps = (ra*x - x0)/rm; out = sqrt(a_next)*x0 + c*eps + sigma*noise
// mode 2 (DDIM last): out = x0
// final=1 additionally applies unnormalize + clamp: out = clamp((out+1)/2, 0, 1)   (:640, :650)
struct DenoiseCoef { float sa, s1, ra, rm, c1, c2, sigma, gw; };
__global__ void video_denoise_kernel(const float* v, const float* v_u, const float* img, const float* noise, float* out,
                                     int B, int f, int HW, DenoiseCoef k, int mode, int final, int ci) {
    const size_t total = (size_t)B * f * ci * HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int hw = (int)(i % HW);
        size_t t = i / HW;
        const int c = (int)(t % ci);
        t /= ci;
        const int fr = (int)(t % f);
        const int b = (int)(t / f);
        const size_t vi = ((((size_t)b * f + fr) * HW) + hw) * ci + c;
        const float x = img[i];
        float x0, eps;
        if (k.gw > 0.f) {
            const float x0c = k.sa * x - k.s1 * v[vi];
            const float x0u = k.sa * x - k.s1 * v_u[vi];
            const float nu = (k.ra * x - x0u) / k.rm;
            const float nc = (k.ra * x - x0c) / k.rm;
            eps = (1.f + k.gw) * nc - k.gw * nu;
            x0 = k.ra * x - k.rm * eps;
        } else {
            x0 = k.sa * x - k.s1 * v[vi];
            eps = (k.ra * x - x0) / k.rm;
        }
        float o;
        if (mode == 0) {
            x0 = fminf(fmaxf(x0, -1.f), 1.f);
            o = k.c1 * x0 + k.c2 * x;
            if (noise) o += k.sigma * noise[i];
        } else if (mode == 1) {
            o = x0 * k.c1 + k.c2 * eps;
            if (noise) o += k.sigma * noise[i];
        } else {
            o = x0;
        }
        if (final) o = fminf(fmaxf((o + 1.f) * 0.5f, 0.f), 1.f);
        out[i] = o;
    }
}

// ----------------------------------------------------------------------------------------- Philox4x32-10 normals
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1], n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
}
// out[i] ~ N(0,1); counter-based: (seed, offset + i/4) so a captured graph can be replayed with a device-side offset.
__global__ void philox_normal_kernel(float* out, size_t n, uint64_t seed, const uint64_t* offset_ptr, uint64_t offset_imm) {
    const uint64_t off = offset_ptr ? *offset_ptr : offset_imm;
    const size_t n4 = (n + 3) / 4;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (size_t)gridDim.x * 256) {
        const uint64_t ctr = off + q;
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
        uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
        for (int r = 0; r < 10; ++r) philox_round(c, k);
        float z[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = ((float)c[2 * h] + 0.5f) * 2.3283064365386963e-10f;
            const float u2 = ((float)c[2 * h + 1] + 0.5f) * 2.3283064365386963e-10f;
            const float rr = sqrtf(-2.0f * logf(u1));
            z[2 * h] = rr * cosf(6.283185307179586f * u2);
            z[2 * h + 1] = rr * sinf(6.283185307179586f * u2);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (q * 4 + j < n) out[q * 4 + j] = z[j];
    }
}
// the same generator with ONE SEED PER ROW of a [rows][row_elems] tensor (row_elems % 4 == 0): element e of row b = the value
// v2a_philox_normal(seeds[b], offset + e / 4) writes -- row b of a batched draw equals the draw of a one-row call with that seed
__global__ void philox_normal_rows_kernel(float* out, int rows, size_t nq_row, const uint64_t* seeds, uint64_t off) {
    const size_t n4 = (size_t)rows * nq_row;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (size_t)gridDim.x * 256) {
        const size_t b = q / nq_row, qq = q - b * nq_row;
        const uint64_t ctr = off + qq, seed = seeds[b];
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
        uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
        for (int r = 0; r < 10; ++r) philox_round(c, k);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = ((float)c[2 * h] + 0.5f) * 2.3283064365386963e-10f;
            const float u2 = ((float)c[2 * h + 1] + 0.5f) * 2.3283064365386963e-10f;
            const float rr = sqrtf(-2.0f * logf(u1));
            out[q * 4 + 2 * h] = rr * cosf(6.283185307179586f * u2);
            out[q * 4 + 2 * h + 1] = rr * sinf(6.283185307179586f * u2);
        }
    }
}
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t ctr, float (&z)[4]) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) philox_round(c, k);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = ((float)c[2 * h] + 0.5f) * 2.3283064365386963e-10f;
        const float u2 = ((float)c[2 * h + 1] + 0.5f) * 2.3283064365386963e-10f;
        const float rr = sqrtf(-2.0f * logf(u1));
        z[2 * h] = rr * cosf(6.283185307179586f * u2);
        z[2 * h + 1] = rr * sinf(6.283185307179586f * u2);
    }
}

// Table-driven denoise step for the whole-loop sampler (goal_diffusion.py:499-559 model_predictions for all three objectives,
// :561-580 p_sample, :617-634 the DDIM update): the coefficients of step s come from row s of a device table and s itself from
// device memory (`state[0]`), so ONE captured hipGraph {UNet forward, this kernel, video_sampler_advance} serves every step of
// every sample() call.  Noise: an explicit tensor (parity tests inject the reference's stream) or drawn in-kernel with Philox
// (`state[1]` = seed, `state[2]` = counter of the initial image; step s uses counters state[2] + (s + 1) * ceil(total / 4) + q, the
// values v2a_philox_normal would write) -- no randn launch, no draw at all when sigma = 0 (DDIM with eta = 0, last ancestral step).
// objective 0 pred_noise, 1 pred_x0, 2 pred_v; CFG (gw > 0): pred_v mixes the two noise estimates (:536-547), the others the outputs.
struct DenoiseRow { float sa, s1, ra, rm, c1, c2, sigma, gw; int mode, final, t, pad; };
__global__ void video_denoise_kernel2(const float* v, const float* v_u, const float* img, const float* noise, float* out, int B, int f,
                                      int HW, int ci, int objective, const DenoiseRow* table, const uint64_t* state, int step_imm,
                                      int use_philox, const uint64_t* row_seeds) {
    const uint64_t step = state ? state[0] : (uint64_t)step_imm;
    const DenoiseRow k = table[step];
    const size_t total = (size_t)B * f * ci * HW;
    const size_t nq = (total + 3) / 4;
    const bool draw = use_philox && !noise && k.sigma != 0.f && k.mode != 2;
    const uint64_t seed = (state && use_philox) ? state[1] : 0ull;
    const uint64_t ctr0 = (state && use_philox) ? state[2] + (step + 1) * (uint64_t)nq : 0ull;
    // row_seeds (one Philox seed per sample; the row length is a multiple of 4): row b draws what a ONE-row call with seed row_seeds[b]
    // draws -- counters state[2] + (step + 1) * nq_row + q_row -- so a batched sample() reproduces its rows sampled one at a time
    const size_t nq_row = ((size_t)f * ci * HW) >> 2;
    const uint64_t ctr0r = (state && use_philox) ? state[2] + (step + 1) * (uint64_t)nq_row : 0ull;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (size_t)gridDim.x * 256) {
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (draw) {
            if (row_seeds) { const size_t b = q / nq_row; philox_normal4(row_seeds[b], ctr0r + (q - b * nq_row), z); }
            else philox_normal4(seed, ctr0 + q, z);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const size_t i = q * 4 + e;
            if (i >= total) break;
            const int hw = (int)(i % HW);
            size_t t = i / HW;
            const int c = (int)(t % ci);
            t /= ci;
            const int fr = (int)(t % f);
            const int b = (int)(t / f);
            const size_t vi = ((((size_t)b * f + fr) * HW) + hw) * ci + c;
            const float x = img[i];
            float x0, eps;
            if (objective == 2) {
                if (k.gw > 0.f && v_u) {
                    const float x0c = k.sa * x - k.s1 * v[vi];
                    const float x0u = k.sa * x - k.s1 * v_u[vi];
                    const float nu = (k.ra * x - x0u) / k.rm;
                    const float nc = (k.ra * x - x0c) / k.rm;
                    eps = (1.f + k.gw) * nc - k.gw * nu;
                    x0 = k.ra * x - k.rm * eps;
                } else {
                    x0 = k.sa * x - k.s1 * v[vi];
                    eps = (k.ra * x - x0) / k.rm;
                }
            } else {
                float mo = v[vi];
                if (k.gw > 0.f && v_u) mo = (1.f + k.gw) * mo - k.gw * v_u[vi];
                if (objective == 0) { eps = mo; x0 = k.ra * x - k.rm * eps; }
                else { x0 = mo; eps = (k.ra * x - x0) / k.rm; }
            }
            float o;
            const float nz = noise ? noise[i] : z[e];
            if (k.mode == 0) {
                x0 = fminf(fmaxf(x0, -1.f), 1.f);
                o = k.c1 * x0 + k.c2 * x;
                if (noise || draw) o += k.sigma * nz;
            } else if (k.mode == 1) {
                o = x0 * k.c1 + k.c2 * eps;
                if (noise || draw) o += k.sigma * nz;
            } else {
                o = x0;
            }
            if (k.final) o = fminf(fmaxf((o + 1.f) * 0.5f, 0.f), 1.f);
            out[i] = o;
        }
    }
}
// state[0] += 1; tt[0..B) = time step of the new row (what the next UNet forward embeds)
__global__ void video_sampler_advance_kernel(uint64_t* state, const DenoiseRow* table, int64_t* tt, int B, int nrows) {
    const uint64_t s = state[0] + 1;
    const int row = s < (uint64_t)nrows ? (int)s : nrows - 1;
    const int t = table[row].t;
    __syncthreads();
    if (threadIdx.x == 0) state[0] = s;
    for (int b = threadIdx.x; b < B; b += blockDim.x) tt[b] = t;
}

// out_i[b][n] = bias_i[n] + sum_k x[b][k] * W_i[n][k] for up to 32 weight matrices sharing the input x [B][K] (B <= 16, K % 256 == 0,
// K <= 1024): the 27 `emb_layers` Linears of the video UNet's ResBlocks (guided_diffusion/unet.py:204-210,248-257) all read the same
// SiLU(emb) -- one launch instead of 27 latency-bound GEMMs.  One wave per output column: the weight row is read once (coalesced),
// x sits in LDS.  Fixed summation order (lane-local chunks, then a shuffle tree).
#define EMB_MAX 32
struct EmbMultiArgs { int n, B, K, total_cols; int col_end[EMB_MAX]; const float* w[EMB_MAX]; const float* bias[EMB_MAX]; float* out[EMB_MAX]; };
__global__ __launch_bounds__(256) void emb_linear_multi_kernel(const float* __restrict__ x, const EmbMultiArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xs[];           // [B][K]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < a.B * a.K; i += 256) xs[i] = x[i];
    __syncthreads();
    const int per = a.K / 64;                                             // floats per lane: 4, 8, 12 or 16
    for (int col = blockIdx.x * 4 + wid; col < a.total_cols; col += gridDim.x * 4) {
        int i = 0;
        while (i + 1 < a.n && col >= a.col_end[i]) ++i;
        const int n = col - (i ? a.col_end[i - 1] : 0);
        const int cout = a.col_end[i] - (i ? a.col_end[i - 1] : 0);
        const float* wr = a.w[i] + (size_t)n * a.K;
        float wv[16];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j * 4 < per) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(wr + j * 256 + lane * 4);
                wv[4 * j] = t[0]; wv[4 * j + 1] = t[1]; wv[4 * j + 2] = t[2]; wv[4 * j + 3] = t[3];
            }
        const float bv = a.bias[i] ? a.bias[i][n] : 0.f;
        for (int b = 0; b < a.B; ++b) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j * 4 < per) {
                    const float* xp = xs + b * a.K + j * 256 + lane * 4;
                    acc += wv[4 * j] * xp[0] + wv[4 * j + 1] * xp[1] + wv[4 * j + 2] * xp[2] + wv[4 * j + 3] * xp[3];
                }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            if (lane == 0) a.out[i][(size_t)b * cout + n] = acc + bv;
        }
    }
}

__global__ void philox_randint_kernel(int64_t* out, int n, int high, uint64_t seed, const uint64_t* offset_ptr, uint64_t offset_imm) {
    const uint64_t off = offset_ptr ? *offset_ptr : offset_imm;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t ctr = off + (uint64_t)i;
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 1u, 0u};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) philox_round(c, k);
    out[i] = (int64_t)(((uint64_t)c[0] * (uint64_t)high) >> 32);
}
__global__ void advance_counter_kernel(uint64_t* ctr, uint64_t inc) { *ctr += inc; }

// backward of a folded nearest x2 upsample: dx[n][h][w][c] = sum of the 2x2 block of du[n][2h..2h+1][2w..2w+1][c]  (C % 4 == 0)
__global__ void sumpool2x2_kernel(const float* __restrict__ du, float* __restrict__ dx, int N, int H, int W, int C4) {
    const size_t total = (size_t)N * H * W * C4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4);
        size_t t = idx / C4;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        const f32x4* src = reinterpret_cast<const f32x4*>(du) + (((size_t)n * 2 * H + 2 * h) * 2 * W + 2 * w) * C4 + c;
        const size_t row = (size_t)2 * W * C4;
        reinterpret_cast<f32x4*>(dx)[idx] = src[0] + src[C4] + src[row] + src[row + C4];
    }
}

// out[b][c] (+)= sum over the `rows` rows of sample b of x[b][r][c]   (gradient of a per-sample broadcast row vector).
// Two passes: grid (C/64, B, slabs) -> fp64 partials [B][slabs][C]; then one thread per (b, c) adds the slabs in order (deterministic).
__global__ __launch_bounds__(256) void colsum_batched_kernel(const float* __restrict__ x, double* __restrict__ partial, int rows, int C,
                                                             int rows_per_slab) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;                       // 4 row partitions per column block
    __shared__ double red[4][64];
    double acc = 0.0;                                        // signed terms that mostly cancel: accumulate in fp64
    const int r0 = blockIdx.z * rows_per_slab;
    const int r1 = min(rows, r0 + rows_per_slab);
    if (c < C) {
        const float* p = x + (size_t)b * rows * C + c;
        int r = r0 + part;
        for (; r + 12 < r1; r += 16) {                       // 4 independent loads in flight
            const float v0 = p[(size_t)r * C], v1 = p[(size_t)(r + 4) * C], v2 = p[(size_t)(r + 8) * C], v3 = p[(size_t)(r + 12) * C];
            acc += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
        }
        for (; r < r1; r += 4) acc += (double)p[(size_t)r * C];
    }
    red[part][threadIdx.x & 63] = acc;
    __syncthreads();
    if (part == 0 && c < C)
        partial[((size_t)b * gridDim.z + blockIdx.z) * C + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void colsum_batched_final_kernel(const double* __restrict__ partial, float* __restrict__ out, int B, int slabs,
                                                                   int C, int accumulate) {
    // one wave per (b, 64-column block): the four waves of a workgroup split the slabs, lanes own columns (coalesced 512-B rows)
    __shared__ double red[4][64];
    const int cb = (C + 63) / 64;
    const int b = blockIdx.x / cb, c = (blockIdx.x % cb) * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    double t = 0.0;
    if (c < C)
        for (int sl = w; sl < slabs; sl += 4) t += partial[((size_t)b * slabs + sl) * C + c];
    red[w][threadIdx.x & 63] = t;
    __syncthreads();
    if (w == 0 && c < C) {
        const float v = (float)(red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
        out[(size_t)b * C + c] = accumulate ? out[(size_t)b * C + c] + v : v;
    }
}
static int colsum_batched_slabs(int B, int rows, int C) {
    const int cb = (C + 63) / 64;
    int slabs = 2048 / (cb * B);                             // ~2048 workgroups in flight (8 per CU)
    if (slabs > rows / 64) slabs = rows / 64;                // at least 64 rows per slab
    return slabs < 1 ? 1 : slabs;
}

extern "C" {

int v2a_act_fwd(const float* x, float* y, size_t n, int act, hipStream_t s) {
    hipLaunchKernelGGL(act_fwd_kernel, GRID_FOR(n), dim3(256), 0, s, x, y, n, act);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_act_bwd(const float* x, const float* dy, float* dx, size_t n, int act, hipStream_t s) {
    hipLaunchKernelGGL(act_bwd_kernel, GRID_FOR(n), dim3(256), 0, s, x, dy, dx, n, act);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_axpy(const float* a, const float* b, float* out, float alpha, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(axpy_kernel, GRID_FOR(n), dim3(256), 0, s, a, b, out, alpha, n);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_scale_by_device_scalar(float* x, size_t n, const float* scalar, hipStream_t s) {
    hipLaunchKernelGGL(scale_by_device_scalar_kernel, GRID_FOR(n), dim3(256), 0, s, x, n, scalar);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_copy2d(const float* src, float* dst, int rows, int cols, int ld_src, int ld_dst, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(copy2d_kernel, GRID_FOR((size_t)rows * cols), dim3(256), 0, s, src, dst, rows, cols, ld_src, ld_dst, accumulate);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_colsum(const float* x, float* out, int rows, int cols, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 15) / 16), dim3(256), 0, s, x, out, rows, cols, accumulate);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_sincos_embed(const int64_t* t, float* out, int B, int dim, int kind, hipStream_t s) {
    const int n = B * (dim / 2);
    hipLaunchKernelGGL(sincos_embed_kernel, dim3((n + 255) / 256), dim3(256), 0, s, t, out, B, dim, kind);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_add_noise(const float* act, const float* noise, const int64_t* t, const float* alphas_cumprod, float* out, int B, int per,
                  const float* act_min, const float* act_max, int act_dim, hipStream_t s) {
    if ((act_min == nullptr) != (act_max == nullptr) || (act_min && (act_dim <= 0 || per % act_dim))) return V2A_ERR_ARG;
    hipLaunchKernelGGL(add_noise_kernel, dim3((B * per + 255) / 256), dim3(256), 0, s, act, noise, t, alphas_cumprod, out, B, per,
                       act_min, act_max, act_dim > 0 ? act_dim : 1);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_mse_loss_scaled(const float* pred, const float* target, float* loss, float* dpred, int n, const float* grad_scale_dev, hipStream_t s) {
    hipLaunchKernelGGL(mse_loss_kernel, dim3(1), dim3(256), 0, s, pred, target, loss, dpred, n, grad_scale_dev);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_mse_loss(const float* pred, const float* target, float* loss, float* dpred, int n, hipStream_t s) {
    return v2a_mse_loss_scaled(pred, target, loss, dpred, n, nullptr, s);
}
int v2a_policy_sched_step(const float* eps, const float* sample, const float* noise, float* out, int n, float c_sb, float c_sa,
                          float c0, float c1, float sigma, int mode, hipStream_t s) {
    hipLaunchKernelGGL(policy_sched_step_kernel, dim3((n + 255) / 256), dim3(256), 0, s, eps, sample, noise, out, n, c_sb, c_sa, c0, c1, sigma, mode);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_unnormalize_action(const float* x, float* out, int n, const float* act_min, const float* act_max, int act_dim, hipStream_t s) {
    if ((act_min == nullptr) != (act_max == nullptr) || (act_min && (act_dim <= 0 || n % act_dim))) return V2A_ERR_ARG;
    hipLaunchKernelGGL(unnormalize_action_kernel, dim3(1), dim3(256), 0, s, x, out, n, act_min, act_max, act_dim > 0 ? act_dim : 1);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_nchw_to_nhwc_f32(const float* src, float* dst, int N, int C, int HW, int normalize, hipStream_t s) {
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<float>), GRID_FOR((size_t)N * C * HW), dim3(256), 0, s, src, dst, N, C, HW, normalize, 1.0f);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_nchw_to_nhwc_u8(const uint8_t* src, float* dst, int N, int C, int HW, int normalize, hipStream_t s) {
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<uint8_t>), GRID_FOR((size_t)N * C * HW), dim3(256), 0, s, src, dst, N, C, HW, normalize, 255.0f);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_nchw_to_nhwc4p(const void* src, int is_u8, float* dst, int N, int H, int W, int pad, int normalize, hipStream_t s) {
    if (!src || !dst || N <= 0 || H <= 0 || W <= 0 || pad < 0 || (((uintptr_t)dst) & 15)) return V2A_ERR_ARG;
    if (is_u8) hipLaunchKernelGGL((nchw_to_nhwc4p_kernel<uint8_t>), GRID_FOR((size_t)N * H * W), dim3(256), 0, s, (const uint8_t*)src, dst, N, H, W, pad, normalize, 255.0f);
    else hipLaunchKernelGGL((nchw_to_nhwc4p_kernel<float>), GRID_FOR((size_t)N * H * W), dim3(256), 0, s, (const float*)src, dst, N, H, W, pad, normalize, 1.0f);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_nhwc_to_nchw_f32(const float* src, float* dst, int N, int C, int HW, hipStream_t s) {
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, GRID_FOR((size_t)N * C * HW), dim3(256), 0, s, src, dst, N, C, HW);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
// img_bstride / cond_bstride: elements between consecutive batch items (lets both be slices of one [B,(f+1)*3,H,W] tensor)
int v2a_video_pack(const float* img, const float* cond, float* xin, int B, int f, int HW, size_t img_bstride, size_t cond_bstride, int frame_ch, hipStream_t s) {
    hipLaunchKernelGGL(video_pack_kernel, GRID_FOR((size_t)B * f * HW * (frame_ch + 3)), dim3(256), 0, s, img, cond, xin, B, f, HW, img_bstride, cond_bstride, frame_ch);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_video_denoise_step(const float* v, const float* v_uncond, const float* img, const float* noise, float* out, int B, int f, int HW,
                           float sa, float s1, float ra, float rm, float c1, float c2, float sigma, float gw, int mode, int final,
                           int frame_ch, hipStream_t s) {
    DenoiseCoef k = {sa, s1, ra, rm, c1, c2, sigma, gw};
    hipLaunchKernelGGL(video_denoise_kernel, GRID_FOR((size_t)B * f * frame_ch * HW), dim3(256), 0, s, v, v_uncond, img, noise, out, B, f, HW, k, mode, final, frame_ch);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_video_qsample(const float* img, const float* noise, const int64_t* t, const float* sqrt_acp, const float* sqrt_1m_acp, float* out, int B,
                      size_t per, int normalize, hipStream_t s) {
    if (!img || !noise || !t || !sqrt_acp || !sqrt_1m_acp || !out || B <= 0) return V2A_ERR_ARG;
    hipLaunchKernelGGL(video_qsample_kernel, GRID_FOR((size_t)B * per), dim3(256), 0, s, img, noise, t, sqrt_acp, sqrt_1m_acp, out, B, per, normalize);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
size_t v2a_video_loss_workspace_bytes(int B) { return (size_t)B * 64 * sizeof(double); }
int v2a_video_loss_fwd(const float* out_cl, const float* img, const float* noise, const int64_t* t, const float* sqrt_acp,
                       const float* sqrt_1m_acp, const float* loss_weight, float* loss, int B, int f, int HW, int ci, int objective, int l1,
                       int normalize, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!out_cl || !img || !noise || !t || !loss || !ws || B <= 0 || objective < 0 || objective > 2) return V2A_ERR_ARG;
    if (ws_bytes < v2a_video_loss_workspace_bytes(B)) return V2A_ERR_ARG;
    const int G = 64;
    hipLaunchKernelGGL(video_loss_partial_kernel, dim3(G, B), dim3(256), 0, s, out_cl, img, noise, t, sqrt_acp, sqrt_1m_acp, (double*)ws, f, HW,
                       ci, objective, l1, normalize);
    hipLaunchKernelGGL(video_loss_final_kernel, dim3(1), dim3(64), 0, s, (const double*)ws, t, loss_weight, loss, B, G, (double)f * HW * ci);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_video_loss_bwd(const float* out_cl, const float* img, const float* noise, const int64_t* t, const float* sqrt_acp,
                       const float* sqrt_1m_acp, const float* loss_weight, const float* gscale, float* dout, int B, int f, int HW, int ci,
                       int objective, int l1, int normalize, hipStream_t s) {
    if (!out_cl || !img || !noise || !t || !dout || B <= 0 || objective < 0 || objective > 2) return V2A_ERR_ARG;
    hipLaunchKernelGGL(video_loss_bwd_kernel, GRID_FOR((size_t)B * f * HW * ci), dim3(256), 0, s, out_cl, img, noise, t, sqrt_acp, sqrt_1m_acp,
                       loss_weight, gscale, dout, B, f, HW, ci, objective, l1, normalize);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_dropout(const float* x, float* y, size_t n, float p, uint64_t seed, uint64_t stream_id, hipStream_t s) {
    if (!x || !y || !(p >= 0.f && p < 1.f)) return V2A_ERR_ARG;
    hipLaunchKernelGGL(dropout_kernel, GRID_FOR(n), dim3(256), 0, s, x, y, n, p, 1.0f / (1.0f - p), (unsigned long long)seed,
                       (unsigned long long)stream_id);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_video_denoise_row_bytes(void) { return (int)sizeof(DenoiseRow); }
// table_dev: [rows] of {sa, s1, ra, rm, c1, c2, sigma, gw (floats), mode, final, t, pad (int32)}; state_dev: uint64[3] = {step, Philox
// seed, Philox counter of the initial image} or null (then row `step_imm`, no in-kernel noise).  `out` may alias `img`.
int v2a_video_denoise_step2(const float* v, const float* v_uncond, const float* img, const float* noise, float* out, int B, int f, int HW,
                            int frame_ch, int objective, const void* table_dev, const uint64_t* state_dev, int step_imm, int use_philox,
                            const uint64_t* row_seeds_dev, hipStream_t s) {
    if (!v || !img || !out || !table_dev || objective < 0 || objective > 2) return V2A_ERR_ARG;
    if (row_seeds_dev && (((size_t)f * frame_ch * HW) & 3)) return V2A_ERR_ARG;        // per-row seeds: whole 4-element draws per row
    // use_philox: bit 0 = draw the noise in the kernel; bit 1 = the device table holds guided rows (gw > 0) -- the host cannot read
    // the table, so the caller says so, and the unconditional half is then mandatory (the kernel would dereference null)
    if ((use_philox & 2) && !v_uncond) return V2A_ERR_ARG;
    const size_t total = (size_t)B * f * frame_ch * HW;
    hipLaunchKernelGGL(video_denoise_kernel2, GRID_FOR((total + 3) / 4), dim3(256), 0, s, v, (use_philox & 2) ? v_uncond : nullptr, img, noise,
                       out, B, f, HW, frame_ch, objective, (const DenoiseRow*)table_dev, state_dev, step_imm, use_philox & 1, row_seeds_dev);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_video_sampler_advance(uint64_t* state_dev, const void* table_dev, int64_t* tt, int B, int nrows, hipStream_t s) {
    if (!state_dev || !table_dev || !tt || B < 1 || nrows < 1) return V2A_ERR_ARG;
    hipLaunchKernelGGL(video_sampler_advance_kernel, dim3(1), dim3(64), 0, s, state_dev, (const DenoiseRow*)table_dev, tt, B, nrows);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_emb_linear_multi_max(void) { return EMB_MAX; }
// HOST arrays w / bias / out / couts of n <= 32 entries (device pointers inside); x [B][K] fp32
int v2a_emb_linear_multi(const float* x, int B, int K, const float* const* w, const float* const* bias, float* const* out, const int* couts,
                         int n, hipStream_t s) {
    if (!x || !w || !bias || !out || !couts || n < 1 || n > EMB_MAX || B < 1 || B > 16 || K % 256 || K < 256 || K > 1024) return V2A_ERR_ARG;
    if (((uintptr_t)x & 15) != 0) return V2A_ERR_ARG;
    EmbMultiArgs a;
    __builtin_memset(&a, 0, sizeof(a));
    a.n = n; a.B = B; a.K = K;
    int tot = 0;
    for (int i = 0; i < n; ++i) {
        if (!w[i] || !out[i] || couts[i] < 1) return V2A_ERR_ARG;
        if (((uintptr_t)w[i] & 15) != 0) return V2A_ERR_ARG;         // the kernel reads weight rows with 16-B loads (K % 256 == 0 keeps rows aligned)
        a.w[i] = w[i]; a.bias[i] = bias[i]; a.out[i] = out[i];
        tot += couts[i];
        a.col_end[i] = tot;
    }
    for (int i = n; i < EMB_MAX; ++i) a.col_end[i] = tot;
    a.total_cols = tot;
    int g = (tot + 3) / 4;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(emb_linear_multi_kernel, dim3(g), dim3(256), (size_t)B * K * sizeof(float), s, x, a);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_philox_normal(float* out, size_t n, uint64_t seed, const uint64_t* offset_dev, uint64_t offset_imm, hipStream_t s) {
    hipLaunchKernelGGL(philox_normal_kernel, GRID_FOR((n + 3) / 4), dim3(256), 0, s, out, n, seed, offset_dev, offset_imm);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_philox_normal_rows(float* out, int rows, size_t row_elems, const uint64_t* seeds_dev, uint64_t offset_imm, hipStream_t s) {
    if (!out || !seeds_dev || rows < 1 || row_elems == 0 || (row_elems & 3)) return V2A_ERR_ARG;
    hipLaunchKernelGGL(philox_normal_rows_kernel, GRID_FOR((size_t)rows * (row_elems / 4)), dim3(256), 0, s, out, rows, row_elems / 4, seeds_dev,
                       offset_imm);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_philox_randint(int64_t* out, int n, int high, uint64_t seed, const uint64_t* offset_dev, uint64_t offset_imm, hipStream_t s) {
    hipLaunchKernelGGL(philox_randint_kernel, dim3((n + 255) / 256), dim3(256), 0, s, out, n, high, seed, offset_dev, offset_imm);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
// measurement aid (tools/, bench.py phase timeline): one lane stores the constant-rate wall clock (100 MHz on gfx950) when the stream
// reaches this point -- the only way to see the REAL overlap of the step's parallel branches (rocprofv3's kernel trace serialises
// the hardware queues: tools/probes/graph_branch_probe.py)
__global__ void timestamp_kernel(unsigned long long* dst) { *dst = wall_clock64(); }
int v2a_debug_timestamp(uint64_t* dst, hipStream_t s) {
    if (!dst) return V2A_ERR_ARG;
    hipLaunchKernelGGL(timestamp_kernel, dim3(1), dim3(1), 0, s, (unsigned long long*)dst);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_advance_counter(uint64_t* ctr, uint64_t inc, hipStream_t s) {
    hipLaunchKernelGGL(advance_counter_kernel, dim3(1), dim3(1), 0, s, ctr, inc);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

int v2a_sumpool2x2(const float* du, float* dx, int N, int H, int W, int C, hipStream_t s) {
    if (!du || !dx || C % 4) return V2A_ERR_ARG;
    const size_t total = (size_t)N * H * W * (C / 4);
    int g = (int)((total + 255) / 256);
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(sumpool2x2_kernel, dim3(g), dim3(256), 0, s, du, dx, N, H, W, C / 4);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

size_t v2a_colsum_batched_workspace_bytes(int B, int rows, int C) {
    return (size_t)B * colsum_batched_slabs(B, rows, C) * C * sizeof(double);
}
int v2a_colsum_batched(const float* x, float* out, int B, int rows, int C, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!x || !out || !ws || B <= 0 || rows <= 0 || C <= 0) return V2A_ERR_ARG;
    if (ws_bytes < v2a_colsum_batched_workspace_bytes(B, rows, C)) return V2A_ERR_ARG;
    const int slabs = colsum_batched_slabs(B, rows, C);
    const int rps = (rows + slabs - 1) / slabs;
    hipLaunchKernelGGL(colsum_batched_kernel, dim3((C + 63) / 64, B, slabs), dim3(256), 0, s, x, (double*)ws, rows, C, rps);
    hipLaunchKernelGGL(colsum_batched_final_kernel, dim3(B * ((C + 63) / 64)), dim3(256), 0, s, (const double*)ws, out, B, slabs, C, accumulate);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

}  // extern "C"
