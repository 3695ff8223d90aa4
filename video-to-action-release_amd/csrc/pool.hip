// MaxPool2d(3, stride 2, pad 1) and SpatialSoftmax of the ResNet-18 image encoders, channels-last fp32.
// replaces torchvision resnet18.maxpool (vision_nets.py:29-39) and SpatialSoftmax.forward (base_nets.py:234-285).
#include "common.h"

// x [N, H, W, C] -> y [N, OH, OW, C]; idx (int8) = winning tap 0..8 (first max in scan order, as torch CPU).
__global__ void maxpool3x3s2_fwd_kernel(const float* x, float* y, int8_t* idx, int N, int H, int W, int C, int OH, int OW) {
    const size_t total = (size_t)N * OH * OW * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t t = i / C;
        const int ow = (int)(t % OW);
        t /= OW;
        const int oh = (int)(t % OH);
        const int n = (int)(t / OH);
        float best = -INFINITY;
        int bi = 0;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
                if (ih >= 0 && ih < H && iw >= 0 && iw < W) {
                    const float v = x[(((size_t)n * H + ih) * W + iw) * C + c];
                    if (v > best || (v != v)) { best = v; bi = kh * 3 + kw; }
                }
            }
        y[i] = best;
        idx[i] = (int8_t)bi;
    }
}
// gather form: dx[n, ih, iw, c] = sum over the <= 4 windows covering (ih, iw) whose argmax is this pixel
__global__ void maxpool3x3s2_bwd_kernel(const float* dy, const int8_t* idx, float* dx, int N, int H, int W, int C, int OH, int OW) {
    const size_t total = (size_t)N * H * W * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t t = i / C;
        const int iw = (int)(t % W);
        t /= W;
        const int ih = (int)(t % H);
        const int n = (int)(t / H);
        float g = 0.f;
        // windows: oh with oh*2-1+kh == ih, kh in 0..2  ->  oh in {(ih+1)/2, (ih+1)/2 - (parity)}
        for (int kh = 0; kh < 3; ++kh) {
            const int num = ih + 1 - kh;
            if (num < 0 || (num & 1)) continue;
            const int oh = num >> 1;
            if (oh >= OH) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int numw = iw + 1 - kw;
                if (numw < 0 || (numw & 1)) continue;
                const int ow = numw >> 1;
                if (ow >= OW) continue;
                const size_t o = (((size_t)n * OH + oh) * OW + ow) * C + c;
                if (idx[o] == kh * 3 + kw) g += dy[o];
            }
        }
        dx[i] = g;
    }
}

// The same two kernels four channels at a time (C % 4 == 0): 16-B loads / stores, one 4-B load of the four tap indices, 32-bit
// index arithmetic (the scalar forms above spend their time in 64-bit divisions: 108 us for the 64 x 64 x 64 x 64 gradient).
__global__ __launch_bounds__(256) void maxpool3x3s2_fwd4_kernel(const float* x, float* y, int8_t* idx, int N, int H, int W, int C4, int OH, int OW) {
    const uint32_t total = (uint32_t)N * OH * OW * C4;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const uint32_t c4 = i % (uint32_t)C4;
        uint32_t t = i / (uint32_t)C4;
        const int ow = (int)(t % (uint32_t)OW);
        t /= (uint32_t)OW;
        const int oh = (int)(t % (uint32_t)OH);
        const int n = (int)(t / (uint32_t)OH);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
                if (ih >= 0 && ih < H && iw >= 0 && iw < W) {
                    const f32x4 v = reinterpret_cast<const f32x4*>(x)[((size_t)(n * H + ih) * W + iw) * C4 + c4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (v[e] > best[e] || (v[e] != v[e])) { best[e] = v[e]; bi[e] = kh * 3 + kw; }
                }
            }
        reinterpret_cast<f32x4*>(y)[i] = best;
        reinterpret_cast<uint32_t*>(idx)[i] = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
    }
}
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd4_kernel(const float* dy, const int8_t* idx, float* dx, int N, int H, int W, int C4, int OH,
                                                                int OW) {
    const uint32_t total = (uint32_t)N * H * W * C4;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const uint32_t c4 = i % (uint32_t)C4;
        uint32_t t = i / (uint32_t)C4;
        const int iw = (int)(t % (uint32_t)W);
        t /= (uint32_t)W;
        const int ih = (int)(t % (uint32_t)H);
        const int n = (int)(t / (uint32_t)H);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int num = ih + 1 - kh;
            if (num < 0 || (num & 1) || (num >> 1) >= OH) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int numw = iw + 1 - kw;
                if (numw < 0 || (numw & 1) || (numw >> 1) >= OW) continue;
                const size_t o = ((size_t)(n * OH + (num >> 1)) * OW + (numw >> 1)) * C4 + c4;
                const uint32_t taps = reinterpret_cast<const uint32_t*>(idx)[o];
                const f32x4 d = reinterpret_cast<const f32x4*>(dy)[o];
                const uint32_t want = (uint32_t)(kh * 3 + kw);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (((taps >> (8 * e)) & 0xffu) == want) g[e] += d[e];
            }
        }
        reinterpret_cast<f32x4*>(dx)[i] = g;
    }
}
// feat [B, P, K] (P = H*W positions, channels-last keypoint logits) -> kp [B, K, 2] = (E[x], E[y]); att saved [B, P, K]
__global__ void spatial_softmax_fwd_kernel(const float* feat, float* kp, float* att, int B, int Hh, int Ww, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * K) return;
    const int b = i / K, k = i % K, P = Hh * Ww;
    const float* f = feat + (size_t)b * P * K + k;
    float m = -INFINITY;
    for (int p = 0; p < P; ++p) m = fmaxf(m, f[(size_t)p * K]);
    float l = 0.f;
    for (int p = 0; p < P; ++p) l += expf(f[(size_t)p * K] - m);
    float ex = 0.f, ey = 0.f;
    for (int p = 0; p < P; ++p) {
        const float a = expf(f[(size_t)p * K] - m) / l;
        att[((size_t)b * P + p) * K + k] = a;
        // np.linspace(-1, 1, n) evaluated in fp64 then cast to fp32 (base_nets.py:206-212)
        const int py = p / Ww, px = p % Ww;
        const float fx = Ww > 1 ? (float)(-1.0 + 2.0 * (double)px / (double)(Ww - 1)) : -1.f;
        const float fy = Hh > 1 ? (float)(-1.0 + 2.0 * (double)py / (double)(Hh - 1)) : -1.f;
        ex += fx * a;
        ey += fy * a;
    }
    kp[(size_t)i * 2] = ex;
    kp[(size_t)i * 2 + 1] = ey;
}
// dfeat[p] = att[p] * ((px - ex) dex + (py - ey) dey)
__global__ void spatial_softmax_bwd_kernel(const float* att, const float* kp, const float* dkp, float* dfeat, int B, int Hh, int Ww, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * K) return;
    const int b = i / K, k = i % K, P = Hh * Ww;
    const float ex = kp[(size_t)i * 2], ey = kp[(size_t)i * 2 + 1];
    const float dex = dkp[(size_t)i * 2], dey = dkp[(size_t)i * 2 + 1];
    for (int p = 0; p < P; ++p) {
        const int py = p / Ww, px = p % Ww;
        const float fx = Ww > 1 ? (float)(-1.0 + 2.0 * (double)px / (double)(Ww - 1)) : -1.f;
        const float fy = Hh > 1 ? (float)(-1.0 + 2.0 * (double)py / (double)(Hh - 1)) : -1.f;
        const size_t o = ((size_t)b * P + p) * K + k;
        dfeat[o] = att[o] * ((fx - ex) * dex + (fy - ey) * dey);
    }
}

extern "C" {
int v2a_maxpool3x3s2_fwd(const float* x, float* y, int8_t* idx, int N, int H, int W, int C, hipStream_t s) {
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    size_t total = (size_t)N * OH * OW * C;
    if (C % 4 == 0 && (double)N * H * W * C < 8.0e9 && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0) && (((uintptr_t)idx & 3) == 0)) {
        int g4 = (int)((total / 4 + 255) / 256);
        if (g4 > 16384) g4 = 16384;
        hipLaunchKernelGGL(maxpool3x3s2_fwd4_kernel, dim3(g4), dim3(256), 0, s, x, y, idx, N, H, W, C / 4, OH, OW);
        V2A_CHECK_LAUNCH();
        return V2A_OK;
    }
    int g = (int)((total + 255) / 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(maxpool3x3s2_fwd_kernel, dim3(g), dim3(256), 0, s, x, y, idx, N, H, W, C, OH, OW);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_maxpool3x3s2_bwd(const float* dy, const int8_t* idx, float* dx, int N, int H, int W, int C, hipStream_t s) {
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    size_t total = (size_t)N * H * W * C;
    if (C % 4 == 0 && (double)N * H * W * C < 8.0e9 && ((((uintptr_t)dy | (uintptr_t)dx) & 15) == 0) && (((uintptr_t)idx & 3) == 0)) {
        int g4 = (int)((total / 4 + 255) / 256);
        if (g4 > 16384) g4 = 16384;
        hipLaunchKernelGGL(maxpool3x3s2_bwd4_kernel, dim3(g4), dim3(256), 0, s, dy, idx, dx, N, H, W, C / 4, OH, OW);
        V2A_CHECK_LAUNCH();
        return V2A_OK;
    }
    int g = (int)((total + 255) / 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3(g), dim3(256), 0, s, dy, idx, dx, N, H, W, C, OH, OW);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_spatial_softmax_fwd(const float* feat, float* kp, float* att, int B, int H, int W, int K, hipStream_t s) {
    hipLaunchKernelGGL(spatial_softmax_fwd_kernel, dim3((B * K + 63) / 64), dim3(64), 0, s, feat, kp, att, B, H, W, K);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
int v2a_spatial_softmax_bwd(const float* att, const float* kp, const float* dkp, float* dfeat, int B, int H, int W, int K, hipStream_t s) {
    hipLaunchKernelGGL(spatial_softmax_bwd_kernel, dim3((B * K + 63) / 64), dim3(64), 0, s, att, kp, dkp, dfeat, B, H, W, K);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
}  // extern "C"
