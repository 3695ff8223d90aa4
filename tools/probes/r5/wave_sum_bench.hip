// Dependent wave-wide sums: ds_bpermute shuffles (__shfl_xor) against DPP row butterflies + readlane, gfx950, one wave per workgroup.
// build: hipcc --offload-arch=gfx950 -O3 -w wave_sum_bench.hip -o wave_sum_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ float sum_shfl(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int CTRL> __device__ __forceinline__ float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float sum_dpp(float v) {
    v += dpp<0xB1>(v); v += dpp<0x4E>(v); v += dpp<0x141>(v); v += dpp<0x140>(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}
template <int MODE> __global__ void k(float* out, int iters, unsigned long long* ticks) {
    float v = (float)threadIdx.x * 1e-3f;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        v = (MODE == 0 ? sum_shfl(v) : sum_dpp(v)) * 1e-2f + (float)threadIdx.x * 1e-3f;
    }
    const unsigned long long t1 = wall_clock64();
    out[blockIdx.x * 64 + threadIdx.x] = v;
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}
int main() {
    float* out; unsigned long long* t; hipMalloc(&out, 1 << 20); hipMalloc(&t, 8);
    for (int nwg : {1, 256, 2048}) for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(64), 0, 0, out, 2000, t);
            else hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(64), 0, 0, out, 2000, t);
            hipDeviceSynchronize();
        }
        unsigned long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
        float r; hipMemcpy(&r, out, 4, hipMemcpyDeviceToHost);
        printf("%-28s %5d workgroups: %.1f ns per dependent wave sum (check %.6g)\n", mode == 0 ? "__shfl_xor x 6" : "4 DPP + 4 readlane", nwg, (double)h * 10.0 / 2000, r);
    }
    return 0;
}
