// Cost of a grid barrier between resident workgroups on gfx950, by flavour (build: hipcc --offload-arch=gfx950 -O3 barrier_bench.hip -o barrier_bench)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int MODE> __global__ __launch_bounds__(256) void k(unsigned* flags, unsigned* counter, float* data, int iters, unsigned long long* ticks) {
    unsigned phase = 0;
    const unsigned long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        data[(size_t)blockIdx.x * 256 + threadIdx.x] = (float)it;          // something to publish
        __syncthreads();
        ++phase;
        if (MODE == 0) {              // flags, release store + acquire fence (agent scope: L2 write-back + invalidate)
            if (threadIdx.x < 64) {
                if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, phase, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                for (;;) {
                    unsigned behind = 0;
                    for (int i = threadIdx.x; i < (int)gridDim.x; i += 64) behind |= (unsigned)(__hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase);
                    if (__all(behind == 0)) break;
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        } else if (MODE == 1) {       // flags, relaxed only (no cache maintenance: data would have to travel with sc1 loads / stores)
            if (threadIdx.x < 64) {
                if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (;;) {
                    unsigned behind = 0;
                    for (int i = threadIdx.x; i < (int)gridDim.x; i += 64) behind |= (unsigned)(__hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase);
                    if (__all(behind == 0)) break;
                }
            }
            __syncthreads();
        } else if (MODE == 2) {       // one counter, release add + acquire fence
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase * gridDim.x) {}
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        } else if (MODE == 3) {       // one counter, relaxed
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase * gridDim.x) {}
            }
            __syncthreads();
        } else if (MODE == 4) {       // fences only, no waiting: what the cache maintenance alone costs
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *ticks = wall_clock64() - t0;
}

template <int MODE> void run(const char* name, int nwg, int iters) {
    unsigned *flags, *counter; float* data; unsigned long long* ticks;
    hipMalloc(&flags, 4096); hipMalloc(&counter, 64); hipMalloc(&data, 1024 * 256 * 4); hipMalloc(&ticks, 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(flags, 0, 4096); hipMemset(counter, 0, 64);
        hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(256), 0, 0, flags, counter, data, iters, ticks);
        hipDeviceSynchronize();
    }
    unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    printf("%-44s nwg %4d: %.2f us per barrier\n", name, nwg, (double)t / 100.0 / iters);
    hipFree(flags); hipFree(counter); hipFree(data); hipFree(ticks);
}

int main() {
    for (int nwg : {256, 128, 64, 32}) {
        run<0>("flags, release store + acquire fence", nwg, 200);
        run<1>("flags, relaxed (no cache maintenance)", nwg, 200);
        run<2>("counter, release add + acquire fence", nwg, 200);
        run<3>("counter, relaxed", nwg, 200);
        run<4>("release + acquire fences only, no wait", nwg, 200);
    }
    return 0;
}
