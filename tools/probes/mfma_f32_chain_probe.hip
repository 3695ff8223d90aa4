// Why do the exact-f32 LDS-DMA conv tiles sit at 46-58 % matrix-pipe utilisation with 60-65 % of their wave cycles in SQ_WAIT_INST?
// Isolated inner loop of conv_igemm_h<64,64,float> / <128,64,float>: operand fragments by ds_read_b128 from a resident LDS tile, chains
// of v_mfma_f32_32x32x2_f32 -- no DMA, no barrier per tile -- in several schedules.  Prints TFLOP/s per variant and workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f32_chain_probe mfma_f32_chain_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// V: 0 = as shipped (2 reads, wait, 4 dependent MFMAs; one accumulator); 1 = two accumulators (even / odd k-steps);
//    2 = all 8 reads of a tile up front, then 16 MFMAs on one accumulator; 3 = 2 + two accumulators; 4 = four accumulators;
//    5 = reads of step h+1 issued before the MFMAs of step h (sched_barrier), one accumulator; 6 = 5 + two accumulators
template <int V, int TM>
__global__ __launch_bounds__(256) void probe(float* out, int iters, int lds_extra) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < (128 + 64) * 32; i += 256) reinterpret_cast<float*>(smem)[i] = (float)((i * 7 + 3) % 13) * 0.01f;
    __syncthreads();
    const int lr = lane & 31, lk = lane >> 5;
    const int rswz = (lr >> 1) & 7;
    const int wm = (wid >> 1) * (TM * 32), wn = (wid & 1) * 32;
    int a_off[TM];
    for (int i = 0; i < TM; ++i) a_off[i] = (wm + i * 32 + lr) * 128;
    const int b_off = 128 * 128 + (wn + lr) * 128;
    f32x16 acc[TM][4];
    for (int i = 0; i < TM; ++i)
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        const unsigned char* base = smem;
        if constexpr (V == 0 || V == 1 || V == 4) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int pos = (((h << 1) | lk) ^ rswz) << 4;
                f32x4 a[TM], b;
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(base + a_off[i] + pos);
                b = *reinterpret_cast<const f32x4*>(base + b_off + pos);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int c = V == 0 ? 0 : (V == 1 ? (e & 1) : e);
                        acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[e], acc[i][c], 0, 0, 0);
                    }
            }
        } else if constexpr (V == 2 || V == 3) {
            f32x4 a[4][TM], b[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int pos = (((h << 1) | lk) ^ rswz) << 4;
#pragma unroll
                for (int i = 0; i < TM; ++i) a[h][i] = *reinterpret_cast<const f32x4*>(base + a_off[i] + pos);
                b[h] = *reinterpret_cast<const f32x4*>(base + b_off + pos);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 4; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int c = V == 2 ? 0 : (e & 1);
                        acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[h][i][e], b[h][e], acc[i][c], 0, 0, 0);
                    }
        } else {
            f32x4 a[2][TM], b[2];
            auto ld = [&](int h, int s) {
                const int pos = (((h << 1) | lk) ^ rswz) << 4;
#pragma unroll
                for (int i = 0; i < TM; ++i) a[s][i] = *reinterpret_cast<const f32x4*>(base + a_off[i] + pos);
                b[s] = *reinterpret_cast<const f32x4*>(base + b_off + pos);
            };
            ld(0, 0);
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                if (h + 1 < 4) ld(h + 1, (h + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int c = V == 5 ? 0 : (e & 1);
                        acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[h & 1][i][e], b[h & 1][e], acc[i][c], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < TM; ++i)
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 16; ++r) s += acc[i][c][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int V, int TM>
static void run(const char* name, float* out) {
    const int iters = 2000;
    for (int per_cu = 1; per_cu <= 3; ++per_cu) {
        // LDS request sized so that exactly per_cu workgroups fit a CU (160 KB)
        const size_t lds = per_cu == 1 ? 100 * 1024 : (per_cu == 2 ? 72 * 1024 : 48 * 1024);
        hipFuncSetAttribute((const void*)probe<V, TM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const int grid = 256 * per_cu;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((probe<V, TM>), dim3(grid), dim3(256), lds, 0, out, 10, 0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<V, TM>), dim3(grid), dim3(256), lds, 0, out, iters, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)grid * 4 /*waves*/ * iters * 16.0 * TM * 4096.0;
        printf("%-44s TM=%d  %d WG/CU: %7.1f TFLOP/s (%.3f ms)\n", name, TM, per_cu, flop / (ms * 1e-3) / 1e12, ms);
    }
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 3 * 256 * sizeof(float));
    run<0, 1>("0 shipped: 2 reads, wait, 4 dependent MFMA", out);
    run<1, 1>("1 two accumulators", out);
    run<4, 1>("4 four accumulators", out);
    run<2, 1>("2 reads up front, one chain of 16", out);
    run<3, 1>("3 reads up front, two accumulators", out);
    run<5, 1>("5 next step's reads before this step's MFMA", out);
    run<6, 1>("6 = 5 + two accumulators", out);
    run<0, 2>("0 shipped (128x64 tile: 2 chains)", out);
    run<2, 2>("2 reads up front (128x64)", out);
    run<3, 2>("3 reads up front, 2 acc per chain (128x64)", out);
    return 0;
}
