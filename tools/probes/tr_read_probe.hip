// Probe of ds_read_b64_tr_b16 (gfx950): which LDS halfwords land in which lane / element for a few per-lane address patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short v4s __attribute__((ext_vector_type(4)));

__global__ void probe(int pattern, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    const int l = threadIdx.x;
    for (int i = l; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    int off;                                   // element (halfword) offset this lane passes
    if (pattern == 0) off = 0;
    else if (pattern == 1) off = l * 4;        // each lane: its own 8 contiguous bytes
    else if (pattern == 2) off = (l & 15) * 4 + (l >> 4) * 256;
    else off = (l & 15) * 64 + (l >> 4) * 4;   // rows of 64 halfwords, lane%16 = row, lane/16 = 4-column group
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)(lds + off);
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    out[l * 4 + 0] = (uint16_t)(r.x & 0xffff); out[l * 4 + 1] = (uint16_t)(r.x >> 16);
    out[l * 4 + 2] = (uint16_t)(r.y & 0xffff); out[l * 4 + 3] = (uint16_t)(r.y >> 16);
}

int main() {
    uint16_t* d;
    (void)hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int p = 0; p < 4; ++p) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, p, d);
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern %d\n", p);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d: %4d %4d %4d %4d", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
            if (l % 4 == 3) printf("\n");
        }
    }
    return 0;
}
