// Direct gradient all-reduce of the data-parallel policy step over peer pointers: reduce-scatter + all-gather in ONE launch per arena slice.
//
// The reference wraps the policy in torch DDP (accelerator.prepare: diffuser/libero/lb_online_trainer_v7.py:153-154), whose hooks all-reduce
// the gradients inside accelerator.backward (:604) before clip_grad_norm_ (:608).  Here the gradients already live in one flat fp32 arena per
// rank; RCCL's all-reduce of it is the default exchange (v2a_hip/dp.py).  This file is the second algorithm behind the same GradReducer
// interface, for a node where RCCL picks a ring for the 349 MB message: xGMI is a full mesh (7 links per GPU), a ring keeps ONE link per
// direction busy (>= 4 ms for 610 MB at ~153 GB/s), the direct exchange keeps all seven busy (each rank pulls 1/W of the arena from every
// peer, twice: ~0.6 ms).
//
// Layout.  Every rank maps every peer's arena and every peer's signal block into its own address space (hipIpc handles, exchanged once by the
// host side).  A slice [lo, hi) is cut into W chunks of 16-byte units; rank r owns chunk r.  Workgroup b of every rank works on the SAME
// relative sub-range b of every chunk, so all ordering is between workgroup b of rank r and workgroup b of its peers: three flag barriers, no
// grid-wide synchronisation, no host round trip, everything stream-ordered.
//   barrier 1  (every peer has entered the kernel => its gradients are final: the kernel is stream-ordered behind the backward)
//   phase RS   rank r: own[chunk r] = sum over p = 0..W-1 (fixed order: every rank computes bit-identical sums) of peer p's [chunk r]
//   barrier 2  (sub-range b of every chunk is summed on its owner)
//   phase AG   rank r: own[chunk p] = peer p's [chunk p]  for p != r
//   barrier 3  (every peer has finished reading this rank's arena: the optimiser behind this launch may zero it)
// Flags are 32-bit counters in an uncached allocation, written with system-scope release stores into the PEER's signal block and polled
// locally; they only grow (3 per launch), so a stale flag of an earlier step can never satisfy a wait.  A wait is bounded by a wall-clock
// budget: on expiry the kernel raises host-visible error words, stops waiting (this launch and every later one run through without their
// barriers: the numbers are garbage and flagged as such) and terminates -- a lost rank is an error message, not a hung GPU.
#include "common.h"
#include <string.h>

#define V2A_DP_MAX_WORLD 8
#define V2A_DP_BLOCKS 256          // flag rows per slot = the most workgroups a launch may use
#define V2A_DP_BLOCKS_DEFAULT 64   // what a launch uses when the caller passes 0 (see the geometry note at the kernel)
#define V2A_DP_THREADS 256
#define V2A_DP_SLOTS 4             // independent flag sets: one per arena slice in flight (the trainer has two)

struct DpPeers {
    float* arena[V2A_DP_MAX_WORLD];
    uint32_t* signal[V2A_DP_MAX_WORLD];
};

__device__ __forceinline__ uint32_t* dp_flag(uint32_t* sig, int slot, int block, int peer) {
    return sig + ((size_t)slot * V2A_DP_BLOCKS + block) * V2A_DP_MAX_WORLD + peer;
}

// One barrier between workgroup `blockIdx.x` of all ranks.  No LDS and no early exit: a workgroup that needs a CU's whole LDS must still fit
// next to a waiting one (see the launch geometry below), so the give-up state lives in the error words, which every later wait also polls.
__device__ __forceinline__ void dp_barrier(const DpPeers& P, int world, int rank, int slot, uint32_t target, int* err, unsigned long long ticks) {
    __syncthreads();                                                   // this workgroup's stores of the phase before are issued
    const int t = threadIdx.x;
    if (t < world) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");                  // system scope: write back what this workgroup wrote (L2 write-back)
        __hip_atomic_store(dp_flag(P.signal[t], slot, blockIdx.x, rank), target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        uint32_t* mine = dp_flag(P.signal[rank], slot, blockIdx.x, t);
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        while ((int32_t)(__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - target) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 1023u) != 0) continue;
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;       // somebody gave up already: do not wait again
            if (wall_clock64() - t0 > ticks) {
                // first report wins: word 0 = 1 + the peer that never arrived, then where this rank stood (slot, flag value awaited, workgroup,
                // flag value seen)
                int expect = 0;
                if (__hip_atomic_compare_exchange_strong(err, &expect, 1 + t, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
                    __hip_atomic_store(err + 1, slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(err + 2, (int)target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(err + 3, (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(err + 4, (int)__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
                }
                break;
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                      // system scope: drop cached lines of the peers' arenas
}

// Launch geometry: `nblocks` workgroups (<= V2A_DP_BLOCKS, default 64), NOT one per CU.  A waiting workgroup holds a wave slot and registers
// on its CU for as long as the slowest peer needs to arrive; the conv kernels of the backward that runs next to slice 0's exchange take
// whole CUs (LDS, register file).  With 64 workgroups three quarters of the chip stay free for them, and 64 x 256 threads x 16 B x W loads in
// flight (>= 1 MB at W = 8) cover the ~400 GB/s x ~3 us of the seven xGMI links.
__global__ __launch_bounds__(V2A_DP_THREADS) void dp_allreduce_direct_kernel(DpPeers P, int world, int rank, size_t lo, size_t hi, int slot,
                                                                            uint32_t epoch, int* err, unsigned long long ticks) {
    const int t = threadIdx.x, b = blockIdx.x, nb = gridDim.x;
    // 16-byte units: `head` scalars up to the first aligned element, n4 float4 units, `tail` scalars behind them.  Every arena has the same
    // alignment (the host side checks it), so all ranks cut the slice alike.
    const size_t n = hi - lo;
    size_t head = ((16 - ((uintptr_t)(P.arena[rank] + lo) & 15)) & 15) / 4;
    if (head > n) head = n;
    const size_t n4 = (n - head) / 4, tail = n - head - 4 * n4;
    const size_t q = (n4 + world - 1) / world;                          // float4 units per chunk
    const size_t qb = (q + nb - 1) / nb;                                // ... per workgroup and chunk
    const uint32_t base = epoch * 3u;

    dp_barrier(P, world, rank, slot, base + 1, err, ticks);

    // ---- reduce-scatter: chunk `rank`, sub-range b, summed over the peers in rank order
    {
        const size_t c0 = (size_t)rank * q, c1 = min(c0 + q, n4);
        const size_t s0 = min(c0 + (size_t)b * qb, c1), s1 = min(s0 + qb, c1);
        const size_t off = lo + head;
        for (size_t i = s0 + t; i < s1; i += V2A_DP_THREADS) {
            f32x4 v[V2A_DP_MAX_WORLD];
#pragma unroll
            for (int p = 0; p < V2A_DP_MAX_WORLD; ++p)
                if (p < world) v[p] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P.arena[p] + off) + i);
            f32x4 acc = v[0];
#pragma unroll
            for (int p = 1; p < V2A_DP_MAX_WORLD; ++p)
                if (p < world) acc += v[p];
            reinterpret_cast<f32x4*>(P.arena[rank] + off)[i] = acc;
        }
        // the unaligned ends (at most 3 + 3 elements) belong to rank 0, workgroup 0
        if (rank == 0 && b == 0 && t < (int)(head + tail)) {
            const size_t e = lo + (t < (int)head ? (size_t)t : head + 4 * n4 + (t - head));
            float acc = P.arena[0][e];
            for (int p = 1; p < world; ++p) acc += P.arena[p][e];
            P.arena[0][e] = acc;
        }
    }

    dp_barrier(P, world, rank, slot, base + 2, err, ticks);

    // ---- all-gather: sub-range b of every other chunk from its owner (four loads in flight per thread)
    {
        const size_t off = lo + head;
        f32x4* dst = reinterpret_cast<f32x4*>(P.arena[rank] + off);
        for (int dp = 1; dp < world; ++dp) {
            const int p = (rank + dp) % world;                          // every rank starts at a different peer: all links busy
            const size_t c0 = (size_t)p * q, c1 = min(c0 + q, n4);
            const size_t s0 = min(c0 + (size_t)b * qb, c1), s1 = min(s0 + qb, c1);
            const f32x4* src = reinterpret_cast<const f32x4*>(P.arena[p] + off);
            size_t i = s0 + t;
            for (; i + 3 * V2A_DP_THREADS < s1; i += 4 * V2A_DP_THREADS) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(src + i + u * V2A_DP_THREADS);
#pragma unroll
                for (int u = 0; u < 4; ++u) dst[i + u * V2A_DP_THREADS] = v[u];
            }
            for (; i < s1; i += V2A_DP_THREADS) dst[i] = __builtin_nontemporal_load(src + i);
        }
        if (rank != 0 && b == 0 && t < (int)(head + tail)) {
            const size_t e = lo + (t < (int)head ? (size_t)t : head + 4 * n4 + (t - head));
            P.arena[rank][e] = P.arena[0][e];
        }
    }

    dp_barrier(P, world, rank, slot, base + 3, err, ticks);
}

extern "C" {

int v2a_dp_max_world(void) { return V2A_DP_MAX_WORLD; }
int v2a_dp_slots(void) { return V2A_DP_SLOTS; }
size_t v2a_dp_signal_bytes(void) { return (size_t)V2A_DP_SLOTS * V2A_DP_BLOCKS * V2A_DP_MAX_WORLD * sizeof(uint32_t); }

// The signal block of this rank: uncached device memory (flags must be seen by peers without a cache in between), zeroed.
int v2a_dp_signal_alloc(void** sig_out) {
    if (!sig_out) return V2A_ERR_ARG;
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, v2a_dp_signal_bytes(), hipDeviceMallocUncached) != hipSuccess) return V2A_ERR_LAUNCH;
    if (hipMemset(p, 0, v2a_dp_signal_bytes()) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(p);
        return V2A_ERR_LAUNCH;
    }
    *sig_out = p;
    return V2A_OK;
}
int v2a_dp_signal_free(void* sig) { return hipFree(sig) == hipSuccess ? V2A_OK : V2A_ERR_LAUNCH; }

// A gradient arena as an allocation of its own (fp32, zeroed): what v2a_dp_ipc_export hands to the peers is then exactly the arena, not a
// block of a caching allocator that other tensors live in (mapping such a block from a second process was measured to never return: the
// 349 MB arena of bench.py at offset 298 MB of a torch segment).
int v2a_dp_arena_alloc(void** arena_out, size_t bytes) {
    if (!arena_out || bytes == 0) return V2A_ERR_ARG;
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return V2A_ERR_LAUNCH;
    if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(p);
        return V2A_ERR_LAUNCH;
    }
    *arena_out = p;
    return V2A_OK;
}
int v2a_dp_arena_free(void* arena) { return hipFree(arena) == hipSuccess ? V2A_OK : V2A_ERR_LAUNCH; }

// The error words (16 ints): pinned host memory the kernel writes and the host reads without synchronising.  [0] = 0 fine / 1 + the peer
// that did not arrive; [1] slot, [2] the flag value waited for (3 * epoch + barrier number), [3] workgroup, [4] the flag value seen.
int v2a_dp_errword_alloc(int** word_out) {
    if (!word_out) return V2A_ERR_ARG;
    void* p = nullptr;
    if (hipHostMalloc(&p, 64, hipHostMallocMapped) != hipSuccess) return V2A_ERR_LAUNCH;
    ::memset(p, 0, 64);
    *word_out = reinterpret_cast<int*>(p);
    return V2A_OK;
}
int v2a_dp_errword_free(int* word) { return hipHostFree(word) == hipSuccess ? V2A_OK : V2A_ERR_LAUNCH; }

// hipIpc handle (64 bytes) of the ALLOCATION that holds `ptr`, ptr's byte offset inside it and the allocation's size (a tensor of a caching
// allocator is a window of a larger block; the handle names the block).
int v2a_dp_ipc_export(const void* ptr, void* handle64_out, uint64_t* offset_out, uint64_t* alloc_bytes_out) {
    if (!ptr || !handle64_out || !offset_out || !alloc_bytes_out) return V2A_ERR_ARG;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size is part of the wire format");
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, const_cast<void*>(ptr)) != hipSuccess) return V2A_ERR_LAUNCH;
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, base) != hipSuccess) return V2A_ERR_LAUNCH;
    ::memcpy(handle64_out, &h, 64);
    *offset_out = (uint64_t)((const char*)ptr - (const char*)base);
    *alloc_bytes_out = (uint64_t)size;
    return V2A_OK;
}
// Map a peer's allocation into this process: *base_out = where the allocation starts here (add the exported offset).  One open per handle and
// process (the host side keeps the table); close with the same base.
int v2a_dp_ipc_open(const void* handle64, void** base_out) {
    if (!handle64 || !base_out) return V2A_ERR_ARG;
    hipIpcMemHandle_t h;
    ::memcpy(&h, handle64, 64);
    void* base = nullptr;
    if (hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return V2A_ERR_LAUNCH;
    *base_out = base;
    return V2A_OK;
}
int v2a_dp_ipc_close(void* base) {
    if (!base) return V2A_ERR_ARG;
    return hipIpcCloseMemHandle(base) == hipSuccess ? V2A_OK : V2A_ERR_LAUNCH;
}

// One direct all-reduce (sum) of arena elements [lo, hi) across `world` ranks.  arenas / signals: HOST arrays of `world` device pointers valid
// in THIS process (own entries local, the others hipIpc mappings), the same peer order on every rank.  slot: flag set (< v2a_dp_slots()); two
// launches that may be in flight together must use different slots.  epoch: 1, 2, 3 ... per slot, the same on every rank.  errword: from
// v2a_dp_errword_alloc (0 = fine; 1 + p = peer p did not arrive within timeout_ms).  blocks: workgroups of the launch (0 = default 64, at
// most 256), the same on every rank.  Stream-ordered; nothing is waited for on the host.
int v2a_dp_allreduce_direct(float* const* arenas, uint32_t* const* signals, int world, int rank, size_t lo, size_t hi, int slot, uint32_t epoch,
                            int* errword, int timeout_ms, int blocks, hipStream_t stream) {
    if (!arenas || !signals || !errword || world < 1 || world > V2A_DP_MAX_WORLD || rank < 0 || rank >= world || hi < lo || slot < 0 ||
        slot >= V2A_DP_SLOTS || epoch == 0 || timeout_ms <= 0 || blocks < 0 || blocks > V2A_DP_BLOCKS)
        return V2A_ERR_ARG;
    if (blocks == 0) blocks = V2A_DP_BLOCKS_DEFAULT;
    DpPeers P = {};
    for (int p = 0; p < world; ++p) {
        if (!arenas[p] || !signals[p]) return V2A_ERR_ARG;
        if (((uintptr_t)arenas[p] & 15) != ((uintptr_t)arenas[0] & 15)) return V2A_ERR_ARG;      // every rank must cut the slice alike
        P.arena[p] = arenas[p];
        P.signal[p] = signals[p];
    }
    if (hi == lo) return V2A_OK;
    int* err_dev = nullptr;
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&err_dev), errword, 0) != hipSuccess) return V2A_ERR_ARG;
    const unsigned long long ticks = (unsigned long long)timeout_ms * 100000ull;                 // wall_clock64: 100 MHz
    hipLaunchKernelGGL(dp_allreduce_direct_kernel, dim3(blocks), dim3(V2A_DP_THREADS), 0, stream, P, world, rank, lo, hi, slot, epoch, err_dev,
                       ticks);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

}  // extern "C"
