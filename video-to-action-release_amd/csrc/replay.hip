// GPU-resident replay store gather + native host index sampler.
// replaces Global_EnvReplayBuffer_Img.sample_random_batch_seq / EnvImg_UnitBuffer.sample_seq
// (reference diffuser/datasets/env_img_replay_buffer.py:68-116, 278-302) and the 'rand_prob' mixing of
// LB_Online_Trainer_V7.sample_from_bufs (lb_online_trainer_v7.py:826-830).
//
// Index stream (bit-exact contract): the host sampler re-implements numpy's legacy MT19937 randint/uniform and
// CPython's random.randint over imported/exported generator states, so a drop-in buffer consumes exactly the
// draws the reference would have consumed from np.random / random.  The payload never leaves HBM: frames are
// stored once ([frame][H][W][3], uint8 or fp32) and a batch is assembled by one coalesced gather kernel that
// also applies the u8 -> [0,1] -> [-1,1] conversion the policy needs (img_utils.py:27-37, normalizer.py:139-146).
#include "common.h"
#include <string.h>

template <typename T>
__global__ void replay_gather_kernel(const T* __restrict__ frames, const float* __restrict__ acts, const int64_t* __restrict__ frame_start,
                                     float* __restrict__ out_start, float* __restrict__ out_goal, float* __restrict__ out_acts,
                                     int B, int frame_elems, int act_len, int act_dim, float denom, int normalize, int chw_out, int HW) {
    const int b = blockIdx.y;
    const int64_t f0 = frame_start[b];
    const T* s0 = frames + (size_t)f0 * frame_elems;
    const T* s1 = frames + (size_t)(f0 + act_len) * frame_elems;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < frame_elems; i += gridDim.x * 256) {
        float a = (float)s0[i] / denom, g = (float)s1[i] / denom;      // true division, as images / 255.0 (img_utils.py:37)
        if (normalize) { a = 2.0f * a - 1.0f; g = 2.0f * g - 1.0f; }
        int o = i;
        if (chw_out) { const int c = i % 3, hw = i / 3; o = c * HW + hw; }   // store is HWC
        out_start[(size_t)b * frame_elems + o] = a;
        out_goal[(size_t)b * frame_elems + o] = g;
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < act_len * act_dim; i += 256)
            out_acts[(size_t)b * act_len * act_dim + i] = acts[(size_t)f0 * act_dim + i];
}

// uint8 HWC store -> CHW fp32 batch, four pixels per thread: 12 contiguous bytes in (three dwords), one 16-B store per colour plane out
// (a wave writes 1 KB runs; the element-per-thread kernel above writes 84-B runs and reads single bytes: 113 us per B = 64 batch).
__global__ __launch_bounds__(256) void replay_gather_u8_chw4_kernel(const uint8_t* __restrict__ frames, const float* __restrict__ acts,
                                                                   const int64_t* __restrict__ frame_start, float* __restrict__ out_start,
                                                                   float* __restrict__ out_goal, float* __restrict__ out_acts, int frame_elems,
                                                                   int act_len, int act_dim, int normalize, int HW) {
    const int b = blockIdx.y;
    const int64_t f0 = frame_start[b];
    const uint8_t* src[2] = {frames + (size_t)f0 * frame_elems, frames + (size_t)(f0 + act_len) * frame_elems};
    float* dst[2] = {out_start + (size_t)b * frame_elems, out_goal + (size_t)b * frame_elems};
    const int q = blockIdx.x * 256 + threadIdx.x;               // pixels 4 q .. 4 q + 3
    if (4 * q < HW) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t* w = reinterpret_cast<const uint32_t*>(src[t] + 12 * (size_t)q);     // (frame_elems % 4 == 0: dword aligned)
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
            const uint32_t bytes[12] = {w0 & 255, (w0 >> 8) & 255, (w0 >> 16) & 255, w0 >> 24, w1 & 255, (w1 >> 8) & 255, (w1 >> 16) & 255, w1 >> 24,
                                        w2 & 255, (w2 >> 8) & 255, (w2 >> 16) & 255, w2 >> 24};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                f32x4 o;
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    float v = (float)bytes[3 * px + c] / 255.0f;      // true division, as images / 255.0 (img_utils.py:37)
                    if (normalize) v = 2.0f * v - 1.0f;
                    o[px] = v;
                }
                *reinterpret_cast<f32x4*>(dst[t] + (size_t)c * HW + 4 * q) = o;
            }
        }
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < act_len * act_dim; i += 256)
            out_acts[(size_t)b * act_len * act_dim + i] = acts[(size_t)f0 * act_dim + i];
}

// ----------------------------------------------------------------------------------------- host sampler
namespace {
struct MT {
    uint32_t mt[624];
    int pos;
};
inline void mt_twist(MT& s) {
    for (int k = 0; k < 624; ++k) {
        uint32_t y = (s.mt[k] & 0x80000000u) | (s.mt[(k + 1) % 624] & 0x7fffffffu);
        s.mt[k] = s.mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    s.pos = 0;
}
inline uint32_t mt_u32(MT& s) {
    if (s.pos >= 624) mt_twist(s);
    uint32_t y = s.mt[s.pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
inline int64_t numpy_bounded(MT& s, uint32_t rng) {   // legacy masked rejection, draws nothing when rng == 0
    if (rng == 0) return 0;
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    do { v = mt_u32(s) & mask; } while (v > rng);
    return v;
}
inline double numpy_double(MT& s) {
    const int32_t a = (int32_t)(mt_u32(s) >> 5), b = (int32_t)(mt_u32(s) >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
}
inline int64_t cpython_randbelow(MT& s, uint32_t n) {
    int k = 32 - __builtin_clz(n);
    uint32_t r;
    do { r = mt_u32(s) >> (32 - k); } while (r >= n);
    return r;
}
}  // namespace

extern "C" {

// np_state / py_state: 625 uint32 each = MT19937 key[624] + position (numpy get_state()[1:3]; random.getstate()[1]).
// Both are updated in place so the caller can write them back into the live generators.
// Returns 0, -1 (bad args), -4 (an episode is shorter than act_len + 1: the reference asserts, :288).
int v2a_replay_sample_indices(uint32_t* np_state, uint32_t* py_state, const int32_t* episode_len, int32_t n_episodes,
                              int32_t batch, int32_t act_len, int64_t* out_episode, int64_t* out_start) {
    if (!np_state || !py_state || !episode_len || n_episodes <= 0 || batch < 0) return V2A_ERR_ARG;
    MT a, b;
    memcpy(a.mt, np_state, 624 * 4); a.pos = (int)np_state[624];
    memcpy(b.mt, py_state, 624 * 4); b.pos = (int)py_state[624];
    for (int i = 0; i < batch; ++i) out_episode[i] = numpy_bounded(a, (uint32_t)(n_episodes - 1));
    int rc = V2A_OK;
    for (int i = 0; i < batch; ++i) {
        const int len = episode_len[out_episode[i]];
        if (!(act_len < len)) { rc = -4; break; }
        out_start[i] = cpython_randbelow(b, (uint32_t)(len - act_len));
    }
    memcpy(np_state, a.mt, 624 * 4); np_state[624] = (uint32_t)a.pos;
    memcpy(py_state, b.mt, 624 * 4); py_state[624] = (uint32_t)b.pos;
    return rc;
}

// np.random.uniform(size=batch) < prob  ->  count (the n_rand of sample_from_bufs 'rand_prob')
int v2a_replay_count_uniform_below(uint32_t* np_state, int32_t batch, double prob) {
    MT a;
    memcpy(a.mt, np_state, 624 * 4); a.pos = (int)np_state[624];
    int n = 0;
    for (int i = 0; i < batch; ++i) n += (0.0 + (1.0 - 0.0) * numpy_double(a)) < prob;
    memcpy(np_state, a.mt, 624 * 4); np_state[624] = (uint32_t)a.pos;
    return n;
}

// seeding helpers (np.random.seed(int) / random.seed(int) semantics) so a fully native loop needs no Python RNG
int v2a_mt_seed_numpy(uint32_t* state, uint32_t seed) {
    state[0] = seed;
    for (int i = 1; i < 624; ++i) state[i] = 1812433253u * (state[i - 1] ^ (state[i - 1] >> 30)) + (uint32_t)i;
    state[624] = 624;
    return V2A_OK;
}
int v2a_mt_seed_python(uint32_t* state, const uint32_t* key, int key_len) {
    v2a_mt_seed_numpy(state, 19650218u);
    int i = 1, j = 0;
    for (int k = (624 > key_len ? 624 : key_len); k; --k) {
        state[i] = (state[i] ^ ((state[i - 1] ^ (state[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        if (++i >= 624) { state[0] = state[623]; i = 1; }
        if (++j >= key_len) j = 0;
    }
    for (int k = 623; k; --k) {
        state[i] = (state[i] ^ ((state[i - 1] ^ (state[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        if (++i >= 624) { state[0] = state[623]; i = 1; }
    }
    state[0] = 0x80000000u;
    state[624] = 624;
    return V2A_OK;
}

// frames: [n_frames][H][W][3] (dtype_u8 ? uint8 : float, HWC); acts: [n_frames][act_dim] fp32 (row f = action taken at frame f);
// frame_start[b] = global index of the start frame.  Outputs are HWC (chw_out = 0, what the encoder consumes) or CHW
// (chw_out = 1, the reference's tensor layout).
int v2a_replay_gather(const void* frames, int dtype_u8, const float* acts, const int64_t* frame_start, float* out_start,
                      float* out_goal, float* out_acts, int B, int H, int W, int act_len, int act_dim, int normalize,
                      int chw_out, hipStream_t s) {
    if (!frames || !acts || !frame_start || !out_start || !out_goal || !out_acts) return V2A_ERR_ARG;
    const int fe = H * W * 3;
    dim3 grid((fe + 256 * 8 - 1) / (256 * 8), B);
    if (dtype_u8 && chw_out && (H * W) % 4 == 0 && ((((uintptr_t)out_start | (uintptr_t)out_goal | (uintptr_t)frames) & 15) == 0))
        hipLaunchKernelGGL(replay_gather_u8_chw4_kernel, dim3((H * W / 4 + 255) / 256, B), dim3(256), 0, s, (const uint8_t*)frames, acts,
                           frame_start, out_start, out_goal, out_acts, fe, act_len, act_dim, normalize, H * W);
    else if (dtype_u8)
        hipLaunchKernelGGL((replay_gather_kernel<uint8_t>), grid, dim3(256), 0, s, (const uint8_t*)frames, acts, frame_start, out_start,
                           out_goal, out_acts, B, fe, act_len, act_dim, 255.0f, normalize, chw_out, H * W);
    else
        hipLaunchKernelGGL((replay_gather_kernel<float>), grid, dim3(256), 0, s, (const float*)frames, acts, frame_start, out_start,
                           out_goal, out_acts, B, fe, act_len, act_dim, 1.0f, normalize, chw_out, H * W);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

}  // extern "C"
