/* oracle/replay_ref.c -- TEST INFRASTRUCTURE (CPU oracle, plain C).
 *
 * Restates the replay-buffer index stream of the reference so that the HIP/host product path can be
 * checked bit-for-bit without Python RNG objects:
 *   Global_EnvReplayBuffer_Img.sample_random_batch_seq
 *        /root/reference/diffuser/datasets/env_img_replay_buffer.py:68-116
 *        -> ONE np.random.randint(0, n_episodes, size=B)                      (:81)
 *   EnvImg_UnitBuffer.sample_seq                                              (:278-302)
 *        -> per selected episode, in order: random.randint(0, len - act_len - 1) (:289), goal = start+act_len
 *   LB_Online_Trainer_V7.sample_from_bufs, 'rand_prob' branch
 *        /root/reference/diffuser/libero/lb_online_trainer_v7.py:826-830
 *        -> probs = np.random.uniform(size=B); n_rand = (probs < p).sum(); rand buffer first, then video buffer
 *
 * The RNG arithmetic itself lives in third-party code (numpy's legacy RandomState over MT19937 and CPython's
 * `random` module).  Both ARE present in this image, and tests/test_replay_oracle.py pins this file against
 * them directly (np.random.seed/randint/uniform, random.seed/randint) for many seeds and ranges.
 *
 *  numpy legacy: seed(int)   = init_genrand(s)
 *                randint     = masked rejection on 32-bit draws: mask = 2^ceil(log2(rng+1))-1;
 *                              do v = next32 & mask while v > rng          (rng = high-1-low; rng==0 draws nothing)
 *                uniform     = low + (high-low) * ((a>>5)*2^26 + (b>>6)) / 2^53 with a,b two 32-bit draws
 *  CPython:      seed(int)   = init_by_array(32-bit words of |s|)
 *                randint(a,b)= a + randbelow(b-a+1); randbelow(n): k = bit_length(n);
 *                              do r = next32 >> (32-k) while r >= n
 */
#include <stdint.h>
#include <stddef.h>

#define MT_N 624
#define MT_M 397

typedef struct {
    uint32_t mt[MT_N];
    int32_t pos;
} mt19937_t;

void mt_init_genrand(mt19937_t *s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < MT_N; ++i)
        s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->pos = MT_N;
}

void mt_init_by_array(mt19937_t *s, const uint32_t *key, int key_len) {
    mt_init_genrand(s, 19650218u);
    int i = 1, j = 0;
    int k = MT_N > key_len ? MT_N : key_len;
    for (; k; --k) {
        s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        ++i; ++j;
        if (i >= MT_N) { s->mt[0] = s->mt[MT_N - 1]; i = 1; }
        if (j >= key_len) j = 0;
    }
    for (k = MT_N - 1; k; --k) {
        s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        ++i;
        if (i >= MT_N) { s->mt[0] = s->mt[MT_N - 1]; i = 1; }
    }
    s->mt[0] = 0x80000000u;
    s->pos = MT_N;
}

static void mt_refill(mt19937_t *s) {
    uint32_t y;
    int kk;
    for (kk = 0; kk < MT_N - MT_M; ++kk) {
        y = (s->mt[kk] & 0x80000000u) | (s->mt[kk + 1] & 0x7fffffffu);
        s->mt[kk] = s->mt[kk + MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; kk < MT_N - 1; ++kk) {
        y = (s->mt[kk] & 0x80000000u) | (s->mt[kk + 1] & 0x7fffffffu);
        s->mt[kk] = s->mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    y = (s->mt[MT_N - 1] & 0x80000000u) | (s->mt[0] & 0x7fffffffu);
    s->mt[MT_N - 1] = s->mt[MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    s->pos = 0;
}

uint32_t mt_next32(mt19937_t *s) {
    if (s->pos >= MT_N) mt_refill(s);
    uint32_t y = s->mt[s->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* numpy legacy RandomState.randint(low, high, size=n) for a range that fits 32 bits */
void np_randint(mt19937_t *s, int64_t low, int64_t high, int64_t *out, int64_t n) {
    uint64_t rng = (uint64_t)(high - 1 - low);
    if (rng == 0) { for (int64_t i = 0; i < n; ++i) out[i] = low; return; }
    uint32_t mask = (uint32_t)rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    for (int64_t i = 0; i < n; ++i) {
        uint32_t v;
        do { v = mt_next32(s) & mask; } while (v > (uint32_t)rng);
        out[i] = low + (int64_t)v;
    }
}

double np_next_double(mt19937_t *s) {
    int32_t a = (int32_t)(mt_next32(s) >> 5), b = (int32_t)(mt_next32(s) >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

void np_uniform(mt19937_t *s, double low, double high, double *out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = low + (high - low) * np_next_double(s);
}

/* CPython random.randint(a, b), b - a + 1 < 2^32 */
int64_t py_randint(mt19937_t *s, int64_t a, int64_t b) {
    uint32_t n = (uint32_t)(b - a + 1);
    int k = 0;
    for (uint32_t t = n; t; t >>= 1) ++k;
    uint32_t r;
    do { r = mt_next32(s) >> (32 - k); } while (r >= n);
    return a + (int64_t)r;
}

/* One sample_random_batch_seq call: episode indices then (start) per row.  Returns 0, or -1 when an
 * episode is too short (the reference asserts act_len < cur_len, :288). */
int replay_sample_seq(mt19937_t *np_state, mt19937_t *py_state, const int32_t *episode_len, int32_t n_episodes,
                      int32_t batch, int32_t act_len, int64_t *out_episode, int64_t *out_start) {
    np_randint(np_state, 0, n_episodes, out_episode, batch);
    for (int32_t i = 0; i < batch; ++i) {
        int32_t len = episode_len[out_episode[i]];
        if (!(act_len < len)) return -1;
        out_start[i] = py_randint(py_state, 0, (int64_t)len - act_len - 1);
    }
    return 0;
}

/* sample_from_bufs 'rand_prob': returns n_rand (rows [0,n_rand) come from the random-action buffer, the rest
 * from the video-rollout buffer), or a negative error. */
int replay_sample_mixed(mt19937_t *np_state, mt19937_t *py_state, const int32_t *len_rand, int32_t n_rand_eps,
                        const int32_t *len_vid, int32_t n_vid_eps, int32_t batch, int32_t act_len, double rand_prob,
                        int64_t *out_episode, int64_t *out_start) {
    int n_rand = 0;
    for (int32_t i = 0; i < batch; ++i) n_rand += (0.0 + 1.0 * np_next_double(np_state)) < rand_prob;
    int rc = replay_sample_seq(np_state, py_state, len_rand, n_rand_eps, n_rand, act_len, out_episode, out_start);
    if (rc) return rc;
    rc = replay_sample_seq(np_state, py_state, len_vid, n_vid_eps, batch - n_rand, act_len, out_episode + n_rand,
                           out_start + n_rand);
    if (rc) return rc;
    return n_rand;
}
