// Halo-tile 3x3 convolution for the bf16-storage video UNet (stride 1, pad 1, one source): the forward counterpart of
// conv_wgrad_halo_f32.
//
// conv_igemm_h / conv_igemm_h2 DMA an A tile (output pixels x 32..64 input channels) for EVERY filter tap: nine fetches of what is,
// up to a one-pixel shift, the same patch of the input.  At bf16 rates those kernels are bound by bytes in flight per CU (29-37 % MFMA
// utilisation at ~3 TB/s of L2 -> LDS traffic, PMC: 2.7x the algorithmic bytes), so the lever is bytes per FLOP, not scheduling.
// Here a 512-thread workgroup owns a 16 x 16 pixel patch of one frame (BM = 256 output rows) x BN output channels and walks the
// reduction as (32-channel chunk c) x (tap t):
//   * per chunk the 18 x 18 pixel HALO of the patch (324 rows x 64 B) is DMA-ed ONCE into one of two halo buffers; the nine taps
//     read it through shifted row windows (output pixel (py, px), tap (kh, kw) -> halo row (py + kh) * 18 + px + kw);
//   * per (chunk, tap) only the weight tile (BN rows x 64 B) moves, through a ring of SB stages.
// Bytes per 256 x 256 x 32 MAC step: 16 KB (B) + 20.7 KB / 9 (A) = 18.3 KB instead of 32 KB; for BN = 128: 10.3 instead of 24 KB.
// Synchronisation as in conv_igemm_h2 (counted `s_waitcnt vmcnt`, one raw `s_barrier` per step, never drained): DMAs of one wave
// retire in issue order, the halo of chunk c+1 is issued in three pieces at taps 1 / 3 / 5 of chunk c (all of them ahead of the
// first weight tile of chunk c+1 in the queue), and EVERY step issues the same number of DMA instructions -- past the end of the
// reduction they read the zero line -- so the wait count of a step depends on its tap alone and is a compile-time constant of the
// unrolled nine-tap body.  LDS: 2 x 24 KB halo + SB x BN x 64 B ring (112 KB at BN = 256, SB = 4; 96 KB at BN = 128, SB = 6).
// 64-B rows: slot (row r, position p) holds the row's 16-B chunk p ^ ((r >> 2) & 3); a 16-lane group reads 16 consecutive halo
// rows starting anywhere, and rows r, r+4, r+8, r+12 always differ in (r >> 2) & 3, so the operand fetches stay conflict-free under
// every tap shift.  Epilogue = conv_igemm_h2's (rows of the patch mapped back to NHWC rows).
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_3;
typedef const __attribute__((address_space(1))) void* gptr3_t;
typedef __attribute__((address_space(3))) void* lptr3_t;

struct ConvDescH3 {
    const uint16_t* x;          // [N, H, W, C] bf16
    const uint16_t* w;          // [Cout][3][3][C] bf16
    const float* bias;
    const float* rowvec;        // [M / rows_per_batch][Cout] or null
    const uint16_t* residual;   // [M][Cout] bf16 or null
    uint16_t* y;                // [M][Cout] bf16
    float* stats;               // optional [M / 64][2][Cout]
    const uint16_t* zeros;
    int N, H, W, C, Cout, M, K, rows_per_batch, tiles_x, tiles_img;
    int ups;                    // 1: nearest x2 upsample folded into the halo gather (Upsample + conv, unet.py:105-115): source frame is (H/2, W/2)
    // GroupNorm + activation applied to the input while it sits in LDS (conv_halo_h3<.., GN = 1>): the conv reads the tensor the
    // GroupNorm would have read, [x | x2] along channels, and normalises it itself
    const uint16_t* x2;         // channels [C1, C) of the input or null
    const float* ab;            // [N / fps][2][C]: per (sample, channel) scale and shift (gn_finalize_h)
    int C1, fps, act;           // channels of x; frames (images) per GroupNorm sample; ACT_SILU / ACT_NONE
};

__device__ __forceinline__ int xcd_remap3(int bid, int nblk) {
    int q = nblk >> 3, r = nblk & 7;
    int xcd = bid & 7, slot = bid >> 3;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// A `ds_read_b128` of a wave is served in four groups of 16 lanes, and within the lower 32 lanes these are {0-3, 12-15, 20-27} and
// {4-11, 16-19, 28-31} (MI355X_MICROARCH.md, LDS) -- not 0-15 / 16-31.  MFMA row m of a 32-row sub-tile therefore takes pixel
// perm(m) of the sub-tile's 2 x 16 pixels, chosen so that each hardware group reads ONE patch row = 16 consecutive halo rows, which
// the (row >> 2) & 3 slot swizzle spreads over all 64 banks under every tap shift.  (With m -> pixel m the groups straddle the two
// patch rows, 18 halo rows apart, and every operand read is a 2-way conflict: 8 LDS cycles instead of 4.)
#ifndef V2A_H3_PERM
#define V2A_H3_PERM 1
#endif
__device__ __forceinline__ int lds_group_perm3(int m) {
#if V2A_H3_PERM
    const int qd = m >> 2;
    return ((__builtin_popcount(qd) & 1) << 4) | ((qd >> 1) << 2) | (m & 3);
#else
    return m;
#endif
}

// value of the other lane of an (even, odd) lane pair
__device__ __forceinline__ float pair_swap3(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));   // quad_perm [1, 0, 3, 2]
}

// One 32 x 32 sub-tile of fp32 accumulators in the MFMA C layout (lane -> column lane & 31, register r -> row
// (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) -> 16-bit values, two adjacent columns per dword, WITHOUT the fp32 round trip through
// LDS: v = (acc + colb) + colr is rounded in registers; the lanes of an (even, odd) pair trade one value per register pair, after
// which the even lane holds row rho(2q) and the odd lane row rho(2q + 1), columns (c & ~1, c | 1), in pk[q].  The per-column sums of
// the ROUNDED values and of their squares accumulate into s2 / q2 (lane: its two columns over its eight rows).
// LDS traffic of an epilogue sub-tile: 8 ds_write_b32 + 2 ds_read_b128 instead of 16 + 4 (+ 4 ds_write_b128 + 16 ds_read_b32 for the
// statistics) -- the write path into LDS moves 64-85 B/clk and was what the epilogues of these kernels waited for.
template <bool F16>
__device__ __forceinline__ void pack_subtile3(const f32x16& acc, float colb, float colr, bool odd, uint32_t (&pk)[8], float (&s2)[2], float (&q2)[2]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float v0 = (acc[2 * q] + colb) + colr, v1 = (acc[2 * q + 1] + colb) + colr;
        const float t = pair_swap3(odd ? v0 : v1);         // even lane: the odd lane's v0 (column c + 1, row rho(2q)); odd lane: the even lane's v1
        pk[q] = v2a_pack_h2<F16>(odd ? t : v0, odd ? v1 : t);
        const float a = v2a_lo_h2<F16>(pk[q]), b = v2a_hi_h2<F16>(pk[q]);
        s2[0] += a; q2[0] += a * a;
        s2[1] += b; q2[1] += b * b;
    }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt3() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// halo piece j of the next chunk goes out at tap halo_tap(j, HP): 1, 3, 5 for three pieces, 1 .. 5 for five -- all of them ahead
// of the first weight tile of the next chunk in the queue as long as SB - 1 <= 3 (issued at tap 9 - (SB - 1) >= 6) ...
// (gn: one tap earlier -- 0, 2, 4 / 0 .. 4 -- so that every piece has landed four taps later, where the fused GroupNorm transforms it)
__host__ __device__ constexpr int halo_tap(int j, int HP, bool gn = false) { return (HP <= 3 ? 1 + 2 * j : 1 + j) - (gn ? 1 : 0); }
__host__ __device__ constexpr int halo_piece_at(int tap, int HP, bool gn = false) {       // piece index issued at `tap`, or -1
    for (int j = 0; j < HP; ++j)
        if (halo_tap(j, HP, gn) == tap) return j;
    return -1;
}
// number of halo pieces issued at steps s-(SB-1) .. s-1 when step s has tap T
template <int T, int SB, int HP, bool GNF = false>
struct HaloLater {
    static constexpr int count() {
        int c = 0;
        for (int d = 1; d <= SB - 1; ++d) {
            int u = ((T - d) % 9 + 9) % 9;
            if (halo_piece_at(u, HP, GNF) >= 0) ++c;
        }
        return c;
    }
    static constexpr int value = count();
};

// BM = 256 (16 x 16 patch) or 512 (32 rows x 16 pixels: the 128-wide instance -- 12.3 KB of DMA per 512 x 128 x 32 step).
// (Measured and not kept, round 3: a 256-thread / 256 x 128 / 80 KB instance with TWO workgroups per CU -- plain layers +5...8 %,
// GroupNorm-fused layers -10 %, sampler unchanged; and a phase-alternating main loop, waves 0-3 fetching while waves 4-7 multiply and
// vice versa with two barriers per step -- 1...4 % slower on plain layers, 10 % on fused ones.  The plain instances sit at 1.2-1.3
// PFLOP/s on random data, which is where the chip's power management holds a dense bf16 MFMA stream (MI355X_MICROARCH.md, DVFS).)
template <int WAVES_M, int WAVES_N, int TM, int TN, int SB, int GN, bool F16>
__global__ __launch_bounds__(512, 1) void conv_halo_h3(const ConvDescH3 p) {
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr int NW = WAVES_M * WAVES_N, NT = NW * 64;
    static_assert((BM == 256 || BM == 512) && NW == 8, "tile shape");
    constexpr int PH = BM / 16;                            // patch: PH rows of 16 pixels
    constexpr int ROWB = 64;                               // bytes per row: 32 bf16
    constexpr int HW_ = 18, HROWS = HW_ * (PH + 2);        // halo: (PH + 2) x 18 pixels
    constexpr int HPIECES = (HROWS * 4 + NT - 1) / NT;      // DMA instructions per thread per halo (3: 24 KB >= 20.7 KB; 5: 40 KB >= 38.3 KB)
    static_assert(HPIECES <= 4 || SB >= 2, "halo pieces");
    constexpr int HBUF = HPIECES * NT * 16;
    constexpr int BL = BN * 4 / NT;                        // DMA instructions per thread per weight tile (NT / 4 rows x 4 chunks per pass)
    constexpr int BSTAGE = BN * ROWB;
    constexpr int PIPE = 2 * HBUF + SB * BSTAGE;
    constexpr int SMEM = PIPE + (GN ? 2 * 1024 * 4 : 0);   // GN: scale / shift of the tile's sample, [2][C <= 1024] fp32, behind the pipeline buffers
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    static_assert((SB - 2) * BL + HPIECES <= 63, "vmcnt is a 6-bit counter");
    static_assert(9 - (SB - 1) > halo_tap(HPIECES - 1, HPIECES, GN != 0), "the last halo piece must precede the next chunk's first weight tile");
    // a piece issued at step u has landed once the wait of step u + SB has passed: that wait leaves the weight tiles of the last SB - 2
    // steps and the halo pieces of the last SB - 1 steps in flight, all of them younger
    constexpr int GN_DELAY = SB;
    static_assert(!GN || halo_tap(HPIECES - 1, HPIECES, true) + GN_DELAY <= 8, "a piece is transformed GN_DELAY taps after it was issued");
    __shared__ __attribute__((aligned(128))) unsigned char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = p.Cout / BN;
    const int tiles_m = p.N * p.tiles_img;
    const int lin = xcd_remap3(blockIdx.x, tiles_m * tiles_n);
    const int tm = lin / tiles_n, n0 = (lin % tiles_n) * BN;
    const int img = tm / p.tiles_img, trem = tm - img * p.tiles_img;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int oy0 = ty * PH, ox0 = tx * 16;
    const int nchunks = p.C >> 5;

    // ---- halo DMA source state: piece q = j * 512 + tid -> halo row q >> 2, position q & 3, carrying chunk (q & 3) ^ ((row >> 2) & 3).
    // 32-bit element offsets from p.x (0xffffffff = the zero line): the kernel sits at the 256-VGPR cap
    uint32_t h_off[HPIECES];                                 // GN: the PIXEL index (the channel part depends on which source the chunk lies in)
    const uint16_t* zsrc = p.zeros;
#pragma unroll
    for (int j = 0; j < HPIECES; ++j) {
        const int q = j * NT + tid;
        const int hr = q >> 2;
        const int hy = hr / HW_, hx = hr - hy * HW_;
        const int ih = oy0 - 1 + hy, iw = ox0 - 1 + hx;
        const int chunk = (q & 3) ^ ((hr >> 2) & 3);
        const bool ok = hr < HROWS && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        const int sh_ = p.ups ? (p.H >> 1) : p.H, sw_ = p.ups ? (p.W >> 1) : p.W;
        const int ihs = p.ups ? (ih >> 1) : ih, iws = p.ups ? (iw >> 1) : iw;
        const uint32_t pix = (uint32_t)(img * sh_ + ihs) * (uint32_t)sw_ + (uint32_t)iws;
        if (GN) h_off[j] = ok ? pix : 0xffffffffu;
        else h_off[j] = ok ? pix * (uint32_t)p.C + (uint32_t)chunk * 8u : 0xffffffffu;
    }
    // ---- weight DMA source state: pass j fills rows j*128 .. +127; slot (row j*128 + tid/4, position tid%4), chunk (tid%4) ^ ((row>>2)&3)
    const uint32_t b_off0 = (uint32_t)(n0 + (tid >> 2)) * (uint32_t)p.K + (uint32_t)(((tid & 3) ^ ((tid >> 4) & 3)) * 8);
    const uint32_t b_step = (uint32_t)(NT / 4) * (uint32_t)p.K;

    auto issue_halo_piece = [&](int j, int chunk_idx, int hb) {     // chunk_idx >= nchunks: the zero line (keeps the DMA count uniform)
        const uint16_t* g = zsrc;
        if (GN) {
            if (h_off[j] != 0xffffffffu && chunk_idx < nchunks) {
                const int q = j * NT + tid;
                const uint32_t cp = (uint32_t)(((q & 3) ^ ((q >> 4) & 3)) * 8);          // (hr >> 2) & 3 with hr = q >> 2
                const uint32_t c0 = (uint32_t)chunk_idx * 32u;
                const uint32_t C2 = (uint32_t)(p.C - p.C1);
                g = (c0 < (uint32_t)p.C1) ? p.x + (size_t)h_off[j] * (uint32_t)p.C1 + c0 + cp
                                          : p.x2 + (size_t)h_off[j] * C2 + (c0 - (uint32_t)p.C1) + cp;
            }
        } else {
            if (h_off[j] != 0xffffffffu && chunk_idx < nchunks) g = p.x + h_off[j] + chunk_idx * 32;
        }
        __builtin_amdgcn_global_load_lds((gptr3_t)g, (lptr3_t)(smem + hb * HBUF + (j * NT + wid * 64) * 16), 16, 0, 0);
    };
    // fused GroupNorm: this thread normalises the piece IT fetched (its own vmcnt wait covers the landing), in place:
    // y = act(x * a[n, c] + b[n, c]) on eight channels, rounded back to bf16 -- the arithmetic of gn_apply_h.  Padding pixels stay 0.
    float* ab_lds = reinterpret_cast<float*>(smem + PIPE);
    // every piece of a thread carries the same eight channels of a chunk ((q >> 4) & 3 does not depend on the piece index: 512 / 16 is a
    // multiple of 4), so the scale / shift of those channels is fetched once per chunk, not once per piece
    const int gn_cp8 = ((tid & 3) ^ ((tid >> 4) & 3)) << 3;
    f32x4 gn_a0, gn_a1, gn_b0, gn_b1;
    auto gn_load_ab = [&](int chunk_idx) {
        const int ch0 = (chunk_idx < nchunks ? chunk_idx : 0) * 32 + gn_cp8;
        gn_a0 = *reinterpret_cast<const f32x4*>(ab_lds + ch0); gn_a1 = *reinterpret_cast<const f32x4*>(ab_lds + ch0 + 4);
        gn_b0 = *reinterpret_cast<const f32x4*>(ab_lds + p.C + ch0); gn_b1 = *reinterpret_cast<const f32x4*>(ab_lds + p.C + ch0 + 4);
    };
    // (Pinning this arithmetic between the step's MFMAs -- one channel behind every second MFMA -- was tried: 256 x 256 tiles -2 %,
    // 512 x 128 tiles +5 %, end to end slower; letting the first-dispatched half of the waves normalise BEFORE their MFMAs and the other
    // half after -- role alternation of the two waves of a SIMD -- cost +15 %: the step barrier waits for the slower half.  It runs behind
    // the MFMA issue instead, overlapping the matrix pipe's tail and the partner wave.)
    // split in two so that the LDS round trip of the piece is covered by the step's MFMAs: fetch before them, arithmetic + store after
    auto gn_fetch = [&](int j, int hb) -> uint4 { return *reinterpret_cast<const uint4*>(smem + hb * HBUF + (j * NT + tid) * 16); };
    auto gn_finish = [&](const uint4 u, int j, int chunk_idx, int hb) {
        if (h_off[j] == 0xffffffffu || chunk_idx >= nchunks) return;
        const bool silu = p.act == ACT_SILU;
        uint4 r;
        r.x = v2a_gn_act2<F16>(u.x, v2a_f32x2{gn_a0[0], gn_a0[1]}, v2a_f32x2{gn_b0[0], gn_b0[1]}, silu);
        r.y = v2a_gn_act2<F16>(u.y, v2a_f32x2{gn_a0[2], gn_a0[3]}, v2a_f32x2{gn_b0[2], gn_b0[3]}, silu);
        r.z = v2a_gn_act2<F16>(u.z, v2a_f32x2{gn_a1[0], gn_a1[1]}, v2a_f32x2{gn_b1[0], gn_b1[1]}, silu);
        r.w = v2a_gn_act2<F16>(u.w, v2a_f32x2{gn_a1[2], gn_a1[3]}, v2a_f32x2{gn_b1[2], gn_b1[3]}, silu);
        *reinterpret_cast<uint4*>(smem + hb * HBUF + (j * NT + tid) * 16) = r;
    };
    auto gn_piece = [&](int j, int chunk_idx, int hb) { gn_finish(gn_fetch(j, hb), j, chunk_idx, hb); };
    auto issue_b = [&](int bc, int bt, int stage) {                 // weight tile of (chunk bc, tap bt); bc >= nchunks: the zero line
        const bool live = bc < nchunks;
        const uint32_t koff = (uint32_t)(bt * p.C + bc * 32);
        unsigned char* bbase = smem + 2 * HBUF + stage * BSTAGE;
#pragma unroll
        for (int j = 0; j < BL; ++j) {
            const uint16_t* g = live ? p.w + b_off0 + j * b_step + koff : zsrc;
            __builtin_amdgcn_global_load_lds((gptr3_t)g, (lptr3_t)(bbase + (j * NT + wid * 64) * 16), 16, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm = (wid / WAVES_N) * TM * 32, wn = (wid % WAVES_N) * TN * 32;
    const int lr = lane & 31, lk = lane >> 5;
    int a_hr[TM];                                            // halo row of this lane's output pixel at tap (0, 0)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm + i * 32 + lds_group_perm3(lr);     // row inside the patch: (py, px) = (r / 16, r % 16)
        a_hr[i] = (r >> 4) * HW_ + (r & 15);
    }
    const int brswz = (lr >> 2) & 3;
    int b_off[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) b_off[j] = (wn + j * 32 + lr) * ROWB;

    // ---- prologue: halo of chunk 0, then SB-1 weight tiles
    constexpr int ABV = 2048 / (NT * 4);                      // 2 C <= 2048 floats
    f32x4 abv[ABV];
#pragma unroll
    for (int v = 0; v < ABV; ++v) {
        abv[v] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (GN && (v * NT + tid) * 4 < 2 * p.C) abv[v] = *reinterpret_cast<const f32x4*>(p.ab + (size_t)(img / p.fps) * 2 * p.C + (v * NT + tid) * 4);
    }
#pragma unroll
    for (int j = 0; j < HPIECES; ++j) issue_halo_piece(j, 0, 0);
#pragma unroll
    for (int s = 0; s < SB - 1; ++s) issue_b(s / 9, s % 9, s);
    int cstage = 0, istage = SB - 1;
    if (GN) {
#pragma unroll
        for (int v = 0; v < ABV; ++v)
            if ((v * NT + tid) * 4 < 2 * p.C) *reinterpret_cast<f32x4*>(ab_lds + (v * NT + tid) * 4) = abv[v];
        wait_vmcnt3<(SB - 1) * BL>();                      // this thread's halo pieces of chunk 0 (the weight tiles stay in flight)
        __syncthreads();                                   // scale / shift visible
        gn_load_ab(0);
#pragma unroll
        for (int j = 0; j < HPIECES; ++j) gn_piece(j, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // published by the barrier of the first step
    }

    // one step: all twelve operand fragments are requested first, the step's DMAs are issued while those reads are in flight (a DMA
    // instruction costs the issuing wave 100-185 cycles inside a busy phase), then the sixteen MFMAs run back to back.  Measured
    // +2...4.5 % on every shape over "DMA issue, then per k-half: reads, MFMAs".  (Fetching the next k-half / next step's fragments
    // under the MFMAs with a second register set was 5 % SLOWER: the step is bound by the weight tiles' L2 -> LDS traffic, not by
    // the LDS read latency.)
    auto load_frags = [&](bf16x8_3 (&a)[2][TM], bf16x8_3 (&b)[2][TN], const unsigned char* hbase, const unsigned char* bbase, int shift) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int hr = a_hr[i] + shift;
                a[h][i] = *reinterpret_cast<const bf16x8_3*>(hbase + hr * ROWB + ((((h << 1) | lk) ^ ((hr >> 2) & 3)) << 4));
            }
            const int bpos = (((h << 1) | lk) ^ brswz) << 4;
#pragma unroll
            for (int j = 0; j < TN; ++j) b[h][j] = *reinterpret_cast<const bf16x8_3*>(bbase + b_off[j] + bpos);
        }
    };
    auto mma_frags = [&](const bf16x8_3 (&a)[2][TM], const bf16x8_3 (&b)[2][TN]) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = v2a_mfma_h<F16>(a[h][i], b[h][j], acc[i][j]);
    };

    // the DMAs of step (c, T): weight tile of step + SB - 1, and on some taps a piece of the next chunk's halo
#define V2A_H3_ISSUE(T)                                                                                                \
        issue_b(c + (T + SB - 1) / 9, (T + SB - 1) % 9, istage);                                                       \
        if (halo_piece_at(T, HPIECES, GN != 0) >= 0)                                                                   \
            issue_halo_piece(halo_piece_at(T, HPIECES, GN != 0), c + 1, (c + 1) & 1);
#define V2A_H3_GN_AT(T) (GN && T >= GN_DELAY && halo_piece_at(T - GN_DELAY, HPIECES, true) >= 0)   /* landed: see GN_DELAY */
#define V2A_H3_ADVANCE()                                                                                               \
        cstage = (cstage + 1 == SB) ? 0 : cstage + 1;                                                                  \
        istage = (istage + 1 == SB) ? 0 : istage + 1;

    // lock-step: all twelve operand fragments are requested first, the step's DMAs are issued while those reads are in flight (a DMA
    // instruction costs the issuing wave 100-185 cycles inside a busy phase), then the sixteen MFMAs run back to back.
#define V2A_H3_TAP(T)                                                                                                  \
    {                                                                                                                  \
        if (c == 0) wait_vmcnt3<(SB - 2) * BL>();      /* start-up: no halo pieces of a previous chunk in the queue */  \
        else wait_vmcnt3<(SB - 2) * BL + HaloLater<T, SB, HPIECES, GN != 0>::value>();                                 \
        __builtin_amdgcn_s_barrier();                                                                                  \
        bf16x8_3 a[2][TM], b[2][TN];                                                                                   \
        load_frags(a, b, smem + (c & 1) * HBUF, smem + 2 * HBUF + cstage * BSTAGE, (T / 3) * HW_ + (T % 3));           \
        uint4 gu = {0u, 0u, 0u, 0u};                                                                                   \
        if (V2A_H3_GN_AT(T)) gu = gn_fetch(halo_piece_at(T - GN_DELAY, HPIECES, true), (c + 1) & 1);                   \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        V2A_H3_ISSUE(T)                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        mma_frags(a, b);                                                                                               \
        if (V2A_H3_GN_AT(T)) {                                                                                         \
            if (T == GN_DELAY) gn_load_ab(c + 1);                                                                      \
            gn_finish(gu, halo_piece_at(T - GN_DELAY, HPIECES, true), c + 1, (c + 1) & 1);                             \
        }                                                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
        V2A_H3_ADVANCE()                                                                                               \
    }
    for (int c = 0; c < nchunks; ++c) {
        // keep the 72 (sub-tile, tap, k-half) operand addresses out of registers: they are chunk-invariant and the compiler would
        // hoist them all (spilling the accumulators' neighbours); making the row indices opaque per chunk re-derives them under the MFMAs
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(a_hr[i]));
        V2A_H3_TAP(0) V2A_H3_TAP(1) V2A_H3_TAP(2) V2A_H3_TAP(3) V2A_H3_TAP(4) V2A_H3_TAP(5) V2A_H3_TAP(6) V2A_H3_TAP(7) V2A_H3_TAP(8)
    }
#undef V2A_H3_TAP
#undef V2A_H3_ISSUE
#undef V2A_H3_GN_AT
#undef V2A_H3_ADVANCE

    // ---- epilogue (conv_igemm_h2's, with patch rows mapped back to NHWC rows)
    wait_vmcnt3<0>();
    __syncthreads();
    constexpr int WNC = TN * 32, LDC = WNC;
    static_assert(NW * 32 * LDC * 4 <= PIPE, "epilogue staging exceeds the LDS buffers");
    float* cw = reinterpret_cast<float*>(smem) + wid * 32 * LDC;
    constexpr int V = WNC / 8;
    const int vrow = lane / V, vcol = (lane % V) * 8;
    const int n = n0 + wn + vcol;
    float bv[8], ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bv[e] = p.bias ? p.bias[n + e] : 0.f;
        ssum[e] = 0.f;
        ssq[e] = 0.f;
    }
    const size_t img_row0 = (size_t)img * p.H * p.W;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
                cw[row * LDC + ((j * 32 + lr) ^ ((row & 1) << 2))] = acc[i][j][r];
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int rr = 0; rr < 32; rr += 64 / V) {
            const int ml = rr + vrow;
            const int pr = wm + i * 32 + (V2A_H3_PERM ? lds_group_perm3(ml) : ml);   // row inside the patch
            const size_t m = img_row0 + (size_t)(oy0 + (pr >> 4)) * p.W + ox0 + (pr & 15);
            const int sx = (ml & 1) << 2;
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(&cw[ml * LDC + (vcol ^ sx)]);
            const f32x4 c1 = *reinterpret_cast<const f32x4*>(&cw[ml * LDC + ((vcol + 4) ^ sx)]);
            float v[8] = {c0[0] + bv[0], c0[1] + bv[1], c0[2] + bv[2], c0[3] + bv[3], c1[0] + bv[4], c1[1] + bv[5], c1[2] + bv[6], c1[3] + bv[7]};
            const size_t o = m * p.Cout + n;
            if (p.rowvec) {
                const float* rv = p.rowvec + (m / p.rows_per_batch) * p.Cout + n;
                const f32x4 r0 = *reinterpret_cast<const f32x4*>(rv), r1 = *reinterpret_cast<const f32x4*>(rv + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[e + 4] += r1[e]; }
            }
            if (p.residual) {
                const uint4 u = *reinterpret_cast<const uint4*>(p.residual + o);
                v[0] += v2a_lo_h2<F16>(u.x); v[1] += v2a_hi_h2<F16>(u.x);
                v[2] += v2a_lo_h2<F16>(u.y); v[3] += v2a_hi_h2<F16>(u.y);
                v[4] += v2a_lo_h2<F16>(u.z); v[5] += v2a_hi_h2<F16>(u.z);
                v[6] += v2a_lo_h2<F16>(u.w); v[7] += v2a_hi_h2<F16>(u.w);
            }
            uint16_t h[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                h[e] = v2a_f2h<F16>(v[e]);
                const float r = v2a_h2f<F16>(h[e]);
                ssum[e] += r;
                ssq[e] += r * r;
            }
            uint4 u;
            u.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
            u.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
            u.z = (uint32_t)h[4] | ((uint32_t)h[5] << 16);
            u.w = (uint32_t)h[6] | ((uint32_t)h[7] << 16);
            *reinterpret_cast<uint4*>(p.y + o) = u;
        }
        if ((i & 1) == 1 && p.stats) {
            // two 32-row sub-tiles = one 64-row statistics block of this wave.  Blocks are numbered (tile, 64-row group of the
            // patch): another order than NHWC rows / 64, but every block still lies inside ONE frame, which is all the GroupNorm
            // reduction over a sample's blocks needs
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int o2 = V; o2 < 64; o2 <<= 1) {
                    ssum[e] += __shfl_xor(ssum[e], o2, 64);
                    ssq[e] += __shfl_xor(ssq[e], o2, 64);
                }
            }
            if (lane < V) {
                const size_t blk = (size_t)tm * (BM / 64) + ((wm + (i - 1) * 32) >> 6);
                float* dst = p.stats + blk * 2 * p.Cout + n;
                f32x4 a0 = {ssum[0], ssum[1], ssum[2], ssum[3]}, a1 = {ssum[4], ssum[5], ssum[6], ssum[7]};
                f32x4 q0 = {ssq[0], ssq[1], ssq[2], ssq[3]}, q1 = {ssq[4], ssq[5], ssq[6], ssq[7]};
                *reinterpret_cast<f32x4*>(dst) = a0;
                *reinterpret_cast<f32x4*>(dst + 4) = a1;
                *reinterpret_cast<f32x4*>(dst + p.Cout) = q0;
                *reinterpret_cast<f32x4*>(dst + p.Cout + 4) = q1;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------------
// Temporal (3 x 1 x 1) convolution of the factorised Conv3d (unet.py: temporal_conv over [B, F, H*W, C]) -- the frame-stack tile.
// conv_igemm_h / _h2 fetch the input three times (once per tap) and run at 470-770 TFLOP/s on these layers.  Here a 512-thread
// workgroup owns ALL F frames of 64 consecutive pixels of one sample (BM = F * 64 = 448 output rows) x 128 output channels; per
// 32-channel chunk one DMA brings the F x 64 input rows (28 KB) plus the three taps' weight tiles (3 x 8 KB) into one of three LDS
// stages, a wave loads its F operand fragments ONCE into registers and uses each of them for up to three taps
// (output frame f, tap t <- input frame f + t - 1), and the products against the out-of-range frames -1 and F are simply not issued
// (19 instead of 21 MFMAs per k-half: 9.5 % fewer than the zero-padded GEMM).  One barrier per chunk (~2400 MFMA cycles per SIMD),
// a counted vmcnt wait with one newer stage always in flight; 0.53 KB of LDS reads per MFMA (the halo kernel: 0.75).
// K = 3 C is short (4 chunks at C = 128), so the launch is PERSISTENT: one workgroup per CU walks tiles lin, lin + G, ... as one
// chunk stream and the next tile's first two chunks land while this tile's epilogue runs (a one-tile-per-workgroup launch of the
// same body spends a third of its time filling and draining the pipeline; these layers sit near the HBM roofline).
// The two M-waves split every frame's 64 pixels in halves, so a 64-row statistics block is completed through LDS in the epilogue
// (fixed order: deterministic); the per-column sums come from a column pass over the staged tile, not a lane butterfly.
struct ConvDescT3 {
    const uint16_t* x;          // [B, F, HW, C] bf16
    const uint16_t* w;          // [Cout][3][C] bf16
    const float* bias;
    const float* rowvec;        // [M / rows_per_batch][Cout] or null
    const uint16_t* residual;   // [M][Cout] bf16 or null
    uint16_t* y;                // [M][Cout] bf16
    float* stats;               // optional [M / 64][2][Cout]
    const uint16_t* zeros;
    int B, HW, C, Cout, K, rows_per_batch, tiles_b;
};

template <int F, bool F16>
__global__ __launch_bounds__(512, 1) void conv_frames_h3(const ConvDescT3 p) {
    constexpr int PX = 64, BM = F * PX, BN = 128, ROWB = 64, NST = 3;
    constexpr int ABYTES = BM * ROWB, BBYTES = 3 * BN * ROWB, STAGE = ABYTES + BBYTES;
    constexpr int AJ = (BM * 4 + 511) / 512;               // DMA instructions per thread for the input rows; the last one covers
    constexpr int ALAST_WAVES = (BM * 4 - (AJ - 1) * 512) / 64;   // only the first ALAST_WAVES waves (whole waves: BM * 4 % 64 == 0)
    constexpr int LDC = 32, V = 4;
    constexpr int ST_OFF = 8 * 32 * LDC * 4;               // epilogue scratch inside the stage consumed last: staging rows, then statistics
    constexpr int SMEM = NST * STAGE;                      // 156 KB at F = 7
    static_assert(SMEM <= 160 * 1024 && (BM * 4) % 64 == 0, "LDS budget / whole-wave tail");
    static_assert(ST_OFF + 2 * F * 4 * 64 * 4 <= STAGE, "epilogue scratch must fit one stage");
    __shared__ __attribute__((aligned(128))) unsigned char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = p.Cout / BN;
    const int total = p.B * p.tiles_b * tiles_n;
    const int G = gridDim.x;
    int lin = xcd_remap3(blockIdx.x, G);
    if (lin >= total) return;
    const int nchunks = p.C >> 5;
    const bool a_tail = wid < ALAST_WAVES;                 // wave-uniform

    // thread part of the DMA source offsets (BYTES, 32 bit); the tile / chunk part is a wave-uniform 64-bit base, so the DMA takes
    // the scalar-base + 32-bit-offset address form and no per-thread 64-bit pointers stay live across the MFMA loop
    uint32_t a_thr[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int q = j * 512 + tid;
        const int row = q >> 2, f = row >> 6, px = row & 63;
        const int chunk = (q & 3) ^ ((row >> 2) & 3);
        a_thr[j] = (((uint32_t)f * (uint32_t)p.HW + (uint32_t)px) * (uint32_t)p.C + (uint32_t)chunk * 8u) * 2u;
    }
    const uint32_t b_thr = ((uint32_t)(tid >> 2) * (uint32_t)p.K + (uint32_t)(((tid & 3) ^ ((tid >> 4) & 3)) * 8)) * 2u;

    // the chunk stream this workgroup walks: tiles lin, lin + G, ... x chunks 0 .. nchunks-1; `iq_*` is the next chunk to ISSUE (two
    // ahead of the one computed, across tile boundaries: the next tile's first chunks land while this tile's epilogue runs).  Past
    // the end of the stream the DMAs read the zero line, so every step issues the same number and the wait counts stay constants
    int iq_lin = lin, iq_c = 0;
    auto tile_base = [&](int l, size_t& ta, size_t& tb) {
        const int t2 = l / tiles_n;
        const int bb = t2 / p.tiles_b;
        ta = ((size_t)(bb * F) * p.HW + (size_t)(t2 - bb * p.tiles_b) * PX) * p.C;
        tb = (size_t)(l - t2 * tiles_n) * BN * p.K;
    };
    size_t iq_ta, iq_tb;
    tile_base(iq_lin, iq_ta, iq_tb);
    auto issue_next = [&](int stage) {
        unsigned char* base = smem + stage * STAGE;
        const bool live = iq_lin < total;
        const char* xb = reinterpret_cast<const char*>(p.x) + (iq_ta + (size_t)iq_c * 32) * 2;
        const char* wb = reinterpret_cast<const char*>(p.w) + (iq_tb + (size_t)iq_c * 32) * 2;
        const char* zb = reinterpret_cast<const char*>(p.zeros);
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            if (j == AJ - 1 && !a_tail) break;
            __builtin_amdgcn_global_load_lds((gptr3_t)(live ? xb + a_thr[j] : zb), (lptr3_t)(base + (j * 512 + wid * 64) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
            __builtin_amdgcn_global_load_lds((gptr3_t)(live ? wb + (size_t)j * p.C * 2 + b_thr : zb),
                                             (lptr3_t)(base + ABYTES + (j * 512 + wid * 64) * 16), 16, 0, 0);
        if (live && ++iq_c == nchunks) {
            iq_c = 0;
            iq_lin += G;
            if (iq_lin < total) tile_base(iq_lin, iq_ta, iq_tb);
        }
    };

    const int w = wid >> 2, wn = (wid & 3) * 32;           // M half (pixels w*32 .. +31 of every frame), output-channel group
    const int lr = lane & 31, lk = lane >> 5;
    const int swz = (lr >> 2) & 3;
    int a_base = (w * 32 + lr) * ROWB;
    const int b_base = ABYTES + (wn + lr) * ROWB;
    const int vrow = lane / V, vcol = (lane % V) * 8;

    issue_next(0);
    issue_next(1);
    wait_vmcnt3<0>();
    int cstage = 0, istage = 2;
    for (; lin < total; lin += G) {
        const int tmi = lin / tiles_n;
        const int b = tmi / p.tiles_b, p0 = (tmi - b * p.tiles_b) * PX, n0 = (lin - tmi * tiles_n) * BN;
        f32x16 acc[F];
#pragma unroll
        for (int i = 0; i < F; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

        for (int c = 0; c < nchunks; ++c) {
            // chunk c must have landed; chunk c+1 may stay in flight.  The two chunks that were in flight when the previous tile's
            // epilogue began were waited for THERE (below), so steps 0 and 1 need no wait; from step 2 on every DMA this counts was
            // issued after the epilogue's stores.  (Loads and stores share vmcnt and do not retire in one order: a counted wait is
            // exact only for loads with no YOUNGER store in the queue -- older stores can only prolong it.  Waiting for the
            // prefetched chunks after the stores instead, with vmcnt(0), cost 2.5 us of store acknowledgements per tile.)
            if (c >= 2) {
                if (a_tail) wait_vmcnt3<AJ + 3>();
                else wait_vmcnt3<AJ - 1 + 3>();
            }
            __builtin_amdgcn_s_barrier();
            issue_next(istage);
            const unsigned char* sb = smem + cstage * STAGE;
            asm volatile("" : "+v"(a_base));
            // operand fragments: the ten of k-half 0 go out first, the ten of k-half 1 behind the first tap's MFMAs (lgkmcnt is a
            // 4-bit counter: more than 15 reads in flight make the compiler drain the queue).  Left alone the compiler fetches one
            // weight fragment at a time and waits for it before every tap
            bf16x8_3 a[2][F], bt[2][3];
            auto load_half = [&](int h) {
                const int pos = (((h << 1) | lk) ^ swz) << 4;
#pragma unroll
                for (int i = 0; i < F; ++i) a[h][i] = *reinterpret_cast<const bf16x8_3*>(sb + a_base + i * (PX * ROWB) + pos);
#pragma unroll
                for (int t = 0; t < 3; ++t) bt[h][t] = *reinterpret_cast<const bf16x8_3*>(sb + b_base + t * (BN * ROWB) + pos);
            };
            auto mma_tap = [&](int h, int t) {
#pragma unroll
                for (int i = 0; i < F; ++i) {
                    const int src = i + t - 1;
                    if (src >= 0 && src < F) acc[i] = v2a_mfma_h<F16>(a[h][src], bt[h][t], acc[i]);
                }
            };
            load_half(0);
            __builtin_amdgcn_sched_barrier(0);
            mma_tap(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_half(1);
            __builtin_amdgcn_sched_barrier(0);
            mma_tap(0, 1);
            mma_tap(0, 2);
            mma_tap(1, 0);
            mma_tap(1, 1);
            mma_tap(1, 2);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            cstage = (cstage + 1 == NST) ? 0 : cstage + 1;
            istage = (istage + 1 == NST) ? 0 : istage + 1;
        }

        wait_vmcnt3<0>();                                  // the next tile's first two chunks, ahead of this tile's stores
        // `istage` now names the stage consumed last: nothing is in flight to it until the next step's barrier, so -- once every wave
        // has read its last fragments -- it carries the epilogue's staging rows (private to each wave) and the statistics exchange
        __builtin_amdgcn_s_barrier();
        float* cw = reinterpret_cast<float*>(smem + istage * STAGE) + wid * 32 * LDC;
        float* st = reinterpret_cast<float*>(smem + istage * STAGE + ST_OFF);       // [2 (w)][F][4 (wn)][2][32]
        // ---- epilogue: sub-tile i of this wave = frame i, pixels p0 + w*32 .. +31 (staging rows private to the wave)
        const int n = n0 + wn + vcol;
        if (!p.residual && (!p.rowvec || p.rows_per_batch == F * p.HW)) {
            // no residual, and the row vector is one table row for the whole tile: the register path (pack_subtile3)
            const int ncol = n0 + wn + lr;
            const float colb = p.bias ? p.bias[ncol] : 0.f;
            const float colr = p.rowvec ? p.rowvec[(size_t)b * p.Cout + ncol] : 0.f;
            const bool odd = lane & 1;
            uint32_t* ph = reinterpret_cast<uint32_t*>(smem + istage * STAGE) + wid * 32 * 16;    // [32 rows][16 dwords] per wave
#pragma unroll
            for (int i = 0; i < F; ++i) {
                uint32_t pk[8];
                float s2[2] = {0.f, 0.f}, q2[2] = {0.f, 0.f};
                pack_subtile3<F16>(acc[i], colb, colr, odd, pk, s2, q2);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = 2 * q + (odd ? 1 : 0);
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
                    ph[row * 16 + (lr >> 1)] = pk[q];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const size_t m0 = ((size_t)(b * F + i) * p.HW) + p0 + w * 32;
#pragma unroll
                for (int rr = 0; rr < 32; rr += 64 / V) {
                    const int ml = rr + vrow;
                    const uint4 u = *reinterpret_cast<const uint4*>(ph + ml * 16 + (lane % V) * 4);
                    *reinterpret_cast<uint4*>(p.y + (m0 + ml) * p.Cout + n) = u;
                }
                if (p.stats) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        s2[k] += pair_swap3(s2[k]);
                        q2[k] += pair_swap3(q2[k]);
                        s2[k] += __shfl_xor(s2[k], 32, 64);
                        q2[k] += __shfl_xor(q2[k], 32, 64);
                    }
                    if (lane < 32 && !odd) {
                        float* d = st + (((w * F + i) * 4 + (wid & 3)) * 2) * 32 + lr;
                        d[0] = s2[0]; d[1] = s2[1];
                        d[32] = q2[0]; d[33] = q2[1];
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the staging rows are rewritten by the next frame
            }
        } else {
        float bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = p.bias ? p.bias[n + e] : 0.f;
        // Loads and stores share the vmcnt counter, so a load issued behind a frame's stores makes its consumer wait for those stores
        // to be acknowledged -- seven store round trips in
        // a row per tile (PMC: 3.2-3.5 TB/s on the C = 128 layers).  The row vector is one table row per tile when a tile lies inside one
        // batch entry (rows_per_batch = F * HW, the only way the UNet calls it); the residual rows of frame i + 1 are requested before
        // frame i's stores, so their wait leaves those stores in flight.
        const bool rv_tile = p.rowvec && p.rows_per_batch == F * p.HW;
        f32x4 rv0 = {0.f, 0.f, 0.f, 0.f}, rv1 = {0.f, 0.f, 0.f, 0.f};
        if (rv_tile) {
            const float* rv = p.rowvec + (size_t)b * p.Cout + n;
            rv0 = *reinterpret_cast<const f32x4*>(rv);
            rv1 = *reinterpret_cast<const f32x4*>(rv + 4);
        }
        uint4 rnext[2] = {uint4{0u, 0u, 0u, 0u}, uint4{0u, 0u, 0u, 0u}};
        auto load_res = [&](int i) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
                rnext[k] = *reinterpret_cast<const uint4*>(p.residual + (((size_t)(b * F + i) * p.HW) + p0 + w * 32 + k * (64 / V) + vrow) * p.Cout + n);
        };
        if (p.residual) load_res(0);
#pragma unroll
        for (int i = 0; i < F; ++i) {
            const uint4 rcur[2] = {rnext[0], rnext[1]};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
                cw[row * LDC + (lr ^ ((row & 1) << 2))] = acc[i][r];
            }
            if (p.residual && i + 1 < F) load_res(i + 1);          // ahead of this frame's stores
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const size_t m0 = ((size_t)(b * F + i) * p.HW) + p0 + w * 32;
#pragma unroll
            for (int rr = 0; rr < 32; rr += 64 / V) {
                const int ml = rr + vrow;
                const size_t m = m0 + ml;
                const int sx = (ml & 1) << 2;
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(&cw[ml * LDC + (vcol ^ sx)]);
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(&cw[ml * LDC + ((vcol + 4) ^ sx)]);
                float v[8] = {c0[0] + bv[0], c0[1] + bv[1], c0[2] + bv[2], c0[3] + bv[3], c1[0] + bv[4], c1[1] + bv[5], c1[2] + bv[6], c1[3] + bv[7]};
                const size_t o = m * p.Cout + n;
                if (rv_tile) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += rv0[e]; v[e + 4] += rv1[e]; }
                } else if (p.rowvec) {
                    const float* rv = p.rowvec + (size_t)((uint32_t)m / (uint32_t)p.rows_per_batch) * p.Cout + n;
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>(rv), r1 = *reinterpret_cast<const f32x4*>(rv + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[e + 4] += r1[e]; }
                }
                if (p.residual) {
                    const uint4 u = rcur[rr / (64 / V)];
                    v[0] += v2a_lo_h2<F16>(u.x); v[1] += v2a_hi_h2<F16>(u.x);
                    v[2] += v2a_lo_h2<F16>(u.y); v[3] += v2a_hi_h2<F16>(u.y);
                    v[4] += v2a_lo_h2<F16>(u.z); v[5] += v2a_hi_h2<F16>(u.z);
                    v[6] += v2a_lo_h2<F16>(u.w); v[7] += v2a_hi_h2<F16>(u.w);
                }
                uint16_t hh[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) hh[e] = v2a_f2h<F16>(v[e]);
                if (p.stats) {                                        // the stored values go back into the staging rows for the column pass
                    const f32x4 w0 = {v2a_h2f<F16>(hh[0]), v2a_h2f<F16>(hh[1]), v2a_h2f<F16>(hh[2]), v2a_h2f<F16>(hh[3])}, w1 = {v2a_h2f<F16>(hh[4]), v2a_h2f<F16>(hh[5]), v2a_h2f<F16>(hh[6]), v2a_h2f<F16>(hh[7])};
                    *reinterpret_cast<f32x4*>(&cw[ml * LDC + (vcol ^ sx)]) = w0;
                    *reinterpret_cast<f32x4*>(&cw[ml * LDC + ((vcol + 4) ^ sx)]) = w1;
                }
                uint4 u;
                u.x = (uint32_t)hh[0] | ((uint32_t)hh[1] << 16);
                u.y = (uint32_t)hh[2] | ((uint32_t)hh[3] << 16);
                u.z = (uint32_t)hh[4] | ((uint32_t)hh[5] << 16);
                u.w = (uint32_t)hh[6] | ((uint32_t)hh[7] << 16);
                *reinterpret_cast<uint4*>(p.y + o) = u;
            }
            if (p.stats) {
                // column pass: lane = (column lane & 31, rows (lane >> 5) * 16 .. +15) of the 32 x 32 sub-tile, sixteen conflict-free
                // ds_read_b32 and in-lane adds, ONE cross-lane exchange per statistic.  (A 16-lane butterfly over the row-major lanes
                // costs 64 ds_bpermute per sub-tile -- a third of this kernel's time at C = 128.)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int col = lane & 31, r0 = (lane >> 5) * 16;
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    const int row = r0 + rr;
                    const float x = cw[row * LDC + (col ^ ((row & 1) << 2))];
                    s += x;
                    q += x * x;
                }
                s += __shfl_xor(s, 32, 64);
                q += __shfl_xor(q, 32, 64);
                if (lane < 32) {
                    float* d = st + (((w * F + i) * 4 + (wid & 3)) * 2) * 32 + col;
                    d[0] = s;
                    d[32] = q;
                }
            }
        }
        }
        if (p.stats) {
            __syncthreads();
            const int half = lane >> 5, col = lane & 31;
            for (int i = w; i < F; i += 2) {               // the two pixel halves of frame i's 64-row block, added in a fixed order
                const float v = st[(((0 * F + i) * 4 + (wid & 3)) * 2 + half) * 32 + col] + st[(((1 * F + i) * 4 + (wid & 3)) * 2 + half) * 32 + col];
                const size_t blk = (((size_t)(b * F + i) * p.HW) + p0) >> 6;
                p.stats[blk * 2 * p.Cout + (size_t)half * p.Cout + n0 + wn + col] = v;
            }
        }
    }
}

extern "C" {

// 1 when v2a_conv2d_fwd_h3 takes this problem: 3x3 / stride 1 / pad 1, one bf16 source, frames that tile into 16 x 16 patches,
// 128-multiple output width, enough tiles to fill the chip.
int v2a_conv2d_h3_eligible(int N, int H, int W, int C, int Cout, int KH, int KW, int sh, int sw, int ph, int pw, int ups, int C2) {
    if (KH != 3 || KW != 3 || sh != 1 || sw != 1 || ph != 1 || pw != 1 || C2) return 0;
    if (ups) { H *= 2; W *= 2; }                          // H, W: the SOURCE frame; the conv runs over the upsampled one
    if (C % 32 || Cout % 128 || H % 16 || W % 16) return 0;
    const bool wide = Cout % 256 == 0;                    // 256 x 256 tiles; otherwise 512 x 128 (needs H % 32 == 0) or 256 x 128
    const long tiles = wide ? (long)N * (H / 16) * (W / 16) * (Cout / 256)
                            : (long)N * (H / (H % 32 == 0 ? 32 : 16)) * (W / 16) * (Cout / 128);
    if (tiles < 208) return 0;                            // at least ~80 % of the CUs busy in the single round
    if ((double)N * H * W * C >= 4294967296.0 || (double)Cout * 9 * C >= 4294967296.0) return 0;
    return 1;
}

static int conv_h3_launch(const void* x, const void* x2, int C1, const float* gn_ab, int gn_fps, int gn_act, const void* w_packed,
                          const float* bias, const float* rowvec, const void* residual, void* y, const void* zeros, int N, int H, int W, int C,
                          int Cout, int ups, int rows_per_batch, float* stats, hipStream_t stream) {
    if (!x || !w_packed || !zeros || !y || N <= 0) return V2A_ERR_ARG;
    if (!v2a_conv2d_h3_eligible(N, H, W, C, Cout, 3, 3, 1, 1, 1, 1, ups, 0)) return V2A_ERR_ARG;
    if (ups) { H *= 2; W *= 2; }
    if ((((uintptr_t)x | (uintptr_t)x2 | (uintptr_t)w_packed | (uintptr_t)zeros | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)bias |
          (uintptr_t)rowvec | (uintptr_t)gn_ab) & 15) != 0)
        return V2A_ERR_ARG;
    ConvDescH3 p;
    p.x = (const uint16_t*)x; p.w = (const uint16_t*)w_packed; p.bias = bias; p.rowvec = rowvec; p.residual = (const uint16_t*)residual;
    p.y = (uint16_t*)y; p.stats = stats; p.zeros = (const uint16_t*)zeros;
    p.N = N; p.H = H; p.W = W; p.C = C; p.Cout = Cout; p.M = N * H * W; p.K = 9 * C;
    p.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
    p.ups = ups ? 1 : 0;
    p.x2 = (const uint16_t*)x2; p.C1 = x2 ? C1 : C; p.ab = gn_ab; p.fps = gn_fps > 0 ? gn_fps : 1; p.act = gn_act;
    p.tiles_x = W / 16;
    p.tiles_img = (H / 16) * (W / 16);
    const bool gn = gn_ab != nullptr;
#define V2A_H3_LAUNCH_NT(NT_, ...)                                                                                           \
    do {                                                                                                                     \
        if (g_v2a_half_f16) {                                                                                                \
            if (gn) hipLaunchKernelGGL((conv_halo_h3<__VA_ARGS__, 1, true>), dim3(tiles), dim3(NT_), 0, stream, p);          \
            else hipLaunchKernelGGL((conv_halo_h3<__VA_ARGS__, 0, true>), dim3(tiles), dim3(NT_), 0, stream, p);             \
        } else if (gn) hipLaunchKernelGGL((conv_halo_h3<__VA_ARGS__, 1, false>), dim3(tiles), dim3(NT_), 0, stream, p);      \
        else hipLaunchKernelGGL((conv_halo_h3<__VA_ARGS__, 0, false>), dim3(tiles), dim3(NT_), 0, stream, p);                \
    } while (0)
#define V2A_H3_LAUNCH(...) V2A_H3_LAUNCH_NT(512, __VA_ARGS__)
    if (Cout % 256 == 0) {
        const int tiles = N * p.tiles_img * (Cout / 256);
        V2A_H3_LAUNCH(2, 4, 4, 2, 4);                                                                  // 256 x 256, ring of 4 x 16 KB
    } else if (H % 32 == 0) {
        p.tiles_img = (H / 32) * (W / 16);
        const int tiles = N * p.tiles_img * (Cout / 128);
        V2A_H3_LAUNCH(4, 2, 4, 2, 4);                                                                  // 512 x 128, ring of 4 x 8 KB
    } else {
        const int tiles = N * p.tiles_img * (Cout / 128);
        V2A_H3_LAUNCH(4, 2, 2, 2, 4);                                                                  // 256 x 128, ring of 4 x 8 KB
    }
#undef V2A_H3_LAUNCH
#undef V2A_H3_LAUNCH_NT
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

// x [N, H, W, C] (ups: the conv runs over the nearest-x2 upsampled [N, 2H, 2W, C]); bf16 in / bf16 out; bias fp32 [Cout]; rowvec fp32 [M / rows_per_batch][Cout]; residual bf16 [M][Cout]; stats fp32 [M/64][2][Cout].
int v2a_conv2d_fwd_h3(const void* x, const void* w_packed, const float* bias, const float* rowvec, const void* residual, void* y,
                      const void* zeros, int N, int H, int W, int C, int Cout, int ups, int rows_per_batch, float* stats, hipStream_t stream) {
    return conv_h3_launch(x, nullptr, C, nullptr, 1, 0, w_packed, bias, rowvec, residual, y, zeros, N, H, W, C, Cout, ups, rows_per_batch,
                          stats, stream);
}

// The same conv over act(GroupNorm([x | x2])): the normalisation is applied to the input halo while it sits in LDS, from the
// per-(sample, channel) scale / shift table `gn_ab` [N / gn_frames][2][C] that v2a_groupnorm_prep_h leaves (gn_frames images of the
// N belong to one GroupNorm sample; act = 0 none / 1 SiLU).  x [N,H,W,C1], x2 [N,H,W,C-C1] or null; C <= 1024, C1 % 32 == 0.
int v2a_conv2d_fwd_h3_gn(const void* x, const void* x2, int C1, const float* gn_ab, int gn_frames, int act, const void* w_packed,
                         const float* bias, const float* rowvec, const void* residual, void* y, const void* zeros, int N, int H, int W, int C,
                         int Cout, int rows_per_batch, float* stats, hipStream_t stream) {
    if (!gn_ab || C > 1024 || (x2 && (C1 <= 0 || C1 >= C || C1 % 32)) || (act != 0 && act != ACT_SILU) || gn_frames <= 0 || N % gn_frames)
        return V2A_ERR_ARG;
    return conv_h3_launch(x, x2, C1, gn_ab, gn_frames, act, w_packed, bias, rowvec, residual, y, zeros, N, H, W, C, Cout, 0, rows_per_batch,
                          stats, stream);
}

// 1 when v2a_conv2d_fwd_t3 takes this problem: the temporal tap of the factorised Conv3d seen as a 2-d conv over [B, F, HW, C] with
// a 3 x 1 filter, stride 1, pad (1, 0); F = 7 (Libero: seven predicted frames), HW % 64 == 0, C % 32 == 0, Cout % 128 == 0.
int v2a_conv2d_t3_eligible(int N, int H, int W, int C, int Cout, int KH, int KW, int sh, int sw, int ph, int pw, int ups, int C2) {
    if (KH != 3 || KW != 1 || sh != 1 || sw != 1 || ph != 1 || pw != 0 || C2 || ups) return 0;
    if (H != 7 || W % 64 || C % 32 || Cout % 128) return 0;
    if ((long)N * (W / 64) * (Cout / 128) < 208) return 0;
    if ((double)N * H * W * C >= 4294967296.0 || (double)H * W * C >= 1073741824.0 || (double)Cout * 3 * C >= 1073741824.0) return 0;
    return 1;
}

// x [B, F, HW, C] bf16; w_packed [Cout][3][C] bf16; y [B, F, HW, Cout] bf16; bias / rowvec fp32; residual bf16; stats fp32 [M/64][2][Cout].
int v2a_conv2d_fwd_t3(const void* x, const void* w_packed, const float* bias, const float* rowvec, const void* residual, void* y,
                      const void* zeros, int B, int F, int HW, int C, int Cout, int rows_per_batch, float* stats, hipStream_t stream) {
    if (!x || !w_packed || !zeros || !y || B <= 0) return V2A_ERR_ARG;
    if (!v2a_conv2d_t3_eligible(B, F, HW, C, Cout, 3, 1, 1, 1, 1, 0, 0, 0)) return V2A_ERR_ARG;
    if ((((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)zeros | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)bias | (uintptr_t)rowvec) & 15) != 0)
        return V2A_ERR_ARG;
    ConvDescT3 p;
    p.x = (const uint16_t*)x; p.w = (const uint16_t*)w_packed; p.bias = bias; p.rowvec = rowvec; p.residual = (const uint16_t*)residual;
    p.y = (uint16_t*)y; p.stats = stats; p.zeros = (const uint16_t*)zeros;
    p.B = B; p.HW = HW; p.C = C; p.Cout = Cout; p.K = 3 * C;
    p.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
    p.tiles_b = HW / 64;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipGetDevice(&dev);
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (ncu <= 0) ncu = 256;
    }
    const int total = B * p.tiles_b * (Cout / 128);
    if (g_v2a_half_f16) hipLaunchKernelGGL((conv_frames_h3<7, true>), dim3(total < ncu ? total : ncu), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL((conv_frames_h3<7, false>), dim3(total < ncu ? total : ncu), dim3(512), 0, stream, p);      // one persistent workgroup per CU
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

}  // extern "C"
