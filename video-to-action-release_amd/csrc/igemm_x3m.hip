// 3 x 3 / stride 1 / pad 1 convolution over SMALL SQUARE MAPS (32 x 32, 16 x 16, 8 x 8, 4 x 4) on three bf16 planes per fp32 operand: the
// forward and data-gradient convs of the policy's two ResNet-18 image encoders at batch 64 (diffusion_policy/model/vision/
// multi_image_obs_encoder.py, torchvision BasicBlock with GroupNorm; 52 launches per train step), in the phase structure of
// csrc/igemm_x3p.hip conv_patch_x3.
//
// Every one of these layers is the same 4.8 GFLOP (64 images: 64 ch x 32^2, 128 x 16^2, 256 x 8^2, 512 x 4^2) -- 12 us of MFMA time on
// the whole chip -- and conv_halo_x3 (csrc/igemm_h.hip: 128 pixels x 64 channels per 256-thread workgroup, a weight tile split per
// (chunk, tap) step of 24 MFMAs per wave, two barriers around every halo) takes 33 ... 51 us for them (profiles/r06_policy_kernel_stats.csv,
// MFMA utilisation 0.30 - 0.32).  Here a 512-thread workgroup owns 256 CONSECUTIVE output rows (8 rows of a 32-wide map, one 16 x 16 map,
// four 8 x 8 maps, sixteen 4 x 4 maps) x 64 output channels, wave = 64 rows x 32 channels, and walks its slice of the reduction
// (split over 32-channel chunks so that ~256 workgroups exist: 1 / 2 / 4 / 8 slabs) in PHASES (chunk c, 16-channel half h, filter row kh):
// three taps = 36 MFMAs per wave and barrier;
//   * weights: registers -> split -> LDS stage (phase + 1) & 1 behind the phase's first tap, then the request for phase + 2; the first TWO
//     phases are requested together in the prologue (a workgroup lives for 12 phases: one exposed round trip, not two);
//   * halo: the zero-padded (rows + 2) x (width + 2) window of every map of the tile, three bf16 plane images of 64-B slot rows; the
//     16-B pieces that carry channels 16 h .. + 15 form "half h": while the three phases of (c, h) read half h, the next (c, h) is requested
//     (filter row 0) and split into the other half (filter row 2);
//   * half as many weight conversions and weight bytes through L1 per MAC as conv_halo_x3 (256-row tiles), a third of its barriers per MAC.
// LDS slot pitch and the MFMA-row -> pixel permutation are chosen per map width so that each 16-lane group of a ds_read_b128
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}: MI355X_MICROARCH.md, LDS) touches every bank once under all nine tap shifts
// (pitch 34 / 18 / 12 / 6 slots, 4 pad slots behind each 4 x 4 map; exhaustive check: tools/probes/r6/x3m_banks.py).
// Epilogue straight from the accumulators: split-K slab (consumer: conv_splitk_reduce_h or the GroupNorm launch that sums slabs), or
// bias + optional fp32 residual.  Arithmetic: conv_igemm_f32x3's six plane products, smallest first (fp32-equivalent).
#include "common.h"
#include "x3t.h"
#include <stdlib.h>

typedef __attribute__((address_space(1))) f32x4 gf32x4_x3m;
typedef __attribute__((ext_vector_type(8))) __bf16 bfx8_x3m;

struct ConvDescX3M {
    const float* x;          // [N, S, S, C]
    const float* w;          // [Cout][3][3][C]
    const float* bias;       // [Cout] or null            (single-slab launches only)
    const float* residual;   // [N, S, S, Cout] or null   (single-slab launches only)
    float* y;                // [N, S, S, Cout]
    float* partial;          // [splitk][M][Cout] slabs (splitk > 1)
    const float* zeros;
    int M, C, Cout, K, splitk, chunks_per_split;
};

template <int OWC>
struct MX3 {
    static constexpr int BM = 256, BN = 64, NT = 512;
    static constexpr int PIX = OWC * OWC;
    static constexpr int SP = PIX >= BM ? 1 : BM / PIX;                  // maps per tile
    static constexpr int PH = PIX >= BM ? BM / OWC : OWC;                // map rows per tile and map
    static constexpr int P = OWC == 32 ? 34 : OWC == 16 ? 18 : OWC == 8 ? 12 : 6;      // slot pitch of a halo row
    static constexpr int HS = (PH + 2) * P + (OWC == 4 ? 4 : 0);         // slots per map
    static constexpr int NS = SP * HS;                                   // 340 / 324 / 480 / 640
    static constexpr int PHB = (NS + 1) * 64;                            // bytes of one plane of the halo image (+ a dump slot: threads without a
                                                                         // pixel store there, never read -- no branches in the phase body)
    static constexpr int TAPB = BN * 32, PWB = 3 * TAPB, WST = 3 * PWB;  // weight stage: [plane][tap kw][64 rows x 32 B]
    static constexpr int W_OFF = 3 * PHB;
    static constexpr int SMEM = W_OFF + 2 * WST;
    static constexpr int AJ = (NS * 4 + NT - 1) / NT;                    // float4 per thread of a halo half
};

// MFMA row m of a 32-row sub-tile carries pixel x3m_perm<OWC>(m) of the sub-tile's 32 consecutive pixels (see the header)
template <int OWC>
__device__ __forceinline__ int x3m_perm(int m) {
    if constexpr (OWC == 32) return m;
    else if constexpr (OWC == 16) {
        const int qd = m >> 2;
        return ((__builtin_popcount(qd) & 1) << 4) | ((qd >> 1) << 2) | (m & 3);
    } else if constexpr (OWC == 8) {
        // quad -> (row, column half) position: {0, 3, 5, 1, 6, 2, 4, 7}
        return (int)((0x74261530u >> ((m >> 2) * 4)) & 7u) * 4 + (m & 3);
    } else {
        // quad -> (map, row) position: {0, 1, 3, 2, 5, 4, 6, 7}
        return (int)((0x76452310u >> ((m >> 2) * 4)) & 7u) * 4 + (m & 3);
    }
}

template <int OWC>
__global__ __launch_bounds__(512, 1) void conv_maps_x3(const ConvDescX3M p) {
    typedef MX3<OWC> G;
    constexpr int BM = G::BM, BN = G::BN, NT = G::NT, P = G::P, HS = G::HS, NS = G::NS, PHB = G::PHB, TAPB = G::TAPB, PWB = G::PWB,
                  WST = G::WST, W_OFF = G::W_OFF, AJ = G::AJ;
    static_assert(G::SMEM <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(128))) unsigned char smem[G::SMEM];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = p.Cout / BN;
    const int lin = blockIdx.x;
    const int tm = lin / tiles_n;
    const int n0 = (lin - tm * tiles_n) * BN, m0 = tm * BM;
    const int split = blockIdx.y;
    const int nchunks = p.C >> 5;
    const int ck_begin = split * p.chunks_per_split;
    const int ck_end = min(nchunks, ck_begin + p.chunks_per_split);
    const float* zsrc = p.zeros;
    const int img0 = m0 / G::PIX, row0 = (m0 % G::PIX) / OWC;            // (row0 = 0 when the tile holds whole maps)

    // ---- halo loader: item q = j * 512 + tid -> slot q >> 2, float4 q & 3 of the half's 16 channels
    int a_dst[AJ];                                           // LDS byte offset for half 0 (half 1: ^ 32); no such slot / pad slot: the dump slot
    uint32_t a_off[AJ];                                      // element offset of the slot's pixel in x (+ float4 index); ~0: zero line
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int q = j * NT + tid;
        const int slot = q >> 2, c4 = q & 3;
        const int sp = slot / HS, rem = slot - sp * HS;
        const int hy = rem / P, hx = rem - hy * P;
        const bool valid = slot < NS && hy < G::PH + 2 && hx < OWC + 2;
        const int ih = row0 + hy - 1, iw = hx - 1;
        const bool ok = valid && (unsigned)ih < (unsigned)OWC && (unsigned)iw < (unsigned)OWC;
        a_dst[j] = valid ? slot * 64 + ((((c4 >> 1) ^ ((slot >> 2) & 3)) << 4) | ((c4 & 1) << 3)) : NS * 64;
        a_off[j] = ok ? ((uint32_t)((img0 + sp) * OWC + ih) * (uint32_t)OWC + (uint32_t)iw) * (uint32_t)p.C + (uint32_t)(c4 * 4) : 0xffffffffu;
    }
    int la_c = ck_begin, la_h = 0;                           // the (chunk, half) being LOADED: one period ahead of the one computed
    f32x4 ra[AJ];
    // (no branches around the loads: past the end of the slice and outside the map they read the zero line -- a conditional load makes
    // the compiler's wait-count bookkeeping fall back to vmcnt(0) everywhere)
    auto issue_a = [&]() {
        const bool live = la_c < ck_end;
        const float* xb = p.x + la_c * 32 + la_h * 16;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const float* g = (live && a_off[j] != 0xffffffffu) ? xb + a_off[j] : zsrc;
            ra[j] = *(const gf32x4_x3m*)(uint64_t)g;
        }
        la_h ^= 1;
        la_c += la_h == 0 ? 1 : 0;
    };
    auto store_a = [&](int hd, int j0, int j1) {            // items [j0, j1) of the thread (the conversion work is spread over two phases)
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            if (j < j0 || j >= j1) continue;
            uint32_t h0, m0_, l0, h1, m1, l1;
            v2a_split3x2(ra[j][0], ra[j][1], h0, m0_, l0);
            v2a_split3x2(ra[j][2], ra[j][3], h1, m1, l1);
            unsigned char* d = smem + (a_dst[j] ^ (hd << 5));      // half 1: the row's other two 16-B pieces
            *reinterpret_cast<uint2*>(d) = uint2{h0, h1};
            *reinterpret_cast<uint2*>(d + PHB) = uint2{m0_, m1};
            *reinterpret_cast<uint2*>(d + 2 * PHB) = uint2{l0, l1};
        }
    };
    // ---- weight loader of a phase (c, h, kh), 3 taps x 64 rows x 16 channels over 512 threads without a branch: a float4 of tap tid >> 8
    // (row (tid & 255) >> 2, channels 4 (tid & 3) ..) and a float2 of tap 2 (row tid >> 3, channels 2 (tid & 7) ..).  LDS: 32-B rows in pairs,
    // the pair's four 16-B pieces XOR-ed with (n >> 3) & 3 (conv_patch_x3's layout)
    typedef __attribute__((ext_vector_type(2))) float f32x2_x3m;
    typedef __attribute__((address_space(1))) f32x2_x3m gf32x2_x3m;
    const int wr_ld = (tid & 255) >> 2, wc4 = tid & 3;
    const int w_tap0 = tid >> 8;
    const uint32_t w_src = (uint32_t)(n0 + wr_ld) * (uint32_t)p.K + (uint32_t)wc4 * 4u;
    const int w_dst = w_tap0 * TAPB + (wr_ld >> 1) * 64 + ((((((wr_ld & 1) << 1) | (wc4 >> 1)) ^ ((wr_ld >> 3) & 3)) << 4) | ((wc4 & 1) << 3));
    const int wr2 = tid >> 3, we2 = tid & 7;
    const uint32_t w_src2 = (uint32_t)(n0 + wr2) * (uint32_t)p.K + (uint32_t)we2 * 2u;
    const int w_dst2 = 2 * TAPB + (wr2 >> 1) * 64 + ((((((wr2 & 1) << 1) | (we2 >> 2)) ^ ((wr2 >> 3) & 3)) << 4) | ((we2 & 3) << 2));
    struct WSet { f32x4 a; f32x2_x3m b; };
    int lw_c = ck_begin, lw_h = 0, lw_kh = 0;
    auto issue_w = [&](WSet& r) {
        const bool live = lw_c < ck_end;
        const float* wb = p.w + (size_t)(lw_kh * 3) * p.C + lw_c * 32 + lw_h * 16;
        const float* g0 = live ? wb + w_src + (size_t)w_tap0 * p.C : zsrc;
        const float* g1 = live ? wb + w_src2 + (size_t)2 * p.C : zsrc;
        r.a = *(const gf32x4_x3m*)(uint64_t)g0;
        r.b = *(const gf32x2_x3m*)(uint64_t)g1;
        if (++lw_kh == 3) {
            lw_kh = 0;
            lw_h ^= 1;
            lw_c += lw_h == 0 ? 1 : 0;
        }
    };
    auto store_w = [&](const WSet& r, int stage) {
        unsigned char* wbs = smem + W_OFF + stage * WST;
        uint32_t h0, m0_, l0, h1, m1, l1, h2, m2, l2;
        v2a_split3x2(r.a[0], r.a[1], h0, m0_, l0);
        v2a_split3x2(r.a[2], r.a[3], h1, m1, l1);
        v2a_split3x2(r.b[0], r.b[1], h2, m2, l2);
        unsigned char* d = wbs + w_dst;
        *reinterpret_cast<uint2*>(d) = uint2{h0, h1};
        *reinterpret_cast<uint2*>(d + PWB) = uint2{m0_, m1};
        *reinterpret_cast<uint2*>(d + 2 * PWB) = uint2{l0, l1};
        unsigned char* d2 = wbs + w_dst2;
        *reinterpret_cast<uint32_t*>(d2) = h2;
        *reinterpret_cast<uint32_t*>(d2 + PWB) = m2;
        *reinterpret_cast<uint32_t*>(d2 + 2 * PWB) = l2;
    };

    WSet rw0, rw1, rw2;                                      // three weight sets: phase g + 3 is requested while g + 1 is split, g + 2 in flight

    // ---- compute mapping: wave = (64-row group wm, 32-channel group wn); sub-tile i = 32 rows x 32 channels
    const int wm = wid >> 1, wn = (wid & 1) * 32;
    const int lr = lane & 31, lk = lane >> 5;
    int slot0[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pr = wm * 64 + i * 32 + x3m_perm<OWC>(lr);
        const int sp = pr / (G::PH * OWC), q = pr - sp * (G::PH * OWC);
        const int py = q / OWC, px = q - py * OWC;
        slot0[i] = sp * HS + py * P + px;
    }
    const int b_off = ((wn + lr) >> 1) * 64 + (((((lr & 1) << 1) | lk) ^ ((lr >> 3) & 3)) << 4);

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // one tap (kh, kw) of a phase = nine operand fragments + 12 MFMAs (the two accumulators' chains interleaved).  The fragments of tap
    // kw + 1 are requested BEFORE the MFMAs of tap kw are issued (two fragment sets): a wave's LDS round trips run under its own MFMAs
    // instead of relying on the other wave of the SIMD being out of step
    struct Frag { bfx8_x3m a[2][3], b[3]; };
    auto frag_load = [&](Frag& f, int h, int stage, int kh, int kw) {
        const unsigned char* wb = smem + W_OFF + stage * WST + kw * TAPB + b_off;
        const int kp = (h << 1) | lk;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int sl = slot0[i] + kh * P + kw;
            const unsigned char* ab = smem + sl * 64 + ((kp ^ ((sl >> 2) & 3)) << 4);
#pragma unroll
            for (int q = 0; q < 3; ++q) f.a[i][q] = *reinterpret_cast<const bfx8_x3m*>(ab + q * PHB);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) f.b[q] = *reinterpret_cast<const bfx8_x3m*>(wb + q * PWB);
    };
    auto frag_mfma = [&](const Frag& f) {
#define V2A_X3M_PROD(QA, QB)                                                                        \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                   \
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][QA], f.b[QB], acc[i], 0, 0, 0);
        V2A_X3M_PROD(2, 0)      // lo  * hi
        V2A_X3M_PROD(0, 2)      // hi  * lo
        V2A_X3M_PROD(1, 1)      // mid * mid
        V2A_X3M_PROD(1, 0)      // mid * hi
        V2A_X3M_PROD(0, 1)      // hi  * mid
        V2A_X3M_PROD(0, 0)      // hi  * hi
#undef V2A_X3M_PROD
    };
    Frag f0, f1;

    // ---- prologue: halo (first chunk, half 0) and weight phases 0, 1, 2 requested together; phase 0 into LDS, its first tap requested
    issue_a();
    issue_w(rw0);
    issue_w(rw1);
    issue_w(rw2);
    store_a(0, 0, AJ);
    store_w(rw0, 0);
    __syncthreads();
    frag_load(f0, 0, 0, 0, 0);

    // A phase's barrier sits between its second and third tap: the operands of phase + 1 (stored behind the first tap) are visible from
    // there on, so the first tap of phase + 1 is requested before the MFMAs of this phase's third tap -- every fragment request has a tap
    // of MFMAs to land.  (All reads of this phase's weight stage / halo half are complete at that barrier (lgkmcnt(0) in front of it):
    // the stores of phase + 1, which re-use the stage, come after it.)  Fragment sets alternate: on entry X holds tap 0 of the phase.
    constexpr int AJ1 = (AJ + 1) / 2;                        // halo items converted in the group's second phase (the rest in its third)
    constexpr int AV1 = (22 * AJ1 + 11) / 12, AV2 = (22 * (AJ - AJ1) + 11) / 12;
    for (int c = ck_begin; c < ck_end; ++c) {
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(slot0[i]));
    // (the 12 MFMAs of a tap with NV_ VALU instructions behind each -- conversions run in the MFMAs' issue shadow, at most five
    // single-issue instructions fit per 32-cycle MFMA: MI355X_MICROARCH.md -- then the NW_ LDS stores)
#define V2A_X3M_MIX(NV_, NW_)                                                                                                   \
    _Pragma("unroll") for (int g_ = 0; g_ < 12; ++g_) {                                                                          \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x002, NV_, 0);                                                                    \
    }                                                                                                                           \
    __builtin_amdgcn_sched_group_barrier(0x200, NW_, 0);
#define V2A_X3M_SET(I_) ((I_) % 3 == 0 ? rw0 : (I_) % 3 == 1 ? rw1 : rw2)
#define V2A_X3M_PHASE(H_, KH_, X_, Y_)                                                                                          \
    {                                                                                                                           \
        constexpr int idx_ = (H_) * 3 + (KH_);          /* six phases per chunk: stage = idx & 1, register set = idx % 3 */          \
        constexpr int stage_ = idx_ & 1;                                                                                        \
        constexpr int nh_ = (KH_) == 2 ? ((H_) ^ 1) : (H_), nkh_ = (KH_) == 2 ? 0 : (KH_) + 1;   /* the next phase */                 \
        frag_load(Y_, H_, stage_, KH_, 1);                                                                                      \
        issue_w(V2A_X3M_SET(idx_));                      /* phase + 3 into the set whose phase went into LDS one phase ago */       \
        if ((KH_) == 0) issue_a();                       /* the halo of the next (chunk, half) ... */                             \
        __builtin_amdgcn_sched_barrier(0);               /* (requests stay in front of the MFMAs they hide under) */                \
        frag_mfma(X_);                                                                                                          \
        store_w(V2A_X3M_SET(idx_ + 1), stage_ ^ 1);      /* phase + 1: its stage's readers finished before the last barrier */       \
        V2A_X3M_MIX(3, 9)                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
        frag_load(X_, H_, stage_, KH_, 2);                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
        frag_mfma(Y_);                                                                                                          \
        /* ... into the half whose readers finished two barriers ago, the conversions spread over the group's last two phases */  \
        if ((KH_) == 1) { store_a((H_) ^ 1, 0, AJ1); V2A_X3M_MIX(AV1, 3 * AJ1) }                                                    \
        if ((KH_) == 2) { store_a((H_) ^ 1, AJ1, AJ); V2A_X3M_MIX(AV2, 3 * (AJ - AJ1)) }                                            \
        __syncthreads();                                                                                                        \
        frag_load(Y_, nh_, stage_ ^ 1, nkh_, 0);                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
        frag_mfma(X_);                                                                                                          \
    }
        V2A_X3M_PHASE(0, 0, f0, f1) V2A_X3M_PHASE(0, 1, f1, f0) V2A_X3M_PHASE(0, 2, f0, f1)
        V2A_X3M_PHASE(1, 0, f1, f0) V2A_X3M_PHASE(1, 1, f0, f1) V2A_X3M_PHASE(1, 2, f1, f0)
#undef V2A_X3M_PHASE
#undef V2A_X3M_SET
#undef V2A_X3M_MIX
    }

    // ---- epilogue: lane = output column n0 + wn + lr; register r = row (r & 3) + 8 (r >> 2) + 4 lk of sub-tile i
    const int ncol = n0 + wn + lr;
    if (p.splitk > 1) {
        float* slab = p.partial + (size_t)split * p.M * p.Cout;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + x3m_perm<OWC>((r & 3) + 8 * (r >> 2) + 4 * lk);
                slab[(size_t)m * p.Cout + ncol] = acc[i][r];
            }
    } else {
        const float colb = p.bias ? p.bias[ncol] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + x3m_perm<OWC>((r & 3) + 8 * (r >> 2) + 4 * lk);
                const size_t o = (size_t)m * p.Cout + ncol;
                float v = acc[i][r] + colb;
                if (p.residual) v += p.residual[o];
                p.y[o] = v;
            }
    }
}

// slabs (over 32-channel chunks) of conv_maps_x3: about 256 workgroups of >= 1 chunk, at most 8 slabs
int conv_maps_x3_split(int M, int Cout, int C) {
    const int tiles = (M / 256) * (Cout / 64), nchunks = C / 32;
    int s = 1;
    if (tiles > 0 && tiles < 208) {
        s = (256 + tiles - 1) / tiles;
        if (s > 8) s = 8;
        if (s > nchunks) s = nchunks;
        if (s < 1) s = 1;
    }
    const int cps = (nchunks + s - 1) / s;
    return (nchunks + cps - 1) / cps;
}

// 1 when conv_maps_x3 takes the problem: N square maps of S = 32 / 16 / 8 / 4, whole 256-row tiles, C % 32 == 0, Cout % 64 == 0 and enough
// (tile, slab) workgroups to cover most of the chip (smaller launches stay on conv_halo_x3, which cuts 128-row tiles)
int conv_maps_x3_eligible(int N, int S, int C, int Cout) {
    if (!(S == 32 || S == 16 || S == 8 || S == 4) || N <= 0 || C % 32 || Cout % 64) return 0;
    const long M = (long)N * S * S;
    if (M % 256 || M >= 2147483647L) return 0;
    if ((double)M * C >= 4294967296.0 || (double)Cout * 9 * C >= 4294967296.0) return 0;
    const int s = conv_maps_x3_split((int)M, Cout, C);
    const long wgs = (M / 256) * (Cout / 64) * s;
    return wgs >= 192 ? 1 : 0;
}

int conv_maps_x3_launch(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, float* partial,
                        const void* zeros, int N, int S, int C, int Cout, int splitk, hipStream_t stream) {
    if (!x || !w_packed || !zeros || !y || !conv_maps_x3_eligible(N, S, C, Cout)) return V2A_ERR_ARG;
    if ((((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)zeros) & 15) != 0 ||
        (((uintptr_t)y | (uintptr_t)residual | (uintptr_t)bias | (uintptr_t)partial) & 3) != 0)
        return V2A_ERR_ARG;
    const int nchunks = C / 32;
    if (splitk < 1 || splitk > nchunks || (splitk > 1 && !partial)) return V2A_ERR_ARG;
    ConvDescX3M p;
    p.x = x; p.w = w_packed; p.bias = bias; p.residual = residual; p.y = y; p.partial = partial; p.zeros = (const float*)zeros;
    p.M = N * S * S; p.C = C; p.Cout = Cout; p.K = 9 * C; p.splitk = splitk;
    p.chunks_per_split = (nchunks + splitk - 1) / splitk;
    if ((p.chunks_per_split * (splitk - 1)) >= nchunks) return V2A_ERR_ARG;         // an empty slab would go unwritten
    const dim3 grid((p.M / 256) * (Cout / 64), splitk);
    if (S == 32) hipLaunchKernelGGL(conv_maps_x3<32>, grid, dim3(512), 0, stream, p);
    else if (S == 16) hipLaunchKernelGGL(conv_maps_x3<16>, grid, dim3(512), 0, stream, p);
    else if (S == 8) hipLaunchKernelGGL(conv_maps_x3<8>, grid, dim3(512), 0, stream, p);
    else hipLaunchKernelGGL(conv_maps_x3<4>, grid, dim3(512), 0, stream, p);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}
