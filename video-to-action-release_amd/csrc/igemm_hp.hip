// Upsample (nearest x2, guided_diffusion/unet.py:105-115) + 3 x 3 conv of the 16-bit video UNet as FOUR 2 x 2 convs over the SOURCE map,
// one per parity class of the output pixel (2a + ph, 2b + pw): the 16-bit counterpart of conv_patch_x3<.., KS = 2> (csrc/igemm_x3p.hip).
// Behind the upsample two of the three filter rows (columns) read the same source row (column), so their weights are summed in advance
// ([4 classes][Cout][2][2][C], v2a_pack_weight_ups4 + a cast) and an output needs 4 of the 9 products; conv_halo_h3's `ups` form gathers
// the 4 x duplicated halo of the up-sampled map and multiplies all nine taps.
//
// A 512-thread workgroup owns 32 x 16 pixels (a, b) of one class (512 output rows, scattered with stride 2) x 128 output channels; wave =
// 128 rows x 64 channels (8 accumulators).  The reduction runs in phases (32-channel chunk c, filter row kh) of two taps: 32 MFMAs per wave
// and barrier.  Operands are REGISTER-staged (16-B global loads -> ds_write_b128; no conversion): behind a phase's first tap the registers
// holding the next phase's weight tile go into the other LDS stage and the loads of the phase after that are issued; the 33 x 17 source
// window of chunk c + 1 is requested in phase (c, 0) and stored in phase (c, 1) into the other window buffer.  Persistent: one workgroup
// per CU walks tiles lin, lin + G, ... as one phase stream (the next tile's first window and weight phases are in LDS when a tile's
// epilogue starts).  64-B LDS rows, 16-B pieces XOR-ed with (row >> 2) & 3; MFMA rows permuted so that a 16-lane group of a ds_read_b128
// reads one patch row (lds_group_perm3).  Epilogue: bias, round to the storage type, 16-B stores through 32 KB of LDS staging.
#include "common.h"
#include "x3t.h"

typedef unsigned int u32x4_hp __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4_hp guint4_hp;
typedef __attribute__((ext_vector_type(8))) __bf16 bfx8_hp;

struct ConvDescHP {
    const uint16_t* x;       // SOURCE [N, H/2, W/2, C] (16-bit)
    const uint16_t* w;       // [4][Cout][2][2][C] (16-bit)
    const float* bias;       // [Cout] or null
    uint16_t* y;             // [N, H, W, Cout]
    const uint16_t* zeros;
    int N, H, W, C, Cout, K, tiles_x, tiles_img;      // H, W: the OUTPUT map; tiles: 32 x 16 patches of the source map
};

__device__ __forceinline__ int xcd_remap_hp(int bid, int nblk) {
    int q = nblk >> 3, r = nblk & 7;
    int xcd = bid & 7, slot = bid >> 3;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}
__device__ __forceinline__ int patch16_perm_hp(int m) {
    const int qd = m >> 2;
    return ((__builtin_popcount(qd) & 1) << 4) | ((qd >> 1) << 2) | (m & 3);
}

template <bool F16>
__global__ __launch_bounds__(512, 1) void conv_patch_h_ups4(const ConvDescHP p) {
    constexpr int BN = 128, NT = 512, PH = 32;               // 32 x 16 pixel patch
    constexpr int HWD = 17, HROWS = (PH + 1) * HWD;          // source window: 33 x 17 slots of 64 B (32 channels)
    constexpr int ABUF = ((HROWS * 64 + 1023) / 1024) * 1024;
    constexpr int TAPB = BN * 64, WST = 2 * TAPB;            // weight stage: [tap kw][128 rows x 64 B]
    constexpr int W_OFF = 2 * ABUF, ST_OFF = W_OFF + 2 * WST;
    constexpr int SMEM = ST_OFF + 8 * 32 * 64 * 2;           // + per-wave staging: 32 rows x 64 channels, 16-bit
    constexpr int AJ = (HROWS * 4 + NT - 1) / NT;            // 16-B pieces per thread of a window
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(128))) unsigned char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = p.Cout / BN;
    const int total = p.N * 4 * p.tiles_img * tiles_n;
    const int G = gridDim.x;
    int lin = xcd_remap_hp(blockIdx.x, G);
    if (lin >= total) return;
    const int nchunks = p.C >> 5;
    const int srcH = p.H >> 1, srcW = p.W >> 1;
    auto tile_of = [&](int l, int& img, int& cls, int& ty, int& tx) {
        const int tm = l / tiles_n;
        const int per_img = 4 * p.tiles_img;
        img = tm / per_img;
        const int rem = tm - img * per_img;
        cls = rem / p.tiles_img;
        const int t = rem - cls * p.tiles_img;
        ty = t / p.tiles_x;
        tx = t - ty * p.tiles_x;
    };

    // ---- window loader: item q = j * 512 + tid -> slot q >> 2, 16-B piece q & 3 (8 channels)
    int a_dst[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int q = j * NT + tid;
        const int slot = q >> 2, pc = q & 3;
        a_dst[j] = slot < HROWS ? slot * 64 + ((pc ^ ((slot >> 2) & 3)) << 4) : -1;
    }
    int la_lin = lin, la_c = 0;
    uint32_t a_off[AJ];
    auto a_tile_setup = [&](int l) {
        int img, cls, ty, tx;
        tile_of(l, img, cls, ty, tx);
        const int oy = (cls >> 1) - 1, ox = (cls & 1) - 1;   // window origin: one pixel up / left of the patch for the upper / left classes
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int slot = (j * NT + tid) >> 2;           // (recomputed per tile: two registers per piece saved in the main loop)
            const int hy = slot / HWD, hx = slot - hy * HWD;
            const int ih = ty * PH + hy + oy, iw = tx * 16 + hx + ox;
            const bool ok = a_dst[j] >= 0 && (unsigned)ih < (unsigned)srcH && (unsigned)iw < (unsigned)srcW;
            a_off[j] = ok ? ((uint32_t)(img * srcH + ih) * (uint32_t)srcW + (uint32_t)iw) * (uint32_t)p.C + (uint32_t)((tid & 3) * 8) : 0xffffffffu;
        }
    };
    a_tile_setup(la_lin);
    u32x4_hp ra[AJ], rw[2];
    int abase = 0;                                           // parity of the window buffer that holds the current tile's chunk 0
    const uint16_t* zsrc = p.zeros;
    auto issue_a = [&]() {
        const bool live = la_lin < total;
        const uint16_t* xb = p.x + la_c * 32;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const uint16_t* g = (live && a_off[j] != 0xffffffffu) ? xb + a_off[j] : zsrc;
            ra[j] = *(const guint4_hp*)(uint64_t)g;
        }
        if (++la_c == nchunks) {
            la_c = 0;
            la_lin += G;
            if (la_lin < total) a_tile_setup(la_lin);
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int j = 0; j < AJ; ++j)
            if (a_dst[j] >= 0) *reinterpret_cast<u32x4_hp*>(smem + buf * ABUF + a_dst[j]) = ra[j];
    };
    // ---- weight loader of a phase (c, kh): tap kw = t, rows (tid >> 2) and (tid >> 2) + 64... : 128 rows x 4 pieces = 512 items per tap
    const int wn_ld = tid >> 2, wpc = tid & 3;
    const uint32_t w_src = (uint32_t)wn_ld * (uint32_t)p.K + (uint32_t)wpc * 8u;
    const int w_dst = wn_ld * 64 + ((wpc ^ ((wn_ld >> 2) & 3)) << 4);
    int lw_lin = lin, lw_c = 0, lw_kh = 0;
    auto w_base_of = [&](int l) -> size_t {
        int img, cls, ty, tx;
        tile_of(l, img, cls, ty, tx);
        return ((size_t)cls * p.Cout + (size_t)(l % tiles_n) * BN) * p.K;
    };
    size_t lw_base = w_base_of(lin);
    auto issue_w = [&](u32x4_hp (&r)[2]) {
        const bool live = lw_lin < total;
        const uint16_t* wb = p.w + lw_base + (size_t)(lw_kh * 2) * p.C + lw_c * 32 + w_src;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint16_t* g = live ? wb + (size_t)t * p.C : zsrc;
            r[t] = *(const guint4_hp*)(uint64_t)g;
        }
        if (++lw_kh == 2) {
            lw_kh = 0;
            if (++lw_c == nchunks) {
                lw_c = 0;
                lw_lin += G;
                if (lw_lin < total) lw_base = w_base_of(lw_lin);
            }
        }
    };
    auto store_w = [&](const u32x4_hp (&r)[2], int stage) {
#pragma unroll
        for (int t = 0; t < 2; ++t) *reinterpret_cast<u32x4_hp*>(smem + W_OFF + stage * WST + t * TAPB + w_dst) = r[t];
    };

    // ---- compute mapping: wave = (128-row group wm: patch rows 8 wm .. + 7, 64-channel group wn)
    const int wm = wid >> 1, wn = (wid & 1) * 64;
    const int lr = lane & 31, lk = lane >> 5;
    int slot0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pr = wm * 128 + i * 32 + patch16_perm_hp(lr);
        slot0[i] = (pr >> 4) * HWD + (pr & 15);
    }
    int b_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) b_off[j] = (wn + j * 32 + lr) * 64;
    const int brs = (lr >> 2) & 3;

    f32x16 acc[4][2];
    auto tap = [&](int abuf, int stage, int kh, int kw) {
        const unsigned char* ab = smem + abuf * ABUF;
        const unsigned char* wb = smem + W_OFF + stage * WST + kw * TAPB;
#pragma unroll
        for (int h = 0; h < 2; ++h) {                        // (one k-half at a time: 24 fragment registers instead of 48 -- the kernel sits at the VGPR cap)
            const int kp = (h << 1) | lk;
            bfx8_hp a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sl = slot0[i] + kh * HWD + kw;
                a[i] = *reinterpret_cast<const bfx8_hp*>(ab + sl * 64 + ((kp ^ ((sl >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bfx8_hp*>(wb + b_off[j] + ((kp ^ brs) << 4));
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = v2a_mfma_h<F16>(a[i], b[j], acc[i][j]);
        }
    };

    // ---- prologue: window of chunk 0 and weight phases 0 and 1 of the first tile into LDS, weight phase 2 requested
    issue_a();
    issue_w(rw);
    store_a(0);
    store_w(rw, 0);
    issue_w(rw);
    store_w(rw, 1);
    issue_w(rw);
    __syncthreads();

    for (; lin < total; lin += G) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // invariant at a tile's start: LDS holds the window of its chunk 0 (buffer `abase`: windows alternate buffers along the whole
        // chunk stream, across tiles) and its weight phases 0 and 1; weight phase 2 is in flight into the registers
        for (int c = 0; c < nchunks; ++c) {
            const int abuf = (abase + c) & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(slot0[i]));
            // phase (c, 0)
            tap(abuf, 0, 0, 0);
            if (c > 0) {                                     // (weight phase 1 of a tile went into LDS before its epilogue / in the prologue)
                store_w(rw, 1);
                issue_w(rw);
            }
            issue_a();                                       // the window of the next chunk (or of the next tile's first chunk) ...
            tap(abuf, 0, 0, 1);
            __syncthreads();
            // phase (c, 1)
            tap(abuf, 1, 1, 0);
            store_w(rw, 0);                               // the next chunk's (or tile's) phase 0
            issue_w(rw);
            store_a(abuf ^ 1);                               // ... into the other buffer: its readers (chunk c - 1) passed two barriers ago
            tap(abuf, 1, 1, 1);
            __syncthreads();
        }
        abase = (abase + nchunks) & 1;
        store_w(rw, 1);                                   // the next tile's weight phase 1, BEFORE this tile's stores (see conv_frames_x3)

        // ---- epilogue: bias, round, 16-B stores of 8 channels through the wave's staging rows
        const int n0 = (lin % tiles_n) * BN;
        int img, cls, ty, tx;
        tile_of(lin, img, cls, ty, tx);
        const size_t pix0 = ((size_t)img * p.H + (size_t)(ty * PH * 2 + (cls >> 1))) * p.W + tx * 32 + (cls & 1);
        uint16_t* stg = reinterpret_cast<uint16_t*>(smem + ST_OFF) + wid * 32 * 64;      // [32 rows][64 channels]
        float colb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) colb[j] = p.bias ? p.bias[n0 + wn + j * 32 + lr] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
                    stg[row * 64 + ((j * 32 + lr) ^ ((row & 3) << 3))] = v2a_f2h<F16>(acc[i][j][r] + colb[j]);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // lane -> (row = pass * 8 + lane / 8, channels (lane % 8) * 8 .. + 7)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = ps * 8 + (lane >> 3), c8 = (lane & 7) * 8;
                const uint4 u = *reinterpret_cast<const uint4*>(stg + row * 64 + (c8 ^ ((row & 3) << 3)));
                const int pr = wm * 128 + i * 32 + patch16_perm_hp(row);
                const size_t o = (pix0 + (size_t)((pr >> 4) * 2) * p.W + (pr & 15) * 2) * p.Cout + n0 + wn + c8;
                *reinterpret_cast<uint4*>(p.y + o) = u;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the staging rows are rewritten by the next sub-tile
        }
        issue_w(rw);                                      // the next tile's weight phase 2
        __syncthreads();
    }
}

int conv_patch_h_ups4_eligible(int N, int H, int W, int C, int Cout, int ncu) {
    if (H % 64 || W % 32 || C % 32 || Cout % 128) return 0;
    const long tiles = (long)N * 4 * (H / 64) * (W / 32) * (Cout / 128);
    if (ncu <= 0) ncu = 256;
    const long rounds = (tiles + ncu - 1) / ncu;
    if (tiles < 208 || tiles * 100 < rounds * ncu * 85) return 0;
    if ((double)N * H * W * C >= 4294967296.0 || (double)Cout * 16 * C >= 4294967296.0) return 0;
    return 1;
}

static int hp_ncu() {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (ncu <= 0) ncu = 256;
    }
    return ncu;
}

extern "C" {

// Upsample + 3 x 3 conv of the 16-bit video UNet as four 2 x 2 class convs (format: v2a_set_half_format of the calling thread).
// x = the SOURCE [N, H/2, W/2, C]; w_ups4 = [4][Cout][2][2][C] 16-bit (v2a_pack_weight_ups4, cast); y [N, H, W, Cout]; bias fp32.
int v2a_conv2d_hp_ups4_eligible(int N, int H, int W, int C, int Cout) { return conv_patch_h_ups4_eligible(N, H, W, C, Cout, hp_ncu()); }
int v2a_conv2d_fwd_hp_ups4(const void* x, const void* w_ups4, const float* bias, void* y, const void* zeros, int N, int H, int W, int C,
                           int Cout, hipStream_t stream) {
    if (!x || !w_ups4 || !y || !zeros || N <= 0) return V2A_ERR_ARG;
    const int ncu = hp_ncu();
    if (!conv_patch_h_ups4_eligible(N, H, W, C, Cout, ncu)) return V2A_ERR_ARG;
    if ((((uintptr_t)x | (uintptr_t)w_ups4 | (uintptr_t)y | (uintptr_t)zeros) & 15) != 0 || ((uintptr_t)bias & 3) != 0) return V2A_ERR_ARG;
    ConvDescHP p;
    p.x = (const uint16_t*)x; p.w = (const uint16_t*)w_ups4; p.bias = bias; p.y = (uint16_t*)y; p.zeros = (const uint16_t*)zeros;
    p.N = N; p.H = H; p.W = W; p.C = C; p.Cout = Cout; p.K = 4 * C;
    p.tiles_x = (W / 2) / 16;
    p.tiles_img = ((H / 2) / 32) * p.tiles_x;
    const int total = N * 4 * p.tiles_img * (Cout / 128);
    const dim3 grid(total < ncu ? total : ncu);
    if (g_v2a_half_f16) hipLaunchKernelGGL(conv_patch_h_ups4<true>, grid, dim3(512), 0, stream, p);
    else hipLaunchKernelGGL(conv_patch_h_ups4<false>, grid, dim3(512), 0, stream, p);
    V2A_CHECK_LAUNCH();
    return V2A_OK;
}

}  // extern "C"
