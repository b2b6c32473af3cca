// wgrad3x3g.hip - bf16 weight gradient of the 3x3 stride-1 convolution: wgrad3x3r.hip's register reuse of patch rows with
// LDS-DMA staging.
//
// dW[n][kh][kw][c] = sum_p dy[p][n] * x[p + (kh-1, kw-1)][c].  A workgroup (4 waves, each a 32 x 32 (n, c) tile for all 9
// taps) owns a 64 x 64 (n, c) tile and a split-K slice of 8 x 16 pixel patches.  The x patch (10 x 18 pixels x 64 channels)
// and the dy patch (8 x 16 x 64) of the NEXT patch are requested with buffer_load_dwordx4 ... lds while the current one is
// being multiplied (two LDS buffers, one raw s_barrier per patch, no staging registers, no ds_write); out-of-image pixels
// and channel tails are buffer offsets past num_records and read zeros.  LDS rows are unpadded 128-byte pixels; the
// transposed fragment reads (ds_read_b64_tr_b16, 4 consecutive pixels x 64 B per 32-lane half) stay conflict-free because
// the two 64-byte halves of a pixel row are swapped where bit 1 of the pixel's column is set - applied to the DMA source
// granule and to the read address.  Fragment reads are inline asm (the compiler would wait for the outstanding DMA
// before any LDS read it can see) and are issued one K step ahead of the MFMAs that consume them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

#define TRR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

namespace {
constexpr int WTPH = 8, WTPW = 16;
constexpr int WXH = WTPH + 2, WXW = WTPW + 2;          // 10 x 18 x-patch pixels
constexpr int WX_INSTR = (WXH * WXW + 7) / 8;          // 23 wave-instructions of 8 pixels x 128 B
constexpr int WD_INSTR = WTPH * WTPW / 8;              // 16
constexpr int WX_BYTES = WX_INSTR * 1024, WD_BYTES = WD_INSTR * 1024;
constexpr int WBUF = WX_BYTES + WD_BYTES;              // 39936
constexpr uint32_t WOOB = 0xF0000000u;

__device__ __forceinline__ bf16x8 frag(const u32x2& lo, const u32x2& hi) {
    u32x4 v; v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
    return __builtin_bit_cast(bf16x8, v);
}
}  // namespace

// NH = 1: the workgroup above (4 waves, two per CU).  NH = 2: two such wave quartets in ONE workgroup (8 waves, one per CU),
// each with its own pair of LDS buffers and its own half of the workgroup's split-K slice; they advance in lockstep (the
// raw barrier is workgroup wide) and at the end the second quartet hands its 144 accumulators per lane to the first through
// the then idle 160 KB of LDS.  Same occupancy, half as many partial slabs: half the fp32 partial traffic and half the
// split-K reduction (75 MB written + read per launch with NH = 1).
#ifdef UNETRIR_ABLATIONS
// in-kernel clock stamps (ablation build, bit 2048): per workgroup (shader-clock ticks, 100 MHz ticks) around the K loop; they go
// to this array only, no output depends on them (MI355X_MICROARCH.md, DVFS give-back item 6)
__device__ unsigned long long g_stamps_wgrad3x3g[1024][2];
extern "C" int unetrir_abl_stamps_wgrad3x3g(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_wgrad3x3g), (size_t)n * 16);
}
#endif
struct WgSrc { __amdgpu_buffer_rsrc_t rx, rd; int iy0, ix0, py0, px0, xbase, dbase, gv; };     // DMA sources of one patch

template <int NH, int SCH = 1, int STAG = 0, int FAKE16 = 0>
__global__ __launch_bounds__(256 * NH, NH == 1 ? 2 : 1) void wgrad3x3g_bf16_kernel(const Wgrad3ArgsH a, int abl) {   // abl (ablation build only): 1 no DMA after the first patch, 2 no partial stores, 4 no MFMA
    __shared__ __attribute__((aligned(1024))) unsigned char smem_all[NH * 2 * WBUF];
    const int half = NH == 2 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;
    unsigned char* smem = smem_all + half * 2 * WBUF;
    const int tid = threadIdx.x & 255, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;

    // (tile, split) of this workgroup.  Dispatch deals consecutive workgroups round-robin over the 8 XCDs (each with its own L2),
    // which would put the tiles of one split - the workgroups that read the SAME x and dy patches at the same time - on
    // different XCDs: every patch then comes from HBM once per sharer (2.2x the algorithmic bytes at 128 -> 128).  Remapped:
    // XCD x runs splits x ns/8 .. (x+1) ns/8, all tiles of a split next to each other.
    int tile_id = blockIdx.x, split_id = blockIdx.y;
    if (a.xcd_remap) {
        const int tiles = gridDim.x, per_xcd = gridDim.y >> 3;
        const int id = blockIdx.y * tiles + blockIdx.x, k = id >> 3;
        split_id = (id & 7) * per_xcd + k / tiles;
        tile_id = k - (k / tiles) * tiles;
    }
    const int ntC = (a.C + 63) / 64;
    const int rt = tile_id / ntC, ct = tile_id - rt * ntC;
    const int n0 = rt * 64, c0 = ct * 64;
    const int per_img = a.npy * a.npx;
    const int G = a.B * per_img;
    // the workgroup's slice is NH * patches_per_split patches; quartet `half` takes the half-th part.  Every quartet runs
    // n_it iterations (the first quartet's count); patches past the end are all-zero DMAs and add nothing.
    const int G0 = split_id * NH * a.patches_per_split;
    const int g0 = G0 + half * a.patches_per_split;
    const int n_it = min(a.patches_per_split, G - G0);

    // ---- DMA lane constants: instruction i covers patch pixels 8i .. 8i+7, lane = (pixel sub, 16-byte granule g8)
    const int g8 = lane & 7, sub = lane >> 3;
    int xpr[6], xpc[6];
    uint32_t xrel[6];
    bool xcok[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        int i = wave + 4 * j;
        if (i > WX_INSTR - 1) i = WX_INSTR - 1;
        const int p = 8 * i + sub;
        const int pr = p / WXW, pc = p - pr * WXW;
        const int gs = g8 ^ (((pc >> 1) & 1) << 2);
        xpr[j] = p < WXH * WXW ? pr : 1 << 20;          // pixels past the patch: never valid
        xpc[j] = pc;
        xrel[j] = (uint32_t)(((pr * a.IW + pc) * a.ldx + gs * 8) * 2);
        xcok[j] = (c0 + gs * 8) < a.C;
    }
    int dr_[4], dc_[4];
    uint32_t drel[4];
    bool dnok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = 8 * (wave + 4 * j) + sub;
        const int oyl = p >> 4, oxl = p & 15;
        const int gs = g8 ^ (((oxl >> 1) & 1) << 2);
        dr_[j] = oyl; dc_[j] = oxl;
        drel[j] = (uint32_t)(((oyl * a.OW + oxl) * a.lddy + gs * 8) * 2);
        dnok[j] = (n0 + gs * 8) < a.N;
    }
    const size_t x_img = (size_t)a.IH * a.IW * a.ldx, d_img = (size_t)a.OH * a.OW * a.lddy;
    const int x_rec = (int)((((size_t)a.IH * a.IW - 1) * a.ldx + a.C) * 2), d_rec = (int)((((size_t)a.OH * a.OW - 1) * a.lddy + a.N) * 2);

    // A patch's DMA is 10 wave-instructions (6 x, 4 dy).  Issuing them back to back at the top of a patch costs the wave
    // 60-180 cycles each with the matrix pipe idle (both waves of a SIMD run the same program): measured, DMA and MFMA time
    // ADDED UP (154 us = 111 without DMA + 43).  They are issued two at a time between the K steps of the patch instead.
    bool started = false;
    auto patch_src = [&](int g) {
        WgSrc q;
        q.gv = g < G;
        if (!q.gv) g = G - 1;
        const int img = g / per_img;
        const int rem = g - img * per_img;
        const int pyi = rem / a.npx, pxi = rem - pyi * a.npx;
        q.py0 = pyi * WTPH; q.px0 = pxi * WTPW;
        q.iy0 = q.py0 - a.pad_t; q.ix0 = q.px0 - a.pad_l;
        q.rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + img * x_img), (short)0, x_rec, 0x00020000);
        q.rd = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dy + img * d_img), (short)0, d_rec, 0x00020000);
        q.xbase = ((q.iy0 * a.IW + q.ix0) * a.ldx + c0) * 2;       // may be negative; only used for valid pixels
        q.dbase = ((q.py0 * a.OW + q.px0) * a.lddy + n0) * 2;
        return q;
    };
    auto piece = [&](const WgSrc& q, int buf, int j) {             // j = 0..5: x instruction j, 6..9: dy instruction j - 6 (compile-time)
        if (UNETRIR_ABL(abl, 1) && started) return;
        unsigned char* xb = smem + buf * WBUF;
        if (j < 6) {
            int i = wave + 4 * j;
            if (i > WX_INSTR - 1) i = WX_INSTR - 1;
            const int iy = q.iy0 + xpr[j < 6 ? j : 0], ix = q.ix0 + xpc[j < 6 ? j : 0];
            const bool ok = q.gv && xcok[j < 6 ? j : 0] && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
            const uint32_t off = ok ? (uint32_t)(q.xbase + (int)xrel[j < 6 ? j : 0]) : WOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(q.rx, (lptr_t)(xb + i * 1024), 16, off, 0, 0, 0);
        } else {
            const int jj = j >= 6 ? j - 6 : 0;
            const bool ok = q.gv && dnok[jj] && (q.py0 + dr_[jj]) < a.OH && (q.px0 + dc_[jj]) < a.OW;
            const uint32_t off = ok ? (uint32_t)(q.dbase + (int)drel[jj]) : WOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(q.rd, (lptr_t)(xb + WX_BYTES + (wave + 4 * jj) * 1024), 16, off, 0, 0, 0);
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- fragment read addresses (ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies pixel row q, channels 4p..4p+3;
    // groups 0/1 = channels 0-15/16-31 of pixels 0-7 of the 16-pixel K step, groups 2/3 of pixels 8-15)
    const int grp = lane >> 4, li = lane & 15;
    const int tq = li >> 2, tp = li & 3;
    const int lane_px = 8 * h + tq;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    uint32_t xoff[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
        xoff[m] = lane_px * 128 + ((wc * 64 + (grp & 1) * 32 + tp * 8) ^ ((((lane_px + m) >> 1) & 1) << 6));
    const uint32_t doff = lane_px * 128 + ((wr * 64 + (grp & 1) * 32 + tp * 8) ^ (((lane_px >> 1) & 1) << 6));

    // x fragment of patch row R, column shift kw: two reads (pixels +0..3 and +4..7 of this lane's group)
#define RDX(lo, hi, R, KW) do { TRR(lo, xa[(KW) & 3], (R) * (WXW * 128) + (KW) * 128); \
                                TRR(hi, xa[((KW) + 4) & 3], (R) * (WXW * 128) + ((KW) + 4) * 128); } while (0)
#define RDD(lo, hi, R) do { TRR(lo, da, (R) * (WTPW * 128)); TRR(hi, da, (R) * (WTPW * 128) + 4 * 128); } while (0)
    // FAKE16 (ablation build, WRONG RESULTS): the flops of one 32x32x16 issued as two 16x16x32 on the same operand registers - what
    // the matrix-instruction shape alone does to the time and to the clock the chip holds
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
#define MM(T_, A_, B_) do { if (!UNETRIR_ABL(abl, 4)) { if constexpr (FAKE16) { \
        f32x4_ lo_, hi_; _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) { lo_[e_] = acc[T_][e_]; hi_[e_] = acc[T_][4 + e_]; } \
        lo_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_, B_, lo_, 0, 0, 0); hi_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_, B_, hi_, 0, 0, 0); \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) { acc[T_][e_] = lo_[e_]; acc[T_][4 + e_] = hi_[e_]; } \
    } else acc[T_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, acc[T_], 0, 0, 0); } } while (0)
    // one K step: dy row R against x rows R (kh 0), R+1, R+2.  S0 = ring slot of x row R (refilled with row R+3 once the
    // kh = 0 products are issued), S1, S2 = slots of rows R+1, R+2.
#define STEP(R, S0, S1, S2, DC_LO, DC_HI, DN_LO, DN_HI, MORE)                                                     \
    do {                                                                                                            \
        LGKM0();                                                                                                    \
        const bf16x8 fd = frag(DC_LO, DC_HI);                                                                       \
        MM(0, fd, frag(xl[S0][0], xh[S0][0])); MM(1, fd, frag(xl[S0][1], xh[S0][1])); MM(2, fd, frag(xl[S0][2], xh[S0][2])); \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        if (MORE) {                                                                                                 \
            RDX(xl[S0][0], xh[S0][0], (R) + 3, 0); RDX(xl[S0][1], xh[S0][1], (R) + 3, 1); RDX(xl[S0][2], xh[S0][2], (R) + 3, 2); \
            RDD(DN_LO, DN_HI, (R) + 1);                                                                             \
        }                                                                                                           \
        MM(3, fd, frag(xl[S1][0], xh[S1][0])); MM(4, fd, frag(xl[S1][1], xh[S1][1])); MM(5, fd, frag(xl[S1][2], xh[S1][2])); \
        MM(6, fd, frag(xl[S2][0], xh[S2][0])); MM(7, fd, frag(xl[S2][1], xh[S2][1])); MM(8, fd, frag(xl[S2][2], xh[S2][2])); \
    } while (0)

#ifdef UNETRIR_ABLATIONS
    unsigned long long st0 = 0, sr0 = 0;
    if (UNETRIR_ABL(abl, 2048)) { st0 = __builtin_amdgcn_s_memtime(); sr0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    if (n_it > 0) {
        const WgSrc s0 = patch_src(g0);
#pragma unroll
        for (int j = 0; j < 10; ++j) piece(s0, 0, j);
    }
    started = true;
    int k = 0;
    // STAG: the second quartet (waves 4-7, the SIMD partners of waves 0-3) runs half a patch behind the first: two workgroup
    // barriers per patch (before step 0 and before step 4); a quartet's patch boundary is every other one, the other's falls
    // in the middle of its patch.  One quartet's barrier wait, first-rows read burst and DMA wait then sit beside the other's
    // MFMA steps instead of beside the same bubble (MI355X_MICROARCH.md, two waves per SIMD, item 9).
    if (NH == 2 && STAG && half == 1) __builtin_amdgcn_s_barrier();
    for (int it = 0; it < n_it; ++it, k ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's share of patch g has landed ...
        __builtin_amdgcn_s_barrier();                            // ... everybody's has; everybody is done reading the other buffer
        asm volatile("" ::: "memory");
        const bool nxt_on = it + 1 < n_it;
        const WgSrc nx = patch_src(nxt_on ? g0 + it + 1 : g0 + it);
        const int nb_ = k ^ 1;
#define PIECES(J0, J1) do { if (nxt_on) { piece(nx, nb_, J0); piece(nx, nb_, J1); } } while (0)
#define MIDBAR() do { if (NH == 2 && STAG) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } } while (0)
        const uint32_t xb = lds0 + k * WBUF, db = xb + WX_BYTES;
        uint32_t xa[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) xa[m] = xb + xoff[m];
        const uint32_t da = db + doff;
        u32x2 xl[3][3], xh[3][3], d0l, d0h, d1l, d1h;
        RDX(xl[0][0], xh[0][0], 0, 0); RDX(xl[0][1], xh[0][1], 0, 1); RDX(xl[0][2], xh[0][2], 0, 2);
        RDX(xl[1][0], xh[1][0], 1, 0); RDX(xl[1][1], xh[1][1], 1, 1); RDX(xl[1][2], xh[1][2], 1, 2);
        RDX(xl[2][0], xh[2][0], 2, 0); RDX(xl[2][1], xh[2][1], 2, 1); RDX(xl[2][2], xh[2][2], 2, 2);
        RDD(d0l, d0h, 0);
        // DMA placement: ten pieces per wave and patch.  SCH 1 (product): 3-3-2-2 behind the first four K steps - measured 1.5-4 %
        // faster than 2-2-2-2-2 (SCH 0: the last pieces have 1 us less to land), 4-3-3 (SCH 2) and all ten up front (SCH 3: the
        // issue burst stalls the matrix pipe); 0, 2, 3 and the staggered quartets (STAG: +2 %) are instantiated in the ablation build only
        constexpr int sch = SCH;
        if constexpr (sch == 0) {
        STEP(0, 0, 1, 2, d0l, d0h, d1l, d1h, true); PIECES(0, 1);
        STEP(1, 1, 2, 0, d1l, d1h, d0l, d0h, true); PIECES(2, 3);
        STEP(2, 2, 0, 1, d0l, d0h, d1l, d1h, true); PIECES(4, 5);
        STEP(3, 0, 1, 2, d1l, d1h, d0l, d0h, true); PIECES(6, 7);
        MIDBAR(); STEP(4, 1, 2, 0, d0l, d0h, d1l, d1h, true); PIECES(8, 9);
        STEP(5, 2, 0, 1, d1l, d1h, d0l, d0h, true);
        } else if constexpr (sch == 1) {       // 3-3-2-2
        STEP(0, 0, 1, 2, d0l, d0h, d1l, d1h, true); PIECES(0, 1); if (nxt_on) piece(nx, nb_, 2);
        STEP(1, 1, 2, 0, d1l, d1h, d0l, d0h, true); PIECES(3, 4); if (nxt_on) piece(nx, nb_, 5);
        STEP(2, 2, 0, 1, d0l, d0h, d1l, d1h, true); PIECES(6, 7);
        STEP(3, 0, 1, 2, d1l, d1h, d0l, d0h, true); PIECES(8, 9);
        MIDBAR(); STEP(4, 1, 2, 0, d0l, d0h, d1l, d1h, true);
        STEP(5, 2, 0, 1, d1l, d1h, d0l, d0h, true);
        } else if constexpr (sch == 2) {       // 4-3-3
        STEP(0, 0, 1, 2, d0l, d0h, d1l, d1h, true); PIECES(0, 1); PIECES(2, 3);
        STEP(1, 1, 2, 0, d1l, d1h, d0l, d0h, true); PIECES(4, 5); if (nxt_on) piece(nx, nb_, 6);
        STEP(2, 2, 0, 1, d0l, d0h, d1l, d1h, true); PIECES(7, 8); if (nxt_on) piece(nx, nb_, 9);
        STEP(3, 0, 1, 2, d1l, d1h, d0l, d0h, true);
        MIDBAR(); STEP(4, 1, 2, 0, d0l, d0h, d1l, d1h, true);
        STEP(5, 2, 0, 1, d1l, d1h, d0l, d0h, true);
        } else {                     // all ten before the first step
        PIECES(0, 1); PIECES(2, 3); PIECES(4, 5); PIECES(6, 7); PIECES(8, 9);
        STEP(0, 0, 1, 2, d0l, d0h, d1l, d1h, true);
        STEP(1, 1, 2, 0, d1l, d1h, d0l, d0h, true);
        STEP(2, 2, 0, 1, d0l, d0h, d1l, d1h, true);
        STEP(3, 0, 1, 2, d1l, d1h, d0l, d0h, true);
        MIDBAR(); STEP(4, 1, 2, 0, d0l, d0h, d1l, d1h, true);
        STEP(5, 2, 0, 1, d1l, d1h, d0l, d0h, true);
        }
        STEP(6, 0, 1, 2, d0l, d0h, d1l, d1h, true);
        // last step: x row 10 does not exist; nothing left to prefetch
        STEP(7, 1, 2, 0, d1l, d1h, d0l, d0h, false);
    }
    if (NH == 2 && STAG && half == 0) __builtin_amdgcn_s_barrier();
#ifdef UNETRIR_ABLATIONS
    if (UNETRIR_ABL(abl, 2048) && threadIdx.x == 0) {
        const int w = (blockIdx.y * gridDim.x + blockIdx.x) & 1023;
        g_stamps_wgrad3x3g[w][0] = __builtin_amdgcn_s_memtime() - st0;
        g_stamps_wgrad3x3g[w][1] = __builtin_amdgcn_s_memrealtime() - sr0;
    }
#endif
#undef STEP
#undef MIDBAR
#undef PIECES
#undef MM
#undef RDD
#undef RDX

    if constexpr (NH == 2) {
        // second quartet -> first quartet through LDS: [accumulator register][thread], conflict-free in both directions
        __syncthreads();                                 // every fragment read of the last patch has been consumed
        float* red = reinterpret_cast<float*>(smem_all);
        if (half == 1) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 256 + tid] = acc[t][r];
        }
        __syncthreads();
        if (half == 1) return;
    }
    float* part = a.part + (size_t)split_id * a.N * 9 * a.C;
    const int c = c0 + wc * 32 + (lane & 31);
    if (c < a.C && !UNETRIR_ABL(abl, 2)) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float v = acc[t][r];
                if constexpr (NH == 2) v += reinterpret_cast<const float*>(smem_all)[(t * 16 + r) * 256 + tid];
                if (n < a.N) part[((size_t)n * 9 + t) * a.C + c] = v;
            }
    }
}

static void plan_g(int B, int OH, int OW, int N, int C, int* nsplit, int* per_split, int* npy, int* npx) {
    const long long target = 512;                        // split-K workgroups aimed for
    *npy = (OH + WTPH - 1) / WTPH;
    *npx = (OW + WTPW - 1) / WTPW;
    const long long G = (long long)B * (*npy) * (*npx);
    const long long tiles = (long long)((N + 63) / 64) * ((C + 63) / 64);
    long long want = (target + tiles - 1) / tiles;
    long long maxs = (G + 3) / 4;
    if (maxs < 1) maxs = 1;
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    const long long per = (G + want - 1) / want;
    *per_split = (int)per;
    *nsplit = (int)((G + per - 1) / per);
}

// stride-1 3x3 weight gradient; WGRAD3X3R_NOT_TAKEN when this kernel does not take the layer (the caller falls back)
int launch_wgrad3x3g_bf16(Wgrad3ArgsH a, float* dw, float reg, const float* w, void* ws, size_t ws_bytes, hipStream_t s) {
    const bool on = unetrir_cfg().wgrad3x3g != 0;
    const size_t x_bytes = (((size_t)a.IH * a.IW - 1) * a.ldx + a.C) * 2, d_bytes = (((size_t)a.OH * a.OW - 1) * a.lddy + a.N) * 2;
    // heights that are not a multiple of the 8-row patch: the rows past the image are zero-filled DMAs (as the columns past OW);
    // taken while at least 60 % of the patch rows are real (the reference's 144 x 160 geometry: 36 and 18 rows)
    const int rows8 = (a.OH + 7) / 8 * 8;
    if (!on || a.OH * 10 < rows8 * 6 || x_bytes >= 0x70000000u || d_bytes >= 0x70000000u || (a.C & 7) || (a.N & 7)) return WGRAD3X3R_NOT_TAKEN;
    int ns, per;
    plan_g(a.B, a.OH, a.OW, a.N, a.C, &ns, &per, &a.npy, &a.npx);
    const int nh = ns >= 2 ? 2 : 1;                       // two split-K quartets per workgroup share one partial slab
    ns = (ns + nh - 1) / nh;                             // partial slabs = workgroups along the split dimension
    const size_t nout = (size_t)a.N * 9 * a.C;
    const bool direct = (ns == 1 && reg == 0.f);
    if (!direct && ws_bytes < (size_t)ns * nout * sizeof(float)) return WGRAD3X3R_NOT_TAKEN;
    a.part = direct ? dw : (float*)ws;
    a.patches_per_split = per;
    const unsigned tiles = (unsigned)(((a.N + 63) / 64) * ((a.C + 63) / 64));
    a.xcd_remap = (tiles > 1 && ns % 8 == 0 && !UNETRIR_ABL(UNETRIR_ABL_HOST(), 512)) ? 1 : 0;
#ifdef UNETRIR_ABLATIONS
    if (nh == 2 && UNETRIR_ABL(UNETRIR_ABL_HOST(), 16)) hipLaunchKernelGGL((wgrad3x3g_bf16_kernel<2, 0>), dim3(tiles, ns), dim3(512), 0, s, a, 0);
    else if (nh == 2 && UNETRIR_ABL(UNETRIR_ABL_HOST(), 32)) hipLaunchKernelGGL((wgrad3x3g_bf16_kernel<2, 2>), dim3(tiles, ns), dim3(512), 0, s, a, 0);
    else if (nh == 2 && UNETRIR_ABL(UNETRIR_ABL_HOST(), 64)) hipLaunchKernelGGL((wgrad3x3g_bf16_kernel<2, 3>), dim3(tiles, ns), dim3(512), 0, s, a, 0);
    else if (nh == 2 && UNETRIR_ABL(UNETRIR_ABL_HOST(), 4096)) hipLaunchKernelGGL((wgrad3x3g_bf16_kernel<2, 1, 0, 1>), dim3(tiles, ns), dim3(512), 0, s, a, UNETRIR_ABL_HOST() & 2048);
    else if (nh == 2 && UNETRIR_ABL(UNETRIR_ABL_HOST(), 128)) hipLaunchKernelGGL((wgrad3x3g_bf16_kernel<2, 1, 1>), dim3(tiles, ns), dim3(512), 0, s, a, 0);
    else
#endif
    if (nh == 2) hipLaunchKernelGGL(wgrad3x3g_bf16_kernel<2>, dim3(tiles, ns), dim3(512), 0, s, a, UNETRIR_ABL_HOST());
    else hipLaunchKernelGGL(wgrad3x3g_bf16_kernel<1>, dim3(tiles, ns), dim3(256), 0, s, a, UNETRIR_ABL_HOST());
    const int err = (int)hipGetLastError();
    if (err || direct) return err;
    return launch_splitk_reduce((const float*)ws, ns, nout, dw, reg, w, s);
}
