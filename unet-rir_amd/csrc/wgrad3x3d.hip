// wgrad3x3d.hip - bf16 weight gradient of the 3x3 STRIDE-2 convolution (strided Conv2D, dl_models/u_net.py:269-276, and -
// through its adjoint - Conv2DTranspose, :297-304): the LDS-DMA / row-reuse scheme of wgrad3x3g.hip for the "down" geometry.
//
//   dW[n][kh][kw][c] = sum_p dy[p][n] * x[2p + (kh, kw)][c]        (even sizes: TF 'same' pads (0, 1), pad_before = 0)
//
// Stride 2 reads FOUR input pixels per output pixel, so per flop the x operand is 4x the bytes of the stride-1 layer: the
// register-staged kernel (igemm_bf16.hip, 8 x 4 pixel patches, a barrier pair per 18 MFMAs) reached 0.25 of the MFMA peak.
// Here
//   * a workgroup is 8 waves (one per CU) and owns a 128 (n) x 64 (c) tile - wave (wr, wc) a 32 x 32 tile for all 9 taps (144
//     accumulators) - and a split-K slice of 4 x 16 output-pixel patches: both operands of a patch are staged ONCE for all
//     eight waves (the 64 x 64 tiles of the register-staged kernel each re-read them);
//   * the (2*4+1) x 33 pixel x patch and the 4 x 16 dy patch of the NEXT patch are requested by LDS-DMA while the current one is
//     multiplied (two buffers, one raw s_barrier per patch);
//   * the x patch is stored DE-INTERLEAVED by column parity - row = [17 even columns | 16 odd columns | 1 pad] x 128 B - so
//     the 16 input pixels a K step (one output row) pairs with tap column kw are CONSECUTIVE in LDS: even part + 0 (kw 0), odd
//     part + 0 (kw 1), even part + 1 (kw 2), and the transposed fragment reads (ds_read_b64_tr_b16) are those of the stride-1
//     kernel with other offsets.  The de-interleave costs nothing: the DMA's per-lane source address does it;
//   * x row 2R+2 is tap row 2 of output row R and tap row 0 of output row R+1: three register slots per tap column, two rows
//     (12 reads) + one dy fragment (2 reads) fetched per 9 MFMAs;
//   * the dy patch is two 64-channel planes of 128-byte pixels (the stride-1 layout per plane).
// 64-byte halves of a 128-byte pixel are swapped where bit 1 of the pixel's LDS position is set (conflict-free transposed
// reads), on the DMA source granule and on the read address.  Out-of-image pixels / channel tails are buffer offsets past
// num_records and read zeros.  Requires even input sizes, OH % 4 == 0 and OW % 16 == 0; otherwise the caller keeps the
// register-staged kernel.  Partial slabs [split][N][9][C] fp32 + the fixed-order split-K reduce, as the other weight gradients.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

#define TRR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

namespace {
constexpr int DPH = 4, DPW = 16;                       // output-pixel patch
constexpr int DXH = 2 * DPH + 1;                       // 9 x rows
constexpr int DXW = 34;                                // LDS row: 17 even columns, 16 odd columns, 1 pad
constexpr int DXROW = DXW * 128;                       // 4352 B = 17 bank rows
constexpr int DX_INSTR = (DXH * DXW + 7) / 8;          // 39 wave-instructions of 8 pixels x 128 B
constexpr int DD_INSTR = 2 * DPH * DPW / 8;            // 16: two 64-channel planes of 64 pixels
constexpr int DX_BYTES = DX_INSTR * 1024, DD_BYTES = DD_INSTR * 1024;
constexpr int DBUF = DX_BYTES + DD_BYTES;              // 56320
constexpr int DX_PER_WAVE = (DX_INSTR + 7) / 8;        // 5
constexpr uint32_t DOOB = 0xF0000000u;

__device__ __forceinline__ bf16x8 frag(const u32x2& lo, const u32x2& hi) {
    u32x4 v; v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
    return __builtin_bit_cast(bf16x8, v);
}
}  // namespace

__global__ __launch_bounds__(512, 2) void wgrad3x3d_bf16_kernel(const Wgrad3ArgsH a) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * DBUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;             // 32-channel blocks: wr of the 128 output channels, wc of the 64 input channels

    // (tile, split): the tiles of one split read the same patches - keep them on one XCD (see wgrad3x3g.hip)
    int tile_id = blockIdx.x, split_id = blockIdx.y;
    if (a.xcd_remap) {
        const int tiles = gridDim.x, per_xcd = gridDim.y >> 3;
        const int id = blockIdx.y * tiles + blockIdx.x, k = id >> 3;
        split_id = (id & 7) * per_xcd + k / tiles;
        tile_id = k - (k / tiles) * tiles;
    }
    const int ntC = (a.C + 63) / 64;
    const int rt = tile_id / ntC, ct = tile_id - rt * ntC;
    const int n0 = rt * 128, c0 = ct * 64;
    const int per_img = a.npy * a.npx;
    const int G = a.B * per_img;
    const int g0 = split_id * a.patches_per_split;
    const int n_it = min(a.patches_per_split, G - g0);

    // ---- DMA lane constants.  x: instruction i covers LDS pixels 8i .. 8i+7 of the [9][34] image, lane = (pixel sub, granule g8)
    const int g8 = lane & 7, sub = lane >> 3;
    int xr_[DX_PER_WAVE];
    uint32_t xrel[DX_PER_WAVE];
    bool xok[DX_PER_WAVE];
#pragma unroll
    for (int j = 0; j < DX_PER_WAVE; ++j) {
        const int i = wave + 8 * j;
        const int p = 8 * i + sub;
        const int pr = p / DXW, pp = p - pr * DXW;       // LDS row, position in the row
        const int col = pp <= 16 ? 2 * pp : 2 * (pp - 17) + 1;                 // input column of that position (pp = 33: pad)
        const int gs = g8 ^ (((pp >> 1) & 1) << 2);
        xr_[j] = pr;
        xrel[j] = (uint32_t)(((pr * a.IW + col) * a.ldx + gs * 8) * 2);
        xok[j] = i < DX_INSTR && pr < DXH && pp < 33 && (c0 + gs * 8) < a.C;
        if (!xok[j]) xr_[j] = 1 << 20;                   // never a valid row
        else xr_[j] = pr | (col << 8);                   // row | column (for the image-bound checks of a patch)
    }
    // dy: instruction i (0..15): plane i >> 3, pixels 8 (i & 7) .. + 7 of the 4 x 16 patch; wave w issues i = w and w + 8
    int dpix[2];
    uint32_t drel[2];
    bool dok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = wave + 8 * j;
        const int plane = i >> 3, p = 8 * (i & 7) + sub;
        const int oyl = p >> 4, oxl = p & 15;
        const int gs = g8 ^ (((oxl >> 1) & 1) << 2);
        dpix[j] = oyl | (oxl << 8);
        drel[j] = (uint32_t)(((oyl * a.OW + oxl) * a.lddy + plane * 64 + gs * 8) * 2);
        dok[j] = (n0 + plane * 64 + gs * 8) < a.N;
    }
    const size_t x_img = (size_t)a.IH * a.IW * a.ldx, d_img = (size_t)a.OH * a.OW * a.lddy;
    const int x_rec = (int)((((size_t)a.IH * a.IW - 1) * a.ldx + a.C) * 2), d_rec = (int)((((size_t)a.OH * a.OW - 1) * a.lddy + a.N) * 2);

    auto issue = [&](int g, int buf) {
        const bool gv = g < G;
        if (!gv) g = G - 1;
        const int img = g / per_img;
        const int rem = g - img * per_img;
        const int pyi = rem / a.npx, pxi = rem - pyi * a.npx;
        const int py0 = pyi * DPH, px0 = pxi * DPW;
        const int iy0 = 2 * py0, ix0 = 2 * px0;          // pad_before = 0
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + img * x_img), (short)0, x_rec, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dy + img * d_img), (short)0, d_rec, 0x00020000);
        unsigned char* xb = smem + buf * DBUF;
        unsigned char* db = xb + DX_BYTES;
        const int xbase = ((iy0 * a.IW + ix0) * a.ldx + c0) * 2;
        const int dbase = ((py0 * a.OW + px0) * a.lddy + n0) * 2;
#pragma unroll
        for (int j = 0; j < DX_PER_WAVE; ++j) {
            const int i = wave + 8 * j;
            if (i >= DX_INSTR) continue;                 // wave-uniform
            const int iy = iy0 + (xr_[j] & 255), ix = ix0 + (xr_[j] >> 8);
            const bool ok = gv && xok[j] && iy < a.IH && ix < a.IW;
            const uint32_t off = ok ? (uint32_t)(xbase + (int)xrel[j]) : DOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(xb + i * 1024), 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = gv && dok[j] && (py0 + (dpix[j] & 255)) < a.OH && (px0 + (dpix[j] >> 8)) < a.OW;
            const uint32_t off = ok ? (uint32_t)(dbase + (int)drel[j]) : DOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lptr_t)(db + (wave + 8 * j) * 1024), 16, off, 0, 0, 0);
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
    uint32_t xoff[4];                                    // [LDS position offset & 3]: the swizzle bit has period 4 in the position
#pragma unroll
    for (int m = 0; m < 4; ++m)
        xoff[m] = lane_px * 128 + ((wc * 64 + (grp & 1) * 32 + tp * 8) ^ ((((lane_px + m) >> 1) & 1) << 6));
    const uint32_t doff = (wr >> 1) * (DPH * DPW * 128) + lane_px * 128 + (((wr & 1) * 64 + (grp & 1) * 32 + tp * 8) ^ (((lane_px >> 1) & 1) << 6));

    // x fragment of LDS row ROW at position offset O (0: even columns, tap column 0; 17: odd columns, tap column 1; 1: even + 1, tap
    // column 2): two reads (pixels +0..3 and +4..7 of this lane's group)
#define RDX(lo, hi, ROW, O) do { TRR(lo, xa[(O) & 3], (ROW) * DXROW + (O) * 128); TRR(hi, xa[(O) & 3], (ROW) * DXROW + ((O) + 4) * 128); } while (0)
#define RDROW(S, ROW) do { RDX(xl[S][0], xh[S][0], ROW, 0); RDX(xl[S][1], xh[S][1], ROW, 17); RDX(xl[S][2], xh[S][2], ROW, 1); } while (0)
#define RDD(lo, hi, R) do { TRR(lo, da, (R) * (DPW * 128)); TRR(hi, da, (R) * (DPW * 128) + 4 * 128); } while (0)
#define MM(T_, A_, B_) acc[T_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, acc[T_], 0, 0, 0)
    // one K step: dy row R against x rows 2R (tap row 0, slot S0), 2R+1 (S1), 2R+2 (S2).  S2 is the next step's S0; S0 and S1 are
    // refilled with rows 2R+4 and 2R+3 once their products are issued.
#define STEP(R, S0, S1, S2, DC_LO, DC_HI, DN_LO, DN_HI, MORE)                                                     \
    do {                                                                                                            \
        LGKM0();                                                                                                    \
        const bf16x8 fd = frag(DC_LO, DC_HI);                                                                       \
        MM(0, fd, frag(xl[S0][0], xh[S0][0])); MM(1, fd, frag(xl[S0][1], xh[S0][1])); MM(2, fd, frag(xl[S0][2], xh[S0][2])); \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        if (MORE) { RDROW(S0, 2 * (R) + 4); RDD(DN_LO, DN_HI, (R) + 1); }                                           \
        MM(3, fd, frag(xl[S1][0], xh[S1][0])); MM(4, fd, frag(xl[S1][1], xh[S1][1])); MM(5, fd, frag(xl[S1][2], xh[S1][2])); \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        if (MORE) { RDROW(S1, 2 * (R) + 3); }                                                                       \
        MM(6, fd, frag(xl[S2][0], xh[S2][0])); MM(7, fd, frag(xl[S2][1], xh[S2][1])); MM(8, fd, frag(xl[S2][2], xh[S2][2])); \
    } while (0)

    if (n_it > 0) issue(g0, 0);
    int k = 0;
    for (int it = 0; it < n_it; ++it, k ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's share of patch g has landed ...
        __builtin_amdgcn_s_barrier();                            // ... everybody's has; everybody is done reading the other buffer
        asm volatile("" ::: "memory");
        if (it + 1 < n_it) issue(g0 + it + 1, k ^ 1);
        const uint32_t xb = lds0 + k * DBUF, db = xb + DX_BYTES;
        uint32_t xa[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) xa[m] = xb + xoff[m];
        const uint32_t da = db + doff;
        u32x2 xl[3][3], xh[3][3], d0l, d0h, d1l, d1h;
        RDROW(0, 0); RDROW(1, 1); RDROW(2, 2);
        RDD(d0l, d0h, 0);
        STEP(0, 0, 1, 2, d0l, d0h, d1l, d1h, true);
        STEP(1, 2, 1, 0, d1l, d1h, d0l, d0h, true);
        STEP(2, 0, 1, 2, d0l, d0h, d1l, d1h, true);
        STEP(3, 2, 1, 0, d1l, d1h, d0l, d0h, false);
    }
#undef STEP
#undef MM
#undef RDD
#undef RDROW
#undef RDX

    float* part = a.part + (size_t)split_id * a.N * 9 * a.C;
    const int c = c0 + wc * 32 + (lane & 31);
    if (c < a.C) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (n < a.N) part[((size_t)n * 9 + t) * a.C + c] = acc[t][r];
            }
    }
}

namespace {
void plan_d(int B, int OH, int OW, int N, int C, int* nsplit, int* per_split, int* npy, int* npx) {
    *npy = OH / DPH;
    *npx = OW / DPW;
    const long long G = (long long)B * (*npy) * (*npx);
    const long long tiles = (long long)((N + 127) / 128) * ((C + 63) / 64);
    long long want = 256 / tiles;                        // one workgroup per CU: every split costs a fp32 slab of the whole kernel
    if (want < 1) want = 1;
    if (want >= 8) want = want / 8 * 8;                  // a multiple of 8 splits: the XCD grouping applies
    long long maxs = (G + 3) / 4;                        // at least 4 patches per slice
    if (maxs < 1) maxs = 1;
    if (want > maxs) want = maxs;
    const long long per = (G + want - 1) / want;
    *per_split = (int)per;
    *nsplit = (int)((G + per - 1) / per);
}
}  // namespace

bool wgrad3x3d_applies(const Wgrad3ArgsH& a) {
    const size_t x_bytes = (((size_t)a.IH * a.IW - 1) * a.ldx + a.C) * 2, d_bytes = (((size_t)a.OH * a.OW - 1) * a.lddy + a.N) * 2;
    return unetrir_cfg().wgrad3x3d && a.pad_t == 0 && a.pad_l == 0 && a.IH == 2 * a.OH && a.IW == 2 * a.OW && a.OH % DPH == 0 &&
           a.OW % DPW == 0 && (a.C & 7) == 0 && (a.N & 7) == 0 && x_bytes < 0x70000000u && d_bytes < 0x70000000u;
}

size_t wgrad3x3d_ws_bytes(int B, int OH, int OW, int N, int C) {
    if (OH % DPH || OW % DPW) return 0;
    int ns, per, npy, npx;
    plan_d(B, OH, OW, N, C, &ns, &per, &npy, &npx);
    return (size_t)ns * N * 9 * C * sizeof(float);
}

// stride-2 3x3 weight gradient; WGRAD3X3R_NOT_TAKEN when this kernel does not take the layer (the caller falls back)
int launch_wgrad3x3d_bf16(Wgrad3ArgsH a, float* dw, float reg, const float* w, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!wgrad3x3d_applies(a)) return WGRAD3X3R_NOT_TAKEN;
    int ns, per;
    plan_d(a.B, a.OH, a.OW, a.N, a.C, &ns, &per, &a.npy, &a.npx);
    const size_t nout = (size_t)a.N * 9 * a.C;
    const bool direct = (ns == 1 && reg == 0.f);
    if (!direct && ws_bytes < (size_t)ns * nout * sizeof(float)) return WGRAD3X3R_NOT_TAKEN;
    a.part = direct ? dw : (float*)ws;
    a.patches_per_split = per;
    const unsigned tiles = (unsigned)(((a.N + 127) / 128) * ((a.C + 63) / 64));
    a.xcd_remap = (tiles > 1 && ns % 8 == 0) ? 1 : 0;
    hipLaunchKernelGGL(wgrad3x3d_bf16_kernel, dim3(tiles, ns), dim3(512), 0, s, a);
    const int err = (int)hipGetLastError();
    if (err || direct) return err;
    return launch_splitk_reduce((const float*)ws, ns, nout, dw, reg, w, s);
}
