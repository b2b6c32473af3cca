// head_mfma.hip - bf16 fast paths of the output head Conv2D(2, (6,6), padding='same') (dl_models/u_net.py:248) on the
// matrix cores, for images at most 256 pixels wide.
//
// With 2 output channels the plain implicit GEMM wastes 15/16 of every MFMA tile and the direct VALU kernels of head.hip
// run 10x above their HBM floor.  Here the horizontal taps are folded into the MFMA's 16-wide dimension instead:
//     S[y][q][(n,kw)] = sum_kh sum_c  w[n][kh][kw][c] * x[y + kh - 2][q][c]          (12 of 16 rows used, K = 6 * C)
//     out[y][x][n]    = bias[n] + sum_kw S[y][x + kw - 2][(n,kw)]                     (a 6-term shifted sum through LDS)
// so a 16-pixel x fragment feeds 6 * C/32 MFMAs instead of 36 * C/32, every input pixel is loaded from HBM once per
// workgroup (fragment loads straight to registers, no LDS staging) and the kernel is bound by reading x.
// A workgroup (8 waves x 32 pixels) owns up to 256 columns x R output rows of one image and walks the INPUT rows: row iy
// contributes to the six output rows iy - kh + 2, whose accumulators live in a rotating register window (the loop is
// unrolled by 6 so the window index is static); output row iy - 3 is complete after input row iy.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define HM_ROWS 32            // output rows per workgroup

// Images wider than 256 pixels (configs[3]: 512) are cut into ncb column blocks of owb <= 250 OUTPUT columns: a workgroup still
// holds 256 columns of S, those of the input columns g0 .. g0 + 255 with g0 = cb * owb - 2, i.e. its output columns plus the two
// columns to the left and the three to the right that their horizontal taps reach (S of a column depends on that input column only).
// (Round 3 built a variant with BatchNormalization + ReLU of the layer in front of the head on this kernel's load path - the
// activation tensor never written: identical results, head forward 80 -> 133 us, weight gradient 106 -> 164 us against the 105 us
// BatchNorm-apply pass it removed.  Measured slower, removed in round 4; DESIGN.md section 8 item 4 keeps the numbers.)
template <int NCH>
__global__ __launch_bounds__(512) void head_fwd_mfma_kernel(const __bf16* __restrict__ x, int ldx, int B, int H, int W,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ y, int ldy, int ncb, int owb) {
    constexpr int C = 32 * NCH;
    __shared__ __attribute__((aligned(16))) float S[2][256][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int nrb = (H + HM_ROWS - 1) / HM_ROWS;
    const int cb = blockIdx.x % ncb, blk = blockIdx.x / ncb;
    const int img = blk / nrb, rb = blk - img * nrb;
    const int g0 = ncb == 1 ? 0 : cb * owb - 2;                      // image column of local column 0
    const int olo = ncb == 1 ? 0 : cb * owb, ohi = ncb == 1 ? W : min(olo + owb, W);   // output columns of this workgroup
    const int y0 = rb * HM_ROWS;
    const int nrows = (H - y0) < HM_ROWS ? (H - y0) : HM_ROWS;

    // weight fragments (A operand): row = (n,kw) index l15 (12 used), k = 8 channels lq*8.. of chunk ch, per vertical tap kh
    bf16x8 wf[6][NCH];
#pragma unroll
    for (int kh = 0; kh < 6; ++kh)
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
            if (l15 < 12) {
                const int n = l15 / 6, kw = l15 - n * 6;
                const float* src = w + ((size_t)(n * 36 + kh * 6 + kw)) * C + ch * 32 + lq * 8;
                const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
                v[0] = (__bf16)a.x; v[1] = (__bf16)a.y; v[2] = (__bf16)a.z; v[3] = (__bf16)a.w;
                v[4] = (__bf16)b.x; v[5] = (__bf16)b.y; v[6] = (__bf16)b.z; v[7] = (__bf16)b.w;
            }
            wf[kh][ch] = v;
        }

    // x fragments (B operand): column = pixel q, k = 8 channels; two 16-pixel tiles per wave
    const int q0 = wave * 32 + l15;
    const __bf16* xb = x + (size_t)img * H * W * ldx + lq * 8;
    auto load_row = [&](int iy, bf16x8 (&f)[2][NCH]) {
        const bool rok = (unsigned)iy < (unsigned)H;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q = g0 + q0 + 16 * t;                      // image column
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
                if (rok && (unsigned)q < (unsigned)W) v = *reinterpret_cast<const bf16x8*>(xb + ((size_t)iy * W + q) * ldx + ch * 32);
                f[t][ch] = v;
            }
        }
    };

    f32x4 acc[6][2];
#pragma unroll
    for (int s = 0; s < 6; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float b0 = bias ? bias[0] : 0.f, b1 = bias ? bias[1] : 0.f;
    bf16x8 xa[2][NCH], xn[2][NCH];
    load_row(y0 - 2, xa);
    const int nsteps = nrows + 5;                     // input rows y0-2 .. y0+nrows+2
    // one step: input row i (relative to y0-2) -> window slots (U - kh) mod 6; output row i-5 leaves from slot (U+1) mod 6
#define HM_STEP(U, CUR, NXT)                                                                                     \
    if (i + U < nsteps) {                                                                                        \
        const int ii = i + U;                                                                                    \
        load_row(y0 - 2 + ii + 1, NXT);                                                                          \
        if ((unsigned)(y0 - 2 + ii) < (unsigned)H) {                                                             \
            _Pragma("unroll") for (int kh = 0; kh < 6; ++kh) {                                                   \
                const int orow = ii - kh;                                                                        \
                if (orow < 0 || orow >= nrows) continue;                                                         \
                _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                    \
                    _Pragma("unroll") for (int ch = 0; ch < NCH; ++ch)                                           \
                        acc[(U + 6 - kh) % 6][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kh][ch], CUR[t][ch], acc[(U + 6 - kh) % 6][t], 0, 0, 0); \
            }                                                                                                    \
        }                                                                                                        \
        const int orow = ii - 5;                                                                                 \
        if (orow >= 0) {                                                                                         \
            float(*Sb)[16] = S[ii & 1];                                                                          \
            _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                      \
                *reinterpret_cast<f32x4*>(&Sb[wave * 32 + 16 * t + l15][4 * lq]) = acc[(U + 1) % 6][t];          \
                acc[(U + 1) % 6][t] = f32x4{0.f, 0.f, 0.f, 0.f};                                                 \
            }                                                                                                    \
            __syncthreads();                                                                                     \
            if (tid < 256 && g0 + tid >= olo && g0 + tid < ohi) {                                                \
                float o0 = b0, o1 = b1;                                                                          \
                _Pragma("unroll") for (int kw = 0; kw < 6; ++kw) {                                               \
                    const int q = tid + kw - 2;                                                                  \
                    if ((unsigned)q < 256u) { o0 += Sb[q][kw]; o1 += Sb[q][6 + kw]; }                            \
                }                                                                                                \
                const size_t p = ((size_t)img * H + y0 + orow) * W + g0 + tid;                                   \
                if (ldy >= 4) *reinterpret_cast<float4*>(y + p * ldy) = make_float4(o0, o1, 0.f, 0.f);           \
                else { y[p * ldy] = o0; y[p * ldy + 1] = o1; }                                                   \
            }                                                                                                    \
        }                                                                                                        \
    }
    for (int i = 0; i < nsteps; i += 6) {
        HM_STEP(0, xa, xn)
        HM_STEP(1, xn, xa)
        HM_STEP(2, xa, xn)
        HM_STEP(3, xn, xa)
        HM_STEP(4, xa, xn)
        HM_STEP(5, xn, xa)
    }
#undef HM_STEP
}

bool head_mfma_applies(int W, int C) { return W <= 4096 && (C == 32 || C == 64 || C == 128); }

int launch_head_fwd_mfma(const void* x, int ldx, int B, int H, int W, int C, const float* w, const float* bias, float* y, int ldy,
                         hipStream_t s) {
    const int ncb = W <= 256 ? 1 : (W + 249) / 250;                  // column blocks of at most 250 output columns (+ 2 + 3 halo = 255 <= 256)
    const int owb = (W + ncb - 1) / ncb;
    const unsigned grid = (unsigned)(B * ((H + HM_ROWS - 1) / HM_ROWS) * ncb);
    const __bf16* xp = (const __bf16*)x;
    if (C == 32) hipLaunchKernelGGL((head_fwd_mfma_kernel<1>), dim3(grid), dim3(512), 0, s, xp, ldx, B, H, W, w, bias, y, ldy, ncb, owb);
    else if (C == 64) hipLaunchKernelGGL((head_fwd_mfma_kernel<2>), dim3(grid), dim3(512), 0, s, xp, ldx, B, H, W, w, bias, y, ldy, ncb, owb);
    else hipLaunchKernelGGL((head_fwd_mfma_kernel<4>), dim3(grid), dim3(512), 0, s, xp, ldx, B, H, W, w, bias, y, ldy, ncb, owb);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// weight gradient:  dW[n][kh][kw][c] = sum_{y,q} D'[y][(n,kw)][q] * x[y + kh - 2][q][c],   D'[y][(n,kw)][q] = dy[y][q - kw + 2][n]
// One MFMA tile is [16 (n,kw) rows (12 used)] x [16 channels], K = 32 pixels of one image row.  A workgroup (4 waves, one
// 16-channel tile each, 64 channels per blockIdx.y) walks the INPUT rows of a 32-row block: each staged x row (LDS,
// 160-byte pixel stride so ds_read_b64_tr_b16 is conflict-free) meets the six dy rows y = iy - kh + 2, kept as shifted
// copies D' in a 6-slot LDS ring (528-byte (n,kw) stride: conflict-free ds_read_b64).  K order inside a 32-pixel step:
// k = 8*lq + j  <->  pixel 4*lq + j (j < 4) or 16 + 4*lq + j - 4, the same on both operands.
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_h;

#define HW_XL 80              // x row stride in LDS, elements (160 B)
#define HW_DL 264             // D' (n,kw) stride, elements (528 B)
#define HW_ROWS 16            // dy rows per workgroup step: 2 workgroups per CU x 256 CUs = 512 blocks for 32 x 256 rows

__global__ __launch_bounds__(256, 2) void head_wgrad_mfma_kernel(const __bf16* __restrict__ x, int ldx, int B, int H, int W, int C,
                                                                 const __bf16* __restrict__ dy, int lddy,
                                                                 float* __restrict__ part) {
    // 78.2 KB of LDS: two workgroups per CU hide each other's row loads (one x row buffer; (n,kw) rows 12..15 are zero
    // fragments made in registers)
    __shared__ __attribute__((aligned(16))) __bf16 Xs[256 * HW_XL];         // 40960 B
    __shared__ __attribute__((aligned(16))) __bf16 Dp[6][12 * HW_DL];       // 38016 B
    __shared__ __attribute__((aligned(16))) __bf16 dyrow[2][272];           // dy row, planar, 3 zero pixels left / 13 right
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int c0 = blockIdx.y * 64;
    const int nrb = (H + HW_ROWS - 1) / HW_ROWS;

    for (int i = tid; i < 6 * 12 * HW_DL; i += 256) (&Dp[0][0])[i] = (__bf16)0.f;   // the pads stay zero
    for (int i = tid; i < 2 * 272; i += 256) (&dyrow[0][0])[i] = (__bf16)0.f;

    f32x4 acc[6];
#pragma unroll
    for (int kh = 0; kh < 6; ++kh) acc[kh] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g8 = tid & 7, qsub = tid >> 3;          // x staging: 16-byte granule g8 of pixel qsub + 32*j
    uint4 rx[8];
    // images wider than 256 pixels: column blocks of 256 x columns (a partition - the sum over pixels needs no halo of x); the shifted
    // copies D' of a block reach three dy columns to its left and two to its right, which five threads load beside the block's own
    const int ncb = (W + 255) / 256;
    for (int blk = blockIdx.x; blk < B * nrb * ncb; blk += gridDim.x) {
        const int cb = blk % ncb, ib = blk / ncb;
        const int img = ib / nrb, rb = ib - img * nrb;
        const int g0 = cb * 256;                          // image column of local column 0
        const int y0 = rb * HW_ROWS;
        const int nrows = (H - y0) < HW_ROWS ? (H - y0) : HW_ROWS;
        const __bf16* xi = x + (size_t)img * H * W * ldx + c0 + g8 * 8;
        const __bf16* di = dy + (size_t)img * H * W * lddy;
        auto load_x = [&](int iy) {
            const bool rok = (unsigned)iy < (unsigned)H;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = g0 + qsub + 32 * j;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (rok && q < W && c0 + g8 * 8 < C) v = *reinterpret_cast<const uint4*>(xi + ((size_t)iy * W + q) * ldx);   // C = 32: half a block
                rx[j] = v;
            }
        };
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        auto load_dy = [&](int r) {                   // dy row y0 + r (zeros outside the block / image columns)
            bf16x2 d; d[0] = (__bf16)0.f; d[1] = (__bf16)0.f;
            if (r < nrows && g0 + tid < W) d = *reinterpret_cast<const bf16x2*>(di + ((size_t)(y0 + r) * W + g0 + tid) * lddy);
            return d;
        };
        const int hcol = g0 + (tid < 3 ? tid - 3 : 253 + tid);        // threads 0..4: the halo columns g0-3..g0-1, g0+256, g0+257
        auto load_dy_halo = [&](int r) {
            bf16x2 d; d[0] = (__bf16)0.f; d[1] = (__bf16)0.f;
            if (tid < 5 && r < nrows && (unsigned)hcol < (unsigned)W) d = *reinterpret_cast<const bf16x2*>(di + ((size_t)(y0 + r) * W + hcol) * lddy);
            return d;
        };
        load_x(y0 - 2);
        bf16x2 dnext = load_dy(0), dhalo = load_dy_halo(0);
        const int nsteps = nrows + 5;
        for (int i = 0; i < nsteps; ++i) {
            const int iy = y0 - 2 + i;
            __syncthreads();                          // the previous step's MFMAs are done with the x row and with the ring
            __bf16* Xb = Xs;
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<uint4*>(Xb + (qsub + 32 * j) * HW_XL + g8 * 8) = rx[j];
            const bool newdy = i < nrows;             // dy row y0 + i enters the ring (vertical tap kh = 0 of this x row)
            if (newdy) {
                dyrow[0][tid + 3] = dnext[0]; dyrow[1][tid + 3] = dnext[1];
                if (tid < 5) { const int hi_ = tid < 3 ? tid : 256 + tid; dyrow[0][hi_] = dhalo[0]; dyrow[1][hi_] = dhalo[1]; }
            }
            __syncthreads();
            if (newdy) {
                __bf16* Db = Dp[(y0 + i) % 6];
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int kw = 0; kw < 6; ++kw) Db[(n * 6 + kw) * HW_DL + tid] = dyrow[n][tid - kw + 2 + 3];
            }
            if (i + 1 < nsteps) { load_x(iy + 1); dnext = load_dy(i + 1); dhalo = load_dy_halo(i + 1); }
            __syncthreads();
            if ((unsigned)iy < (unsigned)H) {
                // ring slot of each vertical tap's dy row (wave-uniform), -1 where that row is outside the block
                int doff[6];
#pragma unroll
                for (int kh = 0; kh < 6; ++kh) {
                    const int orow = i - kh;
                    doff[kh] = (orow < 0 || orow >= nrows) ? -1 : ((y0 + orow) % 6) * (12 * HW_DL);
                }
                const int r = l15 >> 2, p = l15 & 3;
                const __bf16* xr0 = Xb + (4 * lq + r) * HW_XL + wave * 16 + 4 * p;
                const __bf16* dr0 = &Dp[0][0] + (l15 < 12 ? l15 : 0) * HW_DL + 4 * lq;
#pragma unroll 2
                for (int ks = 0; ks < 8; ++ks) {
                    // B operand (x, transposed read): group lq supplies pixel rows ks*32 + 4*lq + r (+16), channels wave*16 + 4p..
                    const __bf16* xr = xr0 + ks * 32 * HW_XL;
                    const bf16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_h*)xr);
                    const bf16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_h*)(xr + 16 * HW_XL));
                    const bf16x8 fb = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                    for (int kh = 0; kh < 6; ++kh) {
                        if (doff[kh] < 0) continue;
                        const __bf16* dr = dr0 + doff[kh] + ks * 32;
                        bf16x4 alo = *reinterpret_cast<const bf16x4*>(dr);
                        bf16x4 ahi = *reinterpret_cast<const bf16x4*>(dr + 16);
                        if (l15 >= 12) {                  // (n,kw) rows 12..15 do not exist: zero fragments
#pragma unroll
                            for (int e = 0; e < 4; ++e) { alo[e] = (__bf16)0.f; ahi[e] = (__bf16)0.f; }
                        }
                        const bf16x8 fa = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
                        acc[kh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[kh], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
    // acc[kh][j] = dW[(n,kw) = 4*lq + j][kh][channel c0 + wave*16 + l15]
    float* out = part + (size_t)blockIdx.x * 2 * 36 * C;
    const int c = c0 + wave * 16 + l15;
#pragma unroll
    for (int kh = 0; kh < 6; ++kh)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = 4 * lq + j;
            if (idx < 12 && c < C) {
                const int n = idx / 6, kw = idx - n * 6;
                out[(size_t)(n * 36 + kh * 6 + kw) * C + c] = acc[kh][j];
            }
        }
}

int launch_head_wgrad_mfma(const void* x, int ldx, int B, int H, int W, int C, const void* dy, int lddy, float* part, int max_blocks,
                           int* nblk_out, hipStream_t s) {
    int nblk = B * ((H + HW_ROWS - 1) / HW_ROWS) * ((W + 255) / 256);
    if (nblk > max_blocks) nblk = max_blocks;
    *nblk_out = nblk;
    hipLaunchKernelGGL(head_wgrad_mfma_kernel, dim3(nblk, (C + 63) / 64), dim3(256), 0, s, (const __bf16*)x, ldx, B, H, W, C,
                       (const __bf16*)dy, lddy, part);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// data gradient:  dx[y][q][c] = sum_kh sum_(n,kw) Dq[y - kh + 2][q][(n,kw)] * w[n][kh][kw][c],   Dq[r][q][(n,kw)] = dy[r][q - kw + 2][n]
// One MFMA tile is [16 channels] x [16 pixels], K = 32 = two vertical taps x 16 (n,kw) slots (12 used); three K steps
// cover the six vertical taps.  The weight fragments (12) stay in registers; the dy rows are expanded once into their
// 32-byte-per-pixel (n,kw) images in a 6-slot LDS ring, read conflict-free by ds_read_b128.  A workgroup (4 waves x 64
// pixels, C = 64) walks HD_ROWS output rows.  The kernel is bound by writing dx (268 MB at configs[1]): the channels of the
// MFMA rows are permuted (tile ct, row r <-> channel 32 (ct >> 1) + 8 (r >> 2) + 4 (ct & 1) + (r & 3)) so that a lane's
// accumulators of tiles (2m, 2m + 1) are 8 consecutive channels - 16-byte NHWC stores straight from the registers, no LDS
// row image and no barrier for it; 50 KB of LDS and 16-row blocks put two to three workgroups on a CU (round 1: one 32-row
// workgroup per CU with 4 barriers per row, 104 us).
// ------------------------------------------------------------------------------------------------------------------
#define HD_ROWS 16            // output rows per workgroup

// C = 64 channels per workgroup: blockIdx.y picks the 64-channel block of a wider layer (configs[3]: 128); images wider than 256
// pixels run in column blocks of 256 output columns, whose dy rows carry three more columns to the left and two to the right
// (the reach of the horizontal taps), loaded by five threads beside the block's own.
__global__ __launch_bounds__(256) void head_dgrad_mfma_kernel(const __bf16* __restrict__ dy, int lddy, int B, int H, int W,
                                                              const float* __restrict__ w, int C, __bf16* __restrict__ dx, int lddx,
                                                              int ncb) {
    __shared__ __attribute__((aligned(16))) __bf16 Dq[6][256 * 16];        // 49152 B
    __shared__ __attribute__((aligned(16))) __bf16 dyrow[2][272];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int nrb = (H + HD_ROWS - 1) / HD_ROWS;
    const int cb = blockIdx.x % ncb, ib = blockIdx.x / ncb;
    const int img = ib / nrb, rb = ib - img * nrb;
    const int g0 = cb * 256;                              // image column of local column 0
    const int c0 = blockIdx.y * 64;                       // first channel of this workgroup
    const int hcol = g0 + (tid < 3 ? tid - 3 : 253 + tid);        // threads 0..4: halo columns g0-3..g0-1, g0+256, g0+257
    const int y0 = rb * HD_ROWS;
    const int nrows = (H - y0) < HD_ROWS ? (H - y0) : HD_ROWS;
    const __bf16* di = dy + (size_t)img * H * W * lddy;

    for (int i = tid; i < 2 * 272; i += 256) (&dyrow[0][0])[i] = (__bf16)0.f;

    // weight fragments (A operand): row l15 of tile ct = channel 32 (ct >> 1) + 8 (l15 >> 2) + 4 (ct & 1) + (l15 & 3), k = 8*lq + j <->
    // vertical tap 2*kp + (lq>>1), slot 8*(lq&1) + j
    bf16x8 wf[3][4];
#pragma unroll
    for (int kp = 0; kp < 3; ++kp)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            bf16x8 v;
            const int kh = 2 * kp + (lq >> 1);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = 8 * (lq & 1) + j;
                float f = 0.f;
                if (idx < 12 && c0 + 32 * (ct >> 1) < C) {            // C = 32: the upper half of the block does not exist
                    const int n = idx / 6, kw = idx - n * 6;
                    f = w[(size_t)(n * 36 + kh * 6 + kw) * C + c0 + 32 * (ct >> 1) + 8 * (l15 >> 2) + 4 * (ct & 1) + (l15 & 3)];
                }
                v[j] = (__bf16)f;
            }
            wf[kp][ct] = v;
        }

    // expand dy row r into ring slot r mod 6 (all threads).  `guard`: a barrier in front of the staging write - needed when the
    // previous expansion's reads of the staging row are not already behind a barrier (back-to-back calls in the prologue)
    auto push_row = [&](int r, bool guard) {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        bf16x2 d, dh; d[0] = (__bf16)0.f; d[1] = (__bf16)0.f; dh = d;
        if ((unsigned)r < (unsigned)H && g0 + tid < W) d = *reinterpret_cast<const bf16x2*>(di + ((size_t)r * W + g0 + tid) * lddy);
        if (tid < 5 && (unsigned)r < (unsigned)H && (unsigned)hcol < (unsigned)W) dh = *reinterpret_cast<const bf16x2*>(di + ((size_t)r * W + hcol) * lddy);
        if (guard) __syncthreads();
        dyrow[0][tid + 3] = d[0]; dyrow[1][tid + 3] = d[1];
        if (tid < 5) { const int hi_ = tid < 3 ? tid : 256 + tid; dyrow[0][hi_] = dh[0]; dyrow[1][hi_] = dh[1]; }
        __syncthreads();
        bf16x8 lo, hi;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int i0 = e, i1 = 8 + e;             // (n,kw) slots 0..7 and 8..15
            lo[e] = dyrow[i0 / 6][tid - (i0 % 6) + 5];
            hi[e] = i1 < 12 ? dyrow[i1 / 6][tid - (i1 % 6) + 5] : (__bf16)0.f;
        }
        __bf16* dst = Dq[((r % 6) + 6) % 6] + tid * 16;
        *reinterpret_cast<bf16x8*>(dst) = lo;
        *reinterpret_cast<bf16x8*>(dst + 8) = hi;
    };
    for (int r = y0 - 3; r < y0 + 2; ++r) push_row(r, true);

    for (int yy = 0; yy < nrows; ++yy) {
        const int yo = y0 + yy;
        push_row(yo + 2, false);                      // the slot it replaces (row yo - 4) was last read one step ago, behind the barrier inside
        __syncthreads();
        f32x4 acc[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[t][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kp = 0; kp < 3; ++kp) {
            const int r = yo - (2 * kp + (lq >> 1)) + 2;     // dy row of this lane's vertical tap
            const __bf16* src = Dq[((r % 6) + 6) % 6] + (wave * 64 + l15) * 16 + 8 * (lq & 1);
            const bool rok = (unsigned)r < (unsigned)H;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bf16x8 fb = *reinterpret_cast<const bf16x8*>(src + t * 256);
                if (!rok) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) fb[e] = (__bf16)0.f;
                }
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kp][ct], fb, acc[t][ct], 0, 0, 0);
            }
        }
        // acc[t][2m + u][j] = dx[yo][q = wave*64 + 16t + l15][c = 32 m + 8 lq + 4 u + j]: 16 bytes per lane and (t, m)
        __bf16* orow = dx + ((size_t)img * H + yo) * W * lddx;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int q = g0 + wave * 64 + 16 * t + l15;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (__bf16)acc[t][2 * m + (e >> 2)][e & 3];
                if (q < W && c0 + 32 * m < C) *reinterpret_cast<bf16x8*>(orow + (size_t)q * lddx + c0 + 32 * m + 8 * lq) = o;
            }
        }
    }
}

bool head_dgrad_mfma_applies(int W, int C) { return W <= 4096 && C >= 32 && C <= 512 && C % 32 == 0; }

int launch_head_dgrad_mfma(const void* dy, int lddy, int B, int H, int W, const float* w, int C, void* dx, int lddx, hipStream_t s) {
    const int ncb = (W + 255) / 256;
    const unsigned grid = (unsigned)(B * ((H + HD_ROWS - 1) / HD_ROWS) * ncb);
    hipLaunchKernelGGL(head_dgrad_mfma_kernel, dim3(grid, (C + 63) / 64), dim3(256), 0, s, (const __bf16*)dy, lddy, B, H, W, w, C, (__bf16*)dx, lddx,
                       ncb);
    return (int)hipGetLastError();
}
