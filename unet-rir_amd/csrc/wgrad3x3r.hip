// wgrad3x3r.hip - bf16 weight gradient of the 3x3 stride-1 convolution with REGISTER reuse of the input-patch rows.
//
// dW[n][kh][kw][c] = sum_p dy[p][n] * x[p + (kh-1, kw-1)][c]: a GEMM whose K dimension is pixels.  A workgroup (4 waves,
// each a 32 x 32 (n, c) tile for all 9 taps: 144 accumulator registers) owns a 64 x 64 (n, c) tile and a split-K slice of
// TPH x 16 pixel patches.  wgrad3x3_bf16_kernel (igemm_bf16.hip) makes one K step of 2 patch rows x 8 pixels and reads
// 1 + 9 operand fragments for its 9 MFMAs.  Here a K step is ONE patch row of 16 pixels, so the x fragment of tap
// (kh, kw) at row r is the fragment of tap (kh-1, kw) at row r+1: a step reads only the three new fragments of x row
// r + 2 and one dy fragment - 8 ds_read_b64_tr_b16 instead of 20 per 9 v_mfma_f32_32x32x16_bf16.
// LDS rows are 192 B apart (4 consecutive pixel rows of a transposed read land on distinct 64-byte bank groups).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_r;

#define RW_TPW 16
#define RW_LD 96              // LDS pixel stride, elements (192 B)

template <int TPH>
__global__ __launch_bounds__(256, 2) void wgrad3x3r_bf16_kernel(const Wgrad3ArgsH a) {
    constexpr int XH = TPH + 2, XW = RW_TPW + 2;
    constexpr int XN = XH * XW * 8;                  // 16-byte slots of the x patch (8 per pixel: 64 channels)
    constexpr int XJ = (XN + 255) / 256;
    constexpr int DN = TPH * RW_TPW * 8;
    constexpr int DJ = (DN + 255) / 256;
    __shared__ __attribute__((aligned(16))) __bf16 Xs[XH * XW * RW_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Ds[TPH * RW_TPW * RW_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;

    const int ntC = (a.C + 63) / 64;
    const int rt = blockIdx.x / ntC, ct = blockIdx.x - rt * ntC;
    const int n0 = rt * 64, c0 = ct * 64;
    const int per_img = a.npy * a.npx;
    const int G = a.B * per_img;
    const int g0 = blockIdx.y * a.patches_per_split;
    int g1 = g0 + a.patches_per_split;
    if (g1 > G) g1 = G;

    uint4 rx[XJ], rd[DJ];
    const int q8 = tid & 7;
    const bool cok = (c0 + q8 * 8) < a.C, nok = (n0 + q8 * 8) < a.N;

    auto load_patch = [&](int g) {
        const int img = g / per_img;
        const int rem = g - img * per_img;
        const int pyi = rem / a.npx, pxi = rem - pyi * a.npx;
        const int py0 = pyi * TPH, px0 = pxi * RW_TPW;
        const int iy0 = py0 - a.pad_t, ix0 = px0 - a.pad_l;
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            const int i = tid + 256 * j;
            const int pp = i >> 3;
            const int pr = pp / XW, pc = pp - pr * XW;
            const int iy = iy0 + pr, ix = ix0 + pc;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (i < XN && cok && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW)
                v = *reinterpret_cast<const uint4*>(a.x + ((size_t)((long long)img * a.IH + iy) * a.IW + ix) * a.ldx + c0 + q8 * 8);
            rx[j] = v;
        }
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            const int i = tid + 256 * j;
            const int pix = i >> 3;
            const int oy = py0 + (pix >> 4), ox = px0 + (pix & 15);
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (i < DN && nok && oy < a.OH && ox < a.OW)
                v = *reinterpret_cast<const uint4*>(a.dy + ((size_t)((long long)img * a.OH + oy) * a.OW + ox) * a.lddy + n0 + q8 * 8);
            rd[j] = v;
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ds_read_b64_tr_b16: in each 16-lane group lane 4q+p supplies the address of LDS row q (a pixel), columns 4p..4p+3
    // (channels); lane i of the group receives column i of the 4 rows.  Operand lane l wants channel l&31 and pixels
    // 8*(l>>5)+j of the 16-pixel patch row: groups 0/1 cover channels 0-15/16-31 of pixels 0-7, groups 2/3 of pixels 8-15.
    const int grp = lane >> 4, li = lane & 15;
    const int tq = li >> 2, tp = li & 3;
    const int chan = (grp & 1) * 16 + tp * 4;
    const int dlane = (h * 8 + tq) * RW_LD + wr * 32 + chan;         // + (r * 16 + 4 * rd) * RW_LD
    const int xlane = (h * 8 + tq) * RW_LD + wc * 32 + chan;         // + ((r + kh) * XW + kw + 4 * rd) * RW_LD

    auto read_x = [&](int row, int kw) {
        const __bf16* p = Xs + xlane + (row * XW + kw) * RW_LD;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_r*)p);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_r*)(p + 4 * RW_LD));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto read_d = [&](int row) {
        const __bf16* p = Ds + dlane + (row * RW_TPW) * RW_LD;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_r*)p);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_r*)(p + 4 * RW_LD));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    if (g0 < g1) load_patch(g0);
    for (int g = g0; g < g1; ++g) {
        if (g != g0) __syncthreads();
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            const int i = tid + 256 * j;
            if (i < XN) *reinterpret_cast<uint4*>(&Xs[(i >> 3) * RW_LD + q8 * 8]) = rx[j];
        }
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            const int i = tid + 256 * j;
            if (i < DN) *reinterpret_cast<uint4*>(&Ds[(i >> 3) * RW_LD + q8 * 8]) = rd[j];
        }
        __syncthreads();
        if (g + 1 < g1) load_patch(g + 1);
        // rotating window of x-row fragments: xf[row % 3][kw]
        bf16x8 xf[3][3];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) { xf[0][kw] = read_x(0, kw); xf[1][kw] = read_x(1, kw); }
#pragma unroll
        for (int r = 0; r < TPH; ++r) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) xf[(r + 2) % 3][kw] = read_x(r + 2, kw);
            const bf16x8 fa = read_d(r);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, xf[(r + kh) % 3][kw], acc[kh * 3 + kw], 0, 0, 0);
        }
    }

    float* part = a.part + (size_t)blockIdx.y * a.N * 9 * a.C;
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

// split-K plan for TPH x 16 patches: about `target` workgroups, at least 4 patches per slice
static void plan_r(int TPH, int B, int OH, int OW, int N, int C, int* nsplit, int* per_split, int* npy, int* npx) {
    const long long target = 512;                        // split-K workgroups aimed for
    *npy = (OH + TPH - 1) / TPH;
    *npx = (OW + RW_TPW - 1) / RW_TPW;
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

// stride-1 3x3 weight gradient; returns WGRAD3X3R_NOT_TAKEN when this kernel does not take the layer (the caller falls back)
int launch_wgrad3x3r_bf16(Wgrad3ArgsH a, float* dw, float reg, const float* w, void* ws, size_t ws_bytes, hipStream_t s) {
    const bool on = unetrir_cfg().wgrad3x3r != 0;
    if (!on || a.OH % 8 != 0) return WGRAD3X3R_NOT_TAKEN;
    int ns, per;
    plan_r(8, a.B, a.OH, a.OW, a.N, a.C, &ns, &per, &a.npy, &a.npx);
    const size_t nout = (size_t)a.N * 9 * a.C;
    const bool direct = (ns == 1 && reg == 0.f);
    if (!direct && ws_bytes < (size_t)ns * nout * sizeof(float)) return WGRAD3X3R_NOT_TAKEN;
    a.part = direct ? dw : (float*)ws;
    a.patches_per_split = per;
    const unsigned tiles = (unsigned)(((a.N + 63) / 64) * ((a.C + 63) / 64));
    hipLaunchKernelGGL((wgrad3x3r_bf16_kernel<8>), dim3(tiles, ns), dim3(256), 0, s, a);
    const int err = (int)hipGetLastError();
    if (err || direct) return err;
    return launch_splitk_reduce((const float*)ws, ns, nout, dw, reg, w, s);
}
