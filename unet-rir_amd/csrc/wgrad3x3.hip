// wgrad3x3.hip - weight gradient of the 3x3 Conv2D layers (stride 1 or 2, TF 'same'), fp32 MFMA.
//
// dw[n][t][c] = sum_p dy[p][n] * x[p*S + off_t][c].  The generic kernel (igemm.hip) gathers one x tile per tap,
// so every staged byte feeds 1/9 of the taps.  Here the K dimension is walked in 4x8-pixel patches: the x patch
// WITH HALO ((3S+3) x (7S+3) pixels) is staged once in LDS and all 9 taps read it at shifted addresses, and the dy
// patch is staged once for all 9 taps.  One workgroup owns a 64 (Cout) x 64 (Cin) x 9 (taps) slab of dw: each of
// its 4 waves keeps 9 accumulator tiles of 32x32 (144 AGPRs) and issues 9 MFMAs per 10 LDS reads.
// Staged bytes per MFMA fall ~5x against the gathered form; partial slabs + fixed-order reduce keep it deterministic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TPW 8            // patch width (output pixels); patch height TPH_ is 4 (stride 1) or 2 (stride 2)
#define W3_LD 68          // 64 channels + 4 pad (floats)

template <int SI, int TPH>
__global__ __launch_bounds__(256, 2) void wgrad3x3_kernel(const Wgrad3Args a) {
    constexpr int XH = (TPH - 1) * SI + 3, XW = (TPW - 1) * SI + 3;
    constexpr int DJ = TPH * TPW * 16 / 256;         // dy float4 rounds per thread
    constexpr int XN = XH * XW * 16;                 // float4 slots of the x patch (16 quads per pixel)
    constexpr int XJ = (XN + 255) / 256;
    __shared__ __attribute__((aligned(16))) float Xs[XH * XW * W3_LD];
    __shared__ __attribute__((aligned(16))) float Ds[TPH * TPW * W3_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;

    const int ntC = (a.C + 63) / 64;
    const int rt = blockIdx.x / ntC, ct = blockIdx.x - rt * ntC;
    const int n0 = rt * 64, c0 = ct * 64;
    const int per_img = a.npy * a.npx;
    const int G = a.B * per_img;
    const int g0 = blockIdx.y * a.patches_per_split;
    int g1 = g0 + a.patches_per_split;
    if (g1 > G) g1 = G;

    float4 rx[XJ], rd[DJ];
    const int q = tid & 15;
    const bool cok = (c0 + q * 4) < a.C, nok = (n0 + q * 4) < a.N;

    auto load_patch = [&](int g) {
        const int img = g / per_img;
        const int rem = g - img * per_img;
        const int pyi = rem / a.npx, pxi = rem - pyi * a.npx;
        const int py0 = pyi * TPH, px0 = pxi * TPW;
        const int iy0 = py0 * SI - a.pad_t, ix0 = px0 * SI - a.pad_l;
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            const int i = tid + 256 * j;
            const int pp = i >> 4;
            const int pr = pp / XW, pc = pp - pr * XW;
            const int iy = iy0 + pr, ix = ix0 + pc;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < XN && cok && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW)
                v = *reinterpret_cast<const float4*>(a.x + ((size_t)((long long)img * a.IH + iy) * a.IW + ix) * a.ldx + c0 + q * 4);
            rx[j] = v;
        }
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            const int pix = (tid + 256 * j) >> 4;
            const int oy = py0 + (pix >> 3), ox = px0 + (pix & 7);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (nok && oy < a.OH && ox < a.OW)
                v = *reinterpret_cast<const float4*>(a.dy + ((size_t)((long long)img * a.OH + oy) * a.OW + ox) * a.lddy + n0 + q * 4);
            rd[j] = v;
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const float* dbase = &Ds[h * W3_LD + wr * 32 + l31];
    const float* xbase = &Xs[h * SI * W3_LD + wc * 32 + l31];

    if (g0 < g1) load_patch(g0);
    for (int g = g0; g < g1; ++g) {
        if (g != g0) __syncthreads();
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            const int i = tid + 256 * j;
            if (i < XN) *reinterpret_cast<float4*>(&Xs[(i >> 4) * W3_LD + q * 4]) = rx[j];
        }
#pragma unroll
        for (int j = 0; j < DJ; ++j) *reinterpret_cast<float4*>(&Ds[((tid + 256 * j) >> 4) * W3_LD + q * 4]) = rd[j];
        __syncthreads();
        if (g + 1 < g1) load_patch(g + 1);
#pragma unroll
        for (int s = 0; s < TPH * TPW / 2; ++s) {
            const int k0 = 2 * s, r = k0 >> 3, cc = k0 & 7;      // pixel k0 + h = (r, cc + h) inside the patch
            const float fa = dbase[k0 * W3_LD];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int kh = t / 3, kw = t - kh * 3;
                const float fb = xbase[((r * SI + kh) * XW + cc * SI + kw) * W3_LD];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[t], 0, 0, 0);
            }
        }
    }

    // epilogue: rows (registers) = output channel n, cols (lanes) = input channel c
    float* part = a.part + (size_t)blockIdx.y * a.N * 9 * a.C;
    const int c = c0 + wc * 32 + l31;
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

void wgrad3x3_plan_tph(int TPH, int B, int OH, int OW, int N, int C, int* nsplit, int* per_split, int* npy, int* npx);
void wgrad3x3_plan(int stride, int B, int OH, int OW, int N, int C, int* nsplit, int* per_split, int* npy, int* npx) {
    wgrad3x3_plan_tph(stride == 1 ? 4 : 2, B, OH, OW, N, C, nsplit, per_split, npy, npx);
}

void wgrad3x3_plan_tph(int TPH, int B, int OH, int OW, int N, int C, int* nsplit, int* per_split, int* npy, int* npx) {
    *npy = (OH + TPH - 1) / TPH;
    *npx = (OW + TPW - 1) / TPW;
    const long long G = (long long)B * (*npy) * (*npx);
    const long long tiles = (long long)((N + 63) / 64) * ((C + 63) / 64);
    const long long target = 512;                        // split-K workgroups aimed for
    long long want = (target + tiles - 1) / tiles;
    long long maxs = (G + 3) / 4;                       // at least 4 patches per slice
    if (maxs < 1) maxs = 1;
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    const long long per = (G + want - 1) / want;
    *per_split = (int)per;
    *nsplit = (int)((G + per - 1) / per);
}

size_t wgrad3x3_ws_bytes(int stride, int B, int OH, int OW, int N, int C) {
    int ns, per, npy, npx;
    wgrad3x3_plan(stride, B, OH, OW, N, C, &ns, &per, &npy, &npx);
    size_t bytes = (size_t)ns * N * 9 * C * sizeof(float);
    if (stride == 2) {                       // the bf16 LDS-DMA kernel (wgrad3x3d.hip) has its own split plan
        const size_t d = wgrad3x3d_ws_bytes(B, OH, OW, N, C);
        if (d > bytes) bytes = d;
    }
    return bytes;
}

int launch_wgrad3x3(Wgrad3Args a, int stride, float* dw, float reg, const float* w, void* ws, size_t ws_bytes, hipStream_t s) {
    int ns, per;
    wgrad3x3_plan(stride, a.B, a.OH, a.OW, a.N, a.C, &ns, &per, &a.npy, &a.npx);
    const size_t nout = (size_t)a.N * 9 * a.C;
    const bool direct = (ns == 1 && reg == 0.f);
    if (!direct && ws_bytes < (size_t)ns * nout * sizeof(float)) return UNETRIR_EINVAL;
    a.part = direct ? dw : (float*)ws;
    a.patches_per_split = per;
    const unsigned tiles = (unsigned)(((a.N + 63) / 64) * ((a.C + 63) / 64));
    if (stride == 1) hipLaunchKernelGGL((wgrad3x3_kernel<1, 4>), dim3(tiles, ns), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((wgrad3x3_kernel<2, 2>), dim3(tiles, ns), dim3(256), 0, s, a);
    int err = (int)hipGetLastError();
    if (err || direct) return err;
    return launch_splitk_reduce((const float*)ws, ns, nout, dw, reg, w, s);
}
