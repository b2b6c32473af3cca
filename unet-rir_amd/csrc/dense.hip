// dense.hip - data gradient of Dense(N) on a small batch without a transposed weight copy (gfx950).
//
//   dx[b][k] = sum_n dy[b][n] * w[n][k]        w = the [N][K] kernel exactly as the forward pass stores it
//
// Dense(h5*w5*16) of the information-vector branch (dl_models/u_net.py:259) holds 49 % of the network's parameters
// (8192 x 4096 at cfg 2: 134 MB fp32).  Its data gradient used to be the forward kernel on a transposed copy, which costs a
// 268 MB transpose pass per step on top of the 134 MB the product itself has to read.  Here the output columns run along the
// CONTIGUOUS dimension of w, so the matrix streams once with 16-byte loads and no transpose exists:
//   * a workgroup is 4 waves; a wave owns 256 output columns (one float4 per lane) and all B <= 32 batch rows: 128 fp32
//     accumulators per lane, 128 FMAs per 16-byte load (16 flop per byte: the balance point of v_pk_fma_f32 against HBM);
//   * the reduction dimension n is split over blockIdx.y (128 rows per slice): 8 column strips x 32 slices = 256 workgroups
//     at cfg 2; a slice's dy[b][n] values are staged transposed in LDS and broadcast-read ([n][b]: one ds_read_b128 serves 4
//     batch rows of every lane);
//   * 16 rows of w are requested before the first is used (16 KB in flight per wave; splitting them into two register
//     batches so that one is in flight while the other is consumed measured the same 56 us);
//   * raw partial sums per slice, then the fixed-order splitk_rows_reduce_kernel: deterministic, no atomics.
// Algorithmic bytes: N*K*4 (w) once; partials add 2 * slices * B * K * 4 (32 MB + 32 MB at cfg 2).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace {

constexpr int DD_BT = 32;            // batch rows per pass
constexpr int DD_ROWS = 128;         // rows of w (n) per slice
constexpr int DD_COLS = 1024;        // output columns per workgroup (4 waves x 64 lanes x float4)
constexpr int DD_PF = 16;            // rows requested ahead

}  // namespace

// (outside the anonymous namespace: rocprofv3 lists kernels with internal linkage under an empty name)
__global__ __launch_bounds__(256) void dense_dgrad_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ w,
                                                          int B, int K, int N, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float dyT[DD_ROWS][DD_BT];
    const int tid = threadIdx.x;
    const int n0 = blockIdx.y * DD_ROWS;
    const int rows = min(DD_ROWS, N - n0);
    for (int i = tid; i < DD_ROWS * DD_BT; i += 256) {
        const int r = i % DD_ROWS, b = i / DD_ROWS;            // consecutive threads read consecutive n of one batch row
        dyT[r][b] = (b < B && r < rows) ? dy[(size_t)b * lddy + n0 + r] : 0.f;
    }
    __syncthreads();
    const int c = blockIdx.x * DD_COLS + tid * 4;
    if (c >= K) return;                                        // K % 4 == 0: a lane's four columns are in or out together
    float4 acc[DD_BT];
#pragma unroll
    for (int b = 0; b < DD_BT; ++b) acc[b] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* __restrict__ wp = w + (size_t)n0 * K + c;
    for (int r0 = 0; r0 < rows; r0 += DD_PF) {
        float4 wv[DD_PF];
#pragma unroll
        for (int j = 0; j < DD_PF; ++j)
            wv[j] = (r0 + j < rows) ? *reinterpret_cast<const float4*>(wp + (size_t)(r0 + j) * K) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < DD_PF; ++j) {
            const float4* __restrict__ d4 = reinterpret_cast<const float4*>(&dyT[(r0 + j) & (DD_ROWS - 1)][0]);
#pragma unroll
            for (int q = 0; q < DD_BT / 4; ++q) {
                const float4 d = d4[q];
                const float dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float4& a = acc[4 * q + e];
                    a.x += dv[e] * wv[j].x; a.y += dv[e] * wv[j].y; a.z += dv[e] * wv[j].z; a.w += dv[e] * wv[j].w;
                }
            }
        }
    }
    float* __restrict__ po = part + (size_t)blockIdx.y * B * K + c;
#pragma unroll
    for (int b = 0; b < DD_BT; ++b)
        if (b < B) *reinterpret_cast<float4*>(po + (size_t)b * K) = acc[b];
}

bool dense_dgrad_applies(int B, int K, int N) { return B >= 1 && B <= DD_BT && K > 0 && (K & 3) == 0 && N > 0; }

size_t dense_dgrad_ws_bytes(int B, int K, int N) { return (size_t)((N + DD_ROWS - 1) / DD_ROWS) * B * K * sizeof(float); }

int launch_dense_dgrad(const float* dy, int lddy, const float* w, float* dx, int lddx, int B, int K, int N, void* ws, size_t ws_bytes,
                       hipStream_t s) {
    if (!dense_dgrad_applies(B, K, N) || ws_bytes < dense_dgrad_ws_bytes(B, K, N)) return UNETRIR_EINVAL;
    const int nsl = (N + DD_ROWS - 1) / DD_ROWS;
    hipLaunchKernelGGL(dense_dgrad_kernel, dim3((K + DD_COLS - 1) / DD_COLS, nsl), dim3(256), 0, s, dy, lddy, w, B, K, N, (float*)ws);
    int err = (int)hipGetLastError();
    if (err) return err;
    return launch_splitk_rows_reduce((const float*)ws, nsl, B, K, nullptr, dx, lddx, s);
}
