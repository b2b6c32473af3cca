// What a streaming pass over bf16 activation tensors can reach on this chip, by access pattern and loop shape - the ceiling the BatchNorm
// passes (bn_apply: 1 read + 1 write; chan_partial<2>: 2 reads; bn_bwd_apply: 2 reads + 1 write) are measured against.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/micro_stream scripts/micro_stream.hip && /tmp/micro_stream
// Every kernel does the arithmetic of the real pass on 16-byte vectors (8 bf16); U = vectors per thread and loop trip issued before any is
// consumed; NT = non-temporal loads / stores; grid = workgroups of 256 threads.  Prints one JSON line per (pattern, size, U, NT, grid).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NT> __device__ __forceinline__ bf16x8 ld(const bf16x8* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}
template <int NT> __device__ __forceinline__ void st(bf16x8* p, bf16x8 v) {
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// pattern 0: y = relu(x * sc + sh)                      (1 read + 1 write)
// pattern 1: s += g * mask(x), t += g * xhat             (2 reads; sums kept per thread, one atomic-free store of partials at the end)
// pattern 2: dx = sc * (g * mask - c1 - xhat * c2)       (2 reads + 1 write)
template <int PAT, int U, int NT>
__global__ __launch_bounds__(256) void stream_kernel(const bf16x8* __restrict__ x, const bf16x8* __restrict__ g, bf16x8* __restrict__ y,
                                                     long long n, float sc, float sh, float* __restrict__ part) {
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float s0 = 0.f, s1 = 0.f;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        bf16x8 xv[U], gv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            xv[u] = ld<NT>(x + i + u * stride);
            if (PAT) gv[u] = ld<NT>(g + i + u * stride);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bf16x8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float xf = (float)xv[u][k];
                const float a = xf * sc + sh;
                if (PAT == 0) {
                    o[k] = (__bf16)(a > 0.f ? a : 0.f);
                } else {
                    const float gm = a > 0.f ? (float)gv[u][k] : 0.f;
                    const float xh = (xf - sh) * sc;
                    if (PAT == 1) { s0 += gm; s1 += gm * xh; }
                    else o[k] = (__bf16)(sc * (gm - sh - xh * sc));
                }
            }
            if (PAT != 1) st<NT>(y + i + u * stride, o);
        }
    }
    for (; i < n; i += stride) {       // tail
        bf16x8 xv = x[i], o;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float a = (float)xv[k] * sc + sh; o[k] = (__bf16)(a > 0.f ? a : 0.f); s0 += a; }
        if (PAT != 1) y[i] = o;
    }
    if (PAT == 1) { part[(long long)blockIdx.x * 256 + threadIdx.x] = s0 + s1; }
}

template <int PAT, int U, int NT>
static void run(const char* name, const bf16x8* x, const bf16x8* g, bf16x8* y, float* part, long long bytes, int grid, int passes) {
    const long long n = bytes / 16;
    if ((long long)U * grid * 256 * 4 > n) return;          // fewer than four trips of the unrolled loop: not a streaming measurement
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((stream_kernel<PAT, U, NT>), dim3(grid), dim3(256), 0, 0, x, g, y, n, 1.01f, 0.02f, part);
    CK(hipEventRecord(e0, 0));
    const int K = 20;
    for (int w = 0; w < K; ++w) hipLaunchKernelGGL((stream_kernel<PAT, U, NT>), dim3(grid), dim3(256), 0, 0, x, g, y, n, 1.01f, 0.02f, part);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / K;
    printf("{\"pattern\": \"%s\", \"tensor_MB\": %.1f, \"U\": %d, \"nt\": %d, \"grid\": %d, \"us\": %.1f, \"TBps\": %.3f}\n", name, bytes / 1e6, U, NT, grid, us,
           passes * (double)bytes / us / 1e6);
    fflush(stdout);
}

int main() {
    const long long big = 32LL * 256 * 256 * 64 * 2;          // 268 MB: the full-resolution tensors of configs[1]
    bf16x8 *x, *g, *y; float* part;
    CK(hipMalloc(&x, big)); CK(hipMalloc(&g, big)); CK(hipMalloc(&y, big)); CK(hipMalloc(&part, 8192 * 256 * 4));
    std::vector<uint16_t> h(big / 2);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3C00 + ((s >> 9) & 0x3FF)) ^ (uint16_t)((s >> 31) << 15); }
    CK(hipMemcpy(x, h.data(), big, hipMemcpyHostToDevice)); CK(hipMemcpy(g, h.data(), big, hipMemcpyHostToDevice));
    for (long long bytes : {big, big / 2, big / 4, big / 8}) {
        for (int grid : {2048, 4096, 8192}) {
#define ROW(PAT, NAME, PASSES) \
            run<PAT, 1, 0>(NAME, x, g, y, part, bytes, grid, PASSES); run<PAT, 2, 0>(NAME, x, g, y, part, bytes, grid, PASSES); \
            run<PAT, 4, 0>(NAME, x, g, y, part, bytes, grid, PASSES); run<PAT, 1, 1>(NAME, x, g, y, part, bytes, grid, PASSES); \
            run<PAT, 2, 1>(NAME, x, g, y, part, bytes, grid, PASSES); run<PAT, 4, 1>(NAME, x, g, y, part, bytes, grid, PASSES);
            ROW(0, "1r1w", 2) ROW(1, "2r", 2) ROW(2, "2r1w", 3)
        }
    }
    return 0;
}
