// Stand-alone micro-benchmark of the split-K reduction out[i] = sum_s part[s][i] (igemm.hip: splitk_reduce_kernel): block shapes
// (L float4 lanes x G slab groups), slab stride padding, after a writer kernel that leaves the slabs in the caches the way the
// weight-gradient kernel does.   hipcc --offload-arch=gfx950 -O3 scripts/micro_splitk.hip -o /tmp/ms && /tmp/ms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int L, int G>
__global__ __launch_bounds__(L * G) void red_kernel(const float* __restrict__ part, int nsplit, size_t n, size_t stride, float* __restrict__ out) {
    __shared__ float4 red[G][L];
    const int lane = threadIdx.x % L, grp = threadIdx.x / L;
    const size_t i4 = ((size_t)blockIdx.x * L + lane) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i4 < n) {
        const float* p = part + i4;
        int k = grp;
        for (; k + 7 * G < nsplit; k += 8 * G) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(k + u * G) * stride);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; k + G < nsplit; k += 2 * G) {
            const float4 v0 = *reinterpret_cast<const float4*>(p + (size_t)k * stride);
            const float4 v1 = *reinterpret_cast<const float4*>(p + (size_t)(k + G) * stride);
            s.x += v0.x + v1.x; s.y += v0.y + v1.y; s.z += v0.z + v1.z; s.w += v0.w + v1.w;
        }
        for (; k < nsplit; k += G) {
            const float4 v = *reinterpret_cast<const float4*>(p + (size_t)k * stride);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    if (G > 1) {
        red[grp][lane] = s;
        __syncthreads();
        if (grp != 0) return;
#pragma unroll
        for (int g = 1; g < G; ++g) { const float4 v = red[g][lane]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    }
    if (i4 < n) *reinterpret_cast<float4*>(out + i4) = s;
}

// no LDS: a thread owns U float4 outputs (block-strided) and walks ALL slabs for them, nsplit * U loads in flight
template <int U>
__global__ __launch_bounds__(256) void red_flat_kernel(const float* __restrict__ part, int nsplit, size_t n, size_t stride, float* __restrict__ out) {
    const size_t n4 = n / 4;
    size_t i = ((size_t)blockIdx.x * U) * 256 + threadIdx.x;
    float4 s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) s[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < nsplit; k += 4) {
        float4 v[4][U];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t j = i + (size_t)u * 256;
                v[kk][u] = (k + kk < nsplit && j < n4) ? *reinterpret_cast<const float4*>(part + (size_t)(k + kk) * stride + j * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int u = 0; u < U; ++u) { s[u].x += v[kk][u].x; s[u].y += v[kk][u].y; s[u].z += v[kk][u].z; s[u].w += v[kk][u].w; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t j = i + (size_t)u * 256;
        if (j < n4) *reinterpret_cast<float4*>(out + j * 4) = s[u];
    }
}

__global__ void writer_kernel(float* __restrict__ p, size_t total) {          // stands for the kernel that wrote the slabs
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total / 4; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<float4*>(p)[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
    struct Case { size_t n; int nsplit; } cases[] = {{36864, 256}, {73728, 128}, {147456, 64}, {294912, 32}, {589824, 16}, {1179648, 8}, {2359296, 4}, {4718592, 2}, {9216, 256}, {18432, 256}};
    float *part, *out;
    const size_t cap = (size_t)48 << 20;        // floats
    CK(hipMalloc(&part, cap * 4)); CK(hipMalloc(&out, (size_t)8 << 22));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto c : cases) {
        for (size_t pad : {(size_t)0, (size_t)64, (size_t)1088}) {
            const size_t stride = c.n + pad;
            if (stride * c.nsplit > cap) continue;
            const size_t n4 = c.n / 4;
            auto time = [&](auto launch) -> float {
                float tot = 0.f;
                for (int it = 0; it < 12; ++it) {
                    hipLaunchKernelGGL(writer_kernel, dim3(2048), dim3(256), 0, st, part, stride * c.nsplit);
                    hipEventRecord(e0, st);
                    launch();
                    hipEventRecord(e1, st);
                    hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (it >= 2) tot += ms;
                }
                return tot / 10 * 1e3f;
            };
            const float t_cur = time([&] { hipLaunchKernelGGL((red_kernel<64, 8>), dim3((n4 + 63) / 64), dim3(512), 0, st, part, c.nsplit, c.n, stride, out); });
            const float t_w = time([&] { hipLaunchKernelGGL((red_kernel<16, 32>), dim3((n4 + 15) / 16), dim3(512), 0, st, part, c.nsplit, c.n, stride, out); });
            const float t_32 = time([&] { hipLaunchKernelGGL((red_kernel<32, 16>), dim3((n4 + 31) / 32), dim3(512), 0, st, part, c.nsplit, c.n, stride, out); });
            const float t_64_16 = time([&] { hipLaunchKernelGGL((red_kernel<64, 16>), dim3((n4 + 63) / 64), dim3(1024), 0, st, part, c.nsplit, c.n, stride, out); });
            const float t_64_4 = time([&] { hipLaunchKernelGGL((red_kernel<64, 4>), dim3((n4 + 63) / 64), dim3(256), 0, st, part, c.nsplit, c.n, stride, out); });
            const float t_64_2 = time([&] { hipLaunchKernelGGL((red_kernel<64, 2>), dim3((n4 + 63) / 64), dim3(128), 0, st, part, c.nsplit, c.n, stride, out); });
            const float t_64_1 = time([&] { hipLaunchKernelGGL((red_kernel<64, 1>), dim3((n4 + 63) / 64), dim3(64), 0, st, part, c.nsplit, c.n, stride, out); });
            const float t_f1 = time([&] { hipLaunchKernelGGL((red_flat_kernel<1>), dim3((n4 + 255) / 256), dim3(256), 0, st, part, c.nsplit, c.n, stride, out); });
            const float t_f2 = time([&] { hipLaunchKernelGGL((red_flat_kernel<2>), dim3((n4 + 511) / 512), dim3(256), 0, st, part, c.nsplit, c.n, stride, out); });
            const double mb = (double)c.n * 4 * (c.nsplit + 1) / 1e6;
            printf("n %8zu x %3d slabs pad %4zu (%6.1f MB): 64x8 %6.1f | 16x32 %6.1f | 32x16 %6.1f | 64x16 %6.1f | 64x4 %6.1f | 64x2 %6.1f | 64x1 %6.1f | flat1 %6.1f | flat2 %6.1f us\n",
                   c.n, c.nsplit, pad, mb, t_cur, t_w, t_32, t_64_16, t_64_4, t_64_2, t_64_1, t_f1, t_f2);
            fflush(stdout);
        }
    }
    return 0;
}
