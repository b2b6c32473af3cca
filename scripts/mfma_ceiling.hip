// What can the matrix pipes of THIS MI355X sustain on random bf16 data?  (round 4; DESIGN.md section 6, 'the ceiling the MFMA kernels
// sit under').  The dense peak the bench line prices against is 2.5 PFLOP/s = 256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz; under an
// MFMA-dense loop the chip lowers its clock (MI355X_MICROARCH.md, 'DVFS give-back'), so the rate a kernel can reach is the pipe's
// FLOP per cycle x the clock the chip HOLDS.  Three loops, every CU busy (one 512-thread workgroup per CU = two waves per SIMD, the
// occupancy of conv3x3p / conv3x3g / wgrad3x3g), random operands, >= 2 s of back-to-back launches before the timed ones:
//   regs     v_mfma_f32_16x16x32_bf16 back to back, operands in registers (32 accumulator tiles per wave, as conv3x3p holds)
//   lds      the same MFMA stream with conv3x3p's fragment traffic: 24 ds_read_b128 per 96 MFMAs from a 64 KB LDS image, no barriers
//   lds32    `lds` with v_mfma_f32_32x32x16_bf16 (wgrad3x3g's shape): 48 MFMAs of twice the work per 24 reads
// Per loop: TFLOP/s chip-wide, the in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz, median over workgroups) and
// FLOP per cycle per CU.  hipcc --offload-arch=gfx950 -O3 scripts/mfma_ceiling.hip -o /tmp/mfma_ceiling && /tmp/mfma_ceiling
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Stamp { unsigned long long cyc, real; };

// MODE 0: registers only; 1: 16x16x32 with LDS fragment reads; 2: 32x32x16 with LDS fragment reads
template <int MODE>
__global__ __launch_bounds__(512) void loop_kernel(const uint32_t* __restrict__ src, float* __restrict__ sink, int iters, Stamp* stamps) {
    __shared__ __attribute__((aligned(16))) uint32_t img[16384];              // 64 KB of random bf16 pairs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += 512) img[i] = src[(blockIdx.x * 16384 + i) & 0xFFFFF];
    __syncthreads();
    u32x4 wf[12], pf[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        wf[i] = *reinterpret_cast<const u32x4*>(&img[((wave * 24 + i) * 64 + lane) * 4 & 16383]);
        pf[i] = *reinterpret_cast<const u32x4*>(&img[((wave * 24 + 12 + i) * 64 + lane) * 4 & 16383]);
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (MODE == 2) {
        f32x16 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)img + lane * 16;
        uint32_t off = 0;
        for (int it = 0; it < iters; ++it) {
            const uint32_t base = lbase + off;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[i]) : "v"(base), "n"(i * 1024));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pf[i]) : "v"(base), "n"(12288 + i * 1024));
            }
            off = (off + 4096) & 0x7FFF;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int g = 0; g < 6; ++g)
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[2 * g + (t & 1)]),
                                                                     __builtin_bit_cast(bf16x8, pf[2 * g + (t >> 2)]), acc[t], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[i][e];
        if (s == 123.456f) sink[blockIdx.x * 512 + tid] = s;
    } else {
        f32x4 acc[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)img + lane * 16;
        uint32_t off = 0;
        for (int it = 0; it < iters; ++it) {
            if constexpr (MODE == 1) {
                const uint32_t base = lbase + off;
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[i]) : "v"(base), "n"(i * 1024));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pf[i]) : "v"(base), "n"(12288 + i * 1024));
                }
                off = (off + 4096) & 0x7FFF;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int t = 0; t < 32; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[4 * g + (t & 3)]),
                                                                     __builtin_bit_cast(bf16x8, pf[4 * g + ((t >> 2) & 3)]), acc[t], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
        if (s == 123.456f) sink[blockIdx.x * 512 + tid] = s;
    }
    if (tid == 0) { stamps[blockIdx.x].cyc = __builtin_amdgcn_s_memtime() - c0; stamps[blockIdx.x].real = __builtin_amdgcn_s_memrealtime() - r0; }
}

// ---- loop STRUCTURES (round 4): the same arithmetic per CU and iteration (768 MFMAs of 16x16x32, 192 KB of fragment
// reads) organised two ways.
//   barrier8: conv3x3p's shape - 8 waves (two per SIMD), each 24 ds_read_b128 behind the barrier, then 96 MFMAs, one s_barrier per iteration
//   pipe4:    4 waves (ONE per SIMD, 512 registers each: 64 accumulator tiles), 48 reads per iteration of which the second half is
//             issued for the NEXT iteration while this iteration's MFMAs run; 192 MFMAs per wave, one s_barrier per iteration
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void struct_kernel(const uint32_t* __restrict__ src, float* __restrict__ sink, int iters, Stamp* stamps) {
    __shared__ __attribute__((aligned(16))) uint32_t img[16384];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += WAVES * 64) img[i] = src[(blockIdx.x * 16384 + i) & 0xFFFFF];
    __syncthreads();
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)img + lane * 16;
    uint32_t off = 0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (WAVES == 8) {
        f32x4 acc[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 wf[12], pf[12];
        for (int it = 0; it < iters; ++it) {
            const uint32_t base = lbase + off;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[i]) : "v"(base), "n"(i * 1024));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pf[i]) : "v"(base), "n"(12288 + i * 1024));
            }
            off = (off + 4096) & 0x7FFF;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int t = 0; t < 32; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[4 * g + (t & 3)]),
                                                                     __builtin_bit_cast(bf16x8, pf[4 * g + ((t >> 2) & 3)]), acc[t], 0, 0, 0);
            __builtin_amdgcn_s_barrier();
        }
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) sm += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
        if (sm == 123.456f) sink[blockIdx.x * 512 + tid] = sm;
    } else {
        f32x4 acc[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 fa[24], fb[24];                 // first / second half of an iteration's 48 fragments
#pragma unroll
        for (int i = 0; i < 24; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[i]) : "v"(lbase), "n"(i * 1024));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // software pipeline: 8 fragment reads for the group that is 96 MFMAs ahead, then 32 MFMAs on fragments that landed long ago
        // (lgkmcnt is a 4-bit counter: at most the 8 reads just issued are allowed to be outstanding at each wait)
#define GRP(SRC, DSTACC, G)                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                                                              \
    _Pragma("unroll") for (int t = 0; t < 32; ++t)                                                                  \
        acc[DSTACC + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, SRC[8 * G + (t & 3)]), \
                                                                  __builtin_bit_cast(bf16x8, SRC[8 * G + 4 + ((t >> 2) & 3)]), acc[DSTACC + t], 0, 0, 0)
        for (int it = 0; it < iters; ++it) {
            const uint32_t base = lbase + off + 24576;
            off = (off + 4096) & 0x3FFF;
            const uint32_t nbase = lbase + off;
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[i]) : "v"(base), "n"(i * 1024));
            GRP(fa, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[8 + i]) : "v"(base), "n"((8 + i) * 1024));
            GRP(fa, 0, 1);
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[16 + i]) : "v"(base), "n"((16 + i) * 1024));
            GRP(fa, 0, 2);
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[i]) : "v"(nbase), "n"(i * 1024));
            GRP(fb, 32, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[8 + i]) : "v"(nbase), "n"((8 + i) * 1024));
            GRP(fb, 32, 1);
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[16 + i]) : "v"(nbase), "n"((16 + i) * 1024));
            GRP(fb, 32, 2);
            __builtin_amdgcn_s_barrier();
        }
#undef GRP
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) sm += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
        if (sm == 123.456f) sink[blockIdx.x * 512 + tid] = sm;
    }
    if (tid == 0) { stamps[blockIdx.x].cyc = __builtin_amdgcn_s_memtime() - c0; stamps[blockIdx.x].real = __builtin_amdgcn_s_memrealtime() - r0; }
}

template <int WAVES>
static int run_struct(const char* name, const uint32_t* src, float* sink, Stamp* stamps, int cus) {
    const int iters = 4000;
    const double flop_per_iter_cu = 768.0 * 16 * 16 * 32 * 2;                 // 8 x 96 = 4 x 192 MFMAs
    auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.5) {
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(struct_kernel<WAVES>, dim3(cus), dim3(WAVES * 64), 0, 0, src, sink, iters, stamps);
        CK(hipDeviceSynchronize());
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 8;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(struct_kernel<WAVES>, dim3(cus), dim3(WAVES * 64), 0, 0, src, sink, iters, stamps);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> h(cus);
    CK(hipMemcpy(h.data(), stamps, sizeof(Stamp) * cus, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (auto& s : h) ghz.push_back((double)s.cyc / (double)s.real * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double tf = flop_per_iter_cu * iters * cus * reps / (ms * 1e-3) / 1e12;
    const double clk = ghz[ghz.size() / 2];
    printf("{\"loop\": \"%s\", \"tflops\": %.1f, \"clock_ghz\": %.3f, \"flop_per_cycle_per_cu\": %.0f, \"ms_per_launch\": %.3f, \"frac_of_2500\": %.3f}\n",
           name, tf, clk, tf * 1e12 / (clk * 1e9) / cus, ms / reps, tf / 2500.0);
    return 0;
}

template <int MODE>
static int run(const char* name, const uint32_t* src, float* sink, Stamp* stamps, int cus) {
    const int iters = 4000;
    const double flop_per_iter_wave = 96.0 * 16 * 16 * 32 * 2;                // both shapes: 96 x 16384 = 48 x 32768
    auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.5) {        // let the clock settle under this load
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(loop_kernel<MODE>, dim3(cus), dim3(512), 0, 0, src, sink, iters, stamps);
        CK(hipDeviceSynchronize());
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 8;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(loop_kernel<MODE>, dim3(cus), dim3(512), 0, 0, src, sink, iters, stamps);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> h(cus);
    CK(hipMemcpy(h.data(), stamps, sizeof(Stamp) * cus, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (auto& s : h) ghz.push_back((double)s.cyc / (double)s.real * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double flops = flop_per_iter_wave * 8.0 * iters * cus * reps;
    const double tf = flops / (ms * 1e-3) / 1e12;
    const double clk = ghz[ghz.size() / 2];
    printf("{\"loop\": \"%s\", \"tflops\": %.1f, \"clock_ghz\": %.3f, \"clock_ghz_min\": %.3f, \"clock_ghz_max\": %.3f, \"flop_per_cycle_per_cu\": %.0f, "
           "\"ms_per_launch\": %.3f, \"frac_of_2500\": %.3f}\n", name, tf, clk, ghz.front(), ghz.back(), tf * 1e12 / (clk * 1e9) / cus, ms / reps, tf / 2500.0);
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    uint32_t* src; float* sink; Stamp* stamps;
    CK(hipMalloc(&src, 4 << 20)); CK(hipMalloc(&sink, (size_t)cus * 512 * 4)); CK(hipMalloc(&stamps, sizeof(Stamp) * cus));
    std::vector<uint32_t> h(1 << 20);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (auto& v : h) {       // bf16 pairs in (-2, 2): random mantissas and signs, moderate exponents (no inf / nan / denormals)
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
        const uint32_t a = (lo & 0x807Fu) | ((0x7Cu + (lo >> 8 & 3)) << 7), b = (hi & 0x807Fu) | ((0x7Cu + (hi >> 8 & 3)) << 7);
        v = a | (b << 16);
    }
    CK(hipMemcpy(src, h.data(), 4 << 20, hipMemcpyHostToDevice));
    printf("{\"device\": \"%s\", \"cus\": %d}\n", p.name, cus);
    if (run<0>("regs_16x16x32", src, sink, stamps, cus)) return 1;
    if (run<1>("lds_16x16x32", src, sink, stamps, cus)) return 1;
    if (run<2>("lds_32x32x16", src, sink, stamps, cus)) return 1;
    if (run_struct<8>("barrier8: 8 waves, 24 reads behind the barrier, 96 MFMAs, barrier", src, sink, stamps, cus)) return 1;
    if (run_struct<4>("pipe4: 4 waves (one per SIMD), reads one half-iteration ahead, 192 MFMAs, barrier", src, sink, stamps, cus)) return 1;
    return 0;
}
