// stem3x3.hip - bf16 3x3 stride-1 'same' convolution for the network's first layer (gfx950): 8 stored input channels
// (2 real + 6 zero, one 16-byte pixel), output channels in blocks of 64 (encoding_block level 0, dl_models/u_net.py:269-276 with
// resize_factor_0 = [1, 1]).
//
// The layer is 19 GFLOP against a 268 MB bf16 output at cfg 2: HBM-write bound by a factor ~8, so the kernel is built
// around the store stream and nothing is staged in LDS on the way in.  K = 9 taps x 8 channels; one
// v_mfma_f32_32x32x16_bf16 consumes TWO taps: the lower 32 lanes of an operand hold tap 2s, the upper 32 tap 2s+1 (the tenth
// half-step is zero).  A pixel IS one 16-byte load, so the patch operand of a 32-pixel row segment is fetched straight
// from global memory / L2 by the lane that feeds it to the matrix core (the 33 MB input is L2 resident, neighbouring taps
// hit the same lines); the weight operand (64 x 72 bf16) lives in 40 registers per lane for the whole workgroup.
// A wave owns a 32-column strip and walks STEM_ROWS rows of it: 10 MFMAs per row segment, the 32 x 64 result goes through a
// wave-private LDS tile so that every global store instruction writes whole 128-byte pixel rows.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define WAVE_LDS_FENCE() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)

namespace {
constexpr int STEM_ROWS = 8;              // rows per wave; a workgroup (4 waves) covers 32 rows x 32 columns
constexpr int STEM_SROW = 64 * 2 + 16;    // staging row: 64 channels + 16 bytes of padding
}  // namespace

__global__ __launch_bounds__(256) void stem3x3_bf16_kernel(const Conv3Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char stage_all[4 * 32 * STEM_SROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const __bf16* __restrict__ in = (const __bf16*)a.in;
    const __bf16* __restrict__ w = (const __bf16*)a.w;
    __bf16* __restrict__ out = (__bf16*)a.out;
    unsigned char* stage = stage_all + wave * 32 * STEM_SROW;

    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + 4 * STEM_ROWS - 1) / (4 * STEM_ROWS);
    const int ntN = a.N >> 6;                                // 64 output channels per workgroup (N % 64 == 0)
    int id = blockIdx.x;
    const int n0 = (id % ntN) << 6; id /= ntN;
    const int tx = id % tiles_x; id /= tiles_x;
    const int ty = id % tiles_y;
    const int img = id / tiles_y;
    const int x0 = tx * 32, y0 = ty * 4 * STEM_ROWS + wave * STEM_ROWS;

    // weight operand: row n = 32 j + l31, half-step s covers tap 2s + hi; element k of the lane = channel k of that tap
    bf16x8 wf[5][2];
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = 2 * s + hi, n = n0 + 32 * j + l31;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (t < 9 && n < a.N) v = *reinterpret_cast<const uint4*>(w + ((size_t)n * 9 + t) * 8);
            wf[s][j] = __builtin_bit_cast(bf16x8, v);
        }
    // bias of the channels this lane converts: n = 32 j + 8 qd + 4 hi + {0..3}
    float4 bv[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int n = n0 + 32 * j + 8 * qd + 4 * hi;
            bv[j][qd] = (a.bias && n + 3 < a.N) ? *reinterpret_cast<const float4*>(a.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    // per-lane tap geometry: half-step s -> tap t = 2s + hi -> (dy, dx)
    int tdy[5], tdx[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int t = 2 * s + hi;
        tdy[s] = t < 9 ? t / 3 - 1 : 1 << 20;      // the tenth half-step never passes the bounds test
        tdx[s] = t % 3 - 1;
    }
    const __bf16* __restrict__ img_in = in + (size_t)img * a.H * a.W * a.ldi;
    const int cq = lane & 7, pl = lane >> 3;

    for (int ry = 0; ry < STEM_ROWS; ++ry) {
        const int y = y0 + ry;
        if (y >= a.H) break;                       // uniform across the wave
        bf16x8 pf[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int iy = y + tdy[s], ix = x0 + l31 + tdx[s];
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                v = *reinterpret_cast<const uint4*>(img_in + ((size_t)iy * a.W + ix) * a.ldi);
            pf[s] = __builtin_bit_cast(bf16x8, v);
        }
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 5; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s][j], pf[s], acc[j], 0, 0, 0);
        // acc[j][r] = D[n = 32 j + (r & 3) + 8 (r >> 2) + 4 hi][pixel column l31]
        if (ry) WAVE_LDS_FENCE();                  // the previous row's reads of the staging tile are done
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                bf16x4 o;
                o[0] = (__bf16)(acc[j][4 * qd + 0] + bv[j][qd].x); o[1] = (__bf16)(acc[j][4 * qd + 1] + bv[j][qd].y);
                o[2] = (__bf16)(acc[j][4 * qd + 2] + bv[j][qd].z); o[3] = (__bf16)(acc[j][4 * qd + 3] + bv[j][qd].w);
                *reinterpret_cast<bf16x4*>(stage + l31 * STEM_SROW + (32 * j + 8 * qd + 4 * hi) * 2) = o;
            }
        WAVE_LDS_FENCE();
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int p = ps * 8 + pl, x = x0 + p;
            if (x >= a.W) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(stage + p * STEM_SROW + cq * 16);
            *reinterpret_cast<uint4*>(out + (((size_t)img * a.H + y) * a.W + x) * a.ldo + n0 + cq * 8) = v;
        }
    }
}

// bf16, exactly 8 stored input channels, output channels in blocks of 64, plain forward (no addend, no fused statistics)
bool stem3x3_applies(const Conv3Args& a) {
    return a.C == 8 && a.N >= 64 && (a.N & 63) == 0 && a.flip == 0 && !a.addend && !a.colstat && a.ldi >= 8 && (a.ldi & 7) == 0 && (a.ldo & 7) == 0 &&
           (((uintptr_t)a.in | (uintptr_t)a.w | (uintptr_t)a.out) & 15) == 0 && (!a.bias || ((uintptr_t)a.bias & 15) == 0);
}

int launch_stem3x3_bf16(const Conv3Args& a, hipStream_t s) {
    const long long tiles = (long long)a.B * ((a.H + 4 * STEM_ROWS - 1) / (4 * STEM_ROWS)) * ((a.W + 31) / 32) * (a.N >> 6);
    hipLaunchKernelGGL(stem3x3_bf16_kernel, dim3((unsigned)tiles), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}
