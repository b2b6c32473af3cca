// conv3x3r.hip - bf16 3x3 stride-1 'same' convolution (forward / data gradient) with REGISTER reuse of the patch rows.
//
// conv3x3.hip reads one pixel fragment and one weight fragment from LDS per pair of MFMAs (1 KB of LDS reads per
// v_mfma_f32_32x32x16_bf16), which at 128 B/clk/CU ties the LDS pipe to the matrix pipe.  Here a wave owns 4 image rows
// x 32 columns x 64 output channels: for a fixed horizontal tap dx, a patch-row fragment read once from LDS feeds the
// three vertical taps (output rows r, r-1, r-2), and each weight fragment feeds the 4 output rows - 12 fragment reads
// per 24 MFMAs (0.5 KB / MFMA), one barrier per 48 MFMAs per wave.
//   workgroup = 4 waves: <WM=2, WN=2>  8 rows x 32 cols x 128 channels      <WM=4, WN=1>  16 rows x 32 cols x 64 channels
//   K chunk   = 32 input channels (64-byte pixel rows, XOR-swizzled 16-byte granules instead of padding)
//   LDS       = patch (rows+2) x 34 px x 64 B  +  2 x [3 dy taps][64*WN channels][64 B] weights (double-buffered per dx)
// Accumulators are transposed (rows = output channel, lanes = pixel) and leave through an LDS-staged epilogue as
// 16-byte NHWC stores, like conv3x3.hip.  The data gradient is the same kernel with flipped taps.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

// The epilogue staging tile is private to a wave: LDS operations of one wave execute in issue order, so its reads see its
// own earlier writes without a workgroup barrier; this only stops the compiler from moving LDS accesses across the point.
#define WAVE_LDS_FENCE() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define RPC 34          // patch columns (32 + halo)
#define RKE 32          // input channels per chunk

template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv3x3r_bf16_kernel(const Conv3Args a) {
    constexpr int TR = 4 * WM, PR = TR + 2, NPX = PR * RPC, BN = 64 * WN;
    constexpr int AJ = (NPX * 4 + 255) / 256;         // 16-byte slots of the patch per thread
    constexpr int BJ = 3 * BN * 4 / 256;              // 16-byte slots of a 3-tap weight tile per thread
    constexpr int A_BYTES = NPX * 64, B_BYTES = 3 * BN * 64;
    constexpr int SROW = 64 * 2 + 16;                 // epilogue staging row: 64 channels + 16 B pad
    constexpr int STAGE_BYTES = 4 * 64 * SROW;
    constexpr int SMEM = (A_BYTES + 2 * B_BYTES) > STAGE_BYTES ? (A_BYTES + 2 * B_BYTES) : STAGE_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    unsigned char* As = smem;
    unsigned char* Bs = smem + A_BYTES;

    const __bf16* __restrict__ in = (const __bf16*)a.in;
    const __bf16* __restrict__ w = (const __bf16*)a.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;

    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + TR - 1) / TR;
    const int ntN = (a.N + BN - 1) / BN;
    int id = blockIdx.x;
    if ((gridDim.x & 7) == 0) id = (id & 7) * (gridDim.x >> 3) + (id >> 3);   // consecutive tiles on one XCD (shared L2)
    const int nt = id % ntN; id /= ntN;
    const int tx = id % tiles_x; id /= tiles_x;
    const int ty = id % tiles_y;
    const int img = id / tiles_y;
    const int y0 = ty * TR, x0 = tx * 32, n0 = nt * BN;

    const int C = a.C;
    const int nchunks = (C + RKE - 1) / RKE;
    const int ldw = 9 * C;
    const int g4 = tid & 3;                           // 16-byte granule inside a 64-byte row

    uint4 ra[AJ], rb[BJ];
    auto load_a = [&](int c0) {
        const bool cok = (c0 + g4 * 8) < C;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int p = (tid + 256 * j) >> 2;
            const int pr = p / RPC, pc = p - pr * RPC;
            const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (p < NPX && cok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                v = *reinterpret_cast<const uint4*>(in + ((size_t)((long long)img * a.H + iy) * a.W + ix) * a.ldi + c0 + g4 * 8);
            ra[j] = v;
        }
    };
    auto store_a = [&]() {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int p = (tid + 256 * j) >> 2;
            const int pc = p % RPC;                      // swizzle on the COLUMN: one fragment read stays inside a patch row
            if (p < NPX) *reinterpret_cast<uint4*>(As + p * 64 + ((g4 << 4) ^ ((pc & 12) << 2))) = ra[j];
        }
    };
    auto load_b = [&](int step) {                     // step = chunk * 3 + dx: the three dy taps of column dx
        const int ch = step / 3, dx = step - ch * 3;
        const int c0 = ch * RKE;
        const bool cok = (c0 + g4 * 8) < C;
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int row = (tid + 256 * j) >> 2;     // dy * BN + local channel
            const int dy = row / BN, nl = row - dy * BN;
            const int t = dy * 3 + dx;
            const int wt = (a.flip & 1) ? 8 - t : t;
            const int n = n0 + nl;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (cok && n < a.N) v = *reinterpret_cast<const uint4*>(w + (size_t)n * ldw + wt * C + c0 + g4 * 8);
            rb[j] = v;
        }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int row = (tid + 256 * j) >> 2;
            *reinterpret_cast<uint4*>(Bs + buf * B_BYTES + row * 64 + ((g4 << 4) ^ ((row & 12) << 2))) = rb[j];
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nsteps = nchunks * 3;
    load_a(0);
    load_b(0);
    store_a();
    store_b(0);
    __syncthreads();
    if (nsteps > 1) load_b(1);

    const int b_row = wn * 64 + l31;                  // + dy * BN + j * 32 (multiples of 32: the swizzle term is lane-constant)
    const int b_swz = (l31 & 12) << 2;
    int step = 0;
    for (int ch = 0; ch < nchunks; ++ch) {
        if (ch + 1 < nchunks) load_a((ch + 1) * RKE);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx, ++step) {
            if (step + 1 < nsteps) store_b((step + 1) & 1);
            if (step + 2 < nsteps) load_b(step + 2);
            const unsigned char* Bb = Bs + (step & 1) * B_BYTES;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int g16 = (kk * 2 + hi) << 4;
                const unsigned char* Ab = As + (4 * wm * RPC + l31 + dx) * 64 + (g16 ^ (((l31 + dx) & 12) << 2));
                bf16x8 fb[3][2];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        fb[dy][j] = *reinterpret_cast<const bf16x8*>(Bb + (dy * BN + j * 32 + b_row) * 64 + (g16 ^ b_swz));
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    const bf16x8 fa = *reinterpret_cast<const bf16x8*>(Ab + r * (RPC * 64));
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int orow = r - dy;
                        if (orow < 0 || orow > 3) continue;
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[orow][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[dy][j], fa, acc[orow][j], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
        }
        if (ch + 1 < nchunks) {
            store_a();
            __syncthreads();
        }
    }

    // ---- epilogue through LDS, two image rows of the wave at a time: acc[i][j] holds
    // D[n = 32j + (r&3) + 8(r>>2) + 4*hi][pixel column = l31] of image row y0 + 4*wm + i.
    unsigned char* stage = smem + wave * (64 * SROW);
    constexpr int LPP = 8;                            // lanes per pixel (16 B = 8 channels each, 64 channels)
    constexpr int PPP = 8;                            // pixels per pass
    const int cq = lane % LPP, pl = lane / LPP;
    const int nq = n0 + wn * 64 + cq * 8;
    __bf16* __restrict__ out = (__bf16*)a.out;
    const __bf16* __restrict__ addend = (const __bf16*)a.addend;
    float cs_s[8], cs_q[8];                           // fused column statistics (a.colstat, <4,1> tiles only)
#pragma unroll
    for (int e = 0; e < 8; ++e) { cs_s[e] = 0.f; cs_q[e] = 0.f; }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) WAVE_LDS_FENCE();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int nl = 32 * j + 8 * qd + 4 * hi;
                const int n = n0 + wn * 64 + nl;
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.bias && n + 3 < a.N) bv = *reinterpret_cast<const float4*>(a.bias + n);
                else if (a.bias) { float* bp = &bv.x; for (int e = 0; e < 4; ++e) if (n + e < a.N) bp[e] = a.bias[n + e]; }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const f32x16& c = acc[2 * half + i][j];
                    bf16x4 o;
                    o[0] = (__bf16)(c[4 * qd + 0] + bv.x); o[1] = (__bf16)(c[4 * qd + 1] + bv.y);
                    o[2] = (__bf16)(c[4 * qd + 2] + bv.z); o[3] = (__bf16)(c[4 * qd + 3] + bv.w);
                    *reinterpret_cast<bf16x4*>(stage + (32 * i + l31) * SROW + nl * 2) = o;
                }
            }
        }
        WAVE_LDS_FENCE();
#pragma unroll
        for (int ps = 0; ps < 64 / PPP; ++ps) {
            const int p = ps * PPP + pl;
            const int y = y0 + 4 * wm + 2 * half + (p >> 5), x = x0 + (p & 31);
            if (y >= a.H || x >= a.W || nq >= a.N) continue;
            uint4 v = *reinterpret_cast<const uint4*>(stage + p * SROW + cq * 16);
            const size_t pix = ((size_t)img * a.H + y) * a.W + x;
            if (addend) {
                const bf16x8 ad = *reinterpret_cast<const bf16x8*>(addend + pix * a.ldadd + nq);
                bf16x8 vv = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) vv[e] = (__bf16)((float)vv[e] + (float)ad[e]);
                v = __builtin_bit_cast(uint4, vv);
            }
            if (a.colstat) {
                const bf16x8 sv = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)sv[e]; cs_s[e] += f; cs_q[e] += f * f; }
            }
            *reinterpret_cast<uint4*>(out + pix * a.ldo + nq) = v;
        }
    }
    if constexpr (WN == 1) {
        if (a.colstat) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int off = 8; off < 64; off <<= 1) { cs_s[e] += __shfl_xor(cs_s[e], off); cs_q[e] += __shfl_xor(cs_q[e], off); }
            }
            float* red = reinterpret_cast<float*>(smem + 4 * 64 * SROW);   // [4 wm][64 ch][2], past the staging tiles
            if (lane < 8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    red[(wm * 64 + lane * 8 + e) * 2 + 0] = cs_s[e];
                    red[(wm * 64 + lane * 8 + e) * 2 + 1] = cs_q[e];
                }
            }
            __syncthreads();
            if (tid < 128) {
                const int ch = tid >> 1, st = tid & 1;
                const float t = ((red[(0 * 64 + ch) * 2 + st] + red[(1 * 64 + ch) * 2 + st]) + red[(2 * 64 + ch) * 2 + st]) +
                                red[(3 * 64 + ch) * 2 + st];
                const size_t row = ((size_t)img * tiles_y + ty) * tiles_x + tx;
                if (n0 + ch < a.N) a.colstat[(row * a.N + n0 + ch) * 2 + st] = t;
            }
        }
    }
}

int launch_conv3x3r_bf16(const Conv3Args& a, hipStream_t s) {
    const long long tx = (a.W + 31) / 32;
    if (a.N > 64) {
        const long long tiles = (long long)a.B * ((a.H + 7) / 8) * tx * ((a.N + 127) / 128);
        hipLaunchKernelGGL((conv3x3r_bf16_kernel<2, 2>), dim3((unsigned)tiles), dim3(256), 0, s, a);
    } else {
        const long long tiles = (long long)a.B * ((a.H + 15) / 16) * tx;
        hipLaunchKernelGGL((conv3x3r_bf16_kernel<4, 1>), dim3((unsigned)tiles), dim3(256), 0, s, a);
    }
    return (int)hipGetLastError();
}
