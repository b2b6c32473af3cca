// conv3x3.hip - 3x3 stride-1 'same' convolution (forward and its data gradient) with the input patch staged ONCE in LDS.
//
// The tap-table kernels (igemm.hip / igemm_bf16.hip) re-gather the activation tile for every tap, so 9/10 of the bytes
// they stage are re-reads.  Here a workgroup (8 waves) owns an 8 x 32 pixel tile of one image: the (8+2) x (32+2)
// pixel patch of the current 128-byte channel chunk is staged once and all 9 taps read it at shifted LDS addresses;
// only the 128-row weight tile changes per tap (double-buffered, one barrier per tap).  An M sub-tile of the MFMA is
// one 32-pixel image row, so fragment reads have the same conflict-free pattern as the gathered kernels.
// T = float  : v_mfma_f32_32x32x2_f32,  32 channels per chunk
// T = __bf16 : v_mfma_f32_32x32x16_bf16, 64 channels per chunk (fp32 accumulate, bias fp32)
// The data gradient of a stride-1 3x3 conv is the same kernel with the taps flipped (weight tap 8-t at offset t) and the
// [Cin][T][Cout] weight copy.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define TROWS 8
#define TCOLS 32
#define PROWS (TROWS + 2)
#define PCOLS (TCOLS + 2)
#define RSB 144                     // LDS row stride in bytes: 128 B of channels + 16 B pad (conflict-free b128 reads)
#define NPIX (PROWS * PCOLS)        // 340 patch pixels

template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int KE = 32, EPS = 4; };
template <> struct Elem<__bf16> { static constexpr int KE = 64, EPS = 8; };

__device__ __forceinline__ void mma4(f32x16& acc, const uint4& a, const uint4& b, float) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}
// bf16: the weight fragment is the A operand, so the accumulator comes out TRANSPOSED (rows/registers = output channel,
// columns/lanes = pixel): a lane then owns 4 consecutive channels per register quad, which pack into 8-byte LDS writes
// for the staged epilogue (2-byte global stores straight from the MFMA layout cost 35-40 % of the kernel).
__device__ __forceinline__ void mma4(f32x16& acc, const uint4& a, const uint4& b, __bf16) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}

template <typename T, int BN_>
__global__ __launch_bounds__(512) void conv3x3_kernel(const Conv3Args a) {
    constexpr int KE = Elem<T>::KE, EPS = Elem<T>::EPS;
    constexpr int NSUB = BN_ / 64;                   // 32-wide N sub-tiles per wave (waves: 4 along M x 2 along N)
    constexpr int AJ = (NPIX * 8 + 511) / 512;       // 16-byte slots of the patch per thread (6)
    constexpr int BJ = BN_ * 8 / 512;                // 16-byte slots of a weight tile per thread (2 or 1)
    __shared__ __attribute__((aligned(16))) unsigned char smem[NPIX * RSB + 2 * BN_ * RSB];
    unsigned char* As = smem;                                   // NPIX * RSB
    unsigned char* Bs = smem + NPIX * RSB;                      // 2 * BN_ * RSB

    const T* __restrict__ in = (const T*)a.in;
    const T* __restrict__ w = (const T*)a.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hb = (lane >> 5) * 16;

    const int tiles_x = (a.W + TCOLS - 1) / TCOLS, tiles_y = (a.H + TROWS - 1) / TROWS;
    const int ntN = (a.N + BN_ - 1) / BN_;
    int id = blockIdx.x;
    const int nt = id % ntN; id /= ntN;              // N tile fastest: the workgroups sharing a patch are neighbours
    const int tx = id % tiles_x; id /= tiles_x;
    const int ty = id % tiles_y;
    const int img = id / tiles_y;
    const int y0 = ty * TROWS, x0 = tx * TCOLS, n0 = nt * BN_;

    const int C = a.C;
    const int nchunks = (C + KE - 1) / KE;
    const int ldw = 9 * C;
    const int q = tid & 7;                            // 16-byte slot inside a 128-byte row

    uint4 ra[AJ], rb[BJ];
    auto load_a = [&](int c0) {
        const bool cok = (c0 + q * EPS) < C;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int p = (tid + 512 * j) >> 3;
            const int pr = p / PCOLS, pc = p - pr * PCOLS;
            const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (p < NPIX && cok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                v = *reinterpret_cast<const uint4*>(in + ((size_t)((long long)img * a.H + iy) * a.W + ix) * a.ldi + c0 + q * EPS);
            ra[j] = v;
        }
    };
    auto store_a = [&]() {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int p = (tid + 512 * j) >> 3;
            if (p < NPIX) *reinterpret_cast<uint4*>(As + p * RSB + q * 16) = ra[j];
        }
    };
    auto load_b = [&](int step) {                     // step = chunk * 9 + tap
        const int ch = step / 9, t = step - ch * 9;
        const int c0 = ch * KE;
        const int wt = (a.flip & 1) ? 8 - t : t;
        const bool cok = ch < nchunks && (c0 + q * EPS) < C;
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int n = n0 + ((tid + 512 * j) >> 3);
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (cok && n < a.N) v = *reinterpret_cast<const uint4*>(w + (size_t)n * ldw + wt * C + c0 + q * EPS);
            rb[j] = v;
        }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int j = 0; j < BJ; ++j)
            *reinterpret_cast<uint4*>(Bs + buf * (BN_ * RSB) + ((tid + 512 * j) >> 3) * RSB + q * 16) = rb[j];
    };

    f32x16 acc[2][NSUB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NSUB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // prologue: patch of chunk 0 and weight tile of step 0 into LDS; weight tile of step 1 in flight
    load_a(0);
    load_b(0);
    store_a();
    store_b(0);
    __syncthreads();
    load_b(1);

    const int a_lane = ((2 * wm) * PCOLS + l31) * RSB + hb;           // + (i*PCOLS + kh*PCOLS + kw) * RSB + kk*32
    const int b_lane = (wn * (BN_ / 2) + l31) * RSB + hb;             // + j*32*RSB + kk*32
    const int nsteps = nchunks * 9;
    int step = 0;
    for (int ch = 0; ch < nchunks; ++ch) {
        if (ch + 1 < nchunks) load_a((ch + 1) * KE);                  // next patch: in registers until the chunk ends
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw, ++step) {
                // stage the NEXT weight tile into the other buffer (last read two steps ago, a barrier since), refill regs
                if (step + 1 < nsteps) store_b((step + 1) & 1);
                if (step + 2 < nsteps) load_b(step + 2);
                const unsigned char* Ab = As + a_lane + (kh * PCOLS + kw) * RSB;
                const unsigned char* Bb = Bs + (step & 1) * (BN_ * RSB) + b_lane;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    uint4 fa[2], fb[NSUB];
#pragma unroll
                    for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const uint4*>(Ab + i * (PCOLS * RSB) + kk * 32);
#pragma unroll
                    for (int j = 0; j < NSUB; ++j) fb[j] = *reinterpret_cast<const uint4*>(Bb + j * (32 * RSB) + kk * 32);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < NSUB; ++j) mma4(acc[i][j], fa[i], fb[j], T());
                }
                __syncthreads();
            }
        }
        if (ch + 1 < nchunks) {          // every wave has finished reading the patch (barrier above): replace it
            store_a();
            __syncthreads();
        }
    }

    if (a.flip & 2) {   // timing experiments only (scripts/micro_conv.py): skip the epilogue stores, keep the accumulators live
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NSUB; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" :: "v"(acc[i][j]));
#endif
            }
        return;
    }
    if constexpr (sizeof(T) == 2) {
        // ---- bf16 epilogue through LDS: acc[i][j] holds D[n = 32j + (r&3) + 8(r>>2) + 4h][pixel = 32i + l31].
        // (1) + bias, pack 4 consecutive channels -> one ds_write_b64 into this wave's [64 px][BN/2 ch] staging tile;
        // (2) read it back as 16-byte channel runs of one pixel and store 16 B per lane (coalesced NHWC rows).
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        constexpr int WN = BN_ / 2;                       // channels per wave
        constexpr int SROW = WN * 2 + 16;                 // staging row stride in bytes (16-byte pad)
        unsigned char* stage = smem + wave * (64 * SROW); // 8 waves x <= 9216 B, the K loop is over (barrier passed)
        const int hq = lane >> 5;
#pragma unroll
        for (int j = 0; j < NSUB; ++j) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int nl = 32 * j + 8 * qd + 4 * hq;                   // first of 4 consecutive channels (wave-local)
                const int n = n0 + wn * WN + nl;
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.bias && n + 3 < a.N) bv = *reinterpret_cast<const float4*>(a.bias + n);
                else if (a.bias) { float* bp = &bv.x; for (int e = 0; e < 4; ++e) if (n + e < a.N) bp[e] = a.bias[n + e]; }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    bf16x4 o;
                    o[0] = (__bf16)(acc[i][j][4 * qd + 0] + bv.x); o[1] = (__bf16)(acc[i][j][4 * qd + 1] + bv.y);
                    o[2] = (__bf16)(acc[i][j][4 * qd + 2] + bv.z); o[3] = (__bf16)(acc[i][j][4 * qd + 3] + bv.w);
                    *reinterpret_cast<bf16x4*>(stage + (32 * i + l31) * SROW + nl * 2) = o;
                }
            }
        }
        __syncthreads();
        constexpr int LPP = WN / 8;                       // lanes per pixel (16 B = 8 channels each)
        constexpr int PPP = 64 / LPP;                     // pixels per pass
        const int cq = lane % LPP, pl = lane / LPP;
        const int n = n0 + wn * WN + cq * 8;
        T* __restrict__ out = (T*)a.out;
        const T* __restrict__ addend = (const T*)a.addend;
#pragma unroll
        for (int ps = 0; ps < 64 / PPP; ++ps) {
            const int p = ps * PPP + pl;                  // wave-local pixel: image row 2*wm + (p>>5), column p & 31
            const int y = y0 + 2 * wm + (p >> 5), x = x0 + (p & 31);
            if (y >= a.H || x >= a.W || n >= a.N) continue;
            uint4 v = *reinterpret_cast<const uint4*>(stage + p * SROW + cq * 16);
            const size_t pix = ((size_t)img * a.H + y) * a.W + x;
            if (addend) {
                const bf16x8 ad = *reinterpret_cast<const bf16x8*>(addend + pix * a.ldadd + n);
                bf16x8 vv = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) vv[e] = (__bf16)((float)vv[e] + (float)ad[e]);
                v = __builtin_bit_cast(uint4, vv);
            }
            *reinterpret_cast<uint4*>(out + pix * a.ldo + n) = v;
        }
        return;
    }
    // epilogue: accumulator col = lane&31 -> n, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> pixel column inside the image row
    T* __restrict__ out = (T*)a.out;
    const T* __restrict__ addend = (const T*)a.addend;
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
        const int n = n0 + wn * (BN_ / 2) + 32 * j + l31;
        if (n >= a.N) continue;
        const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int y = y0 + 2 * wm + i;
            if (y >= a.H) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (x >= a.W) continue;
                const size_t pix = ((size_t)img * a.H + y) * a.W + x;
                float v = acc[i][j][r] + bias;
                if (addend) v += (float)addend[pix * a.ldadd + n];
                out[pix * a.ldo + n] = (T)v;
            }
        }
    }
}

template <typename T>
static int launch_conv3x3_t(const Conv3Args& a, hipStream_t s) {
    const long long tiles = (long long)a.B * ((a.H + TROWS - 1) / TROWS) * ((a.W + TCOLS - 1) / TCOLS);
    if (a.N > 64) hipLaunchKernelGGL((conv3x3_kernel<T, 128>), dim3((unsigned)(tiles * ((a.N + 127) / 128))), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((conv3x3_kernel<T, 64>), dim3((unsigned)tiles), dim3(512), 0, s, a);
    return (int)hipGetLastError();
}

// true when launch_conv3x3(a, bf16 = 1) lands on a kernel with the fused column statistics (conv3x3g, conv3x3r<4,1>)
bool conv3x3_has_colstat(const Conv3Args& a) {
    const bool rowreuse = unetrir_cfg().conv3x3r != 0, dma = unetrir_cfg().conv3x3g != 0;
    if (dma && conv3x3g_applies(a)) return true;
    if (conv3x3s_applies(a) || conv3x3h_applies(a)) return true;
    return rowreuse && a.N <= 64 && !(a.flip & 2);
}

int launch_conv3x3(const Conv3Args& a, int bf16, hipStream_t s) {
    // bf16: the row-reuse kernel (conv3x3r.hip) unless the conv3x3r switch is off or a timing ablation asks for the no-store variant
    const bool rowreuse = unetrir_cfg().conv3x3r != 0, dma = unetrir_cfg().conv3x3g != 0, stem = unetrir_cfg().stem != 0;
    if (bf16 && stem && stem3x3_applies(a)) return launch_stem3x3_bf16(a, s);          // first layer: 8 stored channels -> 64
    if (bf16 && dma && conv3x3p_applies(a)) return launch_conv3x3p_bf16(a, s);
    if (bf16 && dma && conv3x3g_applies(a)) return launch_conv3x3g_bf16(a, s);
    if (bf16 && conv3x3s_applies(a)) return launch_conv3x3s_bf16(a, s);
    if (bf16 && conv3x3h_applies(a)) return launch_conv3x3h_bf16(a, s);
    if (bf16 && rowreuse && !(a.flip & 2)) return launch_conv3x3r_bf16(a, s);
    return bf16 ? launch_conv3x3_t<__bf16>(a, s) : launch_conv3x3_t<float>(a, s);
}
