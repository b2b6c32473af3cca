// upconv3x3.hip - the 2x up-sampling direction of a 3x3 stride-2 'same' layer pair in ONE launch:
//   * Conv2DTranspose(k=3, s=2, 'same') forward (dl_models/u_net.py:297-304), and
//   * the data gradient of Conv2D(k=3, s=2, 'same') on an even-sized input (dl_models/u_net.py:269-276),
// which are the same operator: out[2p + (kh, kw)] += in[p] . w[kh, kw]  (pad_before = 0, output cropped to 2h x 2w).
//
// The tap-table path runs it as 4 launches, one per output parity class, each with 4/2/2/1 taps: every launch re-reads
// the input and has only 1-4 K steps per channel chunk to hide its prologue.  Here a workgroup (8 waves) owns an 8 x 32
// tile of the COARSE grid: the (8+1) x (32+1) input patch of a 128-byte channel chunk is staged once in LDS, the 9 weight
// taps stream through a double buffer, and tap (kh, kw) accumulates into the accumulators of parity class (kh&1, kw&1)
// reading the patch at offset (-(kh>>1), -(kw>>1)).  Each wave keeps 4 classes x 2 row sub-tiles (128 accumulator regs).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define UR 8
#define UC 32
#define UPR (UR + 1)
#define UPC (UC + 1)
#define UNP (UPR * UPC)           // 297 patch pixels
#define RSB 144
#define UBN 64                    // output channels per workgroup (4 parity classes share the accumulator budget)

template <typename T> struct UElem;
template <> struct UElem<float> { static constexpr int KE = 32, EPS = 4; };
template <> struct UElem<__bf16> { static constexpr int KE = 64, EPS = 8; };

__device__ __forceinline__ void umma(f32x16& acc, const uint4& a, const uint4& b, float) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}
// bf16: weights as the A operand -> transposed accumulator (registers = channel, lanes = pixel) for the staged epilogue
__device__ __forceinline__ void umma(f32x16& acc, const uint4& a, const uint4& b, __bf16) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}

// CHUNKED (bf16): a 32-channel chunk (64-byte rows, 80-byte LDS stride) keeps the weights of ALL 9 taps in LDS, so a chunk
// costs two barriers instead of nine and a patch fragment (2 rows x 2 row offsets x 2 column offsets) is read once for
// the 9 taps: 17 fragment reads per 18 MFMAs instead of 12 per 8, 36 MFMAs per wave between barriers instead of 8.
template <typename T, bool CHUNKED = false>
__global__ __launch_bounds__(512) void upconv3x3_kernel(const Conv3Args a) {
    constexpr int KE = CHUNKED ? 32 : UElem<T>::KE, EPS = UElem<T>::EPS;
    constexpr int RS = CHUNKED ? 80 : RSB;            // LDS row stride, bytes
    constexpr int SLOTS = KE / EPS;                   // 16-byte slots per row (4 chunked, 8 otherwise)
    constexpr int AJ = (UNP * SLOTS + 511) / 512;     // patch slots per thread
    constexpr int BJ = CHUNKED ? (9 * UBN * SLOTS + 511) / 512 : 1;
    constexpr int SMEM_MAIN = CHUNKED ? UNP * RS + 9 * UBN * RS : UNP * RSB + 2 * UBN * RSB;
    constexpr int SMEM_EPI = 8 * 64 * (32 * 2 + 16);
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_MAIN > SMEM_EPI ? SMEM_MAIN : SMEM_EPI];
    unsigned char* As = smem;
    unsigned char* Bs = smem + UNP * RS;

    const T* __restrict__ in = (const T*)a.in;
    const T* __restrict__ w = (const T*)a.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hb = (lane >> 5) * 16;

    // a.H, a.W: coarse (input) grid; the output grid is 2H x 2W
    const int tiles_x = (a.W + UC - 1) / UC, tiles_y = (a.H + UR - 1) / UR;
    const int ntN = (a.N + UBN - 1) / UBN;
    int id = blockIdx.x;
    const int nt = id % ntN; id /= ntN;
    const int tx = id % tiles_x; id /= tiles_x;
    const int ty = id % tiles_y;
    const int img = id / tiles_y;
    const int y0 = ty * UR, x0 = tx * UC, n0 = nt * UBN;

    const int C = a.C;
    const int nchunks = (C + KE - 1) / KE;
    const int ldw = 9 * C;
    const int q = tid & (SLOTS - 1);
    constexpr int SL = CHUNKED ? 2 : 3;               // log2(SLOTS)

    uint4 ra[AJ], rb;
    auto load_a = [&](int c0) {
        const bool cok = (c0 + q * EPS) < C;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int p = (tid + 512 * j) >> SL;
            const int pr = p / UPC, pc = p - pr * UPC;
            const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (p < UNP && cok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                v = *reinterpret_cast<const uint4*>(in + ((size_t)((long long)img * a.H + iy) * a.W + ix) * a.ldi + c0 + q * EPS);
            ra[j] = v;
        }
    };
    auto store_a = [&]() {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int p = (tid + 512 * j) >> SL;
            if (p < UNP) *reinterpret_cast<uint4*>(As + p * RS + q * 16) = ra[j];
        }
    };
    // chunked: all 9 taps of the chunk, row = tap * 64 + channel
    uint4 rbc[BJ];
    auto load_bc = [&](int c0) {
        const bool cok = (c0 + q * EPS) < C;
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int row = (tid + 512 * j) >> SL;
            const int t = row >> 6, n = n0 + (row & 63);
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (row < 9 * UBN && cok && n < a.N) v = *reinterpret_cast<const uint4*>(w + (size_t)n * ldw + t * C + c0 + q * EPS);
            rbc[j] = v;
        }
    };
    auto store_bc = [&]() {
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int row = (tid + 512 * j) >> SL;
            if (row < 9 * UBN) *reinterpret_cast<uint4*>(Bs + row * RS + q * 16) = rbc[j];
        }
    };
    auto load_b = [&](int step) {                      // 64 rows x 8 slots = 512 slots: one per thread
        const int ch = step / 9, t = step - ch * 9;
        const int c0 = ch * KE;
        const int n = n0 + (tid >> 3);
        rb = make_uint4(0u, 0u, 0u, 0u);
        if (ch < nchunks && (c0 + q * EPS) < C && n < a.N) rb = *reinterpret_cast<const uint4*>(w + (size_t)n * ldw + t * C + c0 + q * EPS);
    };
    auto store_b = [&](int buf) { *reinterpret_cast<uint4*>(Bs + buf * (UBN * RSB) + (tid >> 3) * RSB + q * 16) = rb; };

    f32x16 acc[4][2];                                  // [parity class (ay*2+ax)][row sub-tile]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][i][r] = 0.f;

    if constexpr (CHUNKED) {
        load_a(0);
        load_bc(0);
        store_a();
        store_bc();
        __syncthreads();
        const int a_lane = ((2 * wm + 1) * UPC + l31 + 1) * RS + hb;
        const int b_lane = (wn * 32 + l31) * RS + hb;
        for (int ch = 0; ch < nchunks; ++ch) {
            const bool more = ch + 1 < nchunks;
            if (more) { load_a((ch + 1) * KE); load_bc((ch + 1) * KE); }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 fa[2][2][2];                         // [row sub-tile][row offset kh>>1][column offset kw>>1]
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int dr = 0; dr < 2; ++dr)
#pragma unroll
                        for (int dc = 0; dc < 2; ++dc)
                            fa[i][dr][dc] = *reinterpret_cast<const uint4*>(As + a_lane + ((i - dr) * UPC - dc) * RS + kk * 32);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const uint4 fb = *reinterpret_cast<const uint4*>(Bs + b_lane + (kh * 3 + kw) * (UBN * RS) + kk * 32);
                        const int cls = (kh & 1) * 2 + (kw & 1);
#pragma unroll
                        for (int i = 0; i < 2; ++i) umma(acc[cls][i], fa[i][kh >> 1][kw >> 1], fb, T());
                    }
            }
            __syncthreads();
            if (more) {
                store_a();
                store_bc();
                __syncthreads();
            }
        }
    } else {
    load_a(0);
        load_b(0);
        store_a();
        store_b(0);
        __syncthreads();
        load_b(1);
    
        // patch pixel of coarse (row 2wm+i, col l31) at tap offset (-(kh>>1), -(kw>>1)): ((2wm+i+1-(kh>>1)) * UPC + l31+1-(kw>>1))
        const int a_lane = ((2 * wm + 1) * UPC + l31 + 1) * RSB + hb;
        const int b_lane = (wn * 32 + l31) * RSB + hb;
        const int nsteps = nchunks * 9;
        int step = 0;
        for (int ch = 0; ch < nchunks; ++ch) {
            if (ch + 1 < nchunks) load_a((ch + 1) * KE);
    #pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
    #pragma unroll
                for (int kw = 0; kw < 3; ++kw, ++step) {
                    if (step + 1 < nsteps) store_b((step + 1) & 1);
                    if (step + 2 < nsteps) load_b(step + 2);
                    const int cls = (kh & 1) * 2 + (kw & 1);
                    const unsigned char* Ab = As + a_lane - ((kh >> 1) * UPC + (kw >> 1)) * RSB;
                    const unsigned char* Bb = Bs + (step & 1) * (UBN * RSB) + b_lane;
    #pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const uint4 fb = *reinterpret_cast<const uint4*>(Bb + kk * 32);
    #pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const uint4 fa = *reinterpret_cast<const uint4*>(Ab + i * (UPC * RSB) + kk * 32);
                            umma(acc[cls][i], fa, fb, T());
                        }
                    }
                    __syncthreads();
                }
            }
            if (ch + 1 < nchunks) {
                store_a();
                __syncthreads();
            }
        }
    
}

    const int OH = 2 * a.H, OW = 2 * a.W;
    T* __restrict__ out = (T*)a.out;
    const T* __restrict__ addend = (const T*)a.addend;
    if constexpr (sizeof(T) == 2) {
        // bf16: acc[c][i] = D[n = (r&3) + 8(r>>2) + 4h][coarse col l31].  Four rounds, one output image row each
        // (i, ay): stage [64 output cols][32 channels] per wave, then 16-byte channel runs -> coalesced NHWC stores.
        constexpr int SROW = 32 * 2 + 16;                 // bytes per staged output pixel
        unsigned char* stage = smem + wave * (64 * SROW);
        const int hq = lane >> 5;
        const int cq = lane & 3, pl = lane >> 2;          // readback: 4 lanes per pixel, 16 pixels per pass
        const int nrd = n0 + wn * 32 + cq * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int ay = 0; ay < 2; ++ay) {
#pragma unroll
                for (int ax = 0; ax < 2; ++ax) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const int nl = 8 * qd + 4 * hq;
                        const int n = n0 + wn * 32 + nl;
                        float bv[4] = {0.f, 0.f, 0.f, 0.f};
                        if (a.bias) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (n + e < a.N) bv[e] = a.bias[n + e];
                        }
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (__bf16)(acc[ay * 2 + ax][i][4 * qd + e] + bv[e]);
                        *reinterpret_cast<bf16x4*>(stage + (2 * l31 + ax) * SROW + nl * 2) = o;
                    }
                }
                __syncthreads();
                const int cy = y0 + 2 * wm + i;               // coarse row of this sub-tile
                const int oy = 2 * cy + ay;
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int oc = ps * 16 + pl;              // output column inside the wave's 64
                    const int ox = 2 * x0 + oc;
                    if (cy < a.H && ox < OW && nrd < a.N) {
                        uint4 v = *reinterpret_cast<const uint4*>(stage + oc * SROW + cq * 16);
                        const size_t pix = ((size_t)img * OH + oy) * OW + ox;
                        if (addend) {
                            const bf16x8 ad = *reinterpret_cast<const bf16x8*>(addend + pix * a.ldadd + nrd);
                            bf16x8 vv = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                            for (int e = 0; e < 8; ++e) vv[e] = (__bf16)((float)vv[e] + (float)ad[e]);
                            v = __builtin_bit_cast(uint4, vv);
                        }
                        *reinterpret_cast<uint4*>(out + pix * a.ldo + nrd) = v;
                    }
                }
                __syncthreads();
            }
        }
    } else {
        // fp32: standard accumulator (rows/registers = coarse pixel column, lanes = channel): 128-byte runs per store
        const int n = n0 + wn * 32 + l31;
        if (n < a.N) {
            const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ay = c >> 1, ax = c & 1;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int cy = y0 + 2 * wm + i;
                    if (cy >= a.H) continue;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int cx = x0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        if (cx >= a.W) continue;
                        const size_t pix = ((size_t)img * OH + 2 * cy + ay) * OW + 2 * cx + ax;
                        float v = acc[c][i][r] + bias;
                        if (addend) v += (float)addend[pix * a.ldadd + n];
                        out[pix * a.ldo + n] = (T)v;
                    }
                }
            }
        }
    }
}

int launch_upconv3x3(const Conv3Args& a, int bf16, hipStream_t s) {
    const long long tiles = (long long)a.B * ((a.H + UR - 1) / UR) * ((a.W + UC - 1) / UC) * ((a.N + UBN - 1) / UBN);
    if (bf16 && upconv3x3q_applies(a)) return launch_upconv3x3q_bf16(a, s);
    if (bf16 && upconv3x3g_applies(a)) return launch_upconv3x3g_bf16(a, s);
    if (bf16) hipLaunchKernelGGL((upconv3x3_kernel<__bf16, true>), dim3((unsigned)tiles), dim3(512), 0, s, a);
    else hipLaunchKernelGGL(upconv3x3_kernel<float>, dim3((unsigned)tiles), dim3(512), 0, s, a);
    return (int)hipGetLastError();
}
