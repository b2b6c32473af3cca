// head.hip - the output head Conv2D(2, (6,6), padding='same') (dl_models/u_net.py:248) as direct
// (non-MFMA) kernels.  With 2 output channels the implicit GEMM would waste 15/16 of every MFMA
// tile; here each lane owns output pixels (forward) or (tap, channel) pairs (weight gradient) and
// the 21x21-pixel input patch of a 16x16 output tile is staged once in LDS for all 36 taps.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "kernels.h"

#define HT 16                 // output tile edge
#define HK 6                  // kernel size
#define HP (HT + HK - 1)      // 21: patch edge
#define HPAD_T 2              // TF SAME for k=6, s=1: pad (2, 3)

typedef __bf16 bf16x4_h __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_h __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ld4h(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4h(const __bf16* p) {
    const bf16x4_h v = *reinterpret_cast<const bf16x4_h*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ float2 ld2h(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ float2 ld2h(const __bf16* p) {
    const bf16x2_h v = *reinterpret_cast<const bf16x2_h*>(p);
    return make_float2((float)v[0], (float)v[1]);
}

// stage channels [c0, c0+CC) of the patch around tile (ty0, tx0) of image n into LDS, zero outside the image
template <int CC, typename T>
__device__ __forceinline__ void load_patch(const T* __restrict__ x, int ldx, int H, int W, int n, int ty0, int tx0,
                                           int c0, float* __restrict__ As) {
    constexpr int Q = CC / 4, LD = CC + 4;
    for (int i = threadIdx.x; i < HP * HP * Q; i += 256) {
        const int pix = i / Q, q = i - pix * Q;
        const int py = pix / HP, px = pix - py * HP;
        const int iy = ty0 + py - HPAD_T, ix = tx0 + px - HPAD_T;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            v = ld4h(x + ((size_t)((long long)n * H + iy) * W + ix) * ldx + c0 + q * 4);
        *reinterpret_cast<float4*>(&As[pix * LD + q * 4]) = v;
    }
}

// forward: y[p][0..1] = bias + sum_t sum_c x[p + off_t][c] * w[n][t][c]; y is [P][ldy] (ldy >= 2), channels 2.. zeroed up to 4
template <int CC, typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, int ldx, int B, int H, int W, int C,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ y, int ldy) {
    constexpr int LD = CC + 4;
    __shared__ __attribute__((aligned(16))) float As[HP * HP * LD];
    const int tilesx = (W + HT - 1) / HT, tilesy = (H + HT - 1) / HT;
    int bid = blockIdx.x;
    const int n = bid / (tilesx * tilesy);
    bid -= n * tilesx * tilesy;
    const int ty0 = (bid / tilesx) * HT, tx0 = (bid % tilesx) * HT;
    const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15;
    float acc0 = 0.f, acc1 = 0.f;
    const int ldw = HK * HK * C;
    for (int c0 = 0; c0 < C; c0 += CC) {
        if (c0) __syncthreads();
        load_patch<CC, T>(x, ldx, H, W, n, ty0, tx0, c0, As);
        __syncthreads();
#pragma unroll 1
        for (int kh = 0; kh < HK; ++kh) {
#pragma unroll
            for (int kw = 0; kw < HK; ++kw) {
                const float* ap = &As[((ly + kh) * HP + lx + kw) * LD];
                const float* w0 = w + (kh * HK + kw) * C + c0;       // uniform address -> scalar loads
                const float* w1 = w0 + ldw;
#pragma unroll
                for (int q = 0; q < CC / 4; ++q) {
                    const float4 a = *reinterpret_cast<const float4*>(ap + q * 4);
                    acc0 += a.x * w0[q * 4] + a.y * w0[q * 4 + 1] + a.z * w0[q * 4 + 2] + a.w * w0[q * 4 + 3];
                    acc1 += a.x * w1[q * 4] + a.y * w1[q * 4 + 1] + a.z * w1[q * 4 + 2] + a.w * w1[q * 4 + 3];
                }
            }
        }
    }
    const int oy = ty0 + ly, ox = tx0 + lx;
    if (oy < H && ox < W) {
        const size_t p = ((size_t)n * H + oy) * W + ox;
        const float b0 = bias ? bias[0] : 0.f, b1 = bias ? bias[1] : 0.f;
        if (ldy >= 4) {
            *reinterpret_cast<float4*>(y + p * ldy) = make_float4(acc0 + b0, acc1 + b1, 0.f, 0.f);
        } else {
            y[p * ldy] = acc0 + b0;
            y[p * ldy + 1] = acc1 + b1;
        }
    }
}

// weight gradient: part[blk][n][t][c] = sum over the block's tiles of dy[p][n] * x[p + off_t][c]
// thread = (channel c in the CC-chunk, tap group g); tap t belongs to group t % NG
template <int CC, typename T>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const T* __restrict__ x, int ldx, int B, int H, int W, int C,
                                                         const T* __restrict__ dy, int lddy,
                                                         float* __restrict__ part, int ntiles_total) {
    constexpr int LD = CC + 4, NG = 256 / CC, TPG = (HK * HK + NG - 1) / NG;
    __shared__ __attribute__((aligned(16))) float As[HP * HP * LD];
    __shared__ float2 Ds[HT * HT];
    const int c0 = blockIdx.y * CC;
    const int cl = threadIdx.x % CC, g = threadIdx.x / CC;
    const int tilesx = (W + HT - 1) / HT, tilesy = (H + HT - 1) / HT;
    float acc[TPG][2];
    int base[TPG];
#pragma unroll
    for (int i = 0; i < TPG; ++i) {
        acc[i][0] = acc[i][1] = 0.f;
        const int t = g + i * NG;                    // taps beyond 35 read in-bounds garbage and are never stored
        base[i] = (t < HK * HK) ? ((t / HK) * HP + (t % HK)) * LD + cl : cl;
    }
    for (int tile = blockIdx.x; tile < ntiles_total; tile += gridDim.x) {
        const int n = tile / (tilesx * tilesy);
        const int r = tile - n * tilesx * tilesy;
        const int ty0 = (r / tilesx) * HT, tx0 = (r % tilesx) * HT;
        __syncthreads();
        load_patch<CC, T>(x, ldx, H, W, n, ty0, tx0, c0, As);
        {
            const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15;
            const int oy = ty0 + ly, ox = tx0 + lx;
            float2 d = make_float2(0.f, 0.f);
            if (oy < H && ox < W) d = ld2h(dy + (((size_t)n * H + oy) * W + ox) * lddy);
            Ds[threadIdx.x] = d;
        }
        __syncthreads();
        for (int ly = 0; ly < HT; ++ly) {
#pragma unroll 4
            for (int lx = 0; lx < HT; ++lx) {
                const float2 d = Ds[ly * HT + lx];                   // same address in every lane: LDS broadcast
                const int off = (ly * HP + lx) * LD;
#pragma unroll
                for (int i = 0; i < TPG; ++i) {
                    const float a = As[base[i] + off];
                    acc[i][0] += d.x * a; acc[i][1] += d.y * a;
                }
            }
        }
    }
    float* out = part + (size_t)blockIdx.x * 2 * HK * HK * C;
#pragma unroll
    for (int i = 0; i < TPG; ++i) {
        const int t = g + i * NG;
        if (t < HK * HK) {
            out[(size_t)(0 * HK * HK + t) * C + c0 + cl] = acc[i][0];
            out[(size_t)(1 * HK * HK + t) * C + c0 + cl] = acc[i][1];
        }
    }
}

#define HEAD_WGRAD_BLOCKS 512

namespace {

template <typename T>
int head_fwd_impl(const T* x, int ldx, int B, int H, int W, int C, const float* w, const float* bias, float* y, int ldy, hipStream_t s) {
    if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) || ldx < C || (ldx & 3) || ldy < 2 || (ldy >= 4 && (ldy & 3)))
        return UNETRIR_EINVAL;
    if constexpr (sizeof(T) == 2) {      // bf16: matrix-core path (head_mfma.hip; images wider than 256 pixels in column blocks)
        const bool mfma = unetrir_cfg().head_mfma != 0;
        if (mfma && head_mfma_applies(W, C) && (ldx & 7) == 0) return launch_head_fwd_mfma(x, ldx, B, H, W, C, w, bias, y, ldy, s);
    }
    const long long tiles = (long long)B * ((H + HT - 1) / HT) * ((W + HT - 1) / HT);
    if (C % 32 == 0) hipLaunchKernelGGL((head_fwd_kernel<32, T>), dim3((unsigned)tiles), dim3(256), 0, s, x, ldx, B, H, W, C, w, bias, y, ldy);
    else if (C % 16 == 0) hipLaunchKernelGGL((head_fwd_kernel<16, T>), dim3((unsigned)tiles), dim3(256), 0, s, x, ldx, B, H, W, C, w, bias, y, ldy);
    else hipLaunchKernelGGL((head_fwd_kernel<8, T>), dim3((unsigned)tiles), dim3(256), 0, s, x, ldx, B, H, W, C, w, bias, y, ldy);
    return (int)hipGetLastError();
}

template <typename T>
int head_wgrad_impl(const T* x, int ldx, int B, int H, int W, int C, const T* dy, int lddy, float* dw, void* ws, size_t ws_bytes,
                    hipStream_t s) {
    if (!x || !dy || !dw || !ws || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) || ldx < C || (ldx & 3) || lddy < 2 || (lddy & 1) ||
        ws_bytes < (size_t)HEAD_WGRAD_BLOCKS * 2 * HK * HK * C * sizeof(float))
        return UNETRIR_EINVAL;
    if constexpr (sizeof(T) == 2) {      // bf16: matrix-core path (head_mfma.hip)
        const bool mfma = unetrir_cfg().head_mfma != 0;
        if (mfma && W <= 4096 && C % 32 == 0 && (ldx & 7) == 0) {
            int nb = 0;
            const int err = launch_head_wgrad_mfma(x, ldx, B, H, W, C, dy, lddy, (float*)ws, HEAD_WGRAD_BLOCKS, &nb, s);
            if (err) return err;
            return launch_splitk_reduce((const float*)ws, nb, (size_t)2 * HK * HK * C, dw, 0.f, nullptr, s);
        }
    }
    const long long tiles = (long long)B * ((H + HT - 1) / HT) * ((W + HT - 1) / HT);
    const int nblk = tiles < HEAD_WGRAD_BLOCKS ? (int)tiles : HEAD_WGRAD_BLOCKS;
    if (C % 32 == 0)
        hipLaunchKernelGGL((head_wgrad_kernel<32, T>), dim3(nblk, C / 32), dim3(256), 0, s, x, ldx, B, H, W, C, dy, lddy, (float*)ws, (int)tiles);
    else if (C % 16 == 0)
        hipLaunchKernelGGL((head_wgrad_kernel<16, T>), dim3(nblk, C / 16), dim3(256), 0, s, x, ldx, B, H, W, C, dy, lddy, (float*)ws, (int)tiles);
    else
        hipLaunchKernelGGL((head_wgrad_kernel<8, T>), dim3(nblk, C / 8), dim3(256), 0, s, x, ldx, B, H, W, C, dy, lddy, (float*)ws, (int)tiles);
    const size_t nout = (size_t)2 * HK * HK * C;
    return launch_splitk_reduce((const float*)ws, nblk, nout, dw, 0.f, nullptr, s);
}

}  // namespace

extern "C" {

int unetrir_head6x6_supported(int C) { return (C % 8) == 0 ? 1 : 0; }

int unetrir_head6x6_fwd_f32(const float* x, int ldx, int B, int H, int W, int C, const float* w, const float* bias, float* y,
                            int ldy, unetrir_stream_t stream) {
    return head_fwd_impl<float>(x, ldx, B, H, W, C, w, bias, y, ldy, (hipStream_t)stream);
}

size_t unetrir_head6x6_wgrad_ws_bytes(int C) { return (size_t)HEAD_WGRAD_BLOCKS * 2 * HK * HK * C * sizeof(float); }

int unetrir_head6x6_wgrad_f32(const float* x, int ldx, int B, int H, int W, int C, const float* dy, int lddy, float* dw,
                              void* ws, size_t ws_bytes, unetrir_stream_t stream) {
    return head_wgrad_impl<float>(x, ldx, B, H, W, C, dy, lddy, dw, ws, ws_bytes, (hipStream_t)stream);
}

/* bf16 activations in, fp32 logits / fp32 weight gradient out */
int unetrir_head6x6_fwd_bf16(const unetrir_bf16* x, int ldx, int B, int H, int W, int C, const float* w, const float* bias, float* y,
                             int ldy, unetrir_stream_t stream) {
    return head_fwd_impl<__bf16>((const __bf16*)x, ldx, B, H, W, C, w, bias, y, ldy, (hipStream_t)stream);
}

int unetrir_head6x6_wgrad_bf16(const unetrir_bf16* x, int ldx, int B, int H, int W, int C, const unetrir_bf16* dy, int lddy,
                               float* dw, void* ws, size_t ws_bytes, unetrir_stream_t stream) {
    return head_wgrad_impl<__bf16>((const __bf16*)x, ldx, B, H, W, C, (const __bf16*)dy, lddy, dw, ws, ws_bytes, (hipStream_t)stream);
}

int unetrir_head6x6_dgrad_supported(int W, int C) { return head_dgrad_mfma_applies(W, C) ? 1 : 0; }

int unetrir_head6x6_dgrad_bf16(const unetrir_bf16* dy, int lddy, int B, int H, int W, const float* w, int C, unetrir_bf16* dx, int lddx,
                               unetrir_stream_t stream) {
    if (!dy || !w || !dx || B <= 0 || H <= 0 || W <= 0 || !head_dgrad_mfma_applies(W, C) || lddy < 2 || (lddy & 1) || lddx < C ||
        (lddx & 7) || ((uintptr_t)dx & 15) || ((uintptr_t)dy & 3))
        return UNETRIR_EINVAL;
    return launch_head_dgrad_mfma(dy, lddy, B, H, W, w, C, dx, lddx, (hipStream_t)stream);
}

}  // extern "C"
