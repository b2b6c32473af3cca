// elementwise.hip - the HBM-bound kernels of the train step (gfx950): BatchNorm statistics /
// apply / backward, bias gradients, sigmoid + amp/phase loss, embedding, Adam, layout helpers.
// Every reduction is two-stage with a fixed summation order (no float atomics): results are
// bit-reproducible run to run.  Per-channel sums accumulate in fp64 (the VALU has the headroom:
// these kernels are bound by HBM bytes, not by flops).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include "kernels.h"

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
// 4 consecutive channels as float4, for fp32 (16-byte access) and bf16 (8-byte access) tensors
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const __bf16* p) {
    const bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(__bf16* p, float4 v) {
    bf16x4_t o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    *reinterpret_cast<bf16x4_t*>(p) = o;
}

// One 16-byte access per lane: 4 fp32 or 8 bf16 consecutive channels (8-byte accesses reach only ~0.6x the HBM rate).
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <typename T> struct VecOf;
template <> struct VecOf<float> { static constexpr int N = 4; };
template <> struct VecOf<__bf16> { static constexpr int N = 8; };
__device__ __forceinline__ void ldv(const float* p, float (&o)[4]) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void ldv(const __bf16* p, float (&o)[8]) {
    const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (float)v[k];
}
__device__ __forceinline__ void stv(float* p, const float (&o)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
}
__device__ __forceinline__ void stv(__bf16* p, const float (&o)[8]) {
    bf16x8_t v;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (__bf16)o[k];
    *reinterpret_cast<bf16x8_t*>(p) = v;
}

// Loads of the BatchNorm passes.  NT: non-temporal - for tensors larger than the Infinity Cache (the full-resolution levels of configs[1]:
// 268 MB each), which a pass reads once: the lines it would displace are the ones the NEXT kernel reads (round 4: - 0.2 ms per step at
// configs[1]; on the <= 134 MB tensors of configs[4] the same policy costs 0.1 ms, so the launch chooses by size - bn_nt()).
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ void ldv_s(const float* p, float (&o)[4]) {
    const f32x4_t* q = reinterpret_cast<const f32x4_t*>(p);
    const f32x4_t v = NT ? __builtin_nontemporal_load(q) : *q;
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
}
template <bool NT> __device__ __forceinline__ void ldv_s(const __bf16* p, float (&o)[8]) {
    const bf16x8_t* q = reinterpret_cast<const bf16x8_t*>(p);
    const bf16x8_t v = NT ? __builtin_nontemporal_load(q) : *q;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (float)v[k];
}

namespace {

struct ChanPlan { int QB, RB, ngroups, nslab; long long rows_per_slab; };

// Thread layout for [P][C] per-channel reductions: QB channel-quads x RB pixel rows per block.
inline ChanPlan chan_plan(long long P, int C, int vec = 4) {
    ChanPlan p;
    const int CQ = C / vec;
    int qb = 1;
    while (qb * 2 <= CQ && qb * 2 <= 256) qb *= 2;
    p.QB = qb; p.RB = 256 / qb;
    p.ngroups = (CQ + qb - 1) / qb;
    long long want = 2048 / p.ngroups;                    // ~2048 blocks in flight (round 3, small tensors: 256 ... 2048 measured equal)
    if (want < 1) want = 1;
    long long min_rows = (long long)p.RB * 8;             // at least 8 rows per thread
    long long maxslab = (P + min_rows - 1) / min_rows;
    if (maxslab < 1) maxslab = 1;
    if (want > maxslab) want = maxslab;
    p.rows_per_slab = (P + want - 1) / want;
    p.nslab = (int)((P + p.rows_per_slab - 1) / p.rows_per_slab);
    p.rows_per_slab = 0;          // the slab workgroups sweep the tensor together (interleaved rows), not contiguous slabs
    return p;
}

}  // namespace

// -------------------------------------------------------------------------------------------
// stage 1: per-slab per-channel partial sums.  MODE 0: (sum x, sum x^2)   [BN statistics]
//                                               MODE 1: (sum x, -)         [bias gradient]
//                                               MODE 2: (sum g, sum g*xhat) with g = da*[relu mask]
// partial layout: double part[nslab][C][2]
// -------------------------------------------------------------------------------------------
template <int MODE, typename T, bool NT = false>
__global__ __launch_bounds__(256) void chan_partial_kernel(const T* __restrict__ x, int ldx,
                                                           const T* __restrict__ da, int ldda,
                                                           const float* __restrict__ affine,
                                                           const float* __restrict__ saved, int relu,
                                                           long long P, int C, int QB, long long rows_per_slab,
                                                           double* __restrict__ part,
                                                           const T* __restrict__ msk = nullptr, int ldm = 0) {
    // msk (MODE 2, nullable): the activation's OUTPUT where it is not act(x*scale + shift) alone - the Add -> LeakyReLU junction
    // of the residual blocks (dl_models/res_ae.py:334-336): ReLU / LeakyReLU keep the sign, so out > 0 decides the branch.
    constexpr int V = VecOf<T>::N;
    __shared__ double red[256 * 4];
    const int tid = threadIdx.x;
    const int RB = 256 / QB;
    const int ql = tid % QB, rl = tid / QB;
    const int q = blockIdx.y * QB + ql;
    const int c0 = q * V;
    const bool cok = c0 < C;
    // rows_per_slab > 0: block b owns the contiguous rows [b*rows_per_slab, ...).  rows_per_slab <= 0: the blocks sweep the
    // tensor together (block b takes row groups b, b+nslab, ...), so the whole grid reads one moving window of HBM instead
    // of ~2048 separate streams.  Either way a slab is a fixed set of rows summed in a fixed order.
    long long p0 = (long long)blockIdx.x * rows_per_slab, p1 = p0 + rows_per_slab, pstep = RB;
    if (rows_per_slab <= 0) { p0 = (long long)blockIdx.x * RB; p1 = P; pstep = (long long)gridDim.x * RB; }
    if (p1 > P) p1 = P;

    double s0[V], s1[V];
    float sc[V], sh[V], mu[V], rs[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { s0[k] = 0; s1[k] = 0; sc[k] = 1; sh[k] = 0; mu[k] = 0; rs[k] = 1; }
    if (MODE == 2 && cok) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            sc[k] = affine[c0 + k]; sh[k] = affine[C + c0 + k];
            mu[k] = saved[c0 + k]; rs[k] = saved[C + c0 + k];
        }
    }
    if (cok) {
        for (long long p = p0 + rl; p < p1; p += pstep) {
            float xv[V];
            ldv_s<NT>(x + (size_t)p * ldx + c0, xv);
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < V; ++k) { const double d = xv[k]; s0[k] += d; s1[k] += d * d; }
            } else if (MODE == 1) {
#pragma unroll
                for (int k = 0; k < V; ++k) s0[k] += (double)xv[k];
            } else {
                float gv[V], mv[V];
                ldv_s<NT>(da + (size_t)p * ldda + c0, gv);
                if (msk) ldv_s<NT>(msk + (size_t)p * ldm + c0, mv);
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const float a = msk ? mv[k] : xv[k] * sc[k] + sh[k];
                    const float g = (relu && !(a > 0.f)) ? (relu == 2 ? 0.3f * gv[k] : 0.f) : gv[k];
                    const float xh = (xv[k] - mu[k]) * rs[k];
                    s0[k] += (double)g; s1[k] += (double)g * (double)xh;
                }
            }
        }
    }
    // cross-row reduction in the block, four values at a time: 8 KB of LDS instead of 32 KB keeps eight blocks on a CU (the
    // reduction is bandwidth bound: occupancy is what it runs on).  Same fixed summation order as one big exchange.
#pragma unroll
    for (int ch = 0; ch < 2 * V / 4; ++ch) {
        if (ch) __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = ch * 4 + j;
            red[tid * 4 + j] = idx < V ? s0[idx] : s1[idx - V];
        }
        __syncthreads();
        if (rl == 0 && cok) {
            for (int r = 1; r < RB; ++r) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int idx = ch * 4 + j;
                    const double v = red[(r * QB + ql) * 4 + j];
                    if (idx < V) s0[idx] += v; else s1[idx - V] += v;
                }
            }
        }
    }
    if (rl == 0 && cok) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            part[((size_t)blockIdx.x * C + c0 + k) * 2 + 0] = s0[k];
            part[((size_t)blockIdx.x * C + c0 + k) * 2 + 1] = s1[k];
        }
    }
}

// Fixed-order sum of the per-slab partials of FIN_CH consecutive channels by one block of FIN_T threads: thread (lane, ch) adds
// slabs lane, lane + FIN_T/FIN_CH, ... of channel c0 + ch (the FIN_CH threads of a lane read one contiguous 64- or 128-byte run of
// a slab row: with one block per channel every 16-byte pair came from a line of its own, 10 us per launch at 2048 slabs), then a
// tree over the lanes in a fixed pairing order.  Every thread returns the sums of ITS channel (c0 + (threadIdx.x % FIN_CH)).
constexpr int FIN_CH = 8, FIN_T = 1024, FIN_L = FIN_T / FIN_CH;
template <typename PT> struct Pair2;
template <> struct Pair2<double> { using type = double2; };
template <> struct Pair2<float> { using type = float2; };
template <typename PT>
__device__ __forceinline__ void slab_sum(const PT* __restrict__ part, int nslab, int C, int c0, double& s, double& ss, int cend = 0x7fffffff) {
    __shared__ double red[FIN_T * 2];
    using P2 = typename Pair2<PT>::type;
    const int ch = threadIdx.x % FIN_CH, ln = threadIdx.x / FIN_CH, c = c0 + ch;
    double a = 0, b = 0;
    if (c < C && c < cend) {          // C: channels per slab row (the row stride); cend: end of the range asked for
#pragma unroll 8
        for (int k = ln; k < nslab; k += FIN_L) {
            const P2 v = *reinterpret_cast<const P2*>(part + ((size_t)k * C + c) * 2);
            a += (double)v.x; b += (double)v.y;
        }
    }
    red[threadIdx.x * 2] = a; red[threadIdx.x * 2 + 1] = b;
    __syncthreads();
    for (int st = FIN_L / 2; st > 0; st >>= 1) {
        if (ln < st) {
            red[threadIdx.x * 2] += red[(threadIdx.x + st * FIN_CH) * 2];
            red[threadIdx.x * 2 + 1] += red[(threadIdx.x + st * FIN_CH) * 2 + 1];
        }
        __syncthreads();
    }
    s = red[ch * 2]; ss = red[ch * 2 + 1];
}
inline dim3 fin_grid(int C) { return dim3((C + FIN_CH - 1) / FIN_CH); }

// stage 2 (BN statistics): mean / biased variance -> affine, saved, moving statistics
template <typename PT>
__global__ __launch_bounds__(FIN_T) void bn_finalize_kernel(const PT* __restrict__ part, int nslab, long long P, int C,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float momentum, float* __restrict__ moving_mean, float* __restrict__ moving_var,
                                   float* __restrict__ affine, float* __restrict__ saved) {
    const int c = blockIdx.x * FIN_CH + threadIdx.x;
    double s, ss;
    slab_sum<PT>(part, nslab, C, blockIdx.x * FIN_CH, s, ss);
    if (threadIdx.x >= FIN_CH || c >= C) return;
    const double mean = s / (double)P;
    double var = ss / (double)P - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float scale = g * rstd;
    affine[c] = scale;
    affine[C + c] = b - (float)mean * scale;
    saved[c] = (float)mean;
    saved[C + c] = rstd;
    if (moving_mean) moving_mean[c] = momentum * moving_mean[c] + (1.f - momentum) * (float)mean;
    if (moving_var) {
        const double unb = P > 1 ? var * (double)P / (double)(P - 1) : var;
        moving_var[c] = momentum * moving_var[c] + (1.f - momentum) * (float)unb;
    }
}

// stage 2 (column sum): out[c] = sum_slabs of channel c0 + c, c < n
template <typename PT>
__global__ __launch_bounds__(FIN_T) void colsum_finalize_kernel(const PT* __restrict__ part, int nslab, int C, float* __restrict__ out, int c0,
                                                                int n) {
    const int c = blockIdx.x * FIN_CH + threadIdx.x;
    double s, ss;
    slab_sum<PT>(part, nslab, C, c0 + blockIdx.x * FIN_CH, s, ss, c0 + n);
    if (threadIdx.x < FIN_CH && c < n) out[c] = (float)s;
}

// stage 2 (BN backward): dgamma, dbeta and the two means the dx pass needs -> coef[2*C]
__global__ __launch_bounds__(FIN_T) void bn_bwd_finalize_kernel(const double* __restrict__ part, int nslab, long long P, int C,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ coef) {
    const int c = blockIdx.x * FIN_CH + threadIdx.x;
    double s, ss;
    slab_sum(part, nslab, C, blockIdx.x * FIN_CH, s, ss);
    if (threadIdx.x >= FIN_CH || c >= C) return;
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)ss;
    coef[c] = (float)(s / (double)P);
    coef[C + c] = (float)(ss / (double)P);
}

// y = x*scale + shift (ReLU optional), one 16-byte vector per lane over [P][C].  The grid stride is a multiple of the
// channel-vector count whenever C/V divides the 256-thread block (every layer here), so a thread keeps ONE channel group
// and its per-channel parameters live in registers for the whole loop instead of being re-fetched per element.
template <typename T, bool NT = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, int ldx, long long P, int C,
                                                       const float* __restrict__ affine, int relu,
                                                       T* __restrict__ y, int ldy,
                                                       const T* __restrict__ addend = nullptr, int ldadd = 0) {
    constexpr int V = VecOf<T>::N;
    const int CQ = C / V;
    const long long total = P * CQ;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nohoist = relu & 256; relu &= 255;     // bit 8: A/B switch (UNETRIR_BN_HOIST=0), forces the re-fetching loop
    const float sl = relu == 2 ? 0.3f : 0.f;
    if (stride % CQ == 0 && !nohoist && addend != nullptr) {   // measured: hoisting pays for the 3-input forms only
        const int c0 = (int)(i0 % CQ) * V;
        float sc[V], sh[V];
#pragma unroll
        for (int k = 0; k < V; ++k) { sc[k] = affine ? affine[c0 + k] : 1.f; sh[k] = affine ? affine[C + c0 + k] : 0.f; }
        const long long pstep = stride / CQ;
        for (long long p = i0 / CQ; p < P; p += pstep) {
            float r[V];
            ldv_s<NT>(x + (size_t)p * ldx + c0, r);
#pragma unroll
            for (int k = 0; k < V; ++k) r[k] = r[k] * sc[k] + sh[k];
            if (addend) {     // Add() in front of the activation (residual blocks, dl_models/res_ae.py:334, :478)
                float ad[V];
                ldv_s<NT>(addend + (size_t)p * ldadd + c0, ad);
#pragma unroll
                for (int k = 0; k < V; ++k) r[k] += ad[k];
            }
            if (relu) {       // 1: ReLU; 2: keras LeakyReLU() (alpha = 0.3)
#pragma unroll
                for (int k = 0; k < V; ++k) r[k] = r[k] > 0.f ? r[k] : sl * r[k];
            }
            stv(y + (size_t)p * ldy + c0, r);
        }
        return;
    }
    for (long long i = i0; i < total; i += stride) {
        const long long p = i / CQ;
        const int c0 = (int)(i - p * CQ) * V;
        float r[V];
        ldv_s<NT>(x + (size_t)p * ldx + c0, r);
        if (affine) {
#pragma unroll
            for (int k = 0; k < V; ++k) r[k] = r[k] * affine[c0 + k] + affine[C + c0 + k];
        }
        if (addend) {
            float ad[V];
            ldv_s<NT>(addend + (size_t)p * ldadd + c0, ad);
#pragma unroll
            for (int k = 0; k < V; ++k) r[k] += ad[k];
        }
        if (relu) {
#pragma unroll
            for (int k = 0; k < V; ++k) r[k] = r[k] > 0.f ? r[k] : sl * r[k];
        }
        stv(y + (size_t)p * ldy + c0, r);
    }
}

// dx = scale * (g - mean(g) - xhat * mean(g*xhat)),  g = da * relu-mask;  without BN (affine == NULL): dx = g
template <typename T, bool NT = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ da, int ldda,
                                                           const T* __restrict__ x, int ldx, long long P, int C,
                                                           const float* __restrict__ affine,
                                                           const float* __restrict__ saved,
                                                           const float* __restrict__ coef, int relu,
                                                           T* __restrict__ dx, int lddx,
                                                           const T* __restrict__ msk = nullptr, int ldm = 0,
                                                           T* __restrict__ g2 = nullptr, int ldg2 = 0,
                                                           const T* __restrict__ g2add = nullptr, int ldg2a = 0) {
    constexpr int V = VecOf<T>::N;
    const int CQ = C / V;
    const long long total = P * CQ;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nohoist = relu & 256; relu &= 255;
    if (msk) {
        // the Add -> activation junction (dl_models/res_ae.py:334-336, :478-480): g = da * act'(out) feeds BOTH branches - the
        // BatchNorm backward of this one (dx) and the other operand of the Add (g2 = g (+ g2add): its gradient, written or
        // accumulated in place here instead of by two more passes)
        const int c0 = (int)(i0 % CQ) * V;
        const bool hoist = stride % CQ == 0;
        float sc[V], mu[V], rs[V], c1[V], c2[V];
#pragma unroll
        for (int k = 0; k < V; ++k) { sc[k] = affine[c0 + k]; mu[k] = saved[c0 + k]; rs[k] = saved[C + c0 + k]; c1[k] = coef[c0 + k]; c2[k] = coef[C + c0 + k]; }
        for (long long i = i0; i < total; i += stride) {
            const long long p = i / CQ;
            const int cc = hoist ? c0 : (int)(i - p * CQ) * V;
            if (!hoist) {
#pragma unroll
                for (int k = 0; k < V; ++k) { sc[k] = affine[cc + k]; mu[k] = saved[cc + k]; rs[k] = saved[C + cc + k]; c1[k] = coef[cc + k]; c2[k] = coef[C + cc + k]; }
            }
            float xv[V], gv[V], mv[V], out[V], go[V];
            ldv_s<NT>(x + (size_t)p * ldx + cc, xv);
            ldv_s<NT>(da + (size_t)p * ldda + cc, gv);
            ldv_s<NT>(msk + (size_t)p * ldm + cc, mv);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float g = (relu && !(mv[k] > 0.f)) ? (relu == 2 ? 0.3f * gv[k] : 0.f) : gv[k];
                const float xh = (xv[k] - mu[k]) * rs[k];
                out[k] = sc[k] * (g - c1[k] - xh * c2[k]);
                go[k] = g;
            }
            stv(dx + (size_t)p * lddx + cc, out);
            if (g2) {
                if (g2add) {
                    float ad[V];
                    ldv_s<NT>(g2add + (size_t)p * ldg2a + cc, ad);
#pragma unroll
                    for (int k = 0; k < V; ++k) go[k] += ad[k];
                }
                stv(g2 + (size_t)p * ldg2 + cc, go);
            }
        }
        return;
    }
    if (affine && stride % CQ == 0 && !nohoist) {
        const int c0 = (int)(i0 % CQ) * V;
        float sc[V], sh[V], mu[V], rs[V], c1[V], c2[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            sc[k] = affine[c0 + k]; sh[k] = affine[C + c0 + k];
            mu[k] = saved[c0 + k]; rs[k] = saved[C + c0 + k]; c1[k] = coef[c0 + k]; c2[k] = coef[C + c0 + k];
        }
        const long long pstep = stride / CQ;
        for (long long p = i0 / CQ; p < P; p += pstep) {
            float xv[V], gv[V], out[V];
            ldv_s<NT>(x + (size_t)p * ldx + c0, xv);
            ldv_s<NT>(da + (size_t)p * ldda + c0, gv);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float a = xv[k] * sc[k] + sh[k];
                const float g = (relu && !(a > 0.f)) ? (relu == 2 ? 0.3f * gv[k] : 0.f) : gv[k];
                const float xh = (xv[k] - mu[k]) * rs[k];
                out[k] = sc[k] * (g - c1[k] - xh * c2[k]);
            }
            stv(dx + (size_t)p * lddx + c0, out);
        }
        return;
    }
    for (long long i = i0; i < total; i += stride) {
        const long long p = i / CQ;
        const int c0 = (int)(i - p * CQ) * V;
        float xv[V], gv[V], out[V];
        ldv_s<NT>(x + (size_t)p * ldx + c0, xv);
        ldv_s<NT>(da + (size_t)p * ldda + c0, gv);
        if (affine) {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float sc = affine[c0 + k], sh = affine[C + c0 + k];
                const float a = xv[k] * sc + sh;
                const float g = (relu && !(a > 0.f)) ? (relu == 2 ? 0.3f * gv[k] : 0.f) : gv[k];
                const float xh = (xv[k] - saved[c0 + k]) * saved[C + c0 + k];
                out[k] = sc * (g - coef[c0 + k] - xh * coef[C + c0 + k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) out[k] = (relu && !(xv[k] > 0.f)) ? (relu == 2 ? 0.3f * gv[k] : 0.f) : gv[k];
        }
        stv(dx + (size_t)p * lddx + c0, out);
    }
}

// -------------------------------------------------------------------------------------------
// boundary: NCHW [B,C,H,W] -> NHWC [B,H,W,Cpad] (zero fill), one thread per pixel
// -------------------------------------------------------------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_pad_kernel(const float* __restrict__ x, int B, int C, int H, int W,
                                        T* __restrict__ y, int Cpad) {
    const long long hw = (long long)H * W, total = (long long)B * hw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / hw, r = i - b * hw;
        for (int c = 0; c < Cpad; ++c) y[i * Cpad + c] = (T)((c < C) ? x[(b * C + c) * hw + r] : 0.f);
    }
}

// -------------------------------------------------------------------------------------------
// sigmoid + amplitude/phase loss (main_training.py:184-235) + dL/dlogits, one thread per pixel
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }

// phase_ref (nullable): the network INPUT, NCHW - the `diff_loss` switch of main_training.py:214-217 (phase target = phase_true -
// phase_x).  phase_w (nullable): [W] column weights of the phase term - the `sigmoid_loss` switch (:221-222, preprocess.py:116-121).
// part: three doubles per block (amplitude sum, phase sum as the metrics see it, phase sum as the loss sees it = weighted).
template <typename T>
__global__ __launch_bounds__(256) void sigmoid_loss_kernel(const float* __restrict__ logits, int ldl,
                                                           const float* __restrict__ target, int B, int H, int W,
                                                           float alpha, float inv_norm, float* __restrict__ pred,
                                                           T* __restrict__ dlogits, int ldd, double* __restrict__ part,
                                                           const float* __restrict__ phase_ref, const float* __restrict__ phase_w) {
    __shared__ double red[256 * 3];
    const long long hw = (long long)H * W, total = (long long)B * hw;
    const float TWO_PI = 6.283185307179586f, PI = 3.141592653589793f;
    double sa = 0, sp = 0, spw = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / hw, r = i - b * hw;
        const float p0 = sigmoidf_(logits[i * ldl + 0]), p1 = sigmoidf_(logits[i * ldl + 1]);
        pred[(b * 2 + 0) * hw + r] = p0;
        pred[(b * 2 + 1) * hw + r] = p1;
        if (target) {
            const float t0 = target[(b * 2 + 0) * hw + r];
            float t1 = target[(b * 2 + 1) * hw + r];
            if (phase_ref) t1 -= phase_ref[(b * 2 + 1) * hw + r];
            const float wgt = phase_w ? phase_w[(int)(r % W)] : 1.f;
            const float da = t0 - p0;
            const float yt = t1 * TWO_PI - PI, yp = p1 * TWO_PI - PI;
            const float d = (yt - yp) + PI;
            const float ph = (d - floorf(d / TWO_PI) * TWO_PI) - PI;     // python-style modulo
            const float eph = 1.f - cosf(ph);
            sa += (double)(da * da);
            sp += (double)eph;
            spw += (double)(eph * wgt);
            const float g0 = -2.f * da * alpha * inv_norm;
            const float g1 = -(1.f - alpha) * inv_norm * TWO_PI * sinf(ph) * wgt;
            st4(dlogits + i * ldd, make_float4(g0 * p0 * (1.f - p0), g1 * p1 * (1.f - p1), 0.f, 0.f));
            if (ldd > 4) st4(dlogits + i * ldd + 4, make_float4(0.f, 0.f, 0.f, 0.f));
        }
    }
    red[threadIdx.x * 3] = sa; red[threadIdx.x * 3 + 1] = sp; red[threadIdx.x * 3 + 2] = spw;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 256; ++k) { sa += red[k * 3]; sp += red[k * 3 + 1]; spw += red[k * 3 + 2]; }
        part[blockIdx.x * 3] = sa; part[blockIdx.x * 3 + 1] = sp; part[blockIdx.x * 3 + 2] = spw;
    }
}

__global__ void loss_finalize_kernel(const double* __restrict__ part, int nblk, float alpha, float inv_norm,
                                     float* __restrict__ out) {
    __shared__ double red[256 * 3];
    double a = 0, b = 0, c = 0;
    for (int k = threadIdx.x; k < nblk; k += 256) { a += part[k * 3]; b += part[k * 3 + 1]; c += part[k * 3 + 2]; }
    red[threadIdx.x * 3] = a; red[threadIdx.x * 3 + 1] = b; red[threadIdx.x * 3 + 2] = c;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {          // fixed pairing order
        if (threadIdx.x < st) {
#pragma unroll
            for (int j = 0; j < 3; ++j) red[threadIdx.x * 3 + j] += red[(threadIdx.x + st) * 3 + j];
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const double sa = red[0], sp = red[1], spw = red[2];
    out[0] = (float)(((double)alpha * sa + (1.0 - (double)alpha) * spw) * (double)inv_norm);
    out[1] = (float)sa;
    out[2] = (float)sp;
}

__global__ void sigmoid_only_kernel(const float* __restrict__ logits, int ldl, int B, int H, int W,
                                    float* __restrict__ pred) {
    const long long hw = (long long)H * W, total = (long long)B * hw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / hw, r = i - b * hw;
        pred[(b * 2 + 0) * hw + r] = sigmoidf_(logits[i * ldl + 0]);
        pred[(b * 2 + 1) * hw + r] = sigmoidf_(logits[i * ldl + 1]);
    }
}

template <typename T>
__global__ void sigmoid_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ dpred, int B, int H,
                                   int W, T* __restrict__ dlogits, int ldd) {
    const long long hw = (long long)H * W, total = (long long)B * hw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / hw, r = i - b * hw;
        const float p0 = pred[(b * 2 + 0) * hw + r], p1 = pred[(b * 2 + 1) * hw + r];
        const float g0 = dpred[(b * 2 + 0) * hw + r], g1 = dpred[(b * 2 + 1) * hw + r];
        st4(dlogits + i * ldd, make_float4(g0 * p0 * (1.f - p0), g1 * p1 * (1.f - p1), 0.f, 0.f));
        if (ldd > 4) st4(dlogits + i * ldd + 4, make_float4(0.f, 0.f, 0.f, 0.f));
    }
}

// -------------------------------------------------------------------------------------------
// Embedding gather and its deterministic gradient
// -------------------------------------------------------------------------------------------
__global__ void embedding_fwd_kernel(const int* __restrict__ idx, int n_idx, const float* __restrict__ table,
                                     int vocab, int dim, float* __restrict__ out) {
    const int i = blockIdx.x;
    int v = idx[i];
    v = v < 0 ? 0 : (v >= vocab ? vocab - 1 : v);
    for (int d = threadIdx.x; d < dim; d += blockDim.x) out[(size_t)i * dim + d] = table[(size_t)v * dim + d];
}

// dtable[v] = sum over the positions i (in increasing i: a fixed order) with idx[i] == v.  One block per vocabulary row:
// the block first collects the (few) matching positions into LDS, then sums only those rows of dout.
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int* __restrict__ idx, int n_idx, const float* __restrict__ dout,
                                                            int vocab, int dim, float* __restrict__ dtable) {
    __shared__ int list[1024];
    __shared__ int cnt;
    const int v = blockIdx.x;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    bool overflow = false;
    for (int i = threadIdx.x; i < n_idx; i += blockDim.x) {
        int u = idx[i];
        u = u < 0 ? 0 : (u >= vocab ? vocab - 1 : u);
        if (u == v) {
            const int k = atomicAdd(&cnt, 1);
            if (k < 1024) list[k] = i; else overflow = true;
        }
    }
    __syncthreads();
    const int n = cnt;
    if (n > 1024 || __syncthreads_or(overflow)) {          // pathological (one id more than 1024 times): plain ordered scan
        for (int d = threadIdx.x; d < dim; d += blockDim.x) {
            float s = 0.f;
            for (int i = 0; i < n_idx; ++i) {
                int u = idx[i];
                u = u < 0 ? 0 : (u >= vocab ? vocab - 1 : u);
                if (u == v) s += dout[(size_t)i * dim + d];
            }
            dtable[(size_t)v * dim + d] = s;
        }
        return;
    }
    if (threadIdx.x == 0 && n > 1) {                         // the list arrives in atomic order: sort it (n is tiny)
        for (int a = 1; a < n; ++a) {
            const int key = list[a];
            int b = a - 1;
            while (b >= 0 && list[b] > key) { list[b + 1] = list[b]; --b; }
            list[b + 1] = key;
        }
    }
    __syncthreads();
    for (int d = threadIdx.x; d < dim; d += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < n; ++k) s += dout[(size_t)list[k] * dim + d];
        dtable[(size_t)v * dim + d] = s;
    }
}

__global__ void mul_kernel(const float* __restrict__ x, const float* __restrict__ m, float* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = x[i] * m[i];
}

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long long n, double* __restrict__ part) {
    __shared__ double red[256];
    double s = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double v = x[i];
        s += v * v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 256; ++k) s += red[k];
        part[blockIdx.x] = s;
    }
}

__global__ void sumsq_finalize_kernel(const double* __restrict__ part, int nblk, float coef, float* __restrict__ out, int accumulate) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0;
    for (int k = 0; k < nblk; ++k) s += part[k];
    const float r = (float)((double)coef * s);
    out[0] = accumulate ? out[0] + r : r;
}

// Keras Adam: m,v update; theta -= lr_t * m / (sqrt(v) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ theta, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n,
                                                   float lr_t, float b1, float b2, float eps, float gs,
                                                   const float* __restrict__ hyper) {
    if (hyper) { lr_t = hyper[0]; b1 = hyper[1]; b2 = hyper[2]; eps = hyper[3]; gs = hyper[4]; }      // step-dependent values from device memory (HIP graphs)
    const long long n4 = n / 4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    // two independent 16-byte groups per thread and iteration: 8 loads in flight before the first store
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += 2 * stride) {
        const long long i2 = i + stride;
        const bool two = i2 < n4;
        float4 t[2], gg[2], mm[2], vv[2];
        t[0] = reinterpret_cast<float4*>(theta)[i]; gg[0] = reinterpret_cast<const float4*>(g)[i];
        mm[0] = reinterpret_cast<float4*>(m)[i]; vv[0] = reinterpret_cast<float4*>(v)[i];
        if (two) {
            t[1] = reinterpret_cast<float4*>(theta)[i2]; gg[1] = reinterpret_cast<const float4*>(g)[i2];
            mm[1] = reinterpret_cast<float4*>(m)[i2]; vv[1] = reinterpret_cast<float4*>(v)[i2];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            float* tp = &t[u].x; const float* gp = &gg[u].x; float* mp = &mm[u].x; float* vp = &vv[u].x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gk = gp[k] * gs;
                mp[k] = b1 * mp[k] + (1.f - b1) * gk;
                vp[k] = b2 * vp[k] + (1.f - b2) * gk * gk;
                tp[k] -= lr_t * mp[k] / (sqrtf(vp[k]) + eps);
            }
            const long long o = u ? i2 : i;
            reinterpret_cast<float4*>(theta)[o] = t[u];
            reinterpret_cast<float4*>(m)[o] = mm[u];
            reinterpret_cast<float4*>(v)[o] = vv[u];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (long long i = n4 * 4; i < n; ++i) {
            const float gk = g[i] * gs;
            m[i] = b1 * m[i] + (1.f - b1) * gk;
            v[i] = b2 * v[i] + (1.f - b2) * gk * gk;
            theta[i] -= lr_t * m[i] / (sqrtf(v[i]) + eps);
        }
    }
}

// The other two optimizers main_training.py:164-169 can select.  SGD: theta -= lr * g.  Nadam (tf.keras optimizer_v2): the moments
// of Adam, the step lr * (c_g g + c_m m) / (sqrt(c_v v) + eps) with the step-dependent coefficients computed on the host
// (momentum schedule mu_t = beta_1 (1 - 0.5 * 0.96^(0.004 t)): c_g = (1 - mu_t) / (1 - prod mu), c_m = mu_{t+1} / (1 - prod mu * mu_{t+1}),
// c_v = 1 / (1 - beta_2^t)).
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ theta, const float* __restrict__ g, long long n, float lr, float gs) {
    const long long n4 = n / 4, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 t = reinterpret_cast<float4*>(theta)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        t.x -= lr * gs * gg.x; t.y -= lr * gs * gg.y; t.z -= lr * gs * gg.z; t.w -= lr * gs * gg.w;
        reinterpret_cast<float4*>(theta)[i] = t;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long long i = n4 * 4; i < n; ++i) theta[i] -= lr * gs * g[i];
}

__global__ __launch_bounds__(256) void nadam_kernel(float* __restrict__ theta, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                    float cg, float cm, float cv, float gs) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gk = g[i] * gs;
        const float mk = b1 * m[i] + (1.f - b1) * gk;
        const float vk = b2 * v[i] + (1.f - b2) * gk * gk;
        m[i] = mk; v[i] = vk;
        theta[i] -= lr * (cg * gk + cm * mk) / (sqrtf(cv * vk) + eps);
    }
}

// -------------------------------------------------------------------------------------------
// C ABI
// -------------------------------------------------------------------------------------------
static inline unsigned grid_for(long long n, int per_block = 256, int cap = 4096) {
    long long b = (n + per_block - 1) / per_block;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}
static inline unsigned chan_pad_lds() { return 0u; }
static inline int bn_nohoist_flag() { return 0; }
#ifndef UNETRIR_BN_NT_MIN_MB
#define UNETRIR_BN_NT_MIN_MB 100
#endif
// non-temporal loads for a BatchNorm pass over a tensor of this many bytes (see ldv_s)
static inline bool bn_nt(long long P, int C, size_t elem) { return P * C * (long long)elem >= ((long long)UNETRIR_BN_NT_MIN_MB << 20); }
// launches with NT chosen at run time
template <int MODE, typename T, typename... A> static inline void launch_chan_partial(bool nt, dim3 grid, hipStream_t s, A... a) {
    if (nt) hipLaunchKernelGGL((chan_partial_kernel<MODE, T, true>), grid, dim3(256), 0, s, a...);
    else hipLaunchKernelGGL((chan_partial_kernel<MODE, T, false>), grid, dim3(256), 0, s, a...);
}
template <typename T, typename... A> static inline void launch_bn_apply(bool nt, dim3 grid, hipStream_t s, A... a) {
    if (nt) hipLaunchKernelGGL((bn_apply_kernel<T, true>), grid, dim3(256), 0, s, a...);
    else hipLaunchKernelGGL((bn_apply_kernel<T, false>), grid, dim3(256), 0, s, a...);
}
template <typename T, typename... A> static inline void launch_bn_bwd_apply(bool nt, dim3 grid, hipStream_t s, A... a) {
    if (nt) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true>), grid, dim3(256), 0, s, a...);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<T, false>), grid, dim3(256), 0, s, a...);
}
static inline bool chan_ok(const void* x, int ld, long long P, int C, int vec = 4) {
    return x && P > 0 && C > 0 && C % vec == 0 && ld >= C && ld % vec == 0 && ((uintptr_t)x & 15) == 0;
}

namespace {

template <typename T>
int bn_stats_impl(const T* x, int ldx, long long P, int C, const float* gamma, const float* beta, float eps, float momentum,
                  float* moving_mean, float* moving_var, float* affine, float* saved, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!chan_ok(x, ldx, P, C, VecOf<T>::N) || !affine || !saved || !ws || ws_bytes < unetrir_bn_ws_bytes(P, C)) return UNETRIR_EINVAL;
    const ChanPlan pl = chan_plan(P, C, VecOf<T>::N);
    launch_chan_partial<0, T>(bn_nt(P, C, sizeof(T)), dim3(pl.nslab, pl.ngroups), s, x, ldx, (const T*)nullptr, 0,
                       (const float*)nullptr, (const float*)nullptr, 0, P, C, pl.QB, pl.rows_per_slab, (double*)ws);
    hipLaunchKernelGGL(bn_finalize_kernel<double>, fin_grid(C), dim3(FIN_T), 0, s, (const double*)ws, pl.nslab, P, C, gamma,
                       beta, eps, momentum, moving_mean, moving_var, affine, saved);
    return (int)hipGetLastError();
}

template <typename T>
int bn_apply_impl(const T* x, int ldx, long long P, int C, const float* affine, int relu, T* y, int ldy, hipStream_t s) {
    if (!chan_ok(x, ldx, P, C, VecOf<T>::N) || !y || ldy < C || ldy % VecOf<T>::N) return UNETRIR_EINVAL;
    launch_bn_apply<T>(bn_nt(P, C, sizeof(T)), dim3(grid_for(P * (C / VecOf<T>::N))), s, x, ldx, P, C, affine, relu | bn_nohoist_flag(), y, ldy);
    return (int)hipGetLastError();
}

template <typename T>
int bn_bwd_impl(const T* da, int ldda, const T* x, int ldx, long long P, int C, const float* affine, const float* saved,
                int relu, T* dx, int lddx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!chan_ok(x, ldx, P, C, VecOf<T>::N) || !chan_ok(da, ldda, P, C, VecOf<T>::N) || !dx || lddx < C || lddx % VecOf<T>::N || !affine || !saved || !ws ||
        ws_bytes < unetrir_bn_ws_bytes(P, C))
        return UNETRIR_EINVAL;
    const ChanPlan pl = chan_plan(P, C, VecOf<T>::N);
    double* part = (double*)ws;
    float* coef = (float*)(part + (size_t)pl.nslab * C * 2);
    launch_chan_partial<2, T>(bn_nt(P, C, sizeof(T)), dim3(pl.nslab, pl.ngroups), s, x, ldx, da, ldda, affine, saved,
                       relu, P, C, pl.QB, pl.rows_per_slab, part);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, fin_grid(C), dim3(FIN_T), 0, s, (const double*)part, pl.nslab, P, C, dgamma, dbeta, coef);
    launch_bn_bwd_apply<T>(bn_nt(P, C, sizeof(T)), dim3(grid_for(P * (C / VecOf<T>::N))), s, da, ldda, x, ldx, P, C, affine, saved,
                       (const float*)coef, relu | bn_nohoist_flag(), dx, lddx);
    return (int)hipGetLastError();
}

// BatchNormalization -> Add -> activation junction, backward in three launches: reduce, finalize, apply (+ the Add's other operand)
template <typename T>
int bn_bwd_junction_impl(const T* da, int ldda, const T* x, int ldx, const T* out, int ldo, long long P, int C, const float* affine,
                         const float* saved, int act, T* dx, int lddx, T* gskip, int ldgs, const T* gskip_add, int ldga,
                         float* dgamma, float* dbeta, void* ws, size_t ws_bytes, hipStream_t s) {
    constexpr int V = VecOf<T>::N;
    if (!chan_ok(x, ldx, P, C, V) || !chan_ok(da, ldda, P, C, V) || !chan_ok(out, ldo, P, C, V) || !dx || lddx < C || lddx % V || !affine ||
        !saved || !ws || ws_bytes < unetrir_bn_ws_bytes(P, C) || act < 0 || act > 2 || (gskip && (ldgs < C || ldgs % V)) ||
        (gskip_add && (!gskip || ldga < C || ldga % V)))
        return UNETRIR_EINVAL;
    const ChanPlan pl = chan_plan(P, C, V);
    double* part = (double*)ws;
    float* coef = (float*)(part + (size_t)pl.nslab * C * 2);
    launch_chan_partial<2, T>(bn_nt(P, C, sizeof(T)), dim3(pl.nslab, pl.ngroups), s, x, ldx, da, ldda, affine, saved,
                       act, P, C, pl.QB, pl.rows_per_slab, part, out, ldo);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, fin_grid(C), dim3(FIN_T), 0, s, (const double*)part, pl.nslab, P, C, dgamma, dbeta, coef);
    launch_bn_bwd_apply<T>(bn_nt(P, C, sizeof(T)), dim3(grid_for(P * (C / V))), s, da, ldda, x, ldx, P, C, affine, saved,
                       (const float*)coef, act, dx, lddx, out, ldo, gskip, ldgs, gskip_add, ldga);
    return (int)hipGetLastError();
}

template <typename T>
int colsum_impl(const T* x, int ldx, long long P, int C, float* out, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!chan_ok(x, ldx, P, C, VecOf<T>::N) || !out || !ws || ws_bytes < unetrir_bn_ws_bytes(P, C)) return UNETRIR_EINVAL;
    const ChanPlan pl = chan_plan(P, C, VecOf<T>::N);
    launch_chan_partial<1, T>(bn_nt(P, C, sizeof(T)), dim3(pl.nslab, pl.ngroups), s, x, ldx, (const T*)nullptr, 0,
                       (const float*)nullptr, (const float*)nullptr, 0, P, C, pl.QB, pl.rows_per_slab, (double*)ws);
    hipLaunchKernelGGL(colsum_finalize_kernel<double>, fin_grid(C), dim3(FIN_T), 0, s, (const double*)ws, pl.nslab, C, out, 0, C);
    return (int)hipGetLastError();
}

template <typename T>
int relu_bwd_impl(const T* da, int ldda, const T* x, int ldx, long long P, int C, T* dx, int lddx, hipStream_t s) {
    if (!chan_ok(x, ldx, P, C, VecOf<T>::N) || !chan_ok(da, ldda, P, C, VecOf<T>::N) || !dx || lddx < C || lddx % VecOf<T>::N) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(grid_for(P * (C / VecOf<T>::N))), dim3(256), 0, s, da, ldda, x, ldx, P, C,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 1, dx, lddx);
    return (int)hipGetLastError();
}

#define LOSS_BLOCKS 1024
template <typename T>
int sigmoid_loss_impl(const float* logits, int ldl, const float* target, int B, int H, int W, float alpha, float inv_norm,
                      float* pred, T* dlogits, int ldd, float* loss_out, void* ws, size_t ws_bytes, hipStream_t s,
                      const float* phase_ref = nullptr, const float* phase_w = nullptr) {
    if (!logits || ldl < 2 || !target || !pred || !dlogits || !loss_out || !ws || ws_bytes < (size_t)LOSS_BLOCKS * 3 * sizeof(double) ||
        B <= 0 || H <= 0 || W <= 0 || (ldd != 4 && ldd != 8))
        return UNETRIR_EINVAL;
    const unsigned nb = grid_for((long long)B * H * W, 256, LOSS_BLOCKS);
    hipLaunchKernelGGL(sigmoid_loss_kernel<T>, dim3(nb), dim3(256), 0, s, logits, ldl, target, B, H, W, alpha, inv_norm, pred,
                       dlogits, ldd, (double*)ws, phase_ref, phase_w);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, (const double*)ws, (int)nb, alpha, inv_norm, loss_out);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

size_t unetrir_bn_ws_bytes(long long P, int C) {
    if (P <= 0 || C <= 0) return 0;
    const ChanPlan pl = chan_plan(P, C);
    size_t b = ((size_t)pl.nslab * C * 2) * sizeof(double) + (size_t)2 * C * sizeof(float);
    return b;
}

int unetrir_bn_stats_f32(const float* x, int ldx, long long P, int C, const float* gamma, const float* beta, float eps,
                         float momentum, float* moving_mean, float* moving_var, float* affine, float* saved, void* ws,
                         size_t ws_bytes, unetrir_stream_t stream) {
    return bn_stats_impl<float>(x, ldx, P, C, gamma, beta, eps, momentum, moving_mean, moving_var, affine, saved, ws, ws_bytes,
                                (hipStream_t)stream);
}

int unetrir_bn_apply_f32(const float* x, int ldx, long long P, int C, const float* affine, int relu, float* y, int ldy,
                         unetrir_stream_t stream) {
    return bn_apply_impl<float>(x, ldx, P, C, affine, relu, y, ldy, (hipStream_t)stream);
}

int unetrir_bn_bwd_f32(const float* da, int ldda, const float* x, int ldx, long long P, int C, const float* gamma,
                       const float* affine, const float* saved, int relu, float* dx, int lddx, float* dgamma,
                       float* dbeta, void* ws, size_t ws_bytes, unetrir_stream_t stream) {
    (void)gamma;
    return bn_bwd_impl<float>(da, ldda, x, ldx, P, C, affine, saved, relu, dx, lddx, dgamma, dbeta, ws, ws_bytes, (hipStream_t)stream);
}

int unetrir_bn_bwd_junction_f32(const float* da, int ldda, const float* x, int ldx, const float* out, int ldo, long long P, int C,
                                const float* affine, const float* saved, int act, float* dx, int lddx, float* gskip, int ldgs,
                                const float* gskip_add, int ldga, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                                unetrir_stream_t stream) {
    return bn_bwd_junction_impl<float>(da, ldda, x, ldx, out, ldo, P, C, affine, saved, act, dx, lddx, gskip, ldgs, gskip_add, ldga,
                                       dgamma, dbeta, ws, ws_bytes, (hipStream_t)stream);
}

int unetrir_colsum_f32(const float* x, int ldx, long long P, int C, float* out, void* ws, size_t ws_bytes,
                       unetrir_stream_t stream) {
    return colsum_impl<float>(x, ldx, P, C, out, ws, ws_bytes, (hipStream_t)stream);
}

int unetrir_relu_fwd_f32(const float* x, int ldx, long long P, int C, float* y, int ldy, unetrir_stream_t stream) {
    return unetrir_bn_apply_f32(x, ldx, P, C, nullptr, 1, y, ldy, stream);
}

int unetrir_relu_bwd_f32(const float* da, int ldda, const float* x, int ldx, long long P, int C, float* dx, int lddx,
                         unetrir_stream_t stream) {
    return relu_bwd_impl<float>(da, ldda, x, ldx, P, C, dx, lddx, (hipStream_t)stream);
}

int unetrir_nchw_to_nhwc_pad_f32(const float* x, int B, int C, int H, int W, float* y, int Cpad, unetrir_stream_t stream) {
    if (!x || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cpad < C) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel<float>, dim3(grid_for((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream, x, B,
                       C, H, W, y, Cpad);
    return (int)hipGetLastError();
}

size_t unetrir_loss_ws_bytes(long long npix) { (void)npix; return (size_t)LOSS_BLOCKS * 3 * sizeof(double); }

int unetrir_sigmoid_loss_f32(const float* logits, int ldl, const float* target, int B, int H, int W, float alpha,
                             float inv_norm, float* pred, float* dlogits, float* loss_out, void* ws, size_t ws_bytes,
                             unetrir_stream_t stream) {
    return sigmoid_loss_impl<float>(logits, ldl, target, B, H, W, alpha, inv_norm, pred, dlogits, 4, loss_out, ws, ws_bytes,
                                    (hipStream_t)stream);
}

int unetrir_sigmoid_loss_ex_f32(const float* logits, int ldl, const float* target, const float* phase_ref, const float* phase_weight,
                                int B, int H, int W, float alpha, float inv_norm, float* pred, float* dlogits, float* loss_out,
                                void* ws, size_t ws_bytes, unetrir_stream_t stream) {
    return sigmoid_loss_impl<float>(logits, ldl, target, B, H, W, alpha, inv_norm, pred, dlogits, 4, loss_out, ws, ws_bytes,
                                    (hipStream_t)stream, phase_ref, phase_weight);
}

int unetrir_sigmoid_nchw_f32(const float* logits, int ldl, int B, int H, int W, float* pred, unetrir_stream_t stream) {
    if (!logits || ldl < 2 || !pred || B <= 0 || H <= 0 || W <= 0) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(sigmoid_only_kernel, dim3(grid_for((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream, logits,
                       ldl, B, H, W, pred);
    return (int)hipGetLastError();
}

int unetrir_sigmoid_bwd_f32(const float* pred, const float* dpred, int B, int H, int W, float* dlogits,
                            unetrir_stream_t stream) {
    if (!pred || !dpred || !dlogits || B <= 0 || H <= 0 || W <= 0) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(sigmoid_bwd_kernel<float>, dim3(grid_for((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream, pred,
                       dpred, B, H, W, dlogits, 4);
    return (int)hipGetLastError();
}

int unetrir_embedding_fwd_f32(const int* idx, int n_idx, const float* table, int vocab, int dim, float* out,
                              unetrir_stream_t stream) {
    if (!idx || !table || !out || n_idx <= 0 || vocab <= 0 || dim <= 0) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(n_idx), dim3(256), 0, (hipStream_t)stream, idx, n_idx, table, vocab, dim, out);
    return (int)hipGetLastError();
}

int unetrir_embedding_bwd_f32(const int* idx, int n_idx, const float* dout, int vocab, int dim, float* dtable,
                              unetrir_stream_t stream) {
    if (!idx || !dout || !dtable || n_idx <= 0 || vocab <= 0 || dim <= 0) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(vocab), dim3(256), 0, (hipStream_t)stream, idx, n_idx, dout, vocab, dim,
                       dtable);
    return (int)hipGetLastError();
}

int unetrir_mul_f32(const float* x, const float* m, float* y, long long n, unetrir_stream_t stream) {
    if (!x || !m || !y || n <= 0) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(mul_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, m, y, n);
    return (int)hipGetLastError();
}

#define SUMSQ_BLOCKS 512
int unetrir_sumsq_f32(const float* x, long long n, float coef, float* out, int accumulate, void* ws, size_t ws_bytes,
                      unetrir_stream_t stream) {
    if (!x || !out || n <= 0 || !ws || ws_bytes < SUMSQ_BLOCKS * sizeof(double)) return UNETRIR_EINVAL;
    const unsigned nb = grid_for(n, 256, SUMSQ_BLOCKS);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, s, x, n, (double*)ws);
    hipLaunchKernelGGL(sumsq_finalize_kernel, dim3(1), dim3(64), 0, s, (const double*)ws, (int)nb, coef, out, accumulate);
    return (int)hipGetLastError();
}

int unetrir_adam_f32(float* theta, const float* g, float* m, float* v, long long n, float lr_t, float beta1, float beta2,
                     float eps, float grad_scale, unetrir_stream_t stream) {
    if (!theta || !g || !m || !v || n <= 0) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4 + 1, 256, 8192)), dim3(256), 0, (hipStream_t)stream, theta, g, m, v, n,
                       lr_t, beta1, beta2, eps, grad_scale, (const float*)nullptr);
    return (int)hipGetLastError();
}

int unetrir_sgd_f32(float* theta, const float* g, long long n, float lr, float grad_scale, unetrir_stream_t stream) {
    if (!theta || !g || n <= 0) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n / 4 + 1, 256, 8192)), dim3(256), 0, (hipStream_t)stream, theta, g, n, lr, grad_scale);
    return (int)hipGetLastError();
}

int unetrir_nadam_f32(float* theta, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                      float c_g, float c_m, float c_v, float grad_scale, unetrir_stream_t stream) {
    if (!theta || !g || !m || !v || n <= 0) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(nadam_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, theta, g, m, v, n, lr, beta1, beta2,
                       eps, c_g, c_m, c_v, grad_scale);
    return (int)hipGetLastError();
}

int unetrir_adam_dev_f32(float* theta, const float* g, float* m, float* v, long long n, const float* hyper, unetrir_stream_t stream) {
    if (!theta || !g || !m || !v || !hyper || n <= 0) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4 + 1, 256, 8192)), dim3(256), 0, (hipStream_t)stream, theta, g, m, v, n,
                       0.f, 0.f, 0.f, 0.f, 0.f, hyper);
    return (int)hipGetLastError();
}

}  // extern "C"

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 x = ld4(a + i * 4), v = ld4(b + i * 4);
        st4(y + i * 4, make_float4(x.x + v.x, x.y + v.y, x.z + v.z, x.w + v.w));
    }
}

extern "C" {

/* y = act(x*scale + shift + addend): BatchNormalization -> Add -> LeakyReLU of the residual blocks
 * (dl_models/res_ae.py:331-336, :475-480).  act: 0 none, 1 ReLU, 2 LeakyReLU(0.3).  affine may be NULL (identity). */
int unetrir_bn_act_add_f32(const float* x, int ldx, long long P, int C, const float* affine, int act, const float* addend,
                           int ldadd, float* y, int ldy, unetrir_stream_t stream) {
    if (!chan_ok(x, ldx, P, C) || !y || ldy < C || (ldy & 3) || act < 0 || act > 2 || (addend && (ldadd < C || (ldadd & 3))))
        return UNETRIR_EINVAL;
    hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(grid_for(P * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, P, C, affine,
                       act, y, ldy, addend, ldadd);
    return (int)hipGetLastError();
}

/* g = da * act'(out) with the derivative decided by the sign of the activation OUTPUT (valid for ReLU / LeakyReLU):
 * backward of the Add -> LeakyReLU junction, whose result feeds both BatchNorm backward passes. */
int unetrir_act_bwd_f32(const float* da, int ldda, const float* out, int ldo, long long P, int C, int act, float* g, int ldg,
                        unetrir_stream_t stream) {
    if (!chan_ok(out, ldo, P, C) || !chan_ok(da, ldda, P, C) || !g || ldg < C || (ldg & 3) || act < 1 || act > 2) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(grid_for(P * (C / 4))), dim3(256), 0, (hipStream_t)stream, da, ldda, out, ldo,
                       P, C, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, act, g, ldg);
    return (int)hipGetLastError();
}

/* y = a + b over n floats (n % 4 == 0): gradient accumulation where a tensor has two consumers */
int unetrir_add_f32(const float* a, const float* b, float* y, long long n, unetrir_stream_t stream) {
    if (!a || !b || !y || n <= 0 || (n & 3)) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, a, b, y, n / 4);
    return (int)hipGetLastError();
}

}  // extern "C"

// -------------------------------------------------------------------------------------------
// bf16-storage variants (activations bf16, statistics / parameters fp32) and the fp32 <-> bf16 glue of the
// information-vector branch, which stays fp32
// -------------------------------------------------------------------------------------------
__global__ void add_f32_to_bf16_kernel(const __bf16* __restrict__ a, const float* __restrict__ b, __bf16* __restrict__ y, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 x = ld4(a + i * 4), v = ld4(b + i * 4);
        st4(y + i * 4, make_float4(x.x + v.x, x.y + v.y, x.z + v.z, x.w + v.w));
    }
}

__global__ void cast_bf16_to_f32_kernel(const __bf16* __restrict__ a, float* __restrict__ y, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        st4(y + i * 4, ld4(a + i * 4));
}

__global__ void cast_f32_to_bf16_kernel(const float* __restrict__ a, __bf16* __restrict__ y, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        st4(y + i * 4, ld4(a + i * 4));
}

extern "C" {

/* bf16-storage forms of the residual-block glue (unetrir_bn_act_add_f32 / unetrir_act_bwd_f32): C % 8 == 0, 16-byte rows */
int unetrir_bn_act_add_bf16(const unetrir_bf16* x, int ldx, long long P, int C, const float* affine, int act, const unetrir_bf16* addend,
                            int ldadd, unetrir_bf16* y, int ldy, unetrir_stream_t stream) {
    if (!chan_ok(x, ldx, P, C, 8) || !y || ldy < C || (ldy & 7) || act < 0 || act > 2 || (addend && (ldadd < C || (ldadd & 7))))
        return UNETRIR_EINVAL;
    hipLaunchKernelGGL(bn_apply_kernel<__bf16>, dim3(grid_for(P * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, ldx, P, C,
                       affine, act, (__bf16*)y, ldy, (const __bf16*)addend, ldadd);
    return (int)hipGetLastError();
}

int unetrir_act_bwd_bf16(const unetrir_bf16* da, int ldda, const unetrir_bf16* out, int ldo, long long P, int C, int act,
                         unetrir_bf16* g, int ldg, unetrir_stream_t stream) {
    if (!chan_ok(out, ldo, P, C, 8) || !chan_ok(da, ldda, P, C, 8) || !g || ldg < C || (ldg & 7) || act < 1 || act > 2) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(bn_bwd_apply_kernel<__bf16>, dim3(grid_for(P * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const __bf16*)da, ldda,
                       (const __bf16*)out, ldo, P, C, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, act,
                       (__bf16*)g, ldg);
    return (int)hipGetLastError();
}

int unetrir_cast_f32_to_bf16(const float* a, unetrir_bf16* y, long long n, unetrir_stream_t stream) {
    if (!a || !y || n <= 0 || (n & 3)) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(cast_f32_to_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, a, (__bf16*)y, n / 4);
    return (int)hipGetLastError();
}

int unetrir_bn_stats_bf16(const unetrir_bf16* x, int ldx, long long P, int C, const float* gamma, const float* beta, float eps,
                          float momentum, float* moving_mean, float* moving_var, float* affine, float* saved, void* ws,
                          size_t ws_bytes, unetrir_stream_t stream) {
    return bn_stats_impl<__bf16>((const __bf16*)x, ldx, P, C, gamma, beta, eps, momentum, moving_mean, moving_var, affine, saved, ws,
                                 ws_bytes, (hipStream_t)stream);
}

int unetrir_bn_apply_bf16(const unetrir_bf16* x, int ldx, long long P, int C, const float* affine, int relu, unetrir_bf16* y, int ldy,
                          unetrir_stream_t stream) {
    return bn_apply_impl<__bf16>((const __bf16*)x, ldx, P, C, affine, relu, (__bf16*)y, ldy, (hipStream_t)stream);
}

int unetrir_bn_bwd_bf16(const unetrir_bf16* da, int ldda, const unetrir_bf16* x, int ldx, long long P, int C, const float* affine,
                        const float* saved, int relu, unetrir_bf16* dx, int lddx, float* dgamma, float* dbeta, void* ws,
                        size_t ws_bytes, unetrir_stream_t stream) {
    return bn_bwd_impl<__bf16>((const __bf16*)da, ldda, (const __bf16*)x, ldx, P, C, affine, saved, relu, (__bf16*)dx, lddx, dgamma,
                               dbeta, ws, ws_bytes, (hipStream_t)stream);
}

int unetrir_bn_bwd_junction_bf16(const unetrir_bf16* da, int ldda, const unetrir_bf16* x, int ldx, const unetrir_bf16* out, int ldo,
                                 long long P, int C, const float* affine, const float* saved, int act, unetrir_bf16* dx, int lddx,
                                 unetrir_bf16* gskip, int ldgs, const unetrir_bf16* gskip_add, int ldga, float* dgamma, float* dbeta,
                                 void* ws, size_t ws_bytes, unetrir_stream_t stream) {
    return bn_bwd_junction_impl<__bf16>((const __bf16*)da, ldda, (const __bf16*)x, ldx, (const __bf16*)out, ldo, P, C, affine, saved, act,
                                        (__bf16*)dx, lddx, (__bf16*)gskip, ldgs, (const __bf16*)gskip_add, ldga, dgamma, dbeta, ws, ws_bytes,
                                        (hipStream_t)stream);
}

int unetrir_colsum_bf16(const unetrir_bf16* x, int ldx, long long P, int C, float* out, void* ws, size_t ws_bytes,
                        unetrir_stream_t stream) {
    return colsum_impl<__bf16>((const __bf16*)x, ldx, P, C, out, ws, ws_bytes, (hipStream_t)stream);
}

int unetrir_relu_bwd_bf16(const unetrir_bf16* da, int ldda, const unetrir_bf16* x, int ldx, long long P, int C, unetrir_bf16* dx,
                          int lddx, unetrir_stream_t stream) {
    return relu_bwd_impl<__bf16>((const __bf16*)da, ldda, (const __bf16*)x, ldx, P, C, (__bf16*)dx, lddx, (hipStream_t)stream);
}

int unetrir_nchw_to_nhwc_pad_bf16(const float* x, int B, int C, int H, int W, unetrir_bf16* y, int Cpad, unetrir_stream_t stream) {
    if (!x || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cpad < C) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel<__bf16>, dim3(grid_for((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream, x, B,
                       C, H, W, (__bf16*)y, Cpad);
    return (int)hipGetLastError();
}

int unetrir_sigmoid_loss_bf16(const float* logits, int ldl, const float* target, int B, int H, int W, float alpha, float inv_norm,
                              float* pred, unetrir_bf16* dlogits, float* loss_out, void* ws, size_t ws_bytes,
                              unetrir_stream_t stream) {
    return sigmoid_loss_impl<__bf16>(logits, ldl, target, B, H, W, alpha, inv_norm, pred, (__bf16*)dlogits, 8, loss_out, ws, ws_bytes,
                                     (hipStream_t)stream);
}

int unetrir_sigmoid_loss_ex_bf16(const float* logits, int ldl, const float* target, const float* phase_ref, const float* phase_weight,
                                 int B, int H, int W, float alpha, float inv_norm, float* pred, unetrir_bf16* dlogits, float* loss_out,
                                 void* ws, size_t ws_bytes, unetrir_stream_t stream) {
    return sigmoid_loss_impl<__bf16>(logits, ldl, target, B, H, W, alpha, inv_norm, pred, (__bf16*)dlogits, 8, loss_out, ws, ws_bytes,
                                     (hipStream_t)stream, phase_ref, phase_weight);
}

int unetrir_sigmoid_bwd_bf16(const float* pred, const float* dpred, int B, int H, int W, unetrir_bf16* dlogits,
                             unetrir_stream_t stream) {
    if (!pred || !dpred || !dlogits || B <= 0 || H <= 0 || W <= 0) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(sigmoid_bwd_kernel<__bf16>, dim3(grid_for((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream, pred,
                       dpred, B, H, W, (__bf16*)dlogits, 8);
    return (int)hipGetLastError();
}

int unetrir_add_f32_to_bf16(const unetrir_bf16* a, const float* b, unetrir_bf16* y, long long n, unetrir_stream_t stream) {
    if (!a || !b || !y || n <= 0 || (n & 3)) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(add_f32_to_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)a, b,
                       (__bf16*)y, n / 4);
    return (int)hipGetLastError();
}

int unetrir_cast_bf16_to_f32(const unetrir_bf16* a, float* y, long long n, unetrir_stream_t stream) {
    if (!a || !y || n <= 0 || (n & 3)) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(cast_bf16_to_f32_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)a, y, n / 4);
    return (int)hipGetLastError();
}

}  // extern "C"

/* BatchNormalization forward from a convolution's column statistics in ONE call: statistics -> affine (+ moving statistics), then
 * y = act(x * scale + shift (+ addend)): the finalize and the apply launch. */
template <typename T>
static int bn_colstat_act_add_impl(const float* colstat, long long rows, const T* x, int ldx, long long P, int C, const float* gamma,
                                   const float* beta, float eps, float momentum, float* mm, float* mv, float* affine, float* saved, int act,
                                   const T* addend, int ldadd, T* y, int ldy, hipStream_t s) {
    constexpr int V = VecOf<T>::N;
    if (!colstat || rows <= 0 || rows > 0x7fffffffLL || !chan_ok(x, ldx, P, C, V) || !affine || !saved || !y || ldy < C || ldy % V || act < 0 ||
        act > 2 || (addend && (ldadd < C || ldadd % V)))
        return UNETRIR_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel<float>, fin_grid(C), dim3(FIN_T), 0, s, colstat, (int)rows, P, C, gamma, beta, eps, momentum, mm, mv, affine, saved);
    launch_bn_apply<T>(bn_nt(P, C, sizeof(T)), dim3(grid_for(P * (C / V))), s, x, ldx, P, C, (const float*)affine, act, y, ldy, addend, ldadd);
    return (int)hipGetLastError();
}

// -------------------------------------------------------------------------------------------
// statistics from the per-tile partials a convolution epilogue produced (unetrir_conv2d_*_colstat_bf16):
// colstat is [rows][ldc][2] floats (sum, sum of squares); fixed-order fp64 reduction over the rows
// -------------------------------------------------------------------------------------------
extern "C" {

int unetrir_bn_stats_colstat(const float* colstat, long long rows, long long P, int C, const float* gamma, const float* beta, float eps,
                             float momentum, float* moving_mean, float* moving_var, float* affine, float* saved,
                             unetrir_stream_t stream) {
    if (!colstat || rows <= 0 || rows > 0x7fffffffLL || P <= 0 || C <= 0 || !affine || !saved) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel<float>, fin_grid(C), dim3(FIN_T), 0, (hipStream_t)stream, colstat, (int)rows, P, C, gamma, beta, eps,
                       momentum, moving_mean, moving_var, affine, saved);
    return (int)hipGetLastError();
}

int unetrir_bn_colstat_act_add_f32(const float* colstat, long long rows, const float* x, int ldx, long long P, int C, const float* gamma,
                                   const float* beta, float eps, float momentum, float* moving_mean, float* moving_var, float* affine,
                                   float* saved, int act, const float* addend, int ldadd, float* y, int ldy, unetrir_stream_t stream) {
    return bn_colstat_act_add_impl<float>(colstat, rows, x, ldx, P, C, gamma, beta, eps, momentum, moving_mean, moving_var, affine, saved, act,
                                          addend, ldadd, y, ldy, (hipStream_t)stream);
}

int unetrir_bn_colstat_act_add_bf16(const float* colstat, long long rows, const unetrir_bf16* x, int ldx, long long P, int C, const float* gamma,
                                    const float* beta, float eps, float momentum, float* moving_mean, float* moving_var, float* affine,
                                    float* saved, int act, const unetrir_bf16* addend, int ldadd, unetrir_bf16* y, int ldy,
                                    unetrir_stream_t stream) {
    return bn_colstat_act_add_impl<__bf16>(colstat, rows, (const __bf16*)x, ldx, P, C, gamma, beta, eps, momentum, moving_mean, moving_var, affine,
                                           saved, act, (const __bf16*)addend, ldadd, (__bf16*)y, ldy, (hipStream_t)stream);
}

int unetrir_colsum_colstat(const float* colstat, long long rows, int ldc, int c0, int C, float* out, unetrir_stream_t stream) {
    if (!colstat || rows <= 0 || rows > 0x7fffffffLL || C <= 0 || c0 < 0 || c0 + C > ldc || !out) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(colsum_finalize_kernel<float>, fin_grid(C), dim3(FIN_T), 0, (hipStream_t)stream, colstat, (int)rows, ldc, out, c0, C);
    return (int)hipGetLastError();
}

}  // extern "C"

// -------------------------------------------------------------------------------------------
// small host-side glue moved onto the device so that a train step launches no framework kernels:
// BatchNorm inference affine, the Dropout keep mask, the index cast of the information vector
// -------------------------------------------------------------------------------------------
__global__ void bn_inference_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                           const float* __restrict__ mm, const float* __restrict__ mv, float eps, int C,
                                           float* __restrict__ affine) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float scale = (gamma ? gamma[c] : 1.f) / sqrtf(mv[c] + eps);
    affine[c] = scale;
    affine[C + c] = (beta ? beta[c] : 0.f) - mm[c] * scale;
}

// counter-based generator: element i of draw (seed, step) is a fixed function of (seed, step, i) - splitmix64 finaliser
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__global__ void dropout_mask_kernel(float* __restrict__ mask, long long n, float p, float keep_scale, unsigned long long seed,
                                    unsigned long long step, const unsigned long long* __restrict__ step_dev) {
    if (step_dev) step += *step_dev;          // draw number from device memory (a captured HIP graph replays the same arguments)
    const unsigned long long key = mix64(seed * 0x9E3779B97F4A7C15ULL + step);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const unsigned long long r = mix64(key + 0x9E3779B97F4A7C15ULL * (unsigned long long)(i + 1));
        const float u = (float)(r >> 40) * (1.0f / 16777216.0f);          // 24 bits -> [0, 1)
        mask[i] = (u >= p) ? keep_scale : 0.f;
    }
}

// Step counters kept in device memory so that a captured HIP graph can be replayed unchanged: state[0] = Adam step count t,
// state[1] = dropout draws made so far, state[2] = first draw of the CURRENT step (what unetrir_dropout_mask_dev_f32 reads).
// cfg = {lr, beta1, beta2, eps, grad_scale} as the host last set them; hyper = the same with lr replaced by
// lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) for the step that begins (main_training.py:168-169, :268).
__global__ void step_advance_kernel(unsigned long long* __restrict__ state, const float* __restrict__ cfg, float* __restrict__ hyper,
                                    int n_draws, int advance_t) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned long long t = state[0] + (advance_t ? 1ull : 0ull);      // advance_t == 0: a forward-only pass (validation) draws masks too
    state[0] = t;
    const double b1 = (double)cfg[1], b2 = (double)cfg[2];
    // t == 0 (a forward-only pass on a fresh engine): no optimizer step can follow, and 0 / 0 must not reach hyper[0]
    hyper[0] = t == 0 ? 0.f : (float)((double)cfg[0] * sqrt(1.0 - pow(b2, (double)t)) / (1.0 - pow(b1, (double)t)));
    hyper[1] = cfg[1]; hyper[2] = cfg[2]; hyper[3] = cfg[3]; hyper[4] = cfg[4];
    const unsigned long long d = state[1];
    state[2] = d;
    state[1] = d + (unsigned long long)n_draws;
}

__global__ void index_to_i32_kernel(const void* __restrict__ idx, int elem_bytes, long long n, int* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = elem_bytes == 8 ? (int)reinterpret_cast<const long long*>(idx)[i] : reinterpret_cast<const int*>(idx)[i];
}

extern "C" {

int unetrir_bn_inference_affine_f32(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                                    float eps, int C, float* affine, unetrir_stream_t stream) {
    if (!moving_mean || !moving_var || !affine || C <= 0) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(bn_inference_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, moving_mean,
                       moving_var, eps, C, affine);
    return (int)hipGetLastError();
}

int unetrir_dropout_mask_f32(float* mask, long long n, float p, unsigned long long seed, unsigned long long step,
                             unetrir_stream_t stream) {
    if (!mask || n <= 0 || !(p >= 0.f) || !(p < 1.f)) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, mask, n, p, 1.f / (1.f - p), seed, step,
                       (const unsigned long long*)nullptr);
    return (int)hipGetLastError();
}

int unetrir_dropout_mask_dev_f32(float* mask, long long n, float p, unsigned long long seed, const unsigned long long* step_base,
                                 unsigned long long step_offset, unetrir_stream_t stream) {
    if (!mask || !step_base || n <= 0 || !(p >= 0.f) || !(p < 1.f)) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, mask, n, p, 1.f / (1.f - p), seed,
                       step_offset, step_base);           // draw number = *step_base + step_offset
    return (int)hipGetLastError();
}

int unetrir_step_advance(unsigned long long* state, const float* cfg, float* hyper, int n_draws, int advance_t, unetrir_stream_t stream) {
    if (!state || !cfg || !hyper || n_draws < 0) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, cfg, hyper, n_draws, advance_t);
    return (int)hipGetLastError();
}

int unetrir_index_to_i32(const void* idx, int elem_bytes, long long n, int* out, unetrir_stream_t stream) {
    if (!idx || !out || n <= 0 || (elem_bytes != 4 && elem_bytes != 8)) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(index_to_i32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, idx, elem_bytes, n, out);
    return (int)hipGetLastError();
}

}  // extern "C"
