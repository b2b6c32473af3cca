// bnfused.hip - BatchNormalization passes of SMALL tensors in one launch each (gfx950).
//
// The residual graphs (dl_models/res_ae.py, BASELINE.json configs[4]) run 56 BatchNormalization layers per step on tensors of
// 4 ... 34 MB.  As separate launches (elementwise.hip) the backward pass is reduce -> finalize -> apply and the forward pass
// finalize -> apply: at the 32 x 32 and 16 x 16 levels every one of those launches sits at its 5-8 us floor, and at the 128 x 128
// level the apply pass re-reads from HBM what the reduce pass has just streamed.  Here the phases of one direction run in ONE
// kernel, separated by grid barriers:
//
//   backward  (1) per-workgroup per-channel partial sums of g and g * xhat (fp64, fixed row assignment)  | barrier |
//             (2) workgroup c sums channel c over the workgroups' partials in a fixed order -> dgamma, dbeta, the two means
//             | barrier | (3) dx = scale * (g - mean(g) - xhat * mean(g xhat)) (+ the junction's second output), the tensors
//             now coming from L2 / Infinity Cache where they fit;
//   forward   (1) workgroup c reduces channel c of the convolution epilogue's column-statistics rows -> scale / shift, saved mean /
//             rstd, moving statistics | barrier | (2) y = act(x * scale + shift (+ addend)).
//
// The barrier is an arrival counter in the stream's sync slot (kernels.h): a workgroup increments it once its stores are
// acknowledged and thread 0 polls it; one workgroup per CU (all resident at once on an idle device; a workgroup that cannot be
// placed yet is simply late - nothing it waits for depends on it) and the poll gives up after ~2^22 rounds (a flag in the slot)
// instead of hanging the device.  Same arithmetic as the separate launches, partial rows indexed by workgroup (not by arrival):
// bit-reproducible run to run.
//
// MEASURED AND NOT ADOPTED (round 3, scripts/micro_bn.py --resae --bn_fused=0|1, batch 32): OFF by default (switch bn_fused).
//   backward, separate launches / fused:  128 x 128 x 32: 48 / 69 us   64 x 64 x 64: 30 / 46   32 x 32 x 128: 19 / 32   16 x 16 x 256: 17 / 28
//   forward (finalize + apply) / fused:                   16 / 26                    12 / 17                    13 / 16                    13 / 14
// A software grid barrier costs ~8 us at 256 workgroups (256 pollers of one memory-side counter; with agent-scope fences instead of
// write-through stores 25 us, and 80 us at 1024 workgroups), i.e. MORE than the kernel boundary it replaces (~3 us), and 256
// workgroups of 4 waves do not keep enough loads in flight for the streaming phases (the separate kernels run 1024-4096).  On this
// device a kernel boundary is the cheaper grid-wide synchronisation; kept as a tested switch so that the measurement can be repeated.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "kernels.h"

namespace {

typedef __bf16 bfx8 __attribute__((ext_vector_type(8)));
template <typename T> struct Vec;
template <> struct Vec<float> { static constexpr int N = 4; };
template <> struct Vec<__bf16> { static constexpr int N = 8; };
__device__ __forceinline__ void ldvec(const float* p, float (&o)[4]) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void ldvec(const __bf16* p, float (&o)[8]) {
    const bfx8 v = *reinterpret_cast<const bfx8*>(p);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (float)v[k];
}
__device__ __forceinline__ void stvec(float* p, const float (&o)[4]) { *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]); }
__device__ __forceinline__ void stvec(__bf16* p, const float (&o)[8]) {
    bfx8 v;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (__bf16)o[k];
    *reinterpret_cast<bfx8*>(p) = v;
}

// Values that cross workgroups inside a launch (the partial sums, the finished per-channel constants) are written and read with
// agent-scope atomic stores / loads: write-through stores (sc1) and cache-bypassing loads, coherent at the memory side of the 8
// XCDs' L2s WITHOUT cache-wide fences.  (A release / acquire fence pair per workgroup - buffer_wbl2 walks the whole L2 - cost
// 25 us per barrier at 256 workgroups and 80 us at 1024.)
__device__ __forceinline__ void st_dev(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_dev(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_dev(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// all threads of every workgroup of the launch call this; `target` = workgroups x (number of this barrier, 1-based).
// __syncthreads() = every wave's stores are acknowledged (s_waitcnt vmcnt(0): for the write-through stores above that is "visible
// device-wide"); then one arrival per workgroup and thread 0 polls.
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins > (1u << 22)) { __hip_atomic_store(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // give up, leave a mark
        }
    }
    __syncthreads();
}
// the last workgroup to finish leaves the counter at zero for the next launch on this stream
__device__ __forceinline__ void grid_finish(unsigned* ctr, unsigned total_after) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == total_after - 1u) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// fixed-order sum of `n` partial (a, b) pairs of channel c by the 256 threads of a workgroup: thread k takes rows k, k + 256, ...,
// then a tree over the 256 thread sums (the order of elementwise.hip's slab_sum)
template <typename PT, bool DEV>
__device__ __forceinline__ void rows_sum(const PT* part, int n, int C, int c, double* red, double& s, double& ss) {
    double a = 0, b = 0;
    for (int k = threadIdx.x; k < n; k += 256) {
        const PT* q = part + ((size_t)k * C + c) * 2;
        a += (double)(DEV ? ld_dev(q) : q[0]); b += (double)(DEV ? ld_dev(q + 1) : q[1]);
    }
    __syncthreads();
    red[threadIdx.x * 2] = a; red[threadIdx.x * 2 + 1] = b;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) { red[threadIdx.x * 2] += red[(threadIdx.x + st) * 2]; red[threadIdx.x * 2 + 1] += red[(threadIdx.x + st) * 2 + 1]; }
        __syncthreads();
    }
    s = red[0]; ss = red[1];
}

// ---------------------------------------------------------------------------------------------------------------- backward
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_fused_kernel(const T* __restrict__ da, int ldda, const T* __restrict__ x, int ldx,
                                                           const T* __restrict__ msk, int ldm, long long P, int C,
                                                           const float* __restrict__ affine, const float* __restrict__ saved, int act,
                                                           T* __restrict__ dx, int lddx, T* __restrict__ g2, int ldg2,
                                                           const T* g2add, int ldg2a, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           double* part, float* coef, unsigned* sync) {
    constexpr int V = Vec<T>::N;
    __shared__ double red[256 * 4];
    const int tid = threadIdx.x, G = gridDim.x;
    const int CQ = C / V;                              // <= 256 (checked by the launcher): one workgroup spans every channel
    int QB = 1;
    while (QB * 2 <= CQ && QB * 2 <= 256) QB *= 2;     // QB channel vectors x RB rows per trip; CQ not a power of two: a second vector group
    const int RB = 256 / QB;
    const int ql = tid % QB, rl = tid / QB;
    const int ngroups = (CQ + QB - 1) / QB;
    const float sl = act == 2 ? 0.3f : 0.f;

    // ---- phase 1: partial sums, rows blockIdx.x * RB + rl, + G * RB, ...
    for (int gq = 0; gq < ngroups; ++gq) {
        const int q = gq * QB + ql, c0 = q * V;
        const bool cok = q < CQ;
        double s0[V], s1[V];
        float sc[V], sh[V], mu[V], rs[V];
#pragma unroll
        for (int k = 0; k < V; ++k) { s0[k] = 0; s1[k] = 0; sc[k] = 1; sh[k] = 0; mu[k] = 0; rs[k] = 1; }
        if (cok) {
#pragma unroll
            for (int k = 0; k < V; ++k) { sc[k] = affine[c0 + k]; sh[k] = affine[C + c0 + k]; mu[k] = saved[c0 + k]; rs[k] = saved[C + c0 + k]; }
            for (long long p = (long long)blockIdx.x * RB + rl; p < P; p += (long long)G * RB) {
                float xv[V], gv[V], mv[V];
                ldvec(x + (size_t)p * ldx + c0, xv);
                ldvec(da + (size_t)p * ldda + c0, gv);
                if (msk) ldvec(msk + (size_t)p * ldm + c0, mv);
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const float a = msk ? mv[k] : xv[k] * sc[k] + sh[k];
                    const float g = (act && !(a > 0.f)) ? sl * gv[k] : gv[k];
                    const float xh = (xv[k] - mu[k]) * rs[k];
                    s0[k] += (double)g; s1[k] += (double)g * (double)xh;
                }
            }
        }
        // rows of the workgroup -> one partial per channel, four values at a time through 8 KB of LDS, fixed order
#pragma unroll
        for (int ch = 0; ch < 2 * V / 4; ++ch) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int idx = ch * 4 + j; red[tid * 4 + j] = idx < V ? s0[idx] : s1[idx - V]; }
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
                st_dev(part + ((size_t)blockIdx.x * C + c0 + k) * 2 + 0, s0[k]);
                st_dev(part + ((size_t)blockIdx.x * C + c0 + k) * 2 + 1, s1[k]);
            }
        }
    }
    grid_barrier(sync, (unsigned)G);

    // ---- phase 2: workgroup b finishes channels b, b + G, ...
    for (int c = blockIdx.x; c < C; c += G) {
        double s, ss;
        rows_sum<double, true>(part, G, C, c, red, s, ss);
        if (tid == 0) {
            if (dbeta) dbeta[c] = (float)s;
            if (dgamma) dgamma[c] = (float)ss;
            st_dev(coef + c, (float)(s / (double)P));
            st_dev(coef + C + c, (float)(ss / (double)P));
        }
    }
    grid_barrier(sync, 2u * (unsigned)G);

    // ---- phase 3: dx (+ the other operand of a junction); the grid stride is a multiple of the channel-vector count for every
    // power-of-two channel count, so a thread keeps ONE channel group and its six per-channel constants in registers
    const long long total = P * CQ;
    const long long stride = (long long)G * 256;
    const bool hoist = (stride % CQ) == 0;
    float sc[V], sh[V], mu[V], rs[V], c1[V], c2[V];
    auto consts = [&](int cc) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            sc[k] = affine[cc + k]; sh[k] = affine[C + cc + k]; mu[k] = saved[cc + k]; rs[k] = saved[C + cc + k];
            c1[k] = ld_dev(coef + cc + k); c2[k] = ld_dev(coef + C + cc + k);
        }
    };
    const long long i0 = (long long)blockIdx.x * 256 + tid;
    if (hoist) consts((int)(i0 % CQ) * V);
    const long long pstep = stride / CQ;               // hoisted form: the pixel index advances by a constant, no division in the loop
    long long ph = i0 / CQ;
    const int cch = (int)(i0 % CQ) * V;
    for (long long i = i0; i < total; i += stride, ph += pstep) {
        const long long p = hoist ? ph : i / CQ;
        const int cc = hoist ? cch : (int)(i - p * CQ) * V;
        if (!hoist) consts(cc);
        float xv[V], gv[V], mv[V], out[V], go[V];
        ldvec(x + (size_t)p * ldx + cc, xv);
        ldvec(da + (size_t)p * ldda + cc, gv);
        if (msk) ldvec(msk + (size_t)p * ldm + cc, mv);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float a = msk ? mv[k] : xv[k] * sc[k] + sh[k];
            const float g = (act && !(a > 0.f)) ? sl * gv[k] : gv[k];
            const float xh = (xv[k] - mu[k]) * rs[k];
            out[k] = sc[k] * (g - c1[k] - xh * c2[k]);
            go[k] = g;
        }
        stvec(dx + (size_t)p * lddx + cc, out);
        if (g2) {
            if (g2add) {
                float ad[V];
                ldvec(g2add + (size_t)p * ldg2a + cc, ad);
#pragma unroll
                for (int k = 0; k < V; ++k) go[k] += ad[k];
            }
            stvec(g2 + (size_t)p * ldg2 + cc, go);
        }
    }
    grid_finish(sync, 3u * (unsigned)G);
}

// ----------------------------------------------------------------------------------------------------------------- forward
template <typename T>
__global__ __launch_bounds__(256) void bn_fwd_fused_kernel(const float* colstat, int rows, const T* __restrict__ x, int ldx, long long P, int C,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                           float momentum, float* __restrict__ moving_mean, float* __restrict__ moving_var,
                                                           float* affine, float* __restrict__ saved, int act, const T* __restrict__ addend,
                                                           int ldadd, T* __restrict__ y, int ldy, unsigned* sync) {
    constexpr int V = Vec<T>::N;
    __shared__ double red[256 * 2];
    const int tid = threadIdx.x, G = gridDim.x;
    for (int c = blockIdx.x; c < C; c += G) {          // the arithmetic of bn_finalize_kernel (elementwise.hip)
        double s, ss;
        rows_sum<float, false>(colstat, rows, C, c, red, s, ss);      // written by an earlier launch: ordinary loads
        if (tid == 0) {
            const double mean = s / (double)P;
            double var = ss / (double)P - mean * mean;
            if (var < 0) var = 0;
            const float rstd = (float)(1.0 / sqrt(var + (double)eps));
            const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
            const float scale = g * rstd;
            st_dev(affine + c, scale);
            st_dev(affine + C + c, b - (float)mean * scale);
            saved[c] = (float)mean;
            saved[C + c] = rstd;
            if (moving_mean) moving_mean[c] = momentum * moving_mean[c] + (1.f - momentum) * (float)mean;
            if (moving_var) {
                const double unb = P > 1 ? var * (double)P / (double)(P - 1) : var;
                moving_var[c] = momentum * moving_var[c] + (1.f - momentum) * (float)unb;
            }
        }
    }
    grid_barrier(sync, (unsigned)G);
    const int CQ = C / V;
    const long long total = P * CQ, stride = (long long)G * 256;
    const float sl = act == 2 ? 0.3f : 0.f;
    const bool hoist = (stride % CQ) == 0;
    float sc[V], sh[V];
    auto consts = [&](int cc) {
#pragma unroll
        for (int k = 0; k < V; ++k) { sc[k] = ld_dev(affine + cc + k); sh[k] = ld_dev(affine + C + cc + k); }
    };
    const long long i0 = (long long)blockIdx.x * 256 + tid;
    if (hoist) consts((int)(i0 % CQ) * V);
    const long long pstep = stride / CQ;
    long long ph = i0 / CQ;
    const int cch = (int)(i0 % CQ) * V;
    for (long long i = i0; i < total; i += stride, ph += pstep) {
        const long long p = hoist ? ph : i / CQ;
        const int cc = hoist ? cch : (int)(i - p * CQ) * V;
        if (!hoist) consts(cc);
        float r[V];
        ldvec(x + (size_t)p * ldx + cc, r);
#pragma unroll
        for (int k = 0; k < V; ++k) r[k] = r[k] * sc[k] + sh[k];
        if (addend) {
            float ad[V];
            ldvec(addend + (size_t)p * ldadd + cc, ad);
#pragma unroll
            for (int k = 0; k < V; ++k) r[k] += ad[k];
        }
        if (act) {
#pragma unroll
            for (int k = 0; k < V; ++k) r[k] = r[k] > 0.f ? r[k] : sl * r[k];
        }
        stvec(y + (size_t)p * ldy + cc, r);
    }
    grid_finish(sync, 2u * (unsigned)G);
}

// workgroups of a fused launch: as many as the device holds at once (so that every one of them is resident when the first reaches
// a barrier on an idle device), at most 1024, at least one vector row per thread
template <typename K>
int resident_cap(K kernel) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) return 0;
    (void)per_cu;                  // one workgroup per CU: every additional resident workgroup is another poller of the arrival counter
    return cus > 1024 ? 1024 : cus;
}

}  // namespace

bool bn_fused_applies(long long P, int C, int elem_bytes) {
    if (!unetrir_cfg().bn_fused) return false;
    const int V = 16 / elem_bytes;
    if (P <= 0 || C <= 0 || C % V || C / V > 256) return false;
    return (double)P * C * elem_bytes <= 64.0 * 1024 * 1024;          // the tensors the second phase re-reads from cache
}

size_t bn_fused_ws_bytes(int C) { return (size_t)1024 * C * 2 * sizeof(double) + (size_t)2 * C * sizeof(float); }

template <typename T>
static int launch_bwd_t(const void* da, int ldda, const void* x, int ldx, const void* msk, int ldm, long long P, int C, const float* affine,
                        const float* saved, int act, void* dx, int lddx, void* g2, int ldg2, const void* g2add, int ldg2a, float* dgamma,
                        float* dbeta, void* ws, size_t ws_bytes, hipStream_t s) {
    static const int cap = resident_cap(bn_bwd_fused_kernel<T>);
    unsigned* sync = sync_slot(s);
    if (cap < 8 || !sync || ws_bytes < bn_fused_ws_bytes(C)) return BN_FUSED_NOT_TAKEN;
    constexpr int V = Vec<T>::N;
    long long want = (P * (C / V) + 255) / 256;                       // one vector per thread at least
    int G = (int)(want < cap ? want : cap);
    if (G < 1) G = 1;
    double* part = (double*)ws;
    float* coef = (float*)(part + (size_t)1024 * C * 2);
    hipLaunchKernelGGL(bn_bwd_fused_kernel<T>, dim3(G), dim3(256), 0, s, (const T*)da, ldda, (const T*)x, ldx, (const T*)msk, ldm, P, C, affine,
                       saved, act, (T*)dx, lddx, (T*)g2, ldg2, (const T*)g2add, ldg2a, dgamma, dbeta, part, coef, sync + UNETRIR_SYNC_TILES);
    return (int)hipGetLastError();
}

int launch_bn_bwd_fused(int bf16, const void* da, int ldda, const void* x, int ldx, const void* msk, int ldm, long long P, int C,
                        const float* affine, const float* saved, int act, void* dx, int lddx, void* g2, int ldg2, const void* g2add,
                        int ldg2a, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, hipStream_t s) {
    return bf16 ? launch_bwd_t<__bf16>(da, ldda, x, ldx, msk, ldm, P, C, affine, saved, act, dx, lddx, g2, ldg2, g2add, ldg2a, dgamma, dbeta, ws, ws_bytes, s)
                : launch_bwd_t<float>(da, ldda, x, ldx, msk, ldm, P, C, affine, saved, act, dx, lddx, g2, ldg2, g2add, ldg2a, dgamma, dbeta, ws, ws_bytes, s);
}

template <typename T>
static int launch_fwd_t(const float* colstat, long long rows, const void* x, int ldx, long long P, int C, const float* gamma, const float* beta,
                        float eps, float momentum, float* mm, float* mv, float* affine, float* saved, int act, const void* addend, int ldadd,
                        void* y, int ldy, hipStream_t s) {
    static const int cap = resident_cap(bn_fwd_fused_kernel<T>);
    unsigned* sync = sync_slot(s);
    if (cap < 8 || !sync || rows > 0x7fffffffLL) return BN_FUSED_NOT_TAKEN;
    constexpr int V = Vec<T>::N;
    long long want = (P * (C / V) + 255) / 256;
    int G = (int)(want < cap ? want : cap);
    if (G < 1) G = 1;
    hipLaunchKernelGGL(bn_fwd_fused_kernel<T>, dim3(G), dim3(256), 0, s, colstat, (int)rows, (const T*)x, ldx, P, C, gamma, beta, eps, momentum, mm,
                       mv, affine, saved, act, (const T*)addend, ldadd, (T*)y, ldy, sync + UNETRIR_SYNC_TILES + 2);
    return (int)hipGetLastError();
}

int launch_bn_fwd_fused(int bf16, const float* colstat, long long rows, const void* x, int ldx, long long P, int C, const float* gamma,
                        const float* beta, float eps, float momentum, float* mm, float* mv, float* affine, float* saved, int act,
                        const void* addend, int ldadd, void* y, int ldy, hipStream_t s) {
    return bf16 ? launch_fwd_t<__bf16>(colstat, rows, x, ldx, P, C, gamma, beta, eps, momentum, mm, mv, affine, saved, act, addend, ldadd, y, ldy, s)
                : launch_fwd_t<float>(colstat, rows, x, ldx, P, C, gamma, beta, eps, momentum, mm, mv, affine, saved, act, addend, ldadd, y, ldy, s);
}
