// igemm.hip - implicit-GEMM convolution kernels for gfx950 (MI355X), fp32 MFMA.
//
// One forward-type kernel covers Conv2D (stride 1|2), Conv2DTranspose (as 4 output-parity
// sub-convolutions) and every data gradient; one weight-gradient kernel covers every dW.
// Both are driven by a small tap table (TF padding='same' geometry is resolved on the host
// in api.hip), replace what TensorFlow dispatches to cuDNN for dl_models/u_net.py:269-276,
// :297-304, :366, :248, :262 and their tape.gradient counterparts (main_training.py:267).
//
// Math: v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, 64 FLOP/clk/SIMD).  A = activations
// (rows = pixels), B = weights (cols = output channels), so the accumulator has the output
// channel on the lane and the NHWC store is 128 contiguous bytes per (row, half-wave).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 128          // pixels per tile
#define BK 32           // k (tap, channel) values per LDS stage
#define LDS_LD 36       // padded row length (floats): 9 x 16 B, odd -> conflict-free ds_read_b128

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of tiles so
    // neighbouring pixel tiles (shared halo rows, shared weight panel) meet in one L2.
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// ------------------------------------------------------------------------------------------------
// forward-type kernel:  out[opix(p)][n] = bias[n] + addend + sum_t sum_c in[ipix(p,t)][c] * w[n][widx_t][c]
// ------------------------------------------------------------------------------------------------
template <int BN_, bool UNIFORM>
__global__ __launch_bounds__(256) void igemm_fwd_kernel(const IgemmArgs a) {
    constexpr int NSUB = BN_ / 64;        // 32-wide N sub-tiles per wave (waves are 2 x 2)
    constexpr int NB = BN_ / 32;          // B-tile rows loaded per thread
    __shared__ __attribute__((aligned(16))) float As[BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[BN_ * LDS_LD];
    __shared__ uint32_t s_tap[UNETRIR_MAX_TAPS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    if (tid < UNETRIR_MAX_TAPS) s_tap[tid] = a.g.tap[tid];

    const int ntN = (a.g.N + BN_ - 1) / BN_;
    const int nwg = gridDim.x;
    const int id = xcd_remap(blockIdx.x, nwg);
    const int mt = id / ntN, nt = id - mt * ntN;
    const long long M = (long long)a.g.B * a.g.PH * a.g.PW;
    const long long m0 = (long long)mt * BM;
    const int n0 = nt * BN_;

    const int C = a.g.C, ntaps = a.g.ntaps;
    const int IH = a.g.IH, IW = a.g.IW, ldi = a.g.ldi;
    const int ldw = a.g.wtaps * C;
    const int Ktot = ntaps * C;
    int nch = (Ktot + BK - 1) / BK;
    int ch0 = 0;
    if (a.ksplit > 1) {      // split-K (small pixel counts, e.g. the Dense layer): this block owns chunks [ch0, nch)
        const int per = (nch + a.ksplit - 1) / a.ksplit;
        ch0 = blockIdx.y * per;
        nch = min(nch, ch0 + per);
    }

    // loader mapping: 8 threads cover the 32 k-values of one row; 32 rows per pass
    const int quad = tid & 7, lrow = tid >> 3;
    // (tap, channel) of the k-quad being staged.  UNIFORM (C % 32 == 0): a stage never straddles a tap, so the
    // tap index and the channel base are wave-uniform scalars and only quad*4 is per lane.
    int kt = (ch0 * BK + (UNIFORM ? 0 : quad * 4)) / C;
    int kc = (ch0 * BK + (UNIFORM ? 0 : quad * 4)) % C;

    __syncthreads();   // s_tap visible

    // per staged row: pointer to its centre pixel (tap offset 0) and a bit mask of the taps that fall inside the image
    const float* a_ptr[4];
    unsigned long long a_mask[4];
    const int plane = a.g.PH * a.g.PW;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long p = m0 + lrow + 32 * j;
        a_ptr[j] = a.in;
        a_mask[j] = 0ull;
        if (p < M) {
            const int n = (int)(p / plane);
            const int rem = (int)(p - (long long)n * plane);
            const int py = rem / a.g.PW, px = rem - py * a.g.PW;
            const int by = py * a.g.SI, bx = px * a.g.SI;
            a_ptr[j] = a.in + ((long long)((long long)n * IH + by) * IW + bx) * ldi;
            unsigned long long m = 0ull;
            for (int t = 0; t < ntaps; ++t) {
                const uint32_t e = s_tap[t];
                const int iy = by + (int)(int8_t)(e & 0xff), ix = bx + (int)(int8_t)((e >> 8) & 0xff);
                if ((unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW) m |= 1ull << t;
            }
            a_mask[j] = m;
        }
    }
    const float* b_ptr[NB];
    bool b_ok[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + lrow + 32 * j;
        b_ok[j] = n < a.g.N;
        b_ptr[j] = a.w + (size_t)(b_ok[j] ? n : 0) * ldw;
    }

    float4 ra[4], rb[NB];

    auto load_stage = [&]() {
        int t = kt, c = kc;
        if (UNIFORM) { t = __builtin_amdgcn_readfirstlane(t); c = __builtin_amdgcn_readfirstlane(c); }
        const bool kok = t < ntaps;
        uint32_t e = kok ? s_tap[t] : 0u;
        if (UNIFORM) e = __builtin_amdgcn_readfirstlane(e);
        const int dy = (int)(int8_t)(e & 0xff), dx = (int)(int8_t)((e >> 8) & 0xff);
        const int wi = (int)((e >> 16) & 0xff);
        const int aoff = (dy * IW + dx) * ldi + c + (UNIFORM ? quad * 4 : 0);
        const int boff = wi * C + c + (UNIFORM ? quad * 4 : 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kok && ((a_mask[j] >> t) & 1ull)) v = *reinterpret_cast<const float4*>(a_ptr[j] + aoff);
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kok && b_ok[j]) v = *reinterpret_cast<const float4*>(b_ptr[j] + boff);
            rb[j] = v;
        }
    };

    f32x16 acc[2][NSUB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NSUB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_stage();
    const int arow = wm * 64 + (lane & 31), brow = wn * (BN_ / 2) + (lane & 31);
    const int koff = (lane >> 5) * 4;

    for (int ch = ch0; ch < nch; ++ch) {
        if (ch != ch0) __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(&As[(lrow + 32 * j) * LDS_LD + quad * 4]) = ra[j];
#pragma unroll
        for (int j = 0; j < NB; ++j) *reinterpret_cast<float4*>(&Bs[(lrow + 32 * j) * LDS_LD + quad * 4]) = rb[j];
        __syncthreads();
        if (ch + 1 < nch) {
            kc += BK;
            while (kc >= C) { kc -= C; ++kt; }
            load_stage();
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float4 fa[2], fb[NSUB];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const float4*>(&As[(arow + 32 * i) * LDS_LD + kk * 8 + koff]);
#pragma unroll
            for (int j = 0; j < NSUB; ++j) fb[j] = *reinterpret_cast<const float4*>(&Bs[(brow + 32 * j) * LDS_LD + kk * 8 + koff]);
            // k-major order: consecutive MFMAs hit different accumulators (dependent latency == issue interval)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NSUB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NSUB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NSUB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NSUB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
        }
    }

    if (a.ksplit > 1) {      // raw partial sums; bias / reduction happen in splitk_rows_reduce_kernel
        float* part = a.part + (size_t)blockIdx.y * M * a.g.N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long p = m0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (p >= M) continue;
#pragma unroll
                for (int j = 0; j < NSUB; ++j) {
                    const int n = n0 + wn * (BN_ / 2) + 32 * j + (lane & 31);
                    if (n < a.g.N) part[p * a.g.N + n] = acc[i][j][r];
                }
            }
        return;
    }
    // ---- epilogue: accumulator (col = lane&31 -> n, row = (r&3)+8*(r>>2)+4*(lane>>5) -> pixel)
    const bool simple = (a.g.SO == 1 && a.g.ooy == 0 && a.g.oox == 0 && a.g.OH == a.g.PH && a.g.OW == a.g.PW);
    float bias[NSUB];
    int ncol[NSUB];
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
        ncol[j] = n0 + wn * (BN_ / 2) + 32 * j + (lane & 31);
        bias[j] = (a.bias != nullptr && ncol[j] < a.g.N) ? a.bias[ncol[j]] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const long long p = m0 + row;
            if (p >= M) continue;
            long long opix;
            if (simple) {
                opix = p;
            } else {
                const int n = (int)(p / plane);
                const int rem = (int)(p - (long long)n * plane);
                const int py = rem / a.g.PW, px = rem - py * a.g.PW;
                const int oy = py * a.g.SO + a.g.ooy, ox = px * a.g.SO + a.g.oox;
                if (oy >= a.g.OH || ox >= a.g.OW) continue;
                opix = ((long long)n * a.g.OH + oy) * a.g.OW + ox;
            }
#pragma unroll
            for (int j = 0; j < NSUB; ++j) {
                if (ncol[j] < a.g.N) {
                    float v = acc[i][j][r] + bias[j];
                    if (a.addend != nullptr) v += a.addend[opix * a.ldadd + ncol[j]];
                    a.out[opix * a.g.ldo + ncol[j]] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// weight-gradient kernel: part[s][n][widx_t*C + c] = sum_{p in slice s} dy[p][n] * x[ipix(p,t)][c]
// GEMM rows = output channels n, cols = flattened (tap, channel), K = pixels (split over blockIdx.y)
// ------------------------------------------------------------------------------------------------
#define WG_COLS 128
#define WG_PIX 32
#define WG_LDX (WG_COLS + 4)

// operand loads in the storage type of the activations: fp32, or bf16 widened on the way in (the k x k weight gradients of bf16
// graphs that have no bf16 kernel of their own - kernels = 6, the reference's constructor default, dl_models/u_net.py:40-45 - run
// here: fp32 MFMA arithmetic on exactly the stored values)
typedef __bf16 wg_bf16x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ float4 wg_ld4(const void* base, size_t elem_off);
template <> __device__ __forceinline__ float4 wg_ld4<float>(const void* base, size_t elem_off) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem_off);
}
template <> __device__ __forceinline__ float4 wg_ld4<__bf16>(const void* base, size_t elem_off) {
    const wg_bf16x4 v = *reinterpret_cast<const wg_bf16x4*>(reinterpret_cast<const __bf16*>(base) + elem_off);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}

template <int BR_, typename T = float>   // rows (output channels) per tile: 64 or 128
__global__ __launch_bounds__(256) void igemm_wgrad_kernel(const WgradArgs a) {
    constexpr int MSUB = BR_ / 64;        // BR_=128: waves 2x2 of 64x64; BR_=64: waves 2x2 of 32x64
    constexpr int LDD = BR_ + 4;
    constexpr int DQ = BR_ / 4;           // dy quads per pixel row
    constexpr int DPASS = (WG_PIX * DQ) / 256;
    constexpr int DROWS = 256 / DQ;
    __shared__ __attribute__((aligned(16))) float Ds[WG_PIX * LDD];
    __shared__ __attribute__((aligned(16))) float Xs[WG_PIX * WG_LDX];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int C = a.g.C;
    const int Kc = a.g.ntaps * C;                 // valid columns
    const int ntC = (Kc + WG_COLS - 1) / WG_COLS;
    const int id = blockIdx.x;
    const int rt = id / ntC, ct = id - rt * ntC;  // column tile fastest: shares the dy panel in L2
    const int n0 = rt * BR_, j0 = ct * WG_COLS;
    const long long M = (long long)a.g.B * a.g.PH * a.g.PW;
    const long long chunk_per = a.chunks_per_split;
    const long long pk0 = (long long)blockIdx.y * chunk_per * WG_PIX;
    long long pk1 = pk0 + chunk_per * WG_PIX;
    if (pk1 > M) pk1 = M;

    // X loader: this thread always loads the same 4 columns -> fixed (tap, channel)
    const int xq = tid & 31, xr = tid >> 5;      // 32 quads per row, 8 rows per pass
    const int col = j0 + xq * 4;
    const bool colok = col < Kc;
    const int t = colok ? col / C : 0, c = colok ? col - (col / C) * C : 0;
    const uint32_t e = a.g.tap[t];
    const int dy = (int)(int8_t)(e & 0xff), dx = (int)(int8_t)((e >> 8) & 0xff);
    const int plane = a.g.PH * a.g.PW;
    const int IH = a.g.IH, IW = a.g.IW, SI = a.g.SI;

    const int dq = tid % DQ, dr = tid / DQ;
    const bool nok = (n0 + dq * 4) < a.g.N;

    float4 rx[4], rd[DPASS];
    auto load_stage = [&](long long pk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long p = pk + xr + 8 * j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (colok && p < pk1) {
                const int n = (int)(p / plane);
                const int rem = (int)(p - (long long)n * plane);
                const int py = rem / a.g.PW, px = rem - py * a.g.PW;
                const int iy = py * SI + dy, ix = px * SI + dx;
                if ((unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW)
                    v = wg_ld4<T>(a.x, ((size_t)((long long)n * IH + iy) * IW + ix) * a.g.ldi + c);
            }
            rx[j] = v;
        }
#pragma unroll
        for (int j = 0; j < DPASS; ++j) {
            const long long p = pk + dr + DROWS * j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (nok && p < pk1) v = wg_ld4<T>(a.dy, (size_t)p * a.lddy + n0 + dq * 4);
            rd[j] = v;
        }
    };

    f32x16 acc[MSUB][2];
#pragma unroll
    for (int i = 0; i < MSUB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int arow = wm * (BR_ / 2) + (lane & 31);   // output-channel row inside the tile
    const int bcol = wn * 64 + (lane & 31);
    const int kh = lane >> 5;

    if (pk0 < pk1) load_stage(pk0);
    for (long long pk = pk0; pk < pk1; pk += WG_PIX) {
        if (pk != pk0) __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(&Xs[(xr + 8 * j) * WG_LDX + xq * 4]) = rx[j];
#pragma unroll
        for (int j = 0; j < DPASS; ++j) *reinterpret_cast<float4*>(&Ds[(dr + DROWS * j) * LDD + dq * 4]) = rd[j];
        __syncthreads();
        if (pk + WG_PIX < pk1) load_stage(pk + WG_PIX);
#pragma unroll
        for (int s = 0; s < WG_PIX / 2; ++s) {
            float fa[MSUB], fb[2];
#pragma unroll
            for (int i = 0; i < MSUB; ++i) fa[i] = Ds[(2 * s + kh) * LDD + arow + 32 * i];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = Xs[(2 * s + kh) * WG_LDX + bcol + 32 * j];
#pragma unroll
            for (int i = 0; i < MSUB; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: rows (registers) = output channel, cols (lanes) = flattened (tap, channel)
    float* part = a.part + (size_t)blockIdx.y * a.g.N * ((size_t)a.g.wtaps * C);
    const size_t ldp = (size_t)a.g.wtaps * C;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cc = j0 + wn * 64 + 32 * j + (lane & 31);
        if (cc >= Kc) continue;
        const int tt = cc / C, c2 = cc - tt * C;
        const size_t ocol = (size_t)((a.g.tap[tt] >> 16) & 0xff) * C + c2;
#pragma unroll
        for (int i = 0; i < MSUB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm * (BR_ / 2) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (n < a.g.N) part[(size_t)n * ldp + ocol] = acc[i][j][r];
            }
    }
}

// out[i] = sum_s part[s][i] + reg * w[i].  Block = 64 float4 outputs (one full wave: 1 KB contiguous per load) x G split
// groups (waves): group g sums slabs g, g+G, ... in order with 8 independent loads in flight, then the G group sums are
// added in fixed order -> bit-reproducible.  G follows the split count so that no wave idles when there are few slabs.
template <int G>
__device__ __forceinline__ void splitk_reduce_body(float4 (*red)[64], const unsigned block, const float* __restrict__ part, int nsplit, size_t n,
                                                   float* __restrict__ out, float reg, const float* __restrict__ w) {
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    if (grp >= G) return;                          // the batched kernel runs 8 waves whatever G is (a finished wave does not count at the barrier)
    const size_t i4 = ((size_t)block * 64 + lane) * 4;
    const bool full = i4 + 3 < n;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (full) {
        const float* p = part + i4;
        int k = grp;
        for (; k + 7 * G < nsplit; k += 8 * G) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(k + u * G) * n);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; k < nsplit; k += G) {
            const float4 v = *reinterpret_cast<const float4*>(p + (size_t)k * n);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    } else if (i4 < n) {
        float* sp = &s.x;
        for (int k = grp; k < nsplit; k += G)
            for (size_t i = i4; i < n; ++i) sp[i - i4] += part[(size_t)k * n + i];
    }
    if (G > 1) {
        red[grp][lane] = s;
        __syncthreads();
        if (grp != 0) return;
#pragma unroll
        for (int g = 1; g < G; ++g) {
            const float4 v = red[g][lane];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    if (i4 >= n) return;
    if (full) {
        if (reg != 0.f) {
            const float4 v = *reinterpret_cast<const float4*>(w + i4);
            s.x += reg * v.x; s.y += reg * v.y; s.z += reg * v.z; s.w += reg * v.w;
        }
        *reinterpret_cast<float4*>(out + i4) = s;
    } else {
        const float* sp = &s.x;
        for (size_t i = i4; i < n; ++i) out[i] = sp[i - i4] + (reg != 0.f ? reg * w[i] : 0.f);
    }
}
template <int G>
__global__ __launch_bounds__(64 * G) void splitk_reduce_kernel(const float* __restrict__ part, int nsplit, size_t n,
                                                               float* __restrict__ out, float reg, const float* __restrict__ w) {
    __shared__ float4 red[G][64];
    splitk_reduce_body<G>(red, blockIdx.x, part, nsplit, n, out, reg, w);
}

// [N][T][C] -> [C][T][N] through a 32x33 LDS tile per tap
__global__ void transpose_weight_kernel(const float* __restrict__ w, float* __restrict__ wt, int N, int T, int C) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int c0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 8 rows per pass
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, c = c0 + tx;
        tile[r][tx] = (n < N && c < C) ? w[((size_t)n * T + t) * C + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, n = n0 + tx;
        if (n < N && c < C) wt[((size_t)c * T + t) * N + n] = tile[tx][r];
    }
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
int launch_igemm_fwd(const IgemmArgs& a, hipStream_t s) {
    const long long M = (long long)a.g.B * a.g.PH * a.g.PW;
    if (M <= 0 || a.g.N <= 0) return 0;
    const long long mt = (M + BM - 1) / BM;
    const bool uniform = (a.g.C % BK) == 0;
    const unsigned ks = a.ksplit > 1 ? (unsigned)a.ksplit : 1u;
    if (a.g.N > 64) {
        const long long nwg = mt * ((a.g.N + 127) / 128);
        if (uniform) hipLaunchKernelGGL((igemm_fwd_kernel<128, true>), dim3((unsigned)nwg, ks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((igemm_fwd_kernel<128, false>), dim3((unsigned)nwg, ks), dim3(256), 0, s, a);
    } else {
        if (uniform) hipLaunchKernelGGL((igemm_fwd_kernel<64, true>), dim3((unsigned)mt, ks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((igemm_fwd_kernel<64, false>), dim3((unsigned)mt, ks), dim3(256), 0, s, a);
    }
    return (int)hipGetLastError();
}

// y[p][n] = bias[n] + sum_s part[s][p][n]   (fixed order)
__global__ void splitk_rows_reduce_kernel(const float* __restrict__ part, int nsplit, long long M, int N,
                                          const float* __restrict__ bias, float* __restrict__ y, int ldy) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * N) return;
    const long long p = i / N;
    const int n = (int)(i - p * N);
    float s = bias ? bias[n] : 0.f;
    const size_t mn = (size_t)M * N;
    int k = 0;
    for (; k + 7 < nsplit; k += 8) {             // 8 independent loads in flight, added in slab order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(k + u) * mn + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < nsplit; ++k) s += part[(size_t)k * mn + i];
    y[p * ldy + n] = s;
}

int launch_splitk_rows_reduce(const float* part, int nsplit, long long M, int N, const float* bias, float* y, int ldy, hipStream_t s) {
    const long long tot = M * N;
    hipLaunchKernelGGL(splitk_rows_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, part, nsplit, M, N, bias, y, ldy);
    return (int)hipGetLastError();
}

// K slices of the small-batch Dense kernel: with only B <= 128 rows there are ceil(N / 128) (N <= 64: one) output tiles, so K is
// split until ~512 workgroups stream the weight matrix; a slice keeps at least 256 k-values.  (16 for the U-Net's 8192 -> 4096
// layer as before; the residual auto-encoder's 66 560 -> 32 latent layer had ONE tile and ran on 16 workgroups: 184 us.)
static int dense_ksplit(int K, int N) {
    const int tiles = N > 64 ? (N + 127) / 128 : 1;
    int ks = (512 + tiles - 1) / tiles;
    const int maxk = K / 256 > 0 ? K / 256 : 1;
    if (ks > maxk) ks = maxk;
    if (ks > 128) ks = 128;
    if (ks < 1) ks = 1;
    return ks;
}
size_t dense_fwd_ws_bytes(int B, int K, int N) { return (size_t)dense_ksplit(K, N) * B * N * sizeof(float); }

// Dense(N) on a small batch: y[B][N] = x[B][K] . w[N][K]^T + bias.  The weight matrix is streamed once; with only B rows
// there are N/128 output tiles, so the K dimension is split 16 ways to put >= 2 workgroups on every CU.
int launch_dense_fwd(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int B, int K, int N,
                     void* ws, size_t ws_bytes, hipStream_t s) {
    if (ws_bytes < dense_fwd_ws_bytes(B, K, N)) return UNETRIR_EINVAL;
    IgemmArgs a{};
    a.g.B = B; a.g.PH = 1; a.g.PW = 1; a.g.IH = 1; a.g.IW = 1; a.g.C = K; a.g.ldi = ldx;
    a.g.OH = 1; a.g.OW = 1; a.g.N = N; a.g.ldo = ldy; a.g.SI = 1; a.g.SO = 1;
    a.g.ntaps = 1; a.g.wtaps = 1; a.g.tap[0] = 0;
    a.in = x; a.w = w; a.out = y;
    const int ks = dense_ksplit(K, N);
    if (ks == 1) {            // enough output tiles on their own: the kernel's ordinary epilogue writes y (+ bias)
        a.bias = bias; a.ksplit = 0; a.part = nullptr;
        return launch_igemm_fwd(a, s);
    }
    a.ksplit = ks; a.part = (float*)ws;
    int err = launch_igemm_fwd(a, s);
    if (err) return err;
    const long long tot = (long long)B * N;
    hipLaunchKernelGGL(splitk_rows_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, (const float*)ws,
                       ks, (long long)B, N, bias, y, ldy);
    return (int)hipGetLastError();
}

int wgrad_plan(const IgemmGeom& g, int* nsplit, long long* chunks_per_split) {
    const long long M = (long long)g.B * g.PH * g.PW;
    const long long nchunks = (M + WG_PIX - 1) / WG_PIX;
    const int br = g.N > 64 ? 128 : 64;
    const long long tiles = (long long)((g.N + br - 1) / br) * ((g.ntaps * g.C + WG_COLS - 1) / WG_COLS);
    long long want = (1024 + tiles - 1) / tiles;          // aim at >= 1024 workgroups
    long long maxs = (nchunks + 7) / 8;                   // at least 8 chunks (256 pixels) per slice
    if (maxs < 1) maxs = 1;
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    if (want > 1024) want = 1024;
    long long per = (nchunks + want - 1) / want;
    long long ns = (nchunks + per - 1) / per;
    *nsplit = (int)ns;
    *chunks_per_split = per;
    return 0;
}

int launch_igemm_wgrad(WgradArgs a, float* dw, float reg, const float* w, void* ws, size_t ws_bytes, hipStream_t s, int bf16_operands) {
    int nsplit; long long per;
    wgrad_plan(a.g, &nsplit, &per);
    const size_t nout = (size_t)a.g.N * a.g.wtaps * a.g.C;
    const bool direct = (nsplit == 1 && reg == 0.f);
    if (!direct && ws_bytes < (size_t)nsplit * nout * sizeof(float)) return UNETRIR_EINVAL;
    a.part = direct ? dw : (float*)ws;
    a.chunks_per_split = per;
    const int br = a.g.N > 64 ? 128 : 64;
    const unsigned tiles = (unsigned)(((a.g.N + br - 1) / br) * ((a.g.ntaps * a.g.C + WG_COLS - 1) / WG_COLS));
    if (bf16_operands) {      // a.x / a.dy point at bf16 tensors
        if (br == 128) hipLaunchKernelGGL((igemm_wgrad_kernel<128, __bf16>), dim3(tiles, nsplit), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((igemm_wgrad_kernel<64, __bf16>), dim3(tiles, nsplit), dim3(256), 0, s, a);
    } else if (br == 128) hipLaunchKernelGGL(igemm_wgrad_kernel<128>, dim3(tiles, nsplit), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(igemm_wgrad_kernel<64>, dim3(tiles, nsplit), dim3(256), 0, s, a);
    int err = (int)hipGetLastError();
    if (err) return err;
    if (!direct) {
        err = launch_splitk_reduce((const float*)ws, nsplit, nout, dw, reg, w, s);
    }
    return err;
}

// Few outputs, many slabs (the 1x1 and small 3x3 weight gradients of the residual graphs: 1 K .. 64 K floats from up to 512
// slabs): the kernel above would run a handful of workgroups that each walk 64 slabs one after the other.  Here a workgroup
// owns 16 float4 columns and 32 slab groups: group g sums slabs g, g + 32, ... (8 loads in flight), the 32 group sums are added
// in a fixed order.  16 lanes x 16 bytes = one 256-byte piece of a slab per group and step.
__device__ __forceinline__ void splitk_reduce_wide_body(float4 (*red)[16], const unsigned block, const float* __restrict__ part, int nsplit, size_t n,
                                                        float* __restrict__ out, float reg, const float* __restrict__ w) {
    const int lane = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const size_t i4 = ((size_t)block * 16 + lane) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i4 < n) {                                   // n % 4 == 0 (checked by the launcher)
        const float* p = part + i4;
        int k = grp;
        for (; k + 7 * 32 < nsplit; k += 8 * 32) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(k + u * 32) * n);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; k < nsplit; k += 32) {
            const float4 v = *reinterpret_cast<const float4*>(p + (size_t)k * n);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[grp][lane] = s;
    __syncthreads();
    if (grp != 0 || i4 >= n) return;
#pragma unroll
    for (int g = 1; g < 32; ++g) {
        const float4 v = red[g][lane];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (reg != 0.f) {
        const float4 v = *reinterpret_cast<const float4*>(w + i4);
        s.x += reg * v.x; s.y += reg * v.y; s.z += reg * v.z; s.w += reg * v.w;
    }
    *reinterpret_cast<float4*>(out + i4) = s;
}
__global__ __launch_bounds__(512) void splitk_reduce_wide_kernel(const float* __restrict__ part, int nsplit, size_t n,
                                                                 float* __restrict__ out, float reg, const float* __restrict__ w) {
    __shared__ float4 red[32][16];
    splitk_reduce_wide_body(red, blockIdx.x, part, nsplit, n, out, reg, w);
}

// ---- several reductions in ONE launch (round 4).  A weight gradient's fixed-order reduction is 5-20 us of which most is the launch
// itself (37.7 MB of slabs at 1.9 TB/s; 1-64 K outputs at their launch floor), and a train step has 23 (configs[1]) to 57 (configs[4])
// of them.  The weight-gradient entry points can leave their slabs in the caller's workspace and hand back a descriptor instead
// (unetrir_*_wgrad_partials_*); unetrir_splitk_reduce_batched then reduces up to 16 of them per launch.  Every output element is summed
// by the same code over the same slab order as in the single launch (same group count, wide or narrow form per reduction): bit-identical.
#define REDUCE_BATCH 16
struct ReduceBatchArgs {
    unetrir_reduce_desc d[REDUCE_BATCH];
    unsigned first_block[REDUCE_BATCH + 1];       // workgroup range of reduction i: [first_block[i], first_block[i + 1])
    unsigned char kind[REDUCE_BATCH];             // 0: the wide form; 8 / 4 / 2 / 1: the narrow form with that many slab groups
    int n;
};
__global__ __launch_bounds__(512) void splitk_reduce_batched_kernel(const ReduceBatchArgs a) {
    __shared__ float4 red[8 * 64];                 // [8][64] (narrow, 8 slab groups) or [32][16] (wide)
    int i = 0;
    while (i + 1 < a.n && blockIdx.x >= a.first_block[i + 1]) ++i;
    const unetrir_reduce_desc& d = a.d[i];
    const unsigned blk = blockIdx.x - a.first_block[i];
    float4 (*r64)[64] = reinterpret_cast<float4 (*)[64]>(red);
    switch (a.kind[i]) {                           // uniform per workgroup
        case 0: splitk_reduce_wide_body(reinterpret_cast<float4 (*)[16]>(red), blk, d.part, d.nsplit, d.n, d.out, d.reg, d.w); break;
        case 8: splitk_reduce_body<8>(r64, blk, d.part, d.nsplit, d.n, d.out, d.reg, d.w); break;
        case 4: splitk_reduce_body<4>(r64, blk, d.part, d.nsplit, d.n, d.out, d.reg, d.w); break;
        case 2: splitk_reduce_body<2>(r64, blk, d.part, d.nsplit, d.n, d.out, d.reg, d.w); break;
        default: splitk_reduce_body<1>(r64, blk, d.part, d.nsplit, d.n, d.out, d.reg, d.w); break;
    }
}

// A weight-gradient entry point called with a descriptor to fill (unetrir_*_wgrad_partials_*) runs its implementation with this
// pointer set: launch_splitk_reduce then records what it was asked to reduce instead of launching.  Thread-local: the ABI stays
// re-entrant across host threads.
thread_local unetrir_reduce_desc* t_reduce_sink = nullptr;
void set_reduce_sink(unetrir_reduce_desc* d) { t_reduce_sink = d; }

static inline bool reduce_is_wide(int nsplit, size_t n, const float* part, const float* out, const float* w) {
    const size_t n4 = (n + 3) / 4;
    return (n & 3) == 0 && nsplit >= 32 && (n4 + 63) / 64 < 128 && (((uintptr_t)part | (uintptr_t)out | (uintptr_t)w) & 15) == 0;
}

extern "C" int unetrir_splitk_reduce_batched(const unetrir_reduce_desc* desc, int n, unetrir_stream_t stream) {
    if (n < 0 || (n > 0 && !desc)) return UNETRIR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ReduceBatchArgs a;
    a.n = 0;
    a.first_block[0] = 0;
    auto flush = [&]() -> int {
        if (a.n == 0) return 0;
        if (a.n == 1) {          // nothing to batch: the single launch
            const unetrir_reduce_desc& d = a.d[0];
            a.n = 0;
            return launch_splitk_reduce(d.part, d.nsplit, d.n, d.out, d.reg, d.w, s);
        }
        hipLaunchKernelGGL(splitk_reduce_batched_kernel, dim3(a.first_block[a.n]), dim3(512), 0, s, a);
        a.n = 0;
        return (int)hipGetLastError();
    };
    for (int i = 0; i < n; ++i) {
        const unetrir_reduce_desc& d = desc[i];
        if (d.nsplit == 0) continue;                            // the weight gradient went straight into dw: nothing to reduce
        if (!d.part || !d.out || d.nsplit < 0 || d.n == 0 || (d.reg != 0.f && !d.w)) return UNETRIR_EINVAL;
        const bool wide = reduce_is_wide(d.nsplit, d.n, d.part, d.out, d.w);
        const size_t n4 = (d.n + 3) / 4;
        const size_t blocks = wide ? (n4 + 15) / 16 : (n4 + 63) / 64;
        if (blocks > 0x3fffffffu) return UNETRIR_EINVAL;
        if (a.n == REDUCE_BATCH || (size_t)a.first_block[a.n] + blocks > 0x7fffffffu) { const int e = flush(); if (e) return e; a.first_block[0] = 0; }
        a.d[a.n] = d;
        a.kind[a.n] = wide ? 0 : (d.nsplit >= 8 ? 8 : d.nsplit >= 4 ? 4 : d.nsplit >= 2 ? 2 : 1);      // as launch_splitk_reduce picks them
        a.first_block[a.n + 1] = a.first_block[a.n] + (unsigned)blocks;
        ++a.n;
    }
    return flush();
}

int launch_splitk_reduce(const float* part, int nsplit, size_t n, float* out, float reg, const float* w, hipStream_t s) {
    if (t_reduce_sink) {          // deferred: the caller reduces later, together with others (unetrir_splitk_reduce_batched)
        if (t_reduce_sink->nsplit != 0) return UNETRIR_EINVAL;          // one reduction per weight gradient
        t_reduce_sink->part = part; t_reduce_sink->nsplit = nsplit; t_reduce_sink->n = n; t_reduce_sink->out = out;
        t_reduce_sink->reg = reg; t_reduce_sink->w = w;
        return 0;
    }
    const size_t n4 = (n + 3) / 4;
    // (round 3: taking larger outputs too - up to 4096 narrow workgroups - moves single launches by +-8 us and the step by nothing)
    if (reduce_is_wide(nsplit, n, part, out, w)) {
        hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3((unsigned)((n4 + 15) / 16)), dim3(512), 0, s, part, nsplit, n, out, reg, w);
        return (int)hipGetLastError();
    }
    const dim3 grid((unsigned)((n4 + 63) / 64));
    if (nsplit >= 8) hipLaunchKernelGGL(splitk_reduce_kernel<8>, grid, dim3(512), 0, s, part, nsplit, n, out, reg, w);
    else if (nsplit >= 4) hipLaunchKernelGGL(splitk_reduce_kernel<4>, grid, dim3(256), 0, s, part, nsplit, n, out, reg, w);
    else if (nsplit >= 2) hipLaunchKernelGGL(splitk_reduce_kernel<2>, grid, dim3(128), 0, s, part, nsplit, n, out, reg, w);
    else hipLaunchKernelGGL(splitk_reduce_kernel<1>, grid, dim3(64), 0, s, part, nsplit, n, out, reg, w);
    return (int)hipGetLastError();
}

int launch_transpose_weight(const float* w, float* wt, int N, int T, int C, hipStream_t s) {
    dim3 grid((C + 31) / 32, (N + 31) / 32, T);
    hipLaunchKernelGGL(transpose_weight_kernel, grid, dim3(256), 0, s, w, wt, N, T, C);
    return (int)hipGetLastError();
}
