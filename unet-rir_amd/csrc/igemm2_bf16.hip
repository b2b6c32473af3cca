// igemm2_bf16.hip - the tap-table implicit-GEMM forward kernel for SMALL problems (gfx950, bf16 storage, fp32 accumulate).
//
// igemm_bf16.hip serves every layer no patch-staged kernel takes: small grids (the reference's own 36 x 40 / 18 x 20 / 9 x 10
// levels, main_training.py:27), 6 x 6 kernels (the constructor default, dl_models/u_net.py:40-45), stride-2 layers at 16 x 16.  Its
// 128 x 128 tiles with ONE K chunk in flight suit large grids; on a 36 x 40 level there are 360 workgroups for 256 CUs, each
// alone on its CU, and every one of its 18 K chunks pays a full global-memory round trip: 46 us for 13.6 GFLOP (0.3 PFLOP/s).
// Same arithmetic here (v_mfma_f32_32x32x16_bf16 over the same K order: identical bits), re-shaped for few pixels:
//   * 64-pixel tiles (BM = 64) when 128-pixel tiles would not even give every CU one workgroup: four times the workgroups, 37 KB
//     of LDS each, so three to four share a CU and cover each other's memory latency;
//   * TWO K chunks in flight per workgroup: the registers of chunk c + 1 go to the second LDS buffer while chunk c is computed,
//     and are re-loaded for chunk c + 3 at once - one barrier per chunk instead of two, loads issued two chunks ahead;
//   * 32-bit pixel arithmetic in the prologue (the 64-bit divisions of the general kernel cost ~1 us of a 10 us launch);
//   * the same epilogue: bias, optional addend, 16-byte stores through an LDS staging tile, fused column statistics.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define BKH 64
#define LDH 72

namespace {

__device__ __forceinline__ int xcd_remap2(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

template <int BM_, int BN_, bool UNIFORM>
__device__ __forceinline__ void igemm2_body(const IgemmArgsH& a, const int block_id, const int n_blocks) {
    constexpr int MI = BM_ / 64;                          // 32-pixel blocks per wave (waves 2 x 2: BM_/2 pixels x BN_/2 channels each)
    constexpr int NSUB = BN_ / 64;
    constexpr int NA = BM_ / 32, NB = BN_ / 32;           // loader passes (32 rows per pass)
    constexpr int STAGE = (BM_ + BN_) * LDH;
    __shared__ __attribute__((aligned(16))) __bf16 smem_h[2 * STAGE];     // two (A tile, B tile) buffers; reused by the epilogue
    __shared__ uint32_t s_tap[UNETRIR_MAX_TAPS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    if (tid < UNETRIR_MAX_TAPS) s_tap[tid] = a.g.tap[tid];

    const int ntN = (a.g.N + BN_ - 1) / BN_;
    const int id = xcd_remap2(block_id, n_blocks);
    const int mt = id / ntN, nt = id - mt * ntN;
    const unsigned plane = (unsigned)a.g.PH * (unsigned)a.g.PW;
    const unsigned M = (unsigned)a.g.B * plane;           // < 2^31 (checked by the launcher)
    const unsigned m0 = (unsigned)mt * BM_;
    const int n0 = nt * BN_;

    const int C = a.g.C, ntaps = a.g.ntaps;
    const int IH = a.g.IH, IW = a.g.IW, ldi = a.g.ldi;
    const int ldw = a.g.wtaps * C;
    const int nch = (ntaps * C + BKH - 1) / BKH;

    const int quad = tid & 7, lrow = tid >> 3;            // 8 threads x 8 bf16 = one 64-wide row; 32 rows per pass
    int kt = UNIFORM ? 0 : (quad * 8) / C;
    int kc = UNIFORM ? 0 : (quad * 8) % C;
    __syncthreads();

    const __bf16* a_ptr[NA];
    unsigned long long a_mask[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const unsigned p = m0 + lrow + 32 * j;
        a_ptr[j] = a.in;
        a_mask[j] = 0ull;
        if (p < M) {
            const unsigned n = p / plane, rem = p - n * plane;
            const unsigned py = rem / (unsigned)a.g.PW, px = rem - py * (unsigned)a.g.PW;
            const int by = (int)py * a.g.SI, bx = (int)px * a.g.SI;
            a_ptr[j] = a.in + ((size_t)(n * (unsigned)IH + (unsigned)by) * IW + bx) * ldi;
            unsigned long long m = 0ull;
            for (int t = 0; t < ntaps; ++t) {
                const uint32_t e = s_tap[t];
                const int iy = by + (int)(int8_t)(e & 0xff), ix = bx + (int)(int8_t)((e >> 8) & 0xff);
                if ((unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW) m |= 1ull << t;
            }
            a_mask[j] = m;
        }
    }
    const __bf16* b_ptr[NB];
    bool b_ok[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + lrow + 32 * j;
        b_ok[j] = n < a.g.N;
        b_ptr[j] = a.w + (size_t)(b_ok[j] ? n : 0) * ldw;
    }

    uint4 ra[2][NA], rb[2][NB];
    // loads the NEXT chunk in K order (chunks are requested strictly in order 0, 1, 2, ...) into register set `set`
    auto load_stage = [&](auto set_c) {
        constexpr int set = decltype(set_c)::value;
        int t = kt, c = kc;
        if (UNIFORM) { t = __builtin_amdgcn_readfirstlane(t); c = __builtin_amdgcn_readfirstlane(c); }
        const bool kok = t < ntaps;
        uint32_t e = kok ? s_tap[t] : 0u;
        if (UNIFORM) e = __builtin_amdgcn_readfirstlane(e);
        const int dy = (int)(int8_t)(e & 0xff), dx = (int)(int8_t)((e >> 8) & 0xff);
        const int wi = (int)((e >> 16) & 0xff);
        const int aoff = (dy * IW + dx) * ldi + c + (UNIFORM ? quad * 8 : 0);
        const int boff = wi * C + c + (UNIFORM ? quad * 8 : 0);
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (kok && ((a_mask[j] >> t) & 1ull)) v = *reinterpret_cast<const uint4*>(a_ptr[j] + aoff);
            ra[set][j] = v;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (kok && b_ok[j]) v = *reinterpret_cast<const uint4*>(b_ptr[j] + boff);
            rb[set][j] = v;
        }
        kc += BKH;
        while (kc >= C) { kc -= C; ++kt; }
    };
    auto to_lds = [&](auto set_c, auto buf_c) {
        constexpr int set = decltype(set_c)::value, buf = decltype(buf_c)::value;
        __bf16* As = smem_h + buf * STAGE;
        __bf16* Bs = As + BM_ * LDH;
#pragma unroll
        for (int j = 0; j < NA; ++j) *reinterpret_cast<uint4*>(&As[(lrow + 32 * j) * LDH + quad * 8]) = ra[set][j];
#pragma unroll
        for (int j = 0; j < NB; ++j) *reinterpret_cast<uint4*>(&Bs[(lrow + 32 * j) * LDH + quad * 8]) = rb[set][j];
    };

    f32x16 acc[MI][NSUB];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NSUB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int arow = wm * (BM_ / 2) + (lane & 31), brow = wn * (BN_ / 2) + (lane & 31);
    const int koff = (lane >> 5) * 8;

    // prologue: chunks 0 and 1 requested; chunk 0 into LDS buffer 0, its registers re-loaded for chunk 2
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    load_stage(I0{});
    if (nch > 1) load_stage(I1{});
    to_lds(I0{}, I0{});
    if (nch > 2) load_stage(I0{});
    __syncthreads();
    // one K chunk; CUR (compile time: the register sets must not be indexed at run time) = parity of the chunk = its LDS buffer
    auto step = [&](int ch, auto cur_c) {
        constexpr int CUR = decltype(cur_c)::value;
        if (ch + 1 < nch) {                               // chunk ch + 1: registers -> the other buffer; then its registers take chunk ch + 3
            to_lds(std::integral_constant<int, CUR ^ 1>{}, std::integral_constant<int, CUR ^ 1>{});
            if (ch + 3 < nch) load_stage(std::integral_constant<int, CUR ^ 1>{});
        }
        const __bf16* As = smem_h + CUR * STAGE;
        const __bf16* Bs = As + BM_ * LDH;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 fa[MI], fb[NSUB];
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(&As[(arow + 32 * i) * LDH + kk * 16 + koff]);
#pragma unroll
            for (int j = 0; j < NSUB; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(&Bs[(brow + 32 * j) * LDH + kk * 16 + koff]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NSUB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();                                  // buffer CUR^1 is complete, buffer CUR is free
    };
    for (int ch = 0; ch < nch; ch += 2) {
        step(ch, std::integral_constant<int, 0>{});
        if (ch + 1 < nch) step(ch + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue through LDS (as igemm_bf16.hip): acc[i][j] holds D[n = 32j + (r&3) + 8(r>>2) + 4h][pixel = 32i + (lane&31)]
    constexpr int WN = BN_ / 2, WM = BM_ / 2;
    constexpr int SROW = WN + 8;
    __bf16* stage = smem_h + wave * (WM * SROW);
    const int hq = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int nl = 32 * j + 8 * qd + 4 * hq;
            const int n = n0 + wn * WN + nl;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (n + e < a.g.N) bv[e] = a.bias[n + e];
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)(acc[i][j][4 * qd + e] + bv[e]);
                *reinterpret_cast<bf16x4*>(stage + (32 * i + l31) * SROW + nl) = o;
            }
        }
    }
    __syncthreads();
    const bool simple = (a.g.SO == 1 && a.g.ooy == 0 && a.g.oox == 0 && a.g.OH == a.g.PH && a.g.OW == a.g.PW);
    constexpr int LPP = WN / 8;                           // lanes per pixel (8 channels = 16 B each)
    constexpr int PPP = 64 / LPP;                         // pixels per pass
    const int cq = lane % LPP, pl = lane / LPP;
    const int n = n0 + wn * WN + cq * 8;
#pragma unroll
    for (int ps = 0; ps < WM / PPP; ++ps) {
        const int prow = ps * PPP + pl;
        const unsigned p = m0 + wm * WM + prow;
        if (p >= M || n >= a.g.N) continue;
        size_t opix;
        if (simple) {
            opix = p;
        } else {
            const unsigned nimg = p / plane, rem = p - nimg * plane;
            const unsigned py = rem / (unsigned)a.g.PW, px = rem - py * (unsigned)a.g.PW;
            const int oy = (int)py * a.g.SO + a.g.ooy, ox = (int)px * a.g.SO + a.g.oox;
            if (oy >= a.g.OH || ox >= a.g.OW) {           // not stored: not part of the column statistics either
                if (a.colstat != nullptr) *reinterpret_cast<uint4*>(stage + prow * SROW + cq * 8) = make_uint4(0u, 0u, 0u, 0u);
                continue;
            }
            opix = ((size_t)nimg * a.g.OH + oy) * a.g.OW + ox;
        }
        bf16x8 v = *reinterpret_cast<const bf16x8*>(stage + prow * SROW + cq * 8);
        if (n + 7 < a.g.N) {
            if (a.addend != nullptr) {
                const bf16x8 ad = *reinterpret_cast<const bf16x8*>(a.addend + opix * a.ldadd + n);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (__bf16)((float)v[e] + (float)ad[e]);
                if (a.colstat != nullptr) *reinterpret_cast<bf16x8*>(stage + prow * SROW + cq * 8) = v;
            }
            *reinterpret_cast<bf16x8*>(a.out + opix * a.g.ldo + n) = v;
        } else {
            for (int e = 0; e < 8 && n + e < a.g.N; ++e) {
                float f = (float)v[e];
                if (a.addend != nullptr) f += (float)a.addend[opix * a.ldadd + n + e];
                a.out[opix * a.g.ldo + n + e] = (__bf16)f;
            }
        }
    }
    if (a.colstat != nullptr) {       // one row of (sum, sum of squares) per pixel tile, fixed order (igemm_bf16.hip)
        __syncthreads();
        __shared__ float s_cs[4][64][2];
        constexpr int LPC = 64 / WN, RPL = WM / LPC;
        const int ch = lane % WN, half = lane / WN;
        float cs = 0.f, css = 0.f;
#pragma unroll 8
        for (int r = 0; r < RPL; ++r) {
            const int prow = half * RPL + r;
            if (m0 + wm * WM + prow < M) { const float v = (float)stage[prow * SROW + ch]; cs += v; css += v * v; }
        }
        if (LPC == 2) { cs += __shfl_xor(cs, 32); css += __shfl_xor(css, 32); }
        if (lane < WN) { s_cs[wave][ch][0] = cs; s_cs[wave][ch][1] = css; }
        __syncthreads();
        if (tid < BN_) {
            const int wn_ = tid / WN, c = tid % WN, nn = n0 + wn_ * WN + c;
            if (nn < a.g.N) {
                float* row = a.colstat + ((size_t)mt * a.g.N + nn) * 2;
                row[0] = s_cs[wn_][c][0] + s_cs[2 + wn_][c][0];
                row[1] = s_cs[wn_][c][1] + s_cs[2 + wn_][c][1];
            }
        }
    }
}

template <int BM_, int BN_, bool UNIFORM>
__global__ __launch_bounds__(256) void igemm2_fwd_bf16_kernel(const IgemmArgsH a) {
    igemm2_body<BM_, BN_, UNIFORM>(a, blockIdx.x, gridDim.x);
}

struct IgemmArgsH4b { IgemmArgsH a[4]; };
template <int BM_, int BN_, bool UNIFORM>
__global__ __launch_bounds__(256) void igemm2_fwd_bf16_kernel4(const IgemmArgsH4b a4) {
    igemm2_body<BM_, BN_, UNIFORM>(a4.a[blockIdx.y], blockIdx.x, gridDim.x);
}

template <int BM_, int BN_>
int launch_t(const IgemmArgsH* a, int ncls, unsigned nwg, bool uniform, hipStream_t s) {
    if (ncls == 4) {
        IgemmArgsH4b a4;
        for (int i = 0; i < 4; ++i) a4.a[i] = a[i];
        if (uniform) hipLaunchKernelGGL((igemm2_fwd_bf16_kernel4<BM_, BN_, true>), dim3(nwg, 4), dim3(256), 0, s, a4);
        else hipLaunchKernelGGL((igemm2_fwd_bf16_kernel4<BM_, BN_, false>), dim3(nwg, 4), dim3(256), 0, s, a4);
    } else {
        if (uniform) hipLaunchKernelGGL((igemm2_fwd_bf16_kernel<BM_, BN_, true>), dim3(nwg), dim3(256), 0, s, a[0]);
        else hipLaunchKernelGGL((igemm2_fwd_bf16_kernel<BM_, BN_, false>), dim3(nwg), dim3(256), 0, s, a[0]);
    }
    return (int)hipGetLastError();
}

}  // namespace

// Pixel-tile height of the tap-table launch for an iteration grid of M pixels and N output channels, ncls launches sharing the
// grid; 0 = the general kernel (igemm_bf16.hip, 128-pixel tiles).  Measured at batch 32 (scripts/micro_igemm.py, round 3): below
// one 128 x 128 workgroup per CU the 64-pixel tiles win (256 -> 256 @ 18 x 20: 47 -> 38 us, 512 -> 512 @ 9 x 10: 79 -> 51,
// 256 -> 512 stride 2 @ 18 x 20: 44 -> 27); from 360 workgroups on (36 x 40 levels, 16 x 16 with 1024 channels) the general
// kernel's five resident workgroups per CU beat two chunks in flight (35 against 46 us).
int igemm_bf16_tile_m(long long M, int N, int ncls) {
    if (!unetrir_cfg().igemm2 || M <= 0 || M >= (1ll << 31)) return 0;
    const long long wg128 = ((M + 127) / 128) * ((N + 127) / 128) * ncls;
    return wg128 < 256 ? 64 : 0;
}

// a[0 .. ncls): 1 launch or the 4 output-parity classes of a stride-2 transposed layer (same shape, one grid)
int launch_igemm2_fwd_bf16(const IgemmArgsH* a, int ncls, hipStream_t s) {
    const long long M = (long long)a[0].g.B * a[0].g.PH * a[0].g.PW;
    if (M <= 0 || a[0].g.N <= 0) return 0;
    int bm = igemm_bf16_tile_m(M, a[0].g.N, ncls);
    if (bm == 0) bm = 128;
    const bool uniform = (a[0].g.C % BKH) == 0;
    const long long mt = (M + bm - 1) / bm;
    // channel tile: 128 when N > 64 and the 64-pixel x 128-channel tiles still fill the chip twice, else 64
    const bool bn128 = a[0].g.N > 64 && (bm == 128 || mt * ((a[0].g.N + 127) / 128) * ncls >= 2 * 256);
    const unsigned nwg = (unsigned)(mt * (bn128 ? (a[0].g.N + 127) / 128 : (a[0].g.N + 63) / 64));
    if (bm == 128) return bn128 ? launch_t<128, 128>(a, ncls, nwg, uniform, s) : launch_t<128, 64>(a, ncls, nwg, uniform, s);
    return bn128 ? launch_t<64, 128>(a, ncls, nwg, uniform, s) : launch_t<64, 64>(a, ncls, nwg, uniform, s);
}
