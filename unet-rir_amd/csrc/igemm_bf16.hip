// igemm_bf16.hip - bf16-storage / fp32-accumulate variants of the implicit-GEMM convolution kernels (gfx950).
//
// Same tap-table formulation as igemm.hip; activations and the weight work copies are bf16, accumulation is fp32 in
// v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate), bias / BatchNorm parameters / weight gradients stay fp32.
//   forward-type kernel : A = activations [pixel][k], B = weights [n][k], both K-contiguous -> one ds_read_b128 per
//                         operand per MFMA (lane l holds k = 8*(l>>5) .. +7 of row l&31).
//   3x3 weight gradient : K = pixels, which is the STRIDED dimension of NHWC tiles; the operands come out of LDS
//                         through ds_read_b64_tr_b16 (hardware 4x16 transpose), two reads per operand.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define BM 128
#define BKH 64            // bf16 k-values per LDS stage (128 B per row, as the fp32 kernel)
#define LDH 72            // padded row length in bf16 elements (144 B = 9 x 16 B)

__device__ __forceinline__ int xcd_remap_h(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// ------------------------------------------------------------------------------------------------
// forward-type kernel (Conv2D fwd, Conv2DTranspose fwd, every data gradient)
// ------------------------------------------------------------------------------------------------
template <int BN_, bool UNIFORM>
__device__ __forceinline__ void igemm_fwd_bf16_body(const IgemmArgsH& a, const int block_id, const int n_blocks) {
    constexpr int NSUB = BN_ / 64;
    constexpr int NB = BN_ / 32;
    __shared__ __attribute__((aligned(16))) __bf16 smem_h[(BM + BN_) * LDH];     // A tile, B tile; reused by the epilogue
    __bf16* As = smem_h;
    __bf16* Bs = smem_h + BM * LDH;
    __shared__ uint32_t s_tap[UNETRIR_MAX_TAPS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    if (tid < UNETRIR_MAX_TAPS) s_tap[tid] = a.g.tap[tid];

    const int ntN = (a.g.N + BN_ - 1) / BN_;
    const int id = xcd_remap_h(block_id, n_blocks);
    const int mt = id / ntN, nt = id - mt * ntN;
    const long long M = (long long)a.g.B * a.g.PH * a.g.PW;
    const long long m0 = (long long)mt * BM;
    const int n0 = nt * BN_;

    const int C = a.g.C, ntaps = a.g.ntaps;
    const int IH = a.g.IH, IW = a.g.IW, ldi = a.g.ldi;
    const int ldw = a.g.wtaps * C;
    const int nch = (ntaps * C + BKH - 1) / BKH;

    const int quad = tid & 7, lrow = tid >> 3;          // 8 threads x 8 bf16 = one 64-wide row; 32 rows per pass
    int kt = UNIFORM ? 0 : (quad * 8) / C;
    int kc = UNIFORM ? 0 : (quad * 8) % C;
    __syncthreads();

    const __bf16* a_ptr[4];
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
    const __bf16* b_ptr[NB];
    bool b_ok[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + lrow + 32 * j;
        b_ok[j] = n < a.g.N;
        b_ptr[j] = a.w + (size_t)(b_ok[j] ? n : 0) * ldw;
    }

    uint4 ra[4], rb[NB];
    auto load_stage = [&]() {
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
        for (int j = 0; j < 4; ++j) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (kok && ((a_mask[j] >> t) & 1ull)) v = *reinterpret_cast<const uint4*>(a_ptr[j] + aoff);
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (kok && b_ok[j]) v = *reinterpret_cast<const uint4*>(b_ptr[j] + boff);
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
    const int koff = (lane >> 5) * 8;

    for (int ch = 0; ch < nch; ++ch) {
        if (ch) __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4*>(&As[(lrow + 32 * j) * LDH + quad * 8]) = ra[j];
#pragma unroll
        for (int j = 0; j < NB; ++j) *reinterpret_cast<uint4*>(&Bs[(lrow + 32 * j) * LDH + quad * 8]) = rb[j];
        __syncthreads();
        if (ch + 1 < nch) {
            kc += BKH;
            while (kc >= C) { kc -= C; ++kt; }
            load_stage();
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 fa[2], fb[NSUB];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(&As[(arow + 32 * i) * LDH + kk * 16 + koff]);
#pragma unroll
            for (int j = 0; j < NSUB; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(&Bs[(brow + 32 * j) * LDH + kk * 16 + koff]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NSUB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();      // every wave is done with the A/B tiles: the epilogue stages through the same LDS

    // ---- epilogue through LDS.  The weight fragment is the MFMA A operand, so acc[i][j] holds
    // D[n = 32j + (r&3) + 8(r>>2) + 4h][pixel = 32i + (lane&31)]: a lane owns 4 consecutive channels per register quad.
    // (1) + bias, pack 4 channels -> ds_write_b64 into this wave's [64 px][BN/2 ch] staging tile; (2) read back 16-byte
    // channel runs of one pixel, add the optional addend, store 16 B per lane (2-byte stores straight from the MFMA layout
    // cost a third of the kernel).
    constexpr int WN = BN_ / 2;
    constexpr int SROW = WN + 8;                          // staging row stride in bf16 elements (16-byte pad)
    __bf16* stage = smem_h + wave * (64 * SROW);
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
            for (int i = 0; i < 2; ++i) {
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
    for (int ps = 0; ps < 64 / PPP; ++ps) {
        const int prow = ps * PPP + pl;
        const long long p = m0 + wm * 64 + prow;
        if (p >= M || n >= a.g.N) continue;
        long long opix;
        if (simple) {
            opix = p;
        } else {
            const int nimg = (int)(p / plane);
            const int rem = (int)(p - (long long)nimg * plane);
            const int py = rem / a.g.PW, px = rem - py * a.g.PW;
            const int oy = py * a.g.SO + a.g.ooy, ox = px * a.g.SO + a.g.oox;
            if (oy >= a.g.OH || ox >= a.g.OW) {           // not stored: not part of the column statistics either
                if (a.colstat != nullptr) *reinterpret_cast<uint4*>(stage + prow * SROW + cq * 8) = make_uint4(0u, 0u, 0u, 0u);
                continue;
            }
            opix = ((long long)nimg * a.g.OH + oy) * a.g.OW + ox;
        }
        bf16x8 v = *reinterpret_cast<const bf16x8*>(stage + prow * SROW + cq * 8);
        if (n + 7 < a.g.N) {
            if (a.addend != nullptr) {
                const bf16x8 ad = *reinterpret_cast<const bf16x8*>(a.addend + opix * a.ldadd + n);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (__bf16)((float)v[e] + (float)ad[e]);
                if (a.colstat != nullptr) *reinterpret_cast<bf16x8*>(stage + prow * SROW + cq * 8) = v;     // statistics of what is stored
            }
            *reinterpret_cast<bf16x8*>(a.out + opix * a.g.ldo + n) = v;
        } else {          // ragged channel tail (N not a multiple of 8 never happens for activations; kept for safety)
            for (int e = 0; e < 8 && n + e < a.g.N; ++e) {
                float f = (float)v[e];
                if (a.addend != nullptr) f += (float)a.addend[opix * a.ldadd + n + e];
                a.out[opix * a.g.ldo + n + e] = (__bf16)f;
            }
        }
    }
    if (a.colstat != nullptr) {
        __syncthreads();          // the store loop may have put addend sums back into the staging tiles
        // fused column statistics (BatchNormalization batch statistics without re-reading the tensor): per-channel (sum, sum of
        // squares) of the bf16 values this 128-pixel tile is about to store, in a fixed order: each wave sums the 64 pixel rows of
        // its staging tile (WN = 32: two lanes per channel, 32 rows each, combined by one cross-lane add), the two waves that
        // share a channel range are added through LDS.
        __shared__ float s_cs[4][64][2];
        constexpr int LPC = 64 / WN, RPL = 64 / LPC;
        const int ch = lane % WN, half = lane / WN;
        float cs = 0.f, css = 0.f;
#pragma unroll 8
        for (int r = 0; r < RPL; ++r) {
            const int prow = half * RPL + r;
            if (m0 + wm * 64 + prow < M) { const float v = (float)stage[prow * SROW + ch]; cs += v; css += v * v; }
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

template <int BN_, bool UNIFORM>
__global__ __launch_bounds__(256) void igemm_fwd_bf16_kernel(const IgemmArgsH a) {
    igemm_fwd_bf16_body<BN_, UNIFORM>(a, blockIdx.x, gridDim.x);
}

// four launches of the same shape in one grid (blockIdx.y): the output-parity classes of a stride-2 transposed conv
template <int BN_, bool UNIFORM>
__global__ __launch_bounds__(256) void igemm_fwd_bf16_kernel4(const IgemmArgsH4 a4) {
    igemm_fwd_bf16_body<BN_, UNIFORM>(a4.a[blockIdx.y], blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// 3x3 weight gradient, bf16 operands, fp32 partial slabs (see wgrad3x3.hip for the patch scheme)
// ------------------------------------------------------------------------------------------------
#define TPW 8

// KS = 3: the 3x3 layers.  KS = 1: the 1x1 layers of the residual graphs (dl_models/res_ae.py:455-512) - the same patch scheme
// with one tap and no halo (pad_t = pad_l = 0).
template <int SI, int TPH, int PADV = 1, int KS = 3>
__global__ __launch_bounds__(256, 2) void wgrad3x3_bf16_kernel(const Wgrad3ArgsH a) {
    constexpr int NT = KS * KS;
    constexpr int XH = (TPH - 1) * SI + KS, XW = (TPW - 1) * SI + KS;
    constexpr int XN = XH * XW * 8;                  // 16-byte slots of the x patch (8 per pixel: 64 channels)
    constexpr int XJ = (XN + 255) / 256;
    constexpr int DN = TPH * TPW * 8;                // 16-byte slots of the dy patch
    constexpr int DJ = (DN + 255) / 256;
    // Row strides (elements) chosen for the transposed reads: a 16-lane group of ds_read_b64_tr_b16 touches 4 pixel rows x
    // 64 B (two groups share a 32-lane conflict domain), so 4 consecutive rows - SI rows apart in the x patch - must start
    // 64 B apart modulo the 256-B bank row: 192 B for stride-1 rows, 160 B for the stride-2 x patch (144 B is 2-way).
    constexpr int LDD = PADV ? 96 : 72, LDX = PADV ? (SI == 1 ? 96 : 80) : 72;
    __shared__ __attribute__((aligned(16))) __bf16 Xs[XH * XW * LDX];
    __shared__ __attribute__((aligned(16))) __bf16 Ds[TPH * TPW * LDD];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;

    const int ntC = (a.C + 63) / 64;
    const int rt = blockIdx.x / ntC, ct = blockIdx.x - rt * ntC;
    const int n0 = rt * 64, c0 = ct * 64;
    const int per_img = a.npy * a.npx;
    const int G = a.B * per_img;
    const int g0 = blockIdx.y * a.patches_per_split;
    int g1 = g0 + a.patches_per_split;
    if (g1 > G) g1 = G;

    uint4 rx[XJ], rd[DJ];
    const int q8 = tid & 7;
    const bool cok = (c0 + q8 * 8) < a.C, nok = (n0 + q8 * 8) < a.N;

    auto load_patch = [&](int g) {
        const int img = g / per_img;
        const int rem = g - img * per_img;
        const int pyi = rem / a.npx, pxi = rem - pyi * a.npx;
        const int py0 = pyi * TPH, px0 = pxi * TPW;
        const int iy0 = py0 * SI - a.pad_t, ix0 = px0 * SI - a.pad_l;
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            const int i = tid + 256 * j;
            const int pp = i >> 3;
            const int pr = pp / XW, pc = pp - pr * XW;
            const int iy = iy0 + pr, ix = ix0 + pc;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (i < XN && cok && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW)
                v = *reinterpret_cast<const uint4*>(a.x + ((size_t)((long long)img * a.IH + iy) * a.IW + ix) * a.ldx + c0 + q8 * 8);
            rx[j] = v;
        }
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            const int i = tid + 256 * j;
            const int pix = i >> 3;
            const int oy = py0 + (pix >> 3), ox = px0 + (pix & 7);
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (i < DN && nok && oy < a.OH && ox < a.OW)
                v = *reinterpret_cast<const uint4*>(a.dy + ((size_t)((long long)img * a.OH + oy) * a.OW + ox) * a.lddy + n0 + q8 * 8);
            rd[j] = v;
        }
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ds_read_b64_tr_b16: inside each 16-lane group lane 4q+p supplies the address of LDS-matrix row q (a pixel),
    // columns 4p..4p+3 (channels); lane i of the group receives column i of the 4 rows.  Operand lane l wants
    // channel (l&31) and pixels 8*(l>>5)+j: groups 0/1 cover channels 0-15/16-31 of the low k half, 2/3 the high half.
    const int grp = lane >> 4, li = lane & 15;
    const int tq = li >> 2, tp = li & 3;
    const int chan = (grp & 1) * 16 + tp * 4;
    // pixel (inside a 16-pixel K step = 2 patch rows of 8) supplied by this lane for read rd: row h, col 4*rd + tq
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int dlane = (h * 8 + tq) * LDD + wr * 32 + chan;                       // + (s*16 + 4*rd) * LDD
    const int xlane = ((h * SI) * XW + tq * SI) * LDX + wc * 32 + chan;          // + ((2s*SI + kh) * XW + 4*rd*SI + kw) * LDX

    if (g0 < g1) load_patch(g0);
    for (int g = g0; g < g1; ++g) {
        if (g != g0) __syncthreads();
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            const int i = tid + 256 * j;
            if (i < XN) *reinterpret_cast<uint4*>(&Xs[(i >> 3) * LDX + q8 * 8]) = rx[j];
        }
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            const int i = tid + 256 * j;
            if (i < DN) *reinterpret_cast<uint4*>(&Ds[(i >> 3) * LDD + q8 * 8]) = rd[j];
        }
        __syncthreads();
        if (g + 1 < g1) load_patch(g + 1);
#pragma unroll
        for (int s = 0; s < TPH / 2; ++s) {
            bf16x8 fa;
            {
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(Ds + dlane + (s * 16) * LDD));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(Ds + dlane + (s * 16 + 4) * LDD));
                fa = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int kh = t / KS, kw = t - kh * KS;
                const int off0 = ((2 * s * SI + kh) * XW + kw) * LDX;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(Xs + xlane + off0));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(Xs + xlane + off0 + 4 * SI * LDX));
                const bf16x8 fb = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[t], 0, 0, 0);
            }
        }
    }

    float* part = a.part + (size_t)blockIdx.y * a.N * NT * a.C;
    const int c = c0 + wc * 32 + (lane & 31);
    if (c < a.C) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (n < a.N) part[((size_t)n * NT + t) * a.C + c] = acc[t][r];
            }
    }
}

// ------------------------------------------------------------------------------------------------
// weight work copies: fp32 master [N][T][C] -> bf16 [N][T][Cp] (same orientation, channel pad) and bf16 [C][T][Np]
// ------------------------------------------------------------------------------------------------
__global__ void cast_weight_kernel(const float* __restrict__ w, __bf16* __restrict__ o, int N, int T, int C, int Cp) {
    const size_t total = (size_t)N * T * Cp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cp);
        const size_t nt = i / Cp;
        o[i] = (c < C) ? (__bf16)w[nt * C + c] : (__bf16)0.f;
    }
}

__global__ void transpose_cast_weight_kernel(const float* __restrict__ w, __bf16* __restrict__ wt, int N, int T, int C, int Np) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int c0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, c = c0 + tx;
        tile[r][tx] = (n < N && c < C) ? w[((size_t)n * T + t) * C + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, n = n0 + tx;
        if (n < Np && c < C) wt[((size_t)c * T + t) * Np + n] = (__bf16)tile[tx][r];
    }
}

// All layers of a model in two launches: desc[l] describes one fp32 master [N][T][C] and its two bf16 work copies
// (blockIdx.y = layer).  Same element order as cast_weight_kernel / transpose_cast_weight_kernel.
// layers the fused kernel below takes: both copies wanted, no channel padding, 64 x 64 tiles fit exactly
__device__ __forceinline__ bool cast_fused_applies(const unetrir_cast_desc& d) {
    return d.same && d.transposed && d.C == d.Cp && d.N == d.Np && (d.C & 63) == 0 && (d.N & 63) == 0 &&
           ((((uintptr_t)d.w) & 15) | (((uintptr_t)d.same) & 7) | (((uintptr_t)d.transposed) & 7)) == 0;
}

// Both work copies from ONE read of the master: a 64 (n) x 64 (c) tile of tap t is loaded with 16-byte accesses, rounded,
// stored as it lies ([N][T][C]) and, through a padded LDS tile, transposed ([C][T][N]); every global store row is 128 bytes.
__global__ __launch_bounds__(256) void cast_both_batched_kernel(const unetrir_cast_desc* __restrict__ desc) {
    __shared__ __bf16 tile[64][66];                       // 132-byte rows: the column gathers below hit 16 banks
    const unetrir_cast_desc d = desc[blockIdx.y];
    if (!cast_fused_applies(d)) return;
    __bf16* __restrict__ same = (__bf16*)d.same;
    __bf16* __restrict__ wt = (__bf16*)d.transposed;
    __bf16* __restrict__ pk = d.T == 9 ? (__bf16*)d.packed_s2 : nullptr;      // third copy in conv3x3d's DMA order (include/unetrir.h)
    const int ntx = d.C >> 6, nty = d.N >> 6;
    const int ntiles = ntx * nty * d.T;
    const int lr = threadIdx.x >> 4, lc = (threadIdx.x & 15) * 4;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int t = tl / (ntx * nty);
        const int rem = tl - t * (ntx * nty);
        const int c0 = (rem % ntx) << 6, n0 = (rem / ntx) << 6;
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int r = ps * 16 + lr;
            const size_t off = ((size_t)(n0 + r) * d.T + t) * d.C + c0 + lc;
            const float4 v = *reinterpret_cast<const float4*>(d.w + off);
            bf16x4 h;
            h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
            *reinterpret_cast<bf16x4*>(same + off) = h;
            if (pk) {
                const int n = n0 + r, c = c0 + lc;
                const int m = n & 31, rho = (m & 16) | ((m & 8) >> 1) | ((m & 4) << 1) | (m & 3);       // row of the MFMA block that holds channel m
                const size_t po = ((((size_t)(n >> 7) * (d.C >> 4) + (c >> 4)) * 36 + t * 4 + ((n >> 5) & 3)) * 64 + rho + 32 * ((c >> 3) & 1)) * 8 + (c & 7);
                *reinterpret_cast<bf16x4*>(pk + po) = h;
            }
            tile[r][lc + 0] = h[0]; tile[r][lc + 1] = h[1]; tile[r][lc + 2] = h[2]; tile[r][lc + 3] = h[3];
        }
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int c = ps * 16 + lr;
            bf16x4 h;
            h[0] = tile[lc + 0][c]; h[1] = tile[lc + 1][c]; h[2] = tile[lc + 2][c]; h[3] = tile[lc + 3][c];
            *reinterpret_cast<bf16x4*>(wt + ((size_t)(c0 + c) * d.T + t) * d.N + n0 + lc) = h;
        }
    }
}

__global__ __launch_bounds__(256) void cast_weights_batched_kernel(const unetrir_cast_desc* __restrict__ desc) {
    const unetrir_cast_desc d = desc[blockIdx.y];
    if (!d.same || cast_fused_applies(d)) return;
    __bf16* o = (__bf16*)d.same;
    const size_t total = (size_t)d.N * d.T * d.Cp;
    if (d.packed_s2 && d.T == 9 && (d.C & 15) == 0) {     // packed copy for conv3x3d where the fused kernel does not run
        __bf16* pk = (__bf16*)d.packed_s2;
        const size_t tot = (size_t)d.N * 9 * d.C;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
            const int c = (int)(i % d.C);
            const size_t nt_ = i / d.C;
            const int t = (int)(nt_ % 9), n = (int)(nt_ / 9);
            const int m = n & 31, rho = (m & 16) | ((m & 8) >> 1) | ((m & 4) << 1) | (m & 3);
            const size_t po = ((((size_t)(n >> 7) * (d.C >> 4) + (c >> 4)) * 36 + t * 4 + ((n >> 5) & 3)) * 64 + rho + 32 * ((c >> 3) & 1)) * 8 + (c & 7);
            pk[po] = (__bf16)d.w[i];
        }
    }
    if (d.C == d.Cp && (total & 7) == 0 && (((uintptr_t)d.w | (uintptr_t)o) & 15) == 0) {     // flat copy, 8 elements per thread
        const size_t n8 = total >> 3;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
            const float4 a = *reinterpret_cast<const float4*>(d.w + i * 8), b = *reinterpret_cast<const float4*>(d.w + i * 8 + 4);
            bf16x8 v;
            v[0] = (__bf16)a.x; v[1] = (__bf16)a.y; v[2] = (__bf16)a.z; v[3] = (__bf16)a.w;
            v[4] = (__bf16)b.x; v[5] = (__bf16)b.y; v[6] = (__bf16)b.z; v[7] = (__bf16)b.w;
            *reinterpret_cast<bf16x8*>(o + i * 8) = v;
        }
        return;
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % d.Cp);
        const size_t nt = i / d.Cp;
        o[i] = (c < d.C) ? (__bf16)d.w[nt * d.C + c] : (__bf16)0.f;
    }
}

__global__ __launch_bounds__(256) void transpose_cast_weights_batched_kernel(const unetrir_cast_desc* __restrict__ desc) {
    __shared__ float tile[32][33];
    const unetrir_cast_desc d = desc[blockIdx.y];
    if (!d.transposed || cast_fused_applies(d)) return;
    __bf16* wt = (__bf16*)d.transposed;
    const int ntx = (d.C + 31) / 32, nty = (d.Np + 31) / 32;
    const int ntiles = ntx * nty * d.T;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int t = tl / (ntx * nty);
        const int rem = tl - t * (ntx * nty);
        const int c0 = (rem % ntx) * 32, n0 = (rem / ntx) * 32;
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const int n = n0 + r, c = c0 + tx;
            tile[r][tx] = (n < d.N && c < d.C) ? d.w[((size_t)n * d.T + t) * d.C + c] : 0.f;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const int c = c0 + r, n = n0 + tx;
            if (n < d.Np && c < d.C) wt[((size_t)c * d.T + t) * d.Np + n] = (__bf16)tile[tx][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
int launch_igemm_fwd_bf16(const IgemmArgsH& a, hipStream_t s) {
    const long long M = (long long)a.g.B * a.g.PH * a.g.PW;
    if (M <= 0 || a.g.N <= 0) return 0;
    if (igemm_bf16_tile_m(M, a.g.N, 1)) return launch_igemm2_fwd_bf16(&a, 1, s);       // small problems: igemm2_bf16.hip
    const long long mt = (M + BM - 1) / BM;
    const bool uniform = (a.g.C % BKH) == 0;
    if (a.g.N > 64) {
        const long long nwg = mt * ((a.g.N + 127) / 128);
        if (uniform) hipLaunchKernelGGL((igemm_fwd_bf16_kernel<128, true>), dim3((unsigned)nwg), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((igemm_fwd_bf16_kernel<128, false>), dim3((unsigned)nwg), dim3(256), 0, s, a);
    } else {
        if (uniform) hipLaunchKernelGGL((igemm_fwd_bf16_kernel<64, true>), dim3((unsigned)mt), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((igemm_fwd_bf16_kernel<64, false>), dim3((unsigned)mt), dim3(256), 0, s, a);
    }
    return (int)hipGetLastError();
}

int launch_igemm_fwd_bf16_x4(const IgemmArgsH* a, hipStream_t s) {
    const long long M = (long long)a[0].g.B * a[0].g.PH * a[0].g.PW;
    if (M <= 0 || a[0].g.N <= 0) return 0;
    if (igemm_bf16_tile_m(M, a[0].g.N, 4)) return launch_igemm2_fwd_bf16(a, 4, s);
    const long long mt = (M + BM - 1) / BM;
    const bool uniform = (a[0].g.C % BKH) == 0;
    IgemmArgsH4 a4;
    for (int i = 0; i < 4; ++i) a4.a[i] = a[i];
    if (a[0].g.N > 64) {
        const dim3 grid((unsigned)(mt * ((a[0].g.N + 127) / 128)), 4);
        if (uniform) hipLaunchKernelGGL((igemm_fwd_bf16_kernel4<128, true>), grid, dim3(256), 0, s, a4);
        else hipLaunchKernelGGL((igemm_fwd_bf16_kernel4<128, false>), grid, dim3(256), 0, s, a4);
    } else {
        const dim3 grid((unsigned)mt, 4);
        if (uniform) hipLaunchKernelGGL((igemm_fwd_bf16_kernel4<64, true>), grid, dim3(256), 0, s, a4);
        else hipLaunchKernelGGL((igemm_fwd_bf16_kernel4<64, false>), grid, dim3(256), 0, s, a4);
    }
    return (int)hipGetLastError();
}

void wgrad3x3_plan_tph(int TPH, int B, int OH, int OW, int N, int C, int* nsplit, int* per_split, int* npy, int* npx);
// bf16 patch heights: one barrier pair and one global round trip are amortised over TPH*8 pixels of MFMA work
#define TPH_S1 8
#define TPH_S2 4

int launch_wgrad3x3_bf16(Wgrad3ArgsH a, int stride, float* dw, float reg, const float* w, void* ws, size_t ws_bytes, hipStream_t s) {
    if (stride == 1) {       // patch rows reused from registers: LDS-DMA staged (wgrad3x3g.hip), else register staged (wgrad3x3r.hip)
        int err = launch_wgrad3x3g_bf16(a, dw, reg, w, ws, ws_bytes, s);
        if (err != WGRAD3X3R_NOT_TAKEN) return err;
        err = launch_wgrad3x3r_bf16(a, dw, reg, w, ws, ws_bytes, s);
        if (err != WGRAD3X3R_NOT_TAKEN) return err;
    } else {                 // stride 2: LDS-DMA kernel with the column-de-interleaved x patch (wgrad3x3d.hip)
        const int err = launch_wgrad3x3d_bf16(a, dw, reg, w, ws, ws_bytes, s);
        if (err != WGRAD3X3R_NOT_TAKEN) return err;
    }
    int ns, per;
    wgrad3x3_plan_tph(stride == 1 ? TPH_S1 : TPH_S2, a.B, a.OH, a.OW, a.N, a.C, &ns, &per, &a.npy, &a.npx);
    const size_t nout = (size_t)a.N * 9 * a.C;
    const bool direct = (ns == 1 && reg == 0.f);
    if (!direct && ws_bytes < (size_t)ns * nout * sizeof(float)) return UNETRIR_EINVAL;
    a.part = direct ? dw : (float*)ws;
    a.patches_per_split = per;
    const unsigned tiles = (unsigned)(((a.N + 63) / 64) * ((a.C + 63) / 64));
    if (stride == 1) hipLaunchKernelGGL((wgrad3x3_bf16_kernel<1, TPH_S1>), dim3(tiles, ns), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((wgrad3x3_bf16_kernel<2, TPH_S2>), dim3(tiles, ns), dim3(256), 0, s, a);
    int err = (int)hipGetLastError();
    if (err || direct) return err;
    return launch_splitk_reduce((const float*)ws, ns, nout, dw, reg, w, s);
}

// 1x1 weight gradient (stride 1 or 2): 16 x 8 pixel patches (one tap of MFMA work per 16-pixel K step, so taller patches)
#define TPH_1X1 16
#define TPH_1X1_S2 8          // stride 2: the x patch is (2 TPH - 1) x 15 pixels
size_t wgrad1x1_bf16_ws_bytes(int B, int OH, int OW, int N, int C) {
    size_t m = 0;
    for (int tph : {TPH_1X1, TPH_1X1_S2}) {
        int ns, per, npy, npx;
        wgrad3x3_plan_tph(tph, B, OH, OW, N, C, &ns, &per, &npy, &npx);
        const size_t b = (size_t)ns * N * C * sizeof(float);
        if (b > m) m = b;
    }
    return m;
}

int launch_wgrad1x1_bf16(Wgrad3ArgsH a, int stride, float* dw, float reg, const float* w, void* ws, size_t ws_bytes, hipStream_t s) {
    int ns, per;
    wgrad3x3_plan_tph(stride == 1 ? TPH_1X1 : TPH_1X1_S2, a.B, a.OH, a.OW, a.N, a.C, &ns, &per, &a.npy, &a.npx);
    const size_t nout = (size_t)a.N * a.C;
    const bool direct = (ns == 1 && reg == 0.f);
    if (!direct && ws_bytes < (size_t)ns * nout * sizeof(float)) return UNETRIR_EINVAL;
    a.part = direct ? dw : (float*)ws;
    a.patches_per_split = per;
    const unsigned tiles = (unsigned)(((a.N + 63) / 64) * ((a.C + 63) / 64));
    if (stride == 1) hipLaunchKernelGGL((wgrad3x3_bf16_kernel<1, TPH_1X1, 1, 1>), dim3(tiles, ns), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((wgrad3x3_bf16_kernel<2, TPH_1X1_S2, 1, 1>), dim3(tiles, ns), dim3(256), 0, s, a);
    const int err = (int)hipGetLastError();
    if (err || direct) return err;
    return launch_splitk_reduce((const float*)ws, ns, nout, dw, reg, w, s);
}

int launch_cast_weight(const float* w, void* o, int N, int T, int C, int Cp, hipStream_t s) {
    const size_t total = (size_t)N * T * Cp;
    unsigned nb = (unsigned)((total + 255) / 256);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(cast_weight_kernel, dim3(nb), dim3(256), 0, s, w, (__bf16*)o, N, T, C, Cp);
    return (int)hipGetLastError();
}

int launch_transpose_cast_weight(const float* w, void* wt, int N, int T, int C, int Np, hipStream_t s) {
    dim3 grid((C + 31) / 32, (Np + 31) / 32, T);
    hipLaunchKernelGGL(transpose_cast_weight_kernel, grid, dim3(256), 0, s, w, (__bf16*)wt, N, T, C, Np);
    return (int)hipGetLastError();
}

int launch_cast_weights_batched(const unetrir_cast_desc* desc_dev, int n_layers, hipStream_t s) {
    hipLaunchKernelGGL(cast_both_batched_kernel, dim3(256, n_layers), dim3(256), 0, s, desc_dev);
    hipLaunchKernelGGL(cast_weights_batched_kernel, dim3(256, n_layers), dim3(256), 0, s, desc_dev);
    hipLaunchKernelGGL(transpose_cast_weights_batched_kernel, dim3(256, n_layers), dim3(256), 0, s, desc_dev);
    return (int)hipGetLastError();
}
