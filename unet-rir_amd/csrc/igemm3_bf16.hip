// igemm3_bf16.hip - the tap-table implicit-GEMM forward kernel with LDS-DMA staging (gfx950, bf16 storage, fp32 accumulate).
//
// What bounds the tap-table kernels on the layers no patch-staged kernel takes (the reference's own 36 x 40 / 18 x 20 / 9 x 10
// levels, main_training.py:27; 6 x 6 kernels, dl_models/u_net.py:40-45; the stride-2 pair at 16 x 16) is the rate at which a CU
// takes bytes in: a 128 x 128 x 64 K chunk moves 32 KB from L2 into the CU for 2.1 MFLOP - 64 flop per byte, no tap re-uses a
// staged pixel (that is what the patch kernels are for) - and a CU ingests ~12 bytes per cycle (MI355X_MICROARCH.md: 11-13 B/cyc/CU
// for register and LDS-DMA fills alike) = ~30 GB/s: 0.49 PFLOP/s for the chip at this tile shape, whatever the pipelining.  Measured
// here: the register-staged kernel 0.39-0.41 PFLOP/s on 128 -> 128 @ 36 x 40 (212 MB staged in 35 us = 6 TB/s); this file with a ring
// of 3 or 4 stages (96-128 KB: ONE workgroup per CU) 0.30 - three chunks in flight buy nothing, a lone workgroup loses the CU at
// every barrier and 360 tiles take two rounds; with TWO stages (64 KB at 128-pixel tiles, 48 KB at 64: two to three workgroups per
// CU) 0.44-0.47 = 90 % of the ingest bound, 5-17 % faster than the register-staged kernels on every shape tried
// (scripts/micro_igemm.py: 36 x 40 128 -> 128 34.9 -> 30.9 us, 512 -> 1024 s2 @ 32 x 32 102 -> 89 us, 6 x 6 128 -> 128 @ 64 x 64 226 -> 187 us).
// INSIDE THE TRAIN STEP the order reverses (activations and kernels come from HBM / Infinity Cache, not from an L2 the previous
// iteration of a benchmark loop left warm; the register-staged kernel keeps three to four 37 KB workgroups on a CU against two here):
// reference geometry 4.06-4.08 against 3.97-3.99 ms per step, configs[1] single stream 13.36 against 13.25 ms, side-stream schedule
// equal (scripts/time_refgeom.py, scripts/ab_switch.py igemm3).  The kernel therefore stays behind its switch (igemm3, DEFAULT 0): a
// measured refusal, and the record of where the tap-table formulation's ceiling is.
// Same arithmetic (v_mfma_f32_32x32x16_bf16, weights as the row operand, the same K order: IDENTICAL bits), other transport:
//   * K chunks go global -> LDS by DMA (buffer_load_dwordx4 ... lds, 1 KB per wave-instruction, per-lane gather addresses, taps
//     outside the image and rows past M / N as out-of-range offsets = zero fill): no registers on the way, no ds_write;
//   * S = 2 stages of (BM + 128) x 64 bf16: chunk c + 1 streams in while chunk c is multiplied; its DMA instructions are issued one
//     part behind the MFMAs of every K step (a DMA instruction costs 60-180 cycles of issue);
//   * ONE barrier per chunk, in front of it the wait for the chunk's DMAs;
//   * the 16-byte granules of a 128-byte K row are XOR-swizzled with (row & 7) on the DMA SOURCE side (the LDS side of a DMA is
//     lane-linear): fragment reads (ds_read_b128, row = lane & 31) are conflict-free; reads of K step kk + 1 are in flight while
//     the four MFMAs of step kk issue (inline asm: an LDS load the compiler can see would drain the DMA queue first);
//   * the epilogue of igemm_bf16.hip (bias, addend, 16-byte stores through an LDS staging tile, fused column statistics).
// Taken for C % 64 == 0, N > 64 (igemm3_applies); everything else stays on the register-staged kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

#define DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LGKM_WAIT(n) do { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

namespace {
constexpr int G3_BN = 128, G3_BK = 64;
constexpr uint32_t G3_OOB = 0xF0000000u;

__device__ __forceinline__ int xcd_remap3(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

template <int BM_, int S>
__device__ __forceinline__ void igemm3_body(const IgemmArgsH& a, const int block_id, const int n_blocks) {
    constexpr int MI = BM_ / 64;                          // 32-pixel blocks per wave (waves 2 x 2: BM_/2 pixels x 64 channels each)
    constexpr int NA = BM_ / 32;                          // A wave-instructions per wave and chunk (8 rows each)
    constexpr int A_BYTES = BM_ * 128, STAGE = A_BYTES + G3_BN * 128;
    constexpr int PER = NA + 4;                           // DMA instructions per wave and chunk
    __shared__ __attribute__((aligned(1024))) unsigned char smem[S * STAGE];
    __shared__ uint32_t s_tap[UNETRIR_MAX_TAPS];
    __shared__ float s_cs[4][64][2];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    if (tid < UNETRIR_MAX_TAPS) s_tap[tid] = a.g.tap[tid];

    const int ntN = (a.g.N + G3_BN - 1) / G3_BN;
    const int id = xcd_remap3(block_id, n_blocks);
    const int mt = id / ntN, nt = id - mt * ntN;
    const unsigned plane = (unsigned)a.g.PH * (unsigned)a.g.PW;
    const unsigned M = (unsigned)a.g.B * plane;           // < 2^31 (checked by the launcher)
    const unsigned m0 = (unsigned)mt * BM_;
    const int n0 = nt * G3_BN;
    const int C = a.g.C, ntaps = a.g.ntaps, IW = a.g.IW, IH = a.g.IH, ldi = a.g.ldi;
    const int ldw = a.g.wtaps * C;
    const int nch = ntaps * (C / G3_BK);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    const uint32_t tap_a = (uint32_t)(uintptr_t)(lptr_t)s_tap;
    __syncthreads();                                      // the tap table (the only LDS access the compiler sees before the epilogue)

    // ---- DMA sources.  Wave-instruction q of an operand fills LDS rows 8 q .. 8 q + 7 of the stage (128 B per row); lane l is row
    //      8 q + (l >> 3), LDS granule l & 7, and fetches SOURCE granule (l & 7) ^ (row & 7).  This wave issues q = NA wave + j (A)
    //      and q = 4 wave + j (B).
    const int lrow = lane >> 3, sg = (lane & 7) ^ lrow;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, (short)0,
        (int)((((size_t)a.g.B * IH * IW - 1) * ldi + C) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, (int)((size_t)a.g.N * ldw * 2), 0x00020000);
    uint32_t a_base[NA];
    unsigned long long a_mask[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const unsigned p = m0 + 8 * (NA * wave + j) + lrow;
        a_base[j] = 0u; a_mask[j] = 0ull;
        if (p < M) {
            const unsigned n = p / plane, rem = p - n * plane;
            const unsigned py = rem / (unsigned)a.g.PW, px = rem - py * (unsigned)a.g.PW;
            const int by = (int)py * a.g.SI, bx = (int)px * a.g.SI;
            a_base[j] = (uint32_t)((((size_t)n * IH + by) * IW + bx) * ldi * 2 + sg * 16);       // may wrap below zero with a tap offset: uint32 arithmetic
            unsigned long long m = 0ull;
            for (int t = 0; t < ntaps; ++t) {
                const uint32_t e = a.g.tap[t];
                const int iy = by + (int)(int8_t)(e & 0xff), ix = bx + (int)(int8_t)((e >> 8) & 0xff);
                if ((unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW) m |= 1ull << t;
            }
            a_mask[j] = m;
        }
    }
    uint32_t b_base[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + 8 * (4 * wave + j) + lrow;
        b_base[j] = n < a.g.N ? (uint32_t)((size_t)n * ldw * 2 + sg * 16) : G3_OOB;
    }
    int kt = 0, kc = 0;                                   // tap and channel offset of the next chunk to request
    // a chunk's PER DMA instructions are issued in four parts, one behind the MFMAs of every K step (a DMA instruction costs 60 - 180
    // cycles of issue: in one burst they would stand in front of the chunk's matrix work)
    uint32_t q_aoff = 0, q_boff = 0; int q_kt = 0; unsigned char* q_dst = smem;
    auto prepare = [&](int slot) {                        // tap and channel offset of the next chunk to request
        uint32_t e;
        asm volatile("ds_read_b32 %0, %1" : "=v"(e) : "v"(tap_a + (uint32_t)kt * 4u));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        e = (uint32_t)__builtin_amdgcn_readfirstlane((int)e);
        const int dy = (int)(int8_t)(e & 0xff), dx = (int)(int8_t)((e >> 8) & 0xff), wi = (int)((e >> 16) & 0xff);
        q_aoff = (uint32_t)(((dy * IW + dx) * ldi + kc) * 2);
        q_boff = (uint32_t)((wi * C + kc) * 2);
        q_dst = smem + slot * STAGE;
        q_kt = kt;
        kc += G3_BK;
        if (kc == C) { kc = 0; ++kt; }
    };
    auto issue_part = [&](auto part_c) {                  // part 0 .. 3: NA / 4 (or the A rows in parts 0, 1 when NA = 2) + one B instruction
        constexpr int part = decltype(part_c)::value;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            if (j != part) continue;
            const uint32_t off = ((a_mask[j] >> q_kt) & 1ull) ? a_base[j] + q_aoff : G3_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lptr_t)(q_dst + (NA * wave + j) * 1024), 16, off, 0, 0, 0);
        }
        {
            constexpr int j = part;
            const uint32_t off = b_base[j] != G3_OOB ? b_base[j] + q_boff : G3_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lptr_t)(q_dst + A_BYTES + (4 * wave + j) * 1024), 16, off, 0, 0, 0);
        }
    };
    auto issue = [&](int slot) {
        prepare(slot);
        issue_part(std::integral_constant<int, 0>{}); issue_part(std::integral_constant<int, 1>{});
        issue_part(std::integral_constant<int, 2>{}); issue_part(std::integral_constant<int, 3>{});
    };

    // ---- fragment read addresses: row (lane & 31) of the wave's block, K granule 2 kk + (lane >> 5), swizzled with row & 7 = lane & 7
    const uint32_t fa0 = lds0 + (uint32_t)((wm * (BM_ / 2) + (lane & 31)) * 128);
    const uint32_t fb0 = lds0 + A_BYTES + (uint32_t)((wn * 64 + (lane & 31)) * 128);
    uint32_t koff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = (uint32_t)(((2 * kk + (lane >> 5)) ^ (lane & 7)) * 16);

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: chunks 0 .. S-2 requested
#pragma unroll
    for (int c = 0; c < S - 1; ++c)
        if (c < nch) issue(c);

    int slot = 0;
    for (int ch = 0; ch < nch; ++ch) {
        // chunk ch has landed once this wave's DMAs for it have (everybody's, behind the barrier); the min(S - 2, nch - 1 - ch) younger
        // chunks stay in flight
        const int younger = nch - 1 - ch;
        if (S >= 4 && younger >= 2) { if (PER == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
        else if (S >= 3 && younger >= 1) { if (PER == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // ... and everybody is done with chunk ch - 1: its slot is free
        asm volatile("" ::: "memory");
        const bool more = ch + S - 1 < nch;
        if (more) prepare(slot == 0 ? S - 1 : slot - 1);
        const uint32_t fa = fa0 + (uint32_t)slot * STAGE, fb = fb0 + (uint32_t)slot * STAGE;
        u32x4 A[2][MI], B[2][2];
        DSR128(A[0][0], fa + koff[0], 0);
        if (MI == 2) DSR128(A[0][MI - 1], fa + koff[0], 4096);
        DSR128(B[0][0], fb + koff[0], 0); DSR128(B[0][1], fb + koff[0], 4096);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk + 1 < 4) {
                DSR128(A[(kk + 1) & 1][0], fa + koff[(kk + 1) & 3], 0);
                if (MI == 2) DSR128(A[(kk + 1) & 1][MI - 1], fa + koff[(kk + 1) & 3], 4096);
                DSR128(B[(kk + 1) & 1][0], fb + koff[(kk + 1) & 3], 0); DSR128(B[(kk + 1) & 1][1], fb + koff[(kk + 1) & 3], 4096);
                if (MI == 2) LGKM_WAIT(4); else LGKM_WAIT(3);
            } else {
                LGKM_WAIT(0);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, B[kk & 1][j]), __builtin_bit_cast(bf16x8, A[kk & 1][i]),
                                                                        acc[i][j], 0, 0, 0);
            if (more) {
                if (kk == 0) issue_part(std::integral_constant<int, 0>{});
                if (kk == 1) issue_part(std::integral_constant<int, 1>{});
                if (kk == 2) issue_part(std::integral_constant<int, 2>{});
                if (kk == 3) issue_part(std::integral_constant<int, 3>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        slot = slot + 1 == S ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();      // every wave is done with the ring: the epilogue stages through the same LDS

    // ---- epilogue through LDS (as igemm_bf16.hip): acc[i][j] holds D[n = 32j + (r&3) + 8(r>>2) + 4h][pixel = 32i + (lane&31)]
    constexpr int WN = 64, WM = BM_ / 2;
    constexpr int SROW = WN + 8;
    __bf16* stage = reinterpret_cast<__bf16*>(smem) + wave * (WM * SROW);
    const int hq = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
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
        constexpr int RPL = WM;                           // WN = 64: one lane per channel, WM rows each
        const int ch = lane;
        float cs = 0.f, css = 0.f;
#pragma unroll 8
        for (int r = 0; r < RPL; ++r) {
            if (m0 + wm * WM + r < M) { const float v = (float)stage[r * SROW + ch]; cs += v; css += v * v; }
        }
        s_cs[wave][ch][0] = cs; s_cs[wave][ch][1] = css;
        __syncthreads();
        if (tid < G3_BN) {
            const int wn_ = tid / WN, c = tid % WN, nn = n0 + wn_ * WN + c;
            if (nn < a.g.N) {
                float* row = a.colstat + ((size_t)mt * a.g.N + nn) * 2;
                row[0] = s_cs[wn_][c][0] + s_cs[2 + wn_][c][0];
                row[1] = s_cs[wn_][c][1] + s_cs[2 + wn_][c][1];
            }
        }
    }
}

}  // namespace

// (kernels outside the anonymous namespace: profilers print their names)
template <int BM_, int S>
__global__ __launch_bounds__(256) void igemm3_fwd_bf16_kernel(const IgemmArgsH a) {
    igemm3_body<BM_, S>(a, blockIdx.x, gridDim.x);
}
// four launches of the same shape in one grid (blockIdx.y): the output-parity classes of a stride-2 transposed conv
template <int BM_, int S>
__global__ __launch_bounds__(256) void igemm3_fwd_bf16_kernel4(const IgemmArgsH4 a4) {
    igemm3_body<BM_, S>(a4.a[blockIdx.y], blockIdx.x, gridDim.x);
}

// C % 64 == 0 (a chunk is 64 channels of ONE tap), N > 64 (128-channel tiles), 32-bit byte offsets into both operands
bool igemm3_applies(const IgemmArgsH& a) {
    const long long M = (long long)a.g.B * a.g.PH * a.g.PW;
    const size_t in_bytes = (((size_t)a.g.B * a.g.IH * a.g.IW - 1) * a.g.ldi + a.g.C) * 2, w_bytes = (size_t)a.g.N * a.g.wtaps * a.g.C * 2;
    return unetrir_cfg().igemm3 && M > 0 && M < (1ll << 31) && a.g.N > 64 && a.g.C >= 64 && a.g.C % 64 == 0 && (a.g.ldi & 7) == 0 &&
           in_bytes < 0x70000000u && w_bytes < 0x70000000u && a.g.ntaps <= 64 && (((uintptr_t)a.in | (uintptr_t)a.w) & 15) == 0;
}

// pixel-tile height: that of the launch it replaces (64 where 128-pixel tiles would leave CUs without a workgroup: igemm2_bf16.hip),
// so the rows of column statistics (igemm_colstat_rows) do not depend on which kernel runs
static int igemm3_tile_m(long long M, int N, int ncls) { return igemm_bf16_tile_m(M, N, ncls) ? 64 : 128; }

int launch_igemm3_fwd_bf16(const IgemmArgsH* a, int ncls, hipStream_t s) {
    const long long M = (long long)a[0].g.B * a[0].g.PH * a[0].g.PW;
    const int bm = igemm3_tile_m(M, a[0].g.N, ncls);
    const unsigned nwg = (unsigned)(((M + bm - 1) / bm) * ((a[0].g.N + 127) / 128));
    IgemmArgsH4 a4;
    if (ncls == 4) {
        for (int i = 0; i < 4; ++i) a4.a[i] = a[i];
        if (bm == 128) hipLaunchKernelGGL((igemm3_fwd_bf16_kernel4<128, 2>), dim3(nwg, 4), dim3(256), 0, s, a4);
        else hipLaunchKernelGGL((igemm3_fwd_bf16_kernel4<64, 2>), dim3(nwg, 4), dim3(256), 0, s, a4);
    } else {
        if (bm == 128) hipLaunchKernelGGL((igemm3_fwd_bf16_kernel<128, 2>), dim3(nwg), dim3(256), 0, s, a[0]);
        else hipLaunchKernelGGL((igemm3_fwd_bf16_kernel<64, 2>), dim3(nwg), dim3(256), 0, s, a[0]);
    }
    return (int)hipGetLastError();
}
