// conv3x3s.hip - bf16 3x3 stride-1 'same' convolution (forward / data gradient) for the 64 -> 64 channel full-resolution
// layers of the U-Net (enc1.cb1, dec1.cb1b and their data gradients: dl_models/u_net.py:366): the HBM-bound end of the
// 3x3 family (537 MB moved per 155 GFLOP at batch 32, 256 x 256).
//
// What bounds these layers is bytes in flight, not MFMA rate, so the kernel is built around the memory system:
//   * one persistent workgroup per CU (8 waves) walks DOWN a 32-pixel-wide column strip of one image, 8 rows per tile;
//   * the whole 64 x 9 x 64 kernel (73.7 KB bf16) is loaded into LDS ONCE per workgroup, stored in MFMA-fragment order
//     (72 blocks of 1 KB: a fragment read is lane-linear, conflict-free, one instruction);
//   * a pixel's 64 channels are one 128-byte line: the (8+2) x 34 pixel patch of a tile is staged by LDS-DMA in WHOLE lines
//     (buffer_load_dwordx4 ... lds, 8 pixels per wave-instruction), each line requested once - no partial-line requests at
//     different times (the 2.0x read over-fetch of the half-chunk kernel, conv3x3h.hip);
//   * two patch buffers: the patch of tile t+1 streams in during the WHOLE K loop of tile t (43 KB in flight per CU, issued
//     one instruction per K step), and its two halo rows shared with tile t are copied LDS -> LDS instead of fetched again,
//     so the input is read 34/32 times in total;
//   * with kernel and patch resident there is no dependency inside a tile: the K loop (2 chunks x 3 horizontal taps, 24
//     v_mfma_f32_16x16x32_bf16 and 16 fragment reads per step and wave) runs without barriers or counted waits; ONE
//     s_barrier per tile, in front of it a counted vmcnt that leaves the tile's output stores in flight;
//   * a wave owns 2 rows x 16 columns x all 64 channels; the channel order inside the MFMA rows is permuted so that a lane
//     ends up with 2 x 8 consecutive channels of one pixel: stores are 16 bytes per lane straight from registers (no LDS
//     staging), 64-byte half lines per pixel per instruction;
//   * fused column statistics (BatchNormalization batch statistics / bias gradients from the stored bf16 values) are
//     accumulated per lane over the tiles of a JOB (a vertical segment of a strip) and leave the registers at its end: one
//     colstat row per (job, wave), whoever served the job - the sums do not depend on the run-time tile assignment;
//   * WHICH job a workgroup takes next is decided at run time (tickets, as in conv3x3p.hip): a workgroup that cannot be placed
//     at once (a CU held by a collective or a side-stream kernel) leaves its jobs to the ones that run.
// Patch LDS image: pixel-major, 128-byte pixels, the eight 16-byte granules of a pixel XOR-swizzled with (column & 7) on
// the DMA source side and on the fragment read (conflict-free ds_read_b128 for 16 consecutive columns).
// Round 3: the same kernel for 32 -> 32 channels (template CH; the full-resolution layers of the residual graphs, dl_models/res_ae.py:466,
// and of number_filters_0 = 32, main_training.py:154-161, which conv3x3h served with half of every MFMA tile empty and every pixel
// line fetched in two pieces): 64-byte pixels, 16 per DMA instruction; two pixels share a 128-byte LDS line and the four granules of
// a pixel are swizzled with (column >> 2) & 3, so that the 16 columns of a fragment read meet 16 different bank groups; one K chunk
// (9 sub-steps of 4 MFMAs); 62 KB of LDS and 128 registers: two workgroups per CU.
// Requires C == N == 64 or C == N == 32 (conv3x3s_applies); everything else stays on conv3x3h / conv3x3g / conv3x3r.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

#define DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LGKM_WAIT(n) do { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define MMA16(accv, wfrag, pfrag) \
    accv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfrag), __builtin_bit_cast(bf16x8, pfrag), accv, 0, 0, 0)

namespace {
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_sum16(float v) {       // sum over the 16 lanes of a DPP row, in every lane of the row
    v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);     // row_half_mirror
    return dpp_add<0x140>(v);  // row_mirror
}
constexpr int SPC = 34;                         // patch columns
constexpr int STR = 8;                          // tile rows
constexpr int SNPX = (STR + 2) * SPC;           // 340 patch pixels
constexpr uint32_t OOB = 0xF0000000u;
// CH = 64 (the layers the kernel was built for) or 32 (round 3: the full-resolution 32 -> 32 layers of the residual graphs,
// dl_models/res_ae.py:466, and of the reference's own number_filters_0 = 32, main_training.py:154-161): values for 64 / 32 channels
template <int CH> struct SC {
    static constexpr int PIXB = CH * 2;                        // 128 / 64 bytes per pixel
    static constexpr int PPI = 1024 / PIXB;                    // 8 / 16 pixels per DMA wave-instruction
    static constexpr int GPP = PIXB / 16;                      // 8 / 4 16-byte granules per pixel
    static constexpr int KCN = CH / 32;                        // 2 / 1 K chunks of 32 input channels
    static constexpr int NT = CH / 16;                         // 4 / 2 channel tiles of 16
    static constexpr int NH = CH / 32;                         // 2 / 1 groups of 8 channels a lane stores per pixel
    static constexpr int P_INSTR = (SNPX + PPI - 1) / PPI;     // 43 / 22 wave-instructions per patch
    static constexpr int P_BYTES = P_INSTR * 1024;             // 44032 / 22528
    static constexpr int ROWB = SPC * PIXB;                    // 4352 / 2176: patch row pitch
    static constexpr int NBLK = 9 * KCN * NT;                  // 72 / 18 kernel fragment blocks [9 taps][KCN chunks][NT channel tiles]
    static constexpr int W_BYTES = NBLK * 1024;                // 73728 / 18432
    static constexpr int SMEM = W_BYTES + 2 * P_BYTES;         // 161792 (of 163840) / 63488
    static constexpr int TICK = SMEM;                          // 4 words: the first two tickets, then the hand-over words of even / odd jobs
    static constexpr int CARRY = 64 / PPI;                     // 8 / 4 wave-instructions (64 pixels) of the next patch that come from this one through LDS
    static constexpr int NJ = (P_INSTR + 7) / 8;               // 6 / 3 patch DMA instructions per wave
    static constexpr int NU = 9 * KCN;                         // 18 / 9 sub-steps of the K loop
    static constexpr int NG = 3 * KCN;                         // 6 / 3 groups (chunk, horizontal tap)
    static constexpr int WP = 9 * CH * 2;                      // 1152 / 576: byte pitch of a kernel row [n][tap][c]
};

// The K loop is 18 sub-steps u = (group g = (chunk kc, horizontal tap dx), vertical tap dy): 8 MFMAs (2 output rows x 4 channel
// tiles) on 4 kernel fragments and two patch-row fragments.  Fragments of sub-step u+1 are requested before the MFMAs of u:
// 4 kernel blocks into the other of two register sets, and the ONE patch row u+1 adds (two at the start of a group, into
// the other patch set).
template <int CH, int U>
__device__ __forceinline__ void issue_reads(u32x4 (&A)[2][SC<CH>::NT], u32x4 (&B)[2][4], uint32_t wa, uint32_t wa8, const uint32_t (&pa)[SC<CH>::NG]) {
    using T = SC<CH>;
    constexpr int G = U / 3, DY = U % 3, KC = G / 3, DX = G % 3;
    constexpr int BLK = ((DY * 3 + DX) * T::KCN + KC) * T::NT;   // block = (tap * KCN + kc) * NT + t, tap = dy * 3 + dx
    if constexpr (DY == 0) { DSR128(B[G & 1][0], pa[G], 0 * T::ROWB); DSR128(B[G & 1][1], pa[G], 1 * T::ROWB); }
    if constexpr (DY == 1) DSR128(B[G & 1][2], pa[G], 2 * T::ROWB);
    if constexpr (DY == 2) DSR128(B[G & 1][3], pa[G], 3 * T::ROWB);
    if constexpr (BLK < 64) {                                     // 16-bit offsets from wa; blocks from 64 on (tap 8 at 64 channels) from wa8 = wa + 64 KB
        DSR128(A[U & 1][0], wa, (BLK + 0) * 1024); DSR128(A[U & 1][1], wa, (BLK + 1) * 1024);
        if constexpr (T::NT == 4) { DSR128(A[U & 1][2], wa, (BLK + 2) * 1024); DSR128(A[U & 1][3], wa, (BLK + 3) * 1024); }
    } else {
        DSR128(A[U & 1][0], wa8, (BLK - 64 + 0) * 1024); DSR128(A[U & 1][1], wa8, (BLK - 64 + 1) * 1024);
        if constexpr (T::NT == 4) { DSR128(A[U & 1][2], wa8, (BLK - 64 + 2) * 1024); DSR128(A[U & 1][3], wa8, (BLK - 64 + 3) * 1024); }
    }
}

template <int CH, int U>
__device__ __forceinline__ void k_substeps(u32x4 (&A)[2][SC<CH>::NT], u32x4 (&B)[2][4], f32x4 (&acc)[2][SC<CH>::NT], uint32_t wa, uint32_t wa8,
                                           const uint32_t (&pa)[SC<CH>::NG]) {
    using T = SC<CH>;
    constexpr int G = U / 3, DY = U % 3;
    if constexpr (U + 1 < T::NU) {
        issue_reads<CH, U + 1>(A, B, wa, wa8, pa);
        // everything but the reads just issued (NT kernel blocks + two patch rows at the start of a group, else one)
        if constexpr (T::NT == 4) { if constexpr ((U + 1) % 3 == 0) LGKM_WAIT(6); else LGKM_WAIT(5); }
        else { if constexpr ((U + 1) % 3 == 0) LGKM_WAIT(4); else LGKM_WAIT(3); }
    } else {
        LGKM_WAIT(0);
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int t = 0; t < T::NT; ++t) MMA16(acc[o][t], A[U & 1][t], B[G & 1][o + DY]);
    __builtin_amdgcn_s_setprio(0);
    if constexpr (U + 1 < T::NU) k_substeps<CH, U + 1>(A, B, acc, wa, wa8, pa);
}
}  // namespace

// abl (ablation build only): 1 no patch DMA after the first tile, 2 no output stores, 4 no MFMA loop
template <int CH>
__global__ __launch_bounds__(512, CH == 32 ? 4 : 2) void conv3x3s_bf16_kernel(const Conv3Args a, int nseg, int seglen, int per_xcd, unsigned* sched, int abl) {
    using T = SC<CH>;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[T::SMEM + 16];
    const __bf16* __restrict__ in = (const __bf16*)a.in;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rp = wave >> 1, ph = wave & 1;          // row pair, 16-column half of the tile
    const int l15 = lane & 15, lq = lane >> 4;
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + STR - 1) / STR;
    const int nstrips = a.B * tiles_x, njobs = nstrips * nseg;
    // ---- jobs of this workgroup.  Job = strip * nseg + segment.  With tickets (sched != nullptr: kernels.h, sched_slot) XCD x
    // (= blockIdx & 7) owns per_xcd consecutive jobs - neighbouring strips and segments share its L2 - and its workgroups draw them
    // from one counter; without, the fixed assignment wg, wg + grid, ... over all jobs.
    int wg = blockIdx.x;
    if ((gridDim.x & 7) == 0) wg = (wg & 7) * (gridDim.x >> 3) + (wg >> 3);
    const int xcd = blockIdx.x & 7;
    const int pt0 = sched ? xcd * per_xcd : 0;
    const int cnt = sched ? max(0, min(njobs, pt0 + per_xcd) - pt0) : njobs;
    unsigned* ctr = sched ? sched + xcd * 8 : nullptr;
    const uint64_t ctr_addr = (uint64_t)(uintptr_t)ctr;
    int kfix = 0;                                         // fixed assignment: tickets handed out so far (thread 0)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    if (tid == 0) {                                       // the first two tickets: one round trip
        unsigned* tk = reinterpret_cast<unsigned*>(smem + T::TICK);
        if (ctr) { const unsigned t = __hip_atomic_fetch_add(ctr, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); tk[0] = t; tk[1] = t + 1; }
        else { tk[0] = (unsigned)wg; tk[1] = (unsigned)wg + gridDim.x; kfix = 2; }
    }

    // ---- the kernel, once per workgroup, in fragment order.  Block (tap, kc, t), lane l: row r = l & 15 of the MFMA A operand is
    // output channel n(t, r) = 32 (t >> 1) + 8 (r >> 2) + 4 (t & 1) + (r & 3); k group l >> 4 = input channels 32 kc + 8 (l >> 4) ..
    {
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, CH * 9 * CH * 2, 0x00020000);
        const uint32_t vw = (uint32_t)((8 * (l15 >> 2) + (l15 & 3)) * T::WP + lq * 16);
#pragma unroll
        for (int j = 0; j < (T::NBLK + 7) / 8; ++j) {
            const int b = wave + 8 * j;
            if (b >= T::NBLK) continue;                          // wave-uniform
            const int t = b % T::NT, kc = (b / T::NT) % T::KCN, tap = b / (T::NT * T::KCN);
            const int tapsrc = (a.flip & 1) ? 8 - tap : tap;
            const uint32_t so = (uint32_t)((32 * (t >> 1) + 4 * (t & 1)) * T::WP + tapsrc * T::PIXB + kc * 64);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(smem + b * 1024), 16, vw + so, 0, 0, 0);
        }
    }

    // ---- patch DMA: wave-instruction i covers patch pixels PPI i .. PPI i + PPI - 1 (8 of 128 B, or 16 of 64 B), lane l the 16-byte
    // granule l % GPP of pixel PPI i + l / GPP.
    // This wave issues instructions wave, wave + 8, ...: their patch row / column are fixed per lane, the byte offset of a
    // strip's column part is computed once per job, a tile only adds its row offset and the row bound.
    int dma_pr[T::NJ], dma_pc[T::NJ];
#pragma unroll
    for (int j = 0; j < T::NJ; ++j) {
        const int p = T::PPI * (wave + 8 * j) + lane / T::GPP;
        dma_pr[j] = p < SNPX ? p / SPC : 1 << 20;              // past the patch: never a valid row
        dma_pc[j] = p - (p / SPC) * SPC;
    }
    int dma_x[T::NJ];                                        // ((pr - 1) W + ix) ldi 2 + 16 sg for this job's strip, or INT_MIN
    auto image_rsrc = [&](int img) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(in + (size_t)img * a.H * a.W * a.ldi), (short)0,
                                                 (int)((((size_t)a.H * a.W - 1) * a.ldi + CH) * 2), 0x00020000);
    };
    // source granule of LDS slot l % GPP: the granules of a pixel are XOR-swizzled with its column (64 channels: 8 granules, key
    // column & 7; 32 channels: 4 granules, key (column >> 2) & 3 - two pixels share a 128-byte line, four consecutive ones a key)
    auto job_columns = [&](int x0, int (&dx_)[T::NJ]) {
#pragma unroll
        for (int j = 0; j < T::NJ; ++j) {
            const int ix = x0 - 1 + dma_pc[j];
            const int sg = CH == 64 ? ((lane & 7) ^ (dma_pc[j] & 7)) : ((lane & 3) ^ ((dma_pc[j] >> 2) & 3));
            dx_[j] = (unsigned)ix < (unsigned)a.W ? (((dma_pr[j] - 1) * a.W + ix) * a.ldi + sg * 8) * 2 : INT32_MIN;
        }
    };
    auto issue_patch = [&](const __amdgpu_buffer_rsrc_t& rs, const int (&dx_)[T::NJ], int y0, int first, int buf) {
        unsigned char* dst = smem + T::W_BYTES + buf * T::P_BYTES;
        const int rowoff = y0 * a.W * a.ldi * 2;
#pragma unroll
        for (int j = 0; j < T::NJ; ++j) {
            const int i = wave + 8 * j;
            if (i < first || i >= T::P_INSTR) continue;         // wave-uniform
            const bool ok = dx_[j] != INT32_MIN && (unsigned)(y0 - 1 + dma_pr[j]) < (unsigned)a.H;
            const uint32_t off = ok ? (uint32_t)(dx_[j] + rowoff) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + i * 1024), 16, off, 0, 0, 0);
        }
    };
    auto job_origin = [&](int job, int& img, int& x0, int& ty0, int& ty1) {
        const int strip = job / nseg, seg = job - strip * nseg;
        img = strip / tiles_x;
        x0 = (strip - img * tiles_x) * 32;
        ty0 = seg * seglen;
        ty1 = ty0 + seglen < tiles_y ? ty0 + seglen : tiles_y;
    };

    // ---- fragment read addresses
    const uint32_t wa = lds0 + lane * 16, wa8 = wa + 65536;
    uint32_t pbase[T::KCN][3];                        // + buf * P_BYTES + r * ROWB
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int col = 16 * ph + l15 + dx;
#pragma unroll
        for (int kc = 0; kc < T::KCN; ++kc)
            pbase[kc][dx] = lds0 + T::W_BYTES + (uint32_t)((2 * rp * SPC + col) * T::PIXB +
                                                        ((CH == 64 ? ((kc * 4 + lq) ^ (col & 7)) : (lq ^ ((col >> 2) & 3))) << 4));
    }

    // ---- epilogue constants: lane (pixel column l15, quarter lq) holds channels 32 h + 8 lq .. + 8 (h = 0, 1) of its pixel
    float bias_[T::NH][8];
#pragma unroll
    for (int h = 0; h < T::NH; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) bias_[h][e] = a.bias ? a.bias[32 * h + 8 * lq + e] : 0.f;
    float cs_s[T::NH][8], cs_q[T::NH][8];
#pragma unroll
    for (int h = 0; h < T::NH; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) { cs_s[h][e] = 0.f; cs_q[h][e] = 0.f; }
    __bf16* __restrict__ out = (__bf16*)a.out;
    const __bf16* __restrict__ addend = (const __bf16*)a.addend;

    // D row 4 lq + j of channel tile t is channel 32 (t >> 1) + 8 lq + 4 (t & 1) + j of pixel column l15: 16 bytes per lane and
    // (row, half) straight from the accumulators.  Returns the number of store instructions issued (wave-uniform).
    auto epilogue = [&](const f32x4 (&acc)[2][T::NT], int img, int x0, int y0) -> int {
        const int x = x0 + 16 * ph + l15;
        int nst = 0;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int y = y0 + 2 * rp + o;
            if (!(y < a.H && x0 + 16 * ph < a.W) || UNETRIR_ABL(abl, 2)) continue;   // wave-uniform: lane 0 of an issued store is always live
            const bool ok = x < a.W;
            const size_t pix = ((size_t)img * a.H + y) * a.W + x;
#pragma unroll
            for (int h = 0; h < T::NH; ++h) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc[o][2 * h + (e >> 2)][e & 3] + bias_[h][e];
                if (addend && ok) {
                    const bf16x8 ad = *reinterpret_cast<const bf16x8*>(addend + pix * a.ldadd + 32 * h + 8 * lq);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (float)(__bf16)v[e] + (float)ad[e];
                }
                bf16x8 ov;
#pragma unroll
                for (int e = 0; e < 8; ++e) ov[e] = (__bf16)v[e];
                if (ok) {
                    if (a.colstat) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float s = (float)ov[e]; cs_s[h][e] += s; cs_q[h][e] += s * s; }
                    }
                    *reinterpret_cast<bf16x8*>(out + pix * a.ldo + 32 * h + 8 * lq) = ov;
                }
                ++nst;
            }
        }
        return nst;
    };

    // ---- column statistics of a finished job: over the 16 pixel columns of a lane group, then lane l15 == 0 writes its NH x 8 channels
    //      of the row of (job, this wave).  4 NH store instructions (wave-uniform), counted by the caller's vmcnt waits.
    auto flush_stats = [&](int jobid) {
        float* row = a.colstat + ((size_t)jobid * 8 + wave) * (2 * CH);
#pragma unroll
        for (int h = 0; h < T::NH; ++h) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { cs_s[h][e] = row_sum16(cs_s[h][e]); cs_q[h][e] = row_sum16(cs_q[h][e]); }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = {cs_s[h][2 * q], cs_q[h][2 * q], cs_s[h][2 * q + 1], cs_q[h][2 * q + 1]};
                if (l15 == 0) *reinterpret_cast<f32x4*>(row + (32 * h + 8 * lq + 2 * q) * 2) = v;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { cs_s[h][e] = 0.f; cs_q[h][e] = 0.f; }
        }
    };

    // the first two tickets (inline asm: a barrier or an LDS load the compiler can see would drain the kernel's DMAs first)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    unsigned tk0, tk1;
    {
        unsigned v0, v1;
        asm volatile("ds_read_b32 %0, %1" : "=v"(v0) : "v"(lds0 + T::TICK));
        asm volatile("ds_read_b32 %0, %1 offset:4" : "=v"(v1) : "v"(lds0 + T::TICK));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        tk0 = (unsigned)__builtin_amdgcn_readfirstlane((int)v0);
        tk1 = (unsigned)__builtin_amdgcn_readfirstlane((int)v1);
    }
    if (tk0 < (unsigned)cnt) {
    int job = pt0 + (int)tk0, img, x0, ty0, ty1;
    bool have_next = tk1 < (unsigned)cnt;                 // a next job is known to exist
    int njob = pt0 + (int)tk1;
    int par = 0;                                          // hand-over word of the current job
    job_origin(job, img, x0, ty0, ty1);
    job_columns(x0, dma_x);
    __amdgpu_buffer_rsrc_t rs_in = image_rsrc(img);
    issue_patch(rs_in, dma_x, ty0 * STR, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the bias loads are ordinary loads: touch their results here, so that the wait the compiler attaches to their first use
    // lands in the prologue and not inside the tile loop (where it would drain the patch DMA)
#pragma unroll
    for (int h = 0; h < T::NH; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(bias_[h][e]));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int cur = 0;

    // Waves w and w + 4 share a SIMD and run the same program: in lockstep both would convert / store / issue DMA at the same
    // time and leave the matrix pipe idle.  Waves 4-7 therefore DEFER the epilogue of a tile to the start of the next one
    // (accumulators parked in registers): their epilogue runs beside the K loop of waves 0-3 and vice versa.
    const bool defer = wave >= 4;
    f32x4 pacc[2][T::NT];
    int p_img = 0, p_x0 = 0, p_y0 = 0, p_job = -1;        // p_job >= 0: the parked tile is the last of that job (statistics leave with it)
    bool have_prev = false;
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int t = 0; t < T::NT; ++t) pacc[o][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (;;) {
        // ---- the ticket of the job after next: thread 0 draws it at the start of this job (a returning atomic: 1-2 us; inline asm:
        //      the compiler's own atomic sequence waits vmcnt(0) on the spot) and hands it over through LDS in front of the barrier of
        //      the job's first tile, when the DMAs issued after it have landed anyway
        const bool draw = have_next;                      // no further draw after the first ticket past the end
        unsigned tk_mine = 0xFFFFFFFFu;
        if (draw && tid == 0) {
            if (ctr) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(tk_mine) : "v"(ctr_addr), "v"(1u) : "memory");
            else tk_mine = (unsigned)wg + (unsigned)(kfix++) * gridDim.x;
        }
        for (int ty = ty0; ty < ty1; ++ty) {
            const int y0 = ty * STR;
            // ---- the next tile's patch: from this one (two halo rows through LDS, the rest by DMA) or the first of the next job
            const bool same = ty + 1 < ty1;
            if (UNETRIR_ABL(abl, 1)) {
            } else if (same) {
                // inline asm: an LDS load the compiler can see would make it drain every outstanding store first
                u32x4 v;
                const uint32_t src = lds0 + T::W_BYTES + cur * T::P_BYTES + (STR * SPC) * T::PIXB + tid * 16;
                const uint32_t dstc = lds0 + T::W_BYTES + (cur ^ 1) * T::P_BYTES + tid * 16;
                if (CH == 64 || tid < 64 * T::PIXB / 16) {       // 64 pixels: 8 KB (every thread) / 4 KB (waves 0-3)
                    DSR128(v, src, 0);
                    LGKM_WAIT(0);
                    asm volatile("ds_write_b128 %0, %1" :: "v"(dstc), "v"(v) : "memory");
                }
                issue_patch(rs_in, dma_x, y0 + STR, T::CARRY, cur ^ 1);
            } else if (have_next) {
                int img_n, x0_n, ty0_n, ty1_n, dma_n[T::NJ];
                job_origin(njob, img_n, x0_n, ty0_n, ty1_n);
                job_columns(x0_n, dma_n);
                const __amdgpu_buffer_rsrc_t rs_n = image_rsrc(img_n);
                issue_patch(rs_n, dma_n, ty0_n * STR, 0, cur ^ 1);
            }
            int nst = 0;                                    // store instructions issued after the DMAs above
            if (defer && have_prev) {
                nst = epilogue(pacc, p_img, p_x0, p_y0);
                if (a.colstat && p_job >= 0) { flush_stats(p_job); nst += 4 * T::NH; }
            }
            // ---- K loop: NU sub-steps, no synchronisation inside
            f32x4 acc[2][T::NT];
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int t = 0; t < T::NT; ++t) acc[o][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            uint32_t pa[T::NG];
#pragma unroll
            for (int g = 0; g < T::NG; ++g) pa[g] = pbase[g / 3][g % 3] + cur * T::P_BYTES;
            u32x4 A[2][T::NT], B[2][4];
            if (!UNETRIR_ABL(abl, 4)) {
                issue_reads<CH, 0>(A, B, wa, wa8, pa);
                k_substeps<CH, 0>(A, B, acc, wa, wa8, pa);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (defer) {
#pragma unroll
                for (int o = 0; o < 2; ++o)
#pragma unroll
                    for (int t = 0; t < T::NT; ++t) pacc[o][t] = acc[o][t];
                p_img = img; p_x0 = x0; p_y0 = y0; p_job = same ? -1 : job; have_prev = true;
            } else {
                nst = epilogue(acc, img, x0, y0);
                if (a.colstat && !same) { flush_stats(job); nst += 4 * T::NH; }
            }
            // ---- the next patch is complete once this wave's DMAs are: they are older than the nst output stores, which may
            //      stay in flight across the barrier (so is thread 0's ticket)
            //      (64 channels: 0 / 2 / 4 tile stores + 0 / 8 statistics stores; 32 channels: 0 / 1 / 2 + 0 / 4)
            if (nst == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (nst == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if (nst == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (nst == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (nst == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if (nst == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (nst == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (nst == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ty == ty0 && draw && tid == 0)
                asm volatile("ds_write_b32 %0, %1" :: "v"(lds0 + T::TICK + 8 + 4 * par), "v"(tk_mine) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            cur ^= 1;
        }
        if (!have_next) break;
        job = njob;
        have_next = false;
        if (draw) {                                       // the ticket drawn during the job just finished (behind >= 1 barrier)
            unsigned v;
            asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(lds0 + T::TICK + 8 + 4 * par));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const unsigned tk = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
            have_next = tk < (unsigned)cnt;
            njob = pt0 + (int)tk;
        }
        par ^= 1;
        job_origin(job, img, x0, ty0, ty1);
        job_columns(x0, dma_x);
        rs_in = image_rsrc(img);
    }
    if (defer && have_prev) {
        epilogue(pacc, p_img, p_x0, p_y0);
        if (a.colstat && p_job >= 0) flush_stats(p_job);
    }
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no job: the kernel's DMAs must not land in LDS that is no longer ours
    }

    // ---- the last workgroup to leave clears the launch's counters for the next launch on this stream (every workgroup has
    //      drawn its last - failing - ticket before it counts itself out)
    if (sched && tid == 0) {
        const unsigned d = __hip_atomic_fetch_add(sched + 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == gridDim.x - 1) {
            for (int i = 0; i < 65; ++i) __hip_atomic_store(sched + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

namespace {
struct SPlan { int grid, nseg, seglen, njobs; };
// jobs = column strips x vertical segments; persistent workgroups.  Segments so that there are >= 1 024 jobs (four per workgroup at
// 64 channels, two at 32): the unit that tickets hand out (a workgroup placed late takes fewer), and imbalance <= 25 % where the
// strip count is no multiple of the CU count.  A job's first tile fetches its two halo rows itself (10 instead of 8 patch rows: + 3 % at 8 tiles per job).
inline SPlan splan(const Conv3Args& a) {
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + STR - 1) / STR;
    const long long nstrips = (long long)a.B * tiles_x;
    // 64 channels: 158 KB of LDS, one workgroup per CU; 32 channels: 62 KB and 128 registers, two per CU (measured at 128 x 128 / 256 x 256,
    // batch 32: 28.5 / 75.7 us with 256 workgroups, 25.2 / 59.9 us with 512)
    const int cus = a.C == 32 ? 512 : 256;
    int nseg = 1;
    while (nseg < tiles_y && nstrips * nseg < 1024) ++nseg;
    SPlan p;
    p.seglen = (tiles_y + nseg - 1) / nseg;
    p.nseg = (tiles_y + p.seglen - 1) / p.seglen;
    const long long jobs = nstrips * p.nseg;
    p.njobs = (int)jobs;
    p.grid = (int)(jobs < cus ? jobs : cus);
    return p;
}
}  // namespace

bool conv3x3s_applies(const Conv3Args& a) {
    const size_t img_bytes = (((size_t)a.H * a.W - 1) * a.ldi + a.C) * 2;
    return unetrir_cfg().conv3x3s && ((a.C == 64 && a.N == 64) || (a.C == 32 && a.N == 32)) && !(a.flip & 2) && a.ldi >= a.C && (a.ldi & 7) == 0 &&
           img_bytes < 0x70000000u;
}

long long conv3x3s_colstat_rows(const Conv3Args& a) { return (long long)splan(a).njobs * 8; }      // one row per (job, wave)

int launch_conv3x3s_bf16(const Conv3Args& a, hipStream_t s) {
    const SPlan p = splan(a);
    unsigned* sched = (p.grid >= 256 && p.grid % 8 == 0) ? sched_slot(s) : nullptr;          // tickets need workgroups on every XCD
    if (a.C == 64) hipLaunchKernelGGL(conv3x3s_bf16_kernel<64>, dim3((unsigned)p.grid), dim3(512), 0, s, a, p.nseg, p.seglen, (p.njobs + 7) / 8, sched,
                                      UNETRIR_ABL_HOST());
    else hipLaunchKernelGGL(conv3x3s_bf16_kernel<32>, dim3((unsigned)p.grid), dim3(512), 0, s, a, p.nseg, p.seglen, (p.njobs + 7) / 8, sched,
                            UNETRIR_ABL_HOST());
    return (int)hipGetLastError();
}
