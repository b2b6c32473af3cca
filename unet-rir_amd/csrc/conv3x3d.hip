// conv3x3d.hip - bf16 3x3 STRIDE-2 'same' convolution, forward form (even sizes: TF pads (0, 1), pad_before = 0):
//
//   y[b][oy][ox][n] = bias[n] + sum_{kh,kw,c} w[n][kh][kw][c] * x[b][2 oy + kh][2 ox + kw][c]
//
// the strided Conv2D of the U-Net encoder (dl_models/u_net.py:269-276) and - with the [Cin][3][3][Cout] kernel of the layer
// as `w` - the data gradient of Conv2DTranspose (:297-304): 8 launches of 77 GFLOP per train step at BASELINE configs[1].
// The tap-table implicit GEMM (igemm_bf16.hip) reached 0.45-0.68 PFLOP/s on them (register-staged global -> LDS, gathered
// 4-byte reads).  Stride 2 reads FOUR input pixels per output pixel, so per flop the patch is 4x the bytes of a stride-1
// layer and LDS capacity - bytes in flight - bounds the design:
//   * one PERSISTENT workgroup per CU (8 waves) runs a stream of steps (tile, 16-channel chunk) over the jobs (8 x 32 output
//     pixels x 128 output channels) it owns; the (2*8+1) x 65 pixel x 16 channel patch (36 KB) and the [9][128][16] kernel
//     slice (36 KB, stored in MFMA-fragment order) of step s+1 stream in by LDS-DMA during ALL of step s (two buffers, ONE
//     raw s_barrier per step), across tile boundaries too: no prologue / drain per tile;
//   * the patch is stored DE-INTERLEAVED by column parity (row = [33 even | 33 odd] pixels x 32 B) by the DMA's per-lane
//     source address, so the 32 input pixels a fragment pairs with tap column kw are CONSECUTIVE in LDS: even + 0 (kw 0),
//     odd + 0 (kw 1), even + 1 (kw 2);
//   * v_mfma_f32_32x32x16_bf16, K = the 16 channels of the chunk; a wave owns 4 output rows x 32 columns x 32 channels: the
//     9 kernel fragments of a step stay in registers, the 9 patch rows x 3 column variants are read once each (row 2o+2 is
//     tap row 2 of output row o and tap row 0 of o+1) - 36 fragment reads per 36 MFMAs (= 72 of the 16x16x32 size), reads two
//     patch rows ahead, as inline asm so that the compiler does not drain the DMA queue in front of them;
//   * the kernel rows of a 32-channel block are permuted in LDS so that a lane's accumulators are 2 x 8 CONSECUTIVE channels
//     of one pixel: 16-byte stores straight from registers, counted vmcnt so they stay in flight across the next barrier;
//   * jobs are ordered (pixel tile, channel block) and dealt to the XCDs in contiguous ranges: the workgroups that read
//     the same patch (the N / 128 channel blocks of a tile) run at the same time on one XCD and share it in that L2.
// 16-byte granules of a 32-byte pixel are swapped where bit 3 of the pixel's row position is set (conflict-free ds_read_b128),
// on the DMA source granule and on the read address.  Out-of-image pixels carry a buffer offset past num_records and read
// zeros.  Requires even input sizes, OH % 8 == 0, OW % 32 == 0, C % 16 == 0, C >= 32, N % 32 == 0 (conv3x3d_applies).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

#define DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LGKM_WAIT(n) do { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define MMA(accv, wfrag, pfrag) \
    accv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wfrag), __builtin_bit_cast(bf16x8, pfrag), accv, 0, 0, 0)

namespace {
constexpr int DTR = 8, DTC = 32, DNB = 128;     // output tile: rows, columns, channels
constexpr int DPR = 2 * DTR + 1;                // 17 patch rows
constexpr int DPW = 66;                         // patch row: 33 even columns, 33 odd columns (the last one unused)
constexpr int DNPX = DPR * DPW;                 // 1122 pixels of 32 B
constexpr int DP_INSTR = (DNPX + 31) / 32;      // 36 wave-instructions of 32 pixels x 32 B
constexpr int DW_INSTR = 36;                    // [9 taps][4 blocks of 32 channels] fragment blocks of 1 KB
constexpr int DP_BYTES = DP_INSTR * 1024, DW_BYTES = DW_INSTR * 1024;
constexpr int DROW = DPW * 32;                  // 2112
constexpr int DBIAS = 2 * (DP_BYTES + DW_BYTES);            // 147456
constexpr int DMAXN = 4064;                      // output channels whose bias fits beside the rings (and 16 bytes of tile tickets)
constexpr int DSCHED = DBIAS + DMAXN * 4;       // tile tickets handed from thread 0 to the workgroup
constexpr int DSMEM = DSCHED + 16;              // 163728
constexpr uint32_t OOB = 0xF0000000u;
static_assert(DP_INSTR + DW_INSTR == 72, "nine DMA instructions per wave and step");

struct Job { int img, oy0, ox0, n0, redge; uint32_t pbase, wbase; };
}  // namespace

// a.H, a.W: INPUT size (even); output a.H / 2 x a.W / 2
// abl (ablation build only): 1 no patch DMA, 8 no kernel DMA (after the first step), 2 no output stores, 4 no MFMA loop,
// 16 / 32 kernel / patch DMA pieces read contiguous memory (wrong data: what the 32-byte gather granularity costs)
__global__ __launch_bounds__(512, 2) void conv3x3d_bf16_kernel(const Conv3Args a, int njobs, unsigned* sched, int abl) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[DSMEM];
    const __bf16* __restrict__ in = (const __bf16*)a.in;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rh = wave >> 2, nq = wave & 3;          // half of the 8 output rows, 32-channel block of the 128
    const int l31 = lane & 31, hi = lane >> 5;
    const int IH = a.H, IW = a.W, OH = a.H >> 1, OW = a.W >> 1, C = a.C;
    const int tiles_x = OW / DTC, tiles_y = OH / DTR;
    const int ntN = (a.N + DNB - 1) / DNB;
    const int nch = C >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;

    // ---- jobs of this workgroup: XCD x (= blockIdx & 7 under round-robin dispatch) owns a contiguous range of the job list
    //      which job of its XCD's range a workgroup takes next is decided at run time by tickets (kernels.h, sched_slot; conv3x3p.hip
    //      has the why); grids that are not a multiple of 8 (fewer jobs than CUs) keep the fixed assignment
    int job0, cnt, kstep, kfirst;                     // range start, jobs in the range; fixed assignment: kfirst, kfirst + kstep, ...
    unsigned* ctr = nullptr;
    if ((gridDim.x & 7) == 0) {
        const int per = (njobs + 7) >> 3, xcd = blockIdx.x & 7;
        job0 = xcd * per; cnt = max(0, min(njobs, (xcd + 1) * per) - job0);
        kfirst = (int)(blockIdx.x >> 3); kstep = (int)(gridDim.x >> 3);
        if (sched) ctr = sched + xcd * 8;
    } else {
        job0 = 0; cnt = njobs; kfirst = blockIdx.x; kstep = gridDim.x;
    }
    int kstat = 0;
    // ticket -> job of the range.  Consecutive jobs are the channel tiles of one pixel tile and should run at the same time on
    // different CUs (they share the patch in L2), but a workgroup draws its first two tickets at once: within a round of 64 tickets
    // even tickets walk through the first 32 jobs and odd ones through the second 32
    auto job_of = [&](unsigned t) -> int {
        if (!ctr) return job0 + (int)t;
        const unsigned r = t >> 6, i = t & 63;
        return job0 + (int)((r * 64 + 63 < (unsigned)cnt) ? r * 64 + (i >> 1) + (i & 1) * 32 : t);
    };

    // ---- bias of all output channels, once
    {
        float* bl = reinterpret_cast<float*>(smem + DBIAS);
        for (int n = tid; n < a.N; n += 512) bl[n] = a.bias ? a.bias[n] : 0.f;
        if (tid == 0) {                               // the first two tickets (one round trip)
            unsigned* tk = reinterpret_cast<unsigned*>(smem + DSCHED);
            if (ctr) { const unsigned t = __hip_atomic_fetch_add(ctr, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); tk[0] = t; tk[1] = t + 1; }
            else { tk[0] = (unsigned)kfirst; tk[1] = (unsigned)(kfirst + kstep); kstat = 2; }
        }
    }
    __syncthreads();
    const unsigned tk0 = reinterpret_cast<const unsigned*>(smem + DSCHED)[0], tk1 = reinterpret_cast<const unsigned*>(smem + DSCHED)[1];
    // the last workgroup to leave clears the launch's counters (every workgroup has drawn its last - failing - ticket by then)
    auto leave = [&]() {
        if (sched && (gridDim.x & 7) == 0 && tid == 0) {
            const unsigned d = __hip_atomic_fetch_add(sched + 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == gridDim.x - 1)
                for (int i = 0; i < 65; ++i) __hip_atomic_store(sched + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    if (tk0 >= (unsigned)cnt) { leave(); return; }
    const uint32_t sched_a = lds0 + DSCHED;

    // ---- DMA lane constants.  Combined instruction index i = wave + 8 j (j = 0..8): i < 36 patch instruction i, else kernel
    // block i - 36.  Patch instruction i covers LDS pixels 32 i .. 32 i + 31, lane = (pixel sub, 16-byte slot).
    uint32_t pofs[5];
    int pflag[5];                                     // bit 0: never valid (past the patch / unused position), bit 1: input column 2 ox0 + 64
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int i = wave + 8 * j;
        const int p = 32 * i + (lane >> 1);
        const int pr = p / DPW, pp = p - pr * DPW;
        const int col = pp < 33 ? 2 * pp : 2 * (pp - 33) + 1;
        const int gs = (lane & 1) ^ ((pp >> 3) & 1);
        pofs[j] = (uint32_t)(((pr * IW + col) * a.ldi + gs * 8) * 2);
        if (UNETRIR_ABL(abl, 32)) pofs[j] = (uint32_t)(i * 1024 * (C / 16) + lane * 16);   // timing only: contiguous 1 KB pieces
        pflag[j] = (i >= DP_INSTR || p >= DNPX || pp == 65) ? 1 : (pp == 32 ? 2 : 0);
    }
    // kernel block k = tap * 4 + b (b: 32-channel block of the 128): lane (row rho = lane & 31, k half = lane >> 5) reads
    // channel 32 b + perm(rho), perm(rho) = 16 (rho >> 4) + 8 ((rho >> 2) & 1) + 4 ((rho >> 3) & 1) + (rho & 3)
    const int perm = 16 * (l31 >> 4) + 8 * ((l31 >> 2) & 1) + 4 * ((l31 >> 3) & 1) + (l31 & 3);
    uint32_t wofs[9];
    int wnl[9];
#pragma unroll
    for (int j = 4; j < 9; ++j) {
        int k = wave + 8 * j - DP_INSTR;
        if (k < 0) k = 0;
        const int tap = k >> 2, b = k & 3;
        wnl[j] = 32 * b + perm;
        wofs[j] = (uint32_t)(((wnl[j] * 9 + tap) * C + hi * 8) * 2);
        if (UNETRIR_ABL(abl, 16)) wofs[j] = (uint32_t)(k * 1024 + lane * 16);          // timing only: contiguous 1 KB pieces
    }
    const int in_rec = (int)((((size_t)IH * IW - 1) * a.ldi + C) * 2);
    const size_t img_elems = (size_t)IH * IW * a.ldi;
    // packed copy (a.wpk): piece k of (channel tile nt, chunk ch) is the 1 KB at ((nt * nch + ch) * 36 + k) * 1024 - contiguous
    const bool packed = a.wpk != nullptr;
    const __amdgpu_buffer_rsrc_t rs_w = packed
        ? __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, (short)0, (int)((size_t)ntN * 128 * 9 * C * 2), 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, (int)((size_t)a.N * 9 * C * 2), 0x00020000);
    if (packed) {
#pragma unroll
        for (int j = 4; j < 9; ++j) {
            int k = wave + 8 * j - DP_INSTR;
            if (k < 0) k = 0;
            wofs[j] = (uint32_t)(k * 1024 + lane * 16);
            wnl[j] = 0;
        }
    }

    auto job_params = [&](int jb) {
        Job q;
        const int nt = jb % ntN; int t = jb / ntN;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        q.img = t / tiles_y;
        q.oy0 = ty * DTR; q.ox0 = tx * DTC; q.n0 = nt * DNB;
        q.redge = tx == tiles_x - 1;
        q.pbase = (uint32_t)(((2 * q.oy0 * IW + 2 * q.ox0) * a.ldi) * 2);
        q.wbase = packed ? (uint32_t)((size_t)nt * nch * 36 * 1024) : (uint32_t)((size_t)q.n0 * 9 * C * 2);
        return q;
    };
    // One step's DMA = 9 wave-instructions.  They are issued ONE AT A TIME between the MFMA groups of the first patch rows
    // (an LDS-DMA instruction costs the issuing wave 60-180 cycles: nine of them back to back at the top of a step would
    // idle the matrix pipe of both waves of the SIMD, which run the same program); patch pieces first (HBM latency).
    bool started = false;
    struct Src { __amdgpu_buffer_rsrc_t rs; uint32_t pb, wb; int redge, n0, on; };
    auto step_src = [&](const Job& q, int ch, int on) {
        Src r;
        r.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(in + q.img * img_elems), (short)0, in_rec, 0x00020000);
        r.pb = q.pbase + ch * 32; r.wb = q.wbase + (packed ? ch * 36 * 1024 : ch * 32); r.redge = q.redge; r.n0 = q.n0; r.on = on;
        return r;
    };
    auto dma_piece = [&](const Src& q, int j, int buf) {       // j: compile-time after unrolling
        const int i = wave + 8 * j;
        if (j < 5 && i < DP_INSTR) {                     // wave-uniform
            if (UNETRIR_ABL(abl, 1) && started) return;
            const bool bad = (pflag[j < 5 ? j : 0] & (1 | (q.redge << 1))) != 0 || !q.on;   // (no next step: zeros into the idle buffer)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(q.rs, (lptr_t)(smem + buf * DP_BYTES + i * 1024), 16, bad ? OOB : pofs[j < 5 ? j : 0] + q.pb, 0, 0, 0);
        } else if (j >= 4) {
            if (UNETRIR_ABL(abl, 8) && started) return;
            const bool ok = q.n0 + wnl[j >= 4 ? j : 4] < a.N && q.on;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(smem + 2 * DP_BYTES + buf * DW_BYTES + (i - DP_INSTR) * 1024), 16,
                                                     ok ? wofs[j >= 4 ? j : 4] + q.wb : OOB, 0, 0, 0);
        }
    };

    // ---- fragment read addresses (buffer 0)
    const uint32_t wa0 = lds0 + 2 * DP_BYTES + nq * 1024 + lane * 16;             // + buf * DW_BYTES + tap * 4096
    uint32_t pa0[3];                                                               // + buf * DP_BYTES + r * DROW
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int pp = (kw == 0 ? 0 : kw == 1 ? 33 : 1) + l31;
        pa0[kw] = lds0 + (uint32_t)(((8 * rh) * DPW + pp) * 32 + ((hi ^ ((pp >> 3) & 1)) << 4));
    }
    const uint32_t ba = lds0 + DBIAS;
    __bf16* __restrict__ out = (__bf16*)a.out;
    const __bf16* __restrict__ addend = (const __bf16*)a.addend;

    Job cur = job_params(job_of(tk0));
    bool have_next = tk1 < (unsigned)cnt;           // a next job is known to exist
    Job nxt = have_next ? job_params(job_of(tk1)) : cur;
    bool pending = false;                             // the next job's ticket is on its way (drawn in the last job's last step)
    bool drawing = have_next;                         // tickets are drawn until the first one past the end
    {
        const Src s0 = step_src(cur, 0, 1);
#pragma unroll
        for (int j = 0; j < 9; ++j) dma_piece(s0, j, 0);
    }
    started = true;
    int par = 0;
    bool pend = false;                                // output stores issued after the newest DMAs
    for (;;) {
        f32x16 acc[4];
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[o][r] = 0.f;

        for (int ch = 0; ch < nch; ++ch) {
            // ---- this wave's DMAs of the step have landed (they are older than the stores of the last epilogue, which may stay
            //      in flight); behind the barrier everybody's have, and everybody is done with the other buffer
            if (pend) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pend = false;
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // ---- tickets.  The next job's: handed over through LDS in the last epilogue, read behind this job's first barrier.  The one
            //      after: thread 0 draws it at the start of this job's LAST step (returning atomic as inline asm: the compiler's own
            //      sequence waits vmcnt(0) on the spot) and waits for it at the end of that step, leaving the step's 9 DMA pieces in flight.
            if (ch == 0 && pending) {
                unsigned v;
                asm volatile("ds_read_b32 %0, %1 offset:8" : "=v"(v) : "v"(sched_a));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                const unsigned tk = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
                have_next = tk < (unsigned)cnt;
                if (have_next) nxt = job_params(job_of(tk));
                drawing = have_next;
                pending = false;
            }
            unsigned tk_mine = 0xFFFFFFFFu;
            const bool draw = drawing && ch + 1 == nch;
            if (draw && tid == 0) {
                if (ctr) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(tk_mine) : "v"((uint64_t)(uintptr_t)ctr), "v"(1u) : "memory");
                else tk_mine = (unsigned)(kfirst + (kstat++) * kstep);
            }
            const bool same = ch + 1 < nch;
            const Src nx = step_src(same ? cur : nxt, same ? ch + 1 : 0, same || have_next);
            const int nb_ = par ^ 1;

            // ---- 9 patch rows: row r = 2 o + kh feeds output row o with tap row kh
            const uint32_t wa = wa0 + par * DW_BYTES;
            const uint32_t p0 = pa0[0] + par * DP_BYTES, p1 = pa0[1] + par * DP_BYTES, p2 = pa0[2] + par * DP_BYTES;
            u32x4 W[9], P[3][3];
#define RDW(k) DSR128(W[k], wa, (k) * 4096)
#define RDP(set, r) DSR128(P[set][0], p0, (r) * DROW); DSR128(P[set][1], p1, (r) * DROW); DSR128(P[set][2], p2, (r) * DROW)
#define ROW1(set, o, kh) MMA(acc[o], W[3 * (kh) + 0], P[set][0]); MMA(acc[o], W[3 * (kh) + 1], P[set][1]); MMA(acc[o], W[3 * (kh) + 2], P[set][2])
#define ROW2(set, oa, ob) MMA(acc[oa], W[0], P[set][0]); MMA(acc[ob], W[6], P[set][0]); MMA(acc[oa], W[1], P[set][1]); \
                          MMA(acc[ob], W[7], P[set][1]); MMA(acc[oa], W[2], P[set][2]); MMA(acc[ob], W[8], P[set][2])
            RDW(0); RDW(1); RDW(2); RDP(0, 0);
            RDW(3); RDW(4); RDW(5); RDP(1, 1);
            RDW(6); RDW(7); RDW(8); RDP(2, 2);
            __builtin_amdgcn_s_setprio(1);
            if (!UNETRIR_ABL(abl, 4)) {
            LGKM_WAIT(12); ROW1(0, 0, 0);
            RDP(0, 3); dma_piece(nx, 0, nb_); dma_piece(nx, 1, nb_);
            LGKM_WAIT(9); ROW1(1, 0, 1);
            RDP(1, 4); dma_piece(nx, 2, nb_); dma_piece(nx, 3, nb_);
            LGKM_WAIT(6); ROW2(2, 1, 0);
            RDP(2, 5); dma_piece(nx, 4, nb_); dma_piece(nx, 5, nb_);
            LGKM_WAIT(6); ROW1(0, 1, 1);
            RDP(0, 6); dma_piece(nx, 6, nb_); dma_piece(nx, 7, nb_);
            LGKM_WAIT(6); ROW2(1, 2, 1);
            RDP(1, 7); dma_piece(nx, 8, nb_);
            LGKM_WAIT(6); ROW1(2, 2, 1);
            RDP(2, 8);
            LGKM_WAIT(6); ROW2(0, 3, 2);
            LGKM_WAIT(3); ROW1(1, 3, 1);
            LGKM_WAIT(0); ROW1(2, 3, 2);
            } else {
                LGKM_WAIT(0);
#pragma unroll
                for (int j = 0; j < 9; ++j) dma_piece(nx, j, nb_);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
#undef RDW
#undef RDP
#undef ROW1
#undef ROW2
            if (draw && tid < 64) {                    // wave 0: the ticket is older than the step's 9 DMA pieces
                if (tid == 0) {
                    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                    asm volatile("ds_write_b32 %0, %1 offset:8" :: "v"(sched_a), "v"(tk_mine) : "memory");
                }
            }
            par ^= 1;
        }

        // ---- epilogue: register r of acc[o] is D row (r & 3) + 8 (r >> 2) + 4 hi = channel perm(..) = 16 (r >> 3) + 8 hi + (r & 7)
        //      of pixel column l31: two 16-byte stores per output row
        const int nb = cur.n0 + 32 * nq;
        if (nb < a.N && !UNETRIR_ABL(abl, 2)) {        // wave-uniform (N % 32 == 0)
            u32x4 bq[4];
            const uint32_t bad = ba + (uint32_t)((nb + 8 * hi) * 4);
            DSR128(bq[0], bad, 0); DSR128(bq[1], bad, 16); DSR128(bq[2], bad, 64); DSR128(bq[3], bad, 80);
            LGKM_WAIT(0);
            float bias_[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) { const uint32_t u = bq[e >> 2][e & 3]; bias_[e] = __uint_as_float(u); }   // (a bit_cast of the vector element itself reads element 0)
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const size_t pix = ((size_t)cur.img * OH + cur.oy0 + 4 * rh + o) * OW + cur.ox0 + l31;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = acc[o][8 * h + e] + bias_[8 * h + e];
                    if (addend) {
                        const bf16x8 ad = *reinterpret_cast<const bf16x8*>(addend + pix * a.ldadd + nb + 16 * h + 8 * hi);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (float)(__bf16)v[e] + (float)ad[e];
                    }
                    bf16x8 ov;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ov[e] = (__bf16)v[e];
                    *reinterpret_cast<bf16x8*>(out + pix * a.ldo + nb + 16 * h + 8 * hi) = ov;
                }
            }
            pend = true;
        }
        if (!have_next) break;
        cur = nxt;
        pending = drawing;                            // whether a job follows `cur` is known behind its first barrier
        have_next = false;
    }
    leave();
}

bool conv3x3d_applies(const Conv3Args& a) {
    const size_t img_bytes = (((size_t)a.H * a.W - 1) * a.ldi + a.C) * 2, w_bytes = (size_t)((a.N + 127) / 128) * 128 * 9 * a.C * 2;
    return unetrir_cfg().conv3x3d && !a.colstat && (a.H & 1) == 0 && (a.W & 1) == 0 && (a.H / 2) % DTR == 0 && (a.W / 2) % DTC == 0 &&
           a.C >= 32 && a.C % 16 == 0 && a.N >= 32 && a.N % 32 == 0 && a.N <= DMAXN && a.ldi >= a.C && (a.ldi & 7) == 0 &&
           (a.ldo & 7) == 0 && (!a.addend || (a.ldadd & 7) == 0) && img_bytes < 0x70000000u && w_bytes < 0x70000000u;
}

int launch_conv3x3d_bf16(const Conv3Args& a, hipStream_t s) {
    const long long jobs = (long long)a.B * (a.H / 2 / DTR) * (a.W / 2 / DTC) * ((a.N + DNB - 1) / DNB);
    const int cus = 256;
    const int grid = (int)(jobs < cus ? jobs : cus);
    hipLaunchKernelGGL(conv3x3d_bf16_kernel, dim3((unsigned)grid), dim3(512), 0, s, a, (int)jobs, sched_slot(s), UNETRIR_ABL_HOST());
    return (int)hipGetLastError();
}
