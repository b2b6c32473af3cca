// conv3x3g.hip - bf16 3x3 stride-1 'same' convolution (forward / data gradient): LDS-DMA staged, loads in flight across
// barriers, patch rows reused from registers.
//
// Why a third 3x3 kernel: conv3x3.hip (register-staged, one barrier per tap, 16 MFMAs per wave between barriers) tops out
// near 0.9 PFLOP/s - the known ceiling of "stage -> vmcnt(0) -> barrier -> compute" loops on this chip.  This kernel keeps
// the staging loads in flight across the barriers instead:
//   * a workgroup is 8 waves (2 per SIMD, one workgroup per CU) and owns 16 image rows x 32 columns x 128 output channels;
//     a wave owns 4 rows x 32 columns x 64 channels (128 accumulator registers);
//   * K advances in chunks of 32 input channels.  The (16+2) x 34 pixel patch of a chunk is staged once (two buffers), the
//     [3 vertical taps][128 channels][32 input channels] weight tile of one horizontal tap dx per step (ring of three);
//   * everything is staged with buffer_load_dwordx4 ... lds (no staging VGPRs, no ds_write); the LDS images are lane-linear and
//     the 16-byte granules of a 64-byte row are XOR-swizzled on the SOURCE address and again on the fragment read;
//     out-of-image pixels carry a buffer offset past num_records and read zeros;
//   * one raw s_barrier per step, preceded by a COUNTED s_waitcnt vmcnt(N): the weight tile of step s+2 and the patch of
//     the next chunk stay in flight while step s computes.  Fragment reads are inline-asm ds_read_b128, so the compiler
//     does not drain the DMA queue in front of them (it waits vmcnt(0) before any LDS load it can see while an LDS-DMA
//     is outstanding);
//   * for a fixed dx a patch-row fragment feeds the three vertical taps and a weight fragment feeds four output rows:
//     12 fragment reads per 24 v_mfma_f32_32x32x16_bf16.
// Requires C % 32 == 0 (the launcher falls back to conv3x3.hip otherwise).  The data gradient is the same kernel with
// flipped taps on the [Cin][9][Cout] weight copy.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "kernels.h"

// The epilogue staging tile is private to a wave: LDS operations of one wave execute in issue order, so its reads see its
// own earlier writes without a workgroup barrier; this only stops the compiler from moving LDS accesses across the point.
#define WAVE_LDS_FENCE() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LGKM_WAIT(n) do { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define MMA16(accv, wfrag, pfrag) \
    accv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfrag), __builtin_bit_cast(bf16x8, pfrag), accv, 0, 0, 0)
#define MMA(accv, wfrag, pfrag) \
    accv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wfrag), __builtin_bit_cast(bf16x8, pfrag), accv, 0, 0, 0)

namespace {
constexpr int GPC = 34;                    // patch columns
constexpr int GTR = 16;                    // tile rows
constexpr int GNPX = (GTR + 2) * GPC;      // 612 patch pixels
constexpr int GP_INSTR = (GNPX + 15) / 16; // 39 wave-instructions of 16 pixels x 64 B
constexpr int GP_BYTES = GP_INSTR * 1024;  // 39936
constexpr int GBN = 128;
constexpr int GW_BYTES = 3 * GBN * 64;     // 24576 = 24 wave-instructions
constexpr int GP_PER_WAVE = 5;             // ceil(39 / 8)
constexpr int GW_PER_WAVE = 3;             // 24 / 8
constexpr int GSROW = 64 * 2 + 16;
constexpr int GSMEM = 2 * GP_BYTES + 3 * GW_BYTES;   // 153600
}  // namespace

// PAIR (images at most 16 pixels wide, the 16 x 16 level): a tile is 16 rows x 16 columns of TWO consecutive images.  Their
// 18 x 18 patches are stacked in LDS (36 rows of 18 pixels), so the second 16-pixel half of a fragment row is the other
// image, one constant offset away, and nothing else in the pipeline changes - full tiles where the 32-column tile would be
// half empty.
// BN = 64 (paired tiles only): a workgroup owns 64 output channels and its eight waves 2 tile rows each - twice the workgroups
// where 128-channel tiles leave half of the CUs without one (1024 -> 1024 at 16 x 16, batch 32: 128 tiles)
template <int VAR, bool PAIR = false, int BN = GBN>
__global__ __launch_bounds__(512) void conv3x3g_bf16_kernel(const Conv3Args a) {
    constexpr int RPW = BN == 128 ? 4 : 2;                 // tile rows per wave
    constexpr int WB = 3 * BN * 64;                        // one kernel tile: [3 vertical taps][BN channels][32 input channels]
    constexpr int W_INSTR = WB / 1024;                     // 24 / 12 wave-instructions
    constexpr int WPW = (W_INSTR + 7) / 8;                 // 3 / 2 per wave
    static_assert(BN == 128 || (BN == 64 && (VAR & 2)), "64-channel tiles exist for the 16x16x32 body");
    constexpr int PC = PAIR ? 18 : GPC;                    // patch columns
    constexpr int NPX = PAIR ? 36 * 18 : GNPX;             // patch pixels
    constexpr int P_INSTR = (NPX + 15) / 16;               // 41 / 39 wave-instructions
    constexpr int P_BYTES = P_INSTR * 1024;
    constexpr int P_PER_WAVE = (P_INSTR + 7) / 8;
    constexpr int RP = PC * 64;                            // patch row pitch, bytes
    constexpr int HO = PAIR ? 18 * RP : 1024;              // second half of a fragment row: the other image / 16 columns on
    constexpr int SMEM = 2 * P_BYTES + 3 * WB;             // 157696 / 153600 (BN = 64: 120832)
    static_assert(!PAIR || (VAR & 2), "the paired tile exists for the 16x16x32 body only");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];
    const __bf16* __restrict__ in = (const __bf16*)a.in;
    const __bf16* __restrict__ w = (const __bf16*)a.w;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = BN == 128 ? wave >> 1 : wave, wn = BN == 128 ? wave & 1 : 0;
    const int l31 = lane & 31, hi = lane >> 5;
    const int l15 = lane & 15, lq = lane >> 4;
    constexpr bool S16 = (VAR & 2) != 0;              // v_mfma_f32_16x16x32_bf16 body

    const int tiles_x = PAIR ? 1 : (a.W + 31) / 32, tiles_y = (a.H + GTR - 1) / GTR;
    const int ntN = (a.N + BN - 1) / BN;
    int id = blockIdx.x;
    if ((gridDim.x & 7) == 0) id = (id & 7) * (gridDim.x >> 3) + (id >> 3);   // neighbouring tiles on one XCD (shared L2)
    const int nt = id % ntN; id /= ntN;
    const int tx = id % tiles_x; id /= tiles_x;
    const int ty = id % tiles_y;
    const int img = id / tiles_y;                          // PAIR: index of the image pair
    const int img0 = PAIR ? 2 * img : img;
    const int nimg = PAIR ? (a.B - img0 < 2 ? a.B - img0 : 2) : 1;
    const int y0 = ty * GTR, x0 = tx * 32, n0 = nt * BN;
    const int C = a.C;
    const int nch = C / 32;
    const int ldw = 9 * C;

    // ---- per-lane DMA sources (chunk / step invariant part): byte offsets into two raw buffers (this image / the weights);
    // invalid lanes (halo outside the image, channels >= N) carry an offset past num_records and read zeros
    constexpr uint32_t OOB = 0xF0000000u;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(in + (size_t)img0 * a.H * a.W * a.ldi), (short)0, (int)((((size_t)nimg * a.H * a.W - 1) * a.ldi + C) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)w, (short)0, (int)((size_t)a.N * ldw * 2), 0x00020000);
    const int slot = lane & 3, sub = lane >> 2;
    uint32_t pa[6];                                   // (fixed bounds: arrays of a template-dependent size captured by the
    int pi[6];                                        //  lambdas below lose the kernel's host stub with this compiler)
    static_assert(P_PER_WAVE <= 6, "patch instructions per wave");
#pragma unroll
    for (int j = 0; j < P_PER_WAVE; ++j) {
        int i = wave + 8 * j;
        if (i > P_INSTR - 1) i = P_INSTR - 1;            // the last waves repeat the final instruction (uniform DMA counts)
        pi[j] = i;
        const int p = 16 * i + sub;
        const int pr = p / PC, pc = p - pr * PC;
        const int im = PAIR ? pr / 18 : 0;               // PAIR: patch rows 18..35 belong to the second image
        const int gs = S16 ? slot ^ ((pc & 4) >> 1) : slot ^ ((pc >> 2) & 3);
        const int iy = y0 - 1 + pr - 18 * im, ix = x0 - 1 + pc;
        const bool ok = p < NPX && im < nimg && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        pa[j] = ok ? (uint32_t)((((im * a.H + iy) * a.W + ix) * a.ldi + gs * 8) * 2) : OOB;
    }
    uint32_t wp[3];                                   // (fixed bounds: see pa / pi above)
    int wi[3];
    static_assert(WPW <= 3, "kernel-tile instructions per wave");
    const int dxs = ((a.flip & 1) ? -C : C) * 2;        // weight-tap step per dx, bytes
#pragma unroll
    for (int j = 0; j < WPW; ++j) {
        int i = wave + 8 * j;
        if (i > W_INSTR - 1) i = W_INSTR - 1;          // (BN = 64: the last waves repeat the final instruction - uniform DMA counts)
        wi[j] = i;
        const int row = 16 * i + sub;                  // dy * BN + local channel
        const int dy = row / BN, nl = row % BN;
        const int gs = S16 ? slot ^ ((nl & 4) >> 1) : slot ^ ((nl >> 2) & 3);
        const int n = n0 + nl;
        const int tap0 = (a.flip & 1) ? 8 - 3 * dy : 3 * dy;
        wp[j] = n < a.N ? (uint32_t)((n * ldw + tap0 * C + gs * 8) * 2) : OOB;
    }
    auto issue_p = [&](int ch) {
        unsigned char* dst = smem + (ch & 1) * P_BYTES;
        const uint32_t c0b = ch * 64;
#pragma unroll
        for (int j = 0; j < P_PER_WAVE; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lptr_t)(dst + pi[j] * 1024), 16, pa[j] + c0b, 0, 0, 0);
    };
    auto issue_w = [&](int ch, int dx, int buf) {
        unsigned char* dst = smem + 2 * P_BYTES + buf * WB;
        const uint32_t off = ch * 64 + dx * dxs;
#pragma unroll
        for (int j = 0; j < WPW; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(dst + wi[j] * 1024), 16, wp[j] + off, 0, 0, 0);
    };

    f32x16 acc[S16 ? 1 : 4][S16 ? 1 : 2];             // 32x32x16 body: [image row][32-channel tile]
    f32x4 acc16[S16 ? RPW : 1][2][4];                 // 16x16x32 body: [image row][16-pixel half][16-channel tile]
    if constexpr (S16) {
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc16[i][h][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int i = 0; i < (S16 ? 1 : 4); ++i)
#pragma unroll
            for (int j = 0; j < (S16 ? 1 : 2); ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }

    // ---- prologue: patch 0, weight steps 0 and 1
    issue_p(0);
    issue_w(0, 0, 0);
    issue_w(0, 1, 1);
    if constexpr (WPW == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    // fragment addresses: weights  row (dy*128 + wn*64 + j*32 + l31) * 64 + ((kk*2+hi)*16 ^ swz(l31))
    //                     patch    ((4*wm + r) * 34 + l31 + dx) * 64 + ((kk*2+hi)*16 ^ swz(l31 + dx))
    const uint32_t b_lane = lds0 + 2 * P_BYTES + (wn * 64 + l31) * 64;      // (32x32x16 body: BN = 128 only)
    const uint32_t b_swz = (l31 & 12) << 2;
    const uint32_t a_lane = lds0 + (4 * wm * PC + l31) * 64;

    for (int ch = 0; ch < nch; ++ch) {
        const bool more = ch + 1 < nch;
        const uint32_t a_chunk = a_lane + (ch & 1) * P_BYTES;
#pragma unroll 1
        for (int dx = 0; dx < 3; ++dx) {
            // ---- prefetch: weight tile of step s+2 (ring slot (dx+2)%3), patch of the next chunk
            if constexpr (!(VAR & 16)) {              // (VAR bits 2..4: timing ablations only, results are invalid)
            if (dx == 0) {
                issue_w(ch, 2, 2);
                if (more) issue_p(ch + 1);
            } else if (more) {
                issue_w(ch + 1, dx - 1, dx - 1);
            }
            }
            const uint32_t a_dx = a_chunk + dx * 64;
            const uint32_t a_swz = ((l31 + dx) & 12) << 2;
            const uint32_t b_buf = b_lane + dx * WB;
            const uint32_t b_buf16 = lds0 + 2 * P_BYTES + dx * WB + (wn * 64 + l15) * 64;
            if constexpr (S16) {
                // one pass over the 32-channel chunk: lane (l15, lq) reads granule lq of row l15 (weights: channel, patch: pixel)
                const uint32_t ba = b_buf16 + ((lq << 4) ^ ((l15 & 4) << 3));
                const uint32_t aa = lds0 + (ch & 1) * P_BYTES + (RPW * wm * PC + l15 + dx) * 64 + ((lq << 4) ^ (((l15 + dx) & 4) << 3));
                u32x4 wf[3][4], pf[6][2];
#define RDW(dy) DSR128(wf[dy][0], ba, dy * (BN * 64) + 0); DSR128(wf[dy][1], ba, dy * (BN * 64) + 1024); \
                DSR128(wf[dy][2], ba, dy * (BN * 64) + 2048); DSR128(wf[dy][3], ba, dy * (BN * 64) + 3072)
#define RDP(r) DSR128(pf[r][0], aa, r * RP + 0); DSR128(pf[r][1], aa, r * RP + HO)
#define ROWS16(r)                                                                              \
    _Pragma("unroll") for (int dy = 0; dy < 3; ++dy) {                                         \
        if (r - dy < 0 || r - dy > RPW - 1) continue;                                          \
        _Pragma("unroll") for (int h = 0; h < 2; ++h)                                          \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) MMA16(acc16[r - dy][h][t], wf[dy][t], pf[r][h]); \
    }
                if constexpr (RPW == 4) {
                RDW(0); RDP(0); RDW(1); RDP(1); RDW(2); RDP(2);          // 18 reads in flight
                __builtin_amdgcn_s_setprio(1);
                LGKM_WAIT(12); ROWS16(0);
                RDP(3);
                LGKM_WAIT(8); ROWS16(1);
                RDP(4);
                LGKM_WAIT(4); ROWS16(2);
                RDP(5);
                LGKM_WAIT(4); ROWS16(3);
                LGKM_WAIT(2); ROWS16(4);
                LGKM_WAIT(0); ROWS16(5);
                } else {                                                 // two tile rows: patch rows 0..3
                RDW(0); RDP(0); RDW(1); RDP(1); RDW(2); RDP(2); RDP(3);  // 20 reads in flight
                __builtin_amdgcn_s_setprio(1);
                LGKM_WAIT(14); ROWS16(0);
                LGKM_WAIT(8); ROWS16(1);
                LGKM_WAIT(2); ROWS16(2);
                LGKM_WAIT(0); ROWS16(3);
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
#undef RDW
#undef RDP
#undef ROWS16
            } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const uint32_t g16 = (kk * 2 + hi) << 4;
                const uint32_t ba = b_buf + (g16 ^ b_swz);
                const uint32_t aa = a_dx + (g16 ^ a_swz);
                u32x4 w00, w01, w10, w11, w20, w21, p0, p1, p2, p3, p4, p5;
                DSR128(w00, ba, 0 * 8192 + 0);
                DSR128(w01, ba, 0 * 8192 + 2048);
                DSR128(p0, aa, 0 * RP);
                DSR128(w10, ba, 1 * 8192 + 0);
                DSR128(w11, ba, 1 * 8192 + 2048);
                DSR128(p1, aa, 1 * RP);
                DSR128(w20, ba, 2 * 8192 + 0);
                DSR128(w21, ba, 2 * 8192 + 2048);
                DSR128(p2, aa, 2 * RP);
                DSR128(p3, aa, 3 * RP);
                DSR128(p4, aa, 4 * RP);
                DSR128(p5, aa, 5 * RP);
                // counted waits: the reads return in issue order, each MFMA group starts as soon as its fragments are in
                __builtin_amdgcn_s_setprio(1);
                if constexpr (VAR & 1) LGKM_WAIT(9); else if constexpr (9 == 9) LGKM_WAIT(0);
                MMA(acc[0][0], w00, p0); MMA(acc[0][1], w01, p0);
                if constexpr (VAR & 1) LGKM_WAIT(6); else if constexpr (6 == 9) LGKM_WAIT(0);
                MMA(acc[1][0], w00, p1); MMA(acc[1][1], w01, p1);
                MMA(acc[0][0], w10, p1); MMA(acc[0][1], w11, p1);
                if constexpr (VAR & 1) LGKM_WAIT(3); else if constexpr (3 == 9) LGKM_WAIT(0);
                MMA(acc[2][0], w00, p2); MMA(acc[2][1], w01, p2);
                MMA(acc[1][0], w10, p2); MMA(acc[1][1], w11, p2);
                MMA(acc[0][0], w20, p2); MMA(acc[0][1], w21, p2);
                if constexpr (VAR & 1) LGKM_WAIT(2); else if constexpr (2 == 9) LGKM_WAIT(0);
                MMA(acc[3][0], w00, p3); MMA(acc[3][1], w01, p3);
                MMA(acc[2][0], w10, p3); MMA(acc[2][1], w11, p3);
                MMA(acc[1][0], w20, p3); MMA(acc[1][1], w21, p3);
                if constexpr (VAR & 1) LGKM_WAIT(1); else if constexpr (1 == 9) LGKM_WAIT(0);
                MMA(acc[3][0], w10, p4); MMA(acc[3][1], w11, p4);
                MMA(acc[2][0], w20, p4); MMA(acc[2][1], w21, p4);
                LGKM_WAIT(0);
                MMA(acc[3][0], w20, p5); MMA(acc[3][1], w21, p5);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            // ---- retire what the next step reads; younger DMAs stay in flight across the barrier
            if constexpr (!(VAR & 8)) {
            // (counts: a kernel tile is WPW = 3 or 2 instructions per wave, a patch P_PER_WAVE = 5 or 6)
#define VM_WP() do { if constexpr (P_PER_WAVE + WPW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); \
                     else if constexpr (P_PER_WAVE + WPW == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); \
                     else asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); } while (0)
#define VM_W() do { if constexpr (WPW == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); } while (0)
            static_assert(P_PER_WAVE + WPW >= 7 && P_PER_WAVE + WPW <= 9, "vmcnt immediates");
            if (dx == 0) {
                if (more) VM_WP();                                            // W(s+2) + P(ch+1) may remain
                else VM_W();                                                  // W(s+2) may remain
            } else if (dx == 1) {
                if (more) VM_WP();                                            // P(ch+1) + W(s+2) may remain
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (more) VM_W();                                             // W(s+2) may remain; P(ch+1) is older: retired
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#undef VM_WP
#undef VM_W
            }
            if constexpr (!(VAR & 4)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    if constexpr ((VAR & 28) != 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }

    // ---- epilogue through LDS (all DMAs retired, every wave past the last barrier), two image rows of the wave at a time:
    // acc[i][j] holds D[n = 32j + (r&3) + 8(r>>2) + 4*hi][pixel column = l31] of image row y0 + 4*wm + i.
    unsigned char* stage = smem + wave * (64 * GSROW);
    const int cq = lane & 7, pl = lane >> 3;
    const int nq = n0 + wn * 64 + cq * 8;
    __bf16* __restrict__ out = (__bf16*)a.out;
    const __bf16* __restrict__ addend = (const __bf16*)a.addend;
    float cs_s[8], cs_q[8];                           // fused column statistics of this lane's 8 channels (a.colstat)
#pragma unroll
    for (int e = 0; e < 8; ++e) { cs_s[e] = 0.f; cs_q[e] = 0.f; }
#pragma unroll
    for (int half = 0; half < RPW / 2; ++half) {
        if (half) WAVE_LDS_FENCE();
        if constexpr (S16) {
            // acc16[i][h][t][e] = D[n = 16t + 4*lq + e][pixel column = 16h + l15] of image row y0 + 4*wm + i
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int nl = 16 * t + 4 * lq;
                const int n = n0 + wn * 64 + nl;
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.bias && n + 3 < a.N) bv = *reinterpret_cast<const float4*>(a.bias + n);
                else if (a.bias) { float* bp = &bv.x; for (int e = 0; e < 4; ++e) if (n + e < a.N) bp[e] = a.bias[n + e]; }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x4& c = acc16[S16 ? 2 * half + i : 0][h][t];
                        bf16x4 o;
                        o[0] = (__bf16)(c[0] + bv.x); o[1] = (__bf16)(c[1] + bv.y);
                        o[2] = (__bf16)(c[2] + bv.z); o[3] = (__bf16)(c[3] + bv.w);
                        *reinterpret_cast<bf16x4*>(stage + (32 * i + 16 * h + l15) * GSROW + nl * 2) = o;
                    }
            }
        } else
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int nl = 32 * j + 8 * qd + 4 * hi;
                const int n = n0 + wn * 64 + nl;
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.bias && n + 3 < a.N) bv = *reinterpret_cast<const float4*>(a.bias + n);
                else if (a.bias) { float* bp = &bv.x; for (int e = 0; e < 4; ++e) if (n + e < a.N) bp[e] = a.bias[n + e]; }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const f32x16& c = acc[S16 ? 0 : 2 * half + i][S16 ? 0 : j];
                    bf16x4 o;
                    o[0] = (__bf16)(c[4 * qd + 0] + bv.x); o[1] = (__bf16)(c[4 * qd + 1] + bv.y);
                    o[2] = (__bf16)(c[4 * qd + 2] + bv.z); o[3] = (__bf16)(c[4 * qd + 3] + bv.w);
                    *reinterpret_cast<bf16x4*>(stage + (32 * i + l31) * GSROW + nl * 2) = o;
                }
            }
        }
        WAVE_LDS_FENCE();
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int p = ps * 8 + pl;
            const int y = y0 + RPW * wm + 2 * half + (p >> 5), x = PAIR ? (p & 15) : x0 + (p & 31);
            const int imo = PAIR ? img0 + ((p >> 4) & 1) : img;          // PAIR: the second 16-pixel half is the second image
            if (y >= a.H || x >= a.W || nq >= a.N || imo >= a.B) continue;
            uint4 v = *reinterpret_cast<const uint4*>(stage + p * GSROW + cq * 16);
            const size_t pix = ((size_t)imo * a.H + y) * a.W + x;
            if (addend) {
                const bf16x8 ad = *reinterpret_cast<const bf16x8*>(addend + pix * a.ldadd + nq);
                bf16x8 vv = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) vv[e] = (__bf16)((float)vv[e] + (float)ad[e]);
                v = __builtin_bit_cast(uint4, vv);
            }
            if (a.colstat) {
                const bf16x8 sv = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)sv[e]; cs_s[e] += f; cs_q[e] += f * f; }
            }
            *reinterpret_cast<uint4*>(out + pix * a.ldo + nq) = v;
        }
    }
    if (a.colstat) {
        // lanes pl = 0..7 of a wave hold the same 8 channels: fold them, then the 4 row groups (wm) in a fixed order
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) { cs_s[e] += __shfl_xor(cs_s[e], off); cs_q[e] += __shfl_xor(cs_q[e], off); }
        }
        float* red = reinterpret_cast<float*>(smem + 8 * 64 * GSROW);      // [4 wm][128 ch][2], past the staging tiles
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[((wm * BN) + wn * 64 + lane * 8 + e) * 2 + 0] = cs_s[e];
                red[((wm * BN) + wn * 64 + lane * 8 + e) * 2 + 1] = cs_q[e];
            }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int ch = tid >> 1, st = tid & 1;
            float t = ((red[(0 * BN + ch) * 2 + st] + red[(1 * BN + ch) * 2 + st]) + red[(2 * BN + ch) * 2 + st]) +
                      red[(3 * BN + ch) * 2 + st];
            if constexpr (BN == 64)            // eight row groups
                t = (((t + red[(4 * BN + ch) * 2 + st]) + red[(5 * BN + ch) * 2 + st]) + red[(6 * BN + ch) * 2 + st]) + red[(7 * BN + ch) * 2 + st];
            const size_t row = ((size_t)img * tiles_y + ty) * tiles_x + tx;
            if (n0 + ch < a.N) a.colstat[(row * a.N + n0 + ch) * 2 + st] = t;
        }
    }
}

// true when the LDS-DMA kernel takes this layer (bf16, C a multiple of 32, more than 64 output channels)
bool conv3x3g_applies(const Conv3Args& a) {
    const size_t img_bytes = (((size_t)a.H * a.W - 1) * a.ldi + a.C) * 2, w_bytes = (size_t)a.N * 9 * a.C * 2;
    // N = 64 with more than 64 input channels (the 128 -> 64 layer behind the first skip concat): 64-channel tiles, launch_conv3x3g_bf16
    return a.C % 32 == 0 && (a.N > 64 || (a.N == 64 && a.C > 64)) && !(a.flip & 2) && img_bytes < 0x70000000u && w_bytes < 0x70000000u;
}

// images at most 16 pixels wide: the paired-image tile (full tiles where the 32-column tile would be half empty)
bool conv3x3g_pair_applies(const Conv3Args& a) {
    // config switch conv3x3g_pair: 0 = off, 2 = whenever the shape allows (tests), 1 (default) = when the paired tiles give at least
    // 32 workgroups, counted in the 64-channel tiles the launcher falls back to below 256 workgroups.  (Round 1 asked for 128
    // workgroups of 128 channels: 512 -> 512 had 64 and lost 95 : 85 us against the tap-table kernel.  With 64-channel tiles,
    // round 3, batch 32 at 16 x 16: 512 -> 512 61 : 83 us, 256 -> 256 31 : 46, 128 -> 128 18 : 27.)
    const int mode = unetrir_cfg().conv3x3g_pair;
    const size_t pair_bytes = (((size_t)2 * a.H * a.W - 1) * a.ldi + a.C) * 2;
    if (mode == 0 || !conv3x3g_applies(a) || a.W > 16 || a.B < 2 || pair_bytes >= 0x70000000u) return false;
    const long long ptiles = (long long)((a.B + 1) / 2) * ((a.H + GTR - 1) / GTR);
    const long long wgs = ptiles * ((a.N & 63) == 0 ? a.N / 64 : (a.N + GBN - 1) / GBN);
    return mode == 2 || wgs >= 32;
}

long long conv3x3g_colstat_rows(const Conv3Args& a) {
    const long long ty = (a.H + GTR - 1) / GTR;
    return conv3x3g_pair_applies(a) ? (long long)((a.B + 1) / 2) * ty : (long long)a.B * ty * ((a.W + 31) / 32);
}

int launch_conv3x3g_bf16(const Conv3Args& a, hipStream_t s) {
    if (conv3x3g_pair_applies(a)) {
        const long long ptiles = (long long)((a.B + 1) / 2) * ((a.H + GTR - 1) / GTR);
        const long long tiles = ptiles * ((a.N + GBN - 1) / GBN);
        if (tiles < 256 && (a.N & 63) == 0) {             // half of the CUs would stay empty: 64-channel tiles, twice the workgroups
            hipLaunchKernelGGL((conv3x3g_bf16_kernel<2, true, 64>), dim3((unsigned)(ptiles * (a.N / 64))), dim3(512), 0, s, a);
            return (int)hipGetLastError();
        }
        hipLaunchKernelGGL((conv3x3g_bf16_kernel<2, true>), dim3((unsigned)tiles), dim3(512), 0, s, a);
        return (int)hipGetLastError();
    }
    if (a.N <= 64) {                                          // one 64-channel tile per pixel tile
        const long long t64 = (long long)a.B * ((a.H + GTR - 1) / GTR) * ((a.W + 31) / 32);
        hipLaunchKernelGGL((conv3x3g_bf16_kernel<2, false, 64>), dim3((unsigned)t64), dim3(512), 0, s, a);
        return (int)hipGetLastError();
    }
    const long long tiles = (long long)a.B * ((a.H + GTR - 1) / GTR) * ((a.W + 31) / 32) * ((a.N + GBN - 1) / GBN);
    hipLaunchKernelGGL(conv3x3g_bf16_kernel<2>, dim3((unsigned)tiles), dim3(512), 0, s, a);
    return (int)hipGetLastError();
}
