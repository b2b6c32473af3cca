// conv3x3p.hip - the PERSISTENT form of conv3x3g.hip (bf16 3x3 stride-1 'same' convolution, forward / data gradient, more
// than 64 output channels: dl_models/u_net.py:366, :309 and their data gradients).
//
// conv3x3g runs one workgroup per 16 x 32 pixel x 128 channel tile.  Its 150 KB of LDS allow one workgroup per CU, so the
// tiles of a CU run strictly one after the other and every tile pays its prologue (first patch + two kernel tiles through
// HBM / L2 latency before the first MFMA) and its epilogue (accumulators through an LDS staging tile, stores drained at
// s_endpgm) with an idle matrix pipe: 9 of 39 us per tile on the 128 x 128 layers (12 K steps per tile), 11 of 26 us on
// the 256 x 256 data gradient (6 steps).  Here ONE workgroup per CU walks through its tiles and the K loop never stops:
//   * the step stream (tile, 32-channel chunk, horizontal tap) is continuous: the patch of the NEXT TILE's first chunk and
//     its first kernel tiles are requested during the last chunk of the current tile, exactly as a next chunk's are
//     (same ring, same counted vmcnt waits);
//   * the epilogue needs no LDS (which the prefetch now owns): the kernel rows of a wave's 64 channels are permuted in LDS
//     so that a lane's accumulators are 2 x 8 consecutive channels of its pixels - 16-byte stores straight from the
//     accumulators, left in flight across the next tile's first barrier (counted vmcnt);
//   * tiles are dealt to the XCDs in contiguous ranges and inside an XCD round-robin, channel tile fastest: the workgroups
//     that read the same patch run at the same time on one XCD, and a workgroup keeps ONE channel tile for all its tiles
//     (its kernel tiles stay hot in L2, and its fused column statistics add up per lane).
// K loop, LDS images, swizzles, ring and waits are those of conv3x3g.hip (16x16x32 body); see there.  Fused column
// statistics: one row per pixel tile, as conv3x3g writes them (which workgroup serves a tile varies from run to run; the rows do not).
// Requires C % 32 == 0, N / 128 tiles in {1, 2, 4, 8}, at least 512 tiles; otherwise conv3x3g.
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
// sum over the 16 lanes of a DPP row, every lane gets the total: the xor-butterfly order (1, 2, 4, 8) on single vector
// instructions (quad permutes, then the mirrored half / whole row: after two steps the lanes of a quad hold the same value)
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_sum16(float v) {
    v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);     // row_half_mirror
    return dpp_add<0x140>(v);  // row_mirror
}
constexpr int PPC = 34;                     // patch columns
constexpr int PTR = 16;                     // tile rows
constexpr int PNPX = (PTR + 2) * PPC;       // 612 patch pixels
constexpr int PP_INSTR = (PNPX + 15) / 16;  // 39 wave-instructions of 16 pixels x 64 B
constexpr int PP_BYTES = PP_INSTR * 1024;   // 39936
constexpr int PBN = 128;
constexpr int PW_BYTES = 3 * PBN * 64;      // 24576 = 24 wave-instructions: [3 vertical taps][128 channels][32 input channels]
constexpr int PRP = PPC * 64;               // patch row pitch
constexpr int PRING = 2 * PP_BYTES + 3 * PW_BYTES;     // 153600
constexpr int PRED = PRING;                 // column-statistics scratch: 2 buffers x [4 row groups][128 channels][2] floats
constexpr int PRED_BYTES = 4 * 128 * 2 * 4;
constexpr int PBIAS = PRED + 2 * PRED_BYTES;           // bias of the workgroup's 128 channels
constexpr int PSCHED = PBIAS + 128 * 4;                // tile tickets handed from thread 0 to the workgroup
constexpr int PSMEM = PSCHED + 16;                     // 158224
constexpr uint32_t OOB = 0xF0000000u;

struct Tile { int img, y0, x0, n0, pix; };
}  // namespace

#ifdef UNETRIR_ABLATIONS
// in-kernel clock stamps (ablation build, bit 2048): per workgroup (shader-clock ticks, 100 MHz ticks) over the whole persistent
// loop; they go to this array only, no output depends on them (MI355X_MICROARCH.md, DVFS give-back item 6)
__device__ unsigned long long g_stamps_conv3x3p[256][2];
extern "C" int unetrir_abl_stamps_conv3x3p(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_conv3x3p), (size_t)n * 16);
}
#endif
// abl (ablation build only): 1 no patch DMA, 8 no kernel DMA (after the prologue), 2 no output stores, 4 no MFMA loop, 2048 clock stamps
__global__ __launch_bounds__(512) void conv3x3p_bf16_kernel(const Conv3Args a, int pix_tiles, int per_xcd, unsigned* sched, int abl) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[PSMEM];
    const __bf16* __restrict__ in = (const __bf16*)a.in;
#ifdef UNETRIR_ABLATIONS
    unsigned long long st0 = 0, sr0 = 0;
    if (UNETRIR_ABL(abl, 2048)) { st0 = __builtin_amdgcn_s_memtime(); sr0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lq = lane >> 4;
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + PTR - 1) / PTR;
    const int ntN = (a.N + PBN - 1) / PBN;
    const int C = a.C, nch = C / 32, ldw = 9 * C;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;

    // ---- tiles of this workgroup.  Job index = pixel tile * ntN + channel tile; XCD x (= blockIdx & 7) owns per_xcd pixel
    // tiles; its 32 workgroups take jobs slot, slot + 32, ...: ntN divides 32, so a workgroup keeps channel tile slot % ntN.
    //
    // WHICH pixel tile of its XCD range a workgroup takes next is decided at run time: the 32 / ntN workgroups of a group (XCD,
    // channel tile) draw tickets from one counter (kernels.h, SchedSlot).  With a fixed assignment a workgroup that cannot be placed
    // at once - a CU held by another kernel: a collective beside the backward pass, a weight-gradient kernel of the side
    // stream - leaves its tiles for a second round after everybody else has finished (measured with 32 of 256 CUs held:
    // 1.6x the launch time; the per-tile kernel: 1.0-1.26x).  With tickets the workgroups that do run share all tiles.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nt = slot % ntN, n0 = nt * PBN;
    const int pt0 = xcd * per_xcd, cnt = max(0, min(pix_tiles, (xcd + 1) * per_xcd) - pt0);
    unsigned* ctr = sched ? sched + xcd * 8 + nt : nullptr;
    int kstat = 0;                                        // sched == nullptr: the fixed assignment q, q + 32 / ntN, ...
    auto take = [&]() -> unsigned {                       // thread 0 only
        if (ctr) return __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (unsigned)(slot / ntN + (kstat++) * (32 / ntN));
    };
    auto tile_of = [&](unsigned ticket) {
        Tile t;
        int id = pt0 + (int)ticket;
        const int tx = id % tiles_x; id /= tiles_x;
        const int ty = id % tiles_y;
        t.img = id / tiles_y; t.y0 = ty * PTR; t.x0 = tx * 32; t.n0 = n0; t.pix = pt0 + (int)ticket;
        return t;
    };

    // ---- per-lane DMA sources.  Patch: instruction i covers LDS pixels 16 i .. 16 i + 15, lane = (pixel sub, 16-byte slot)
    const int slot4 = lane & 3, sub = lane >> 2;
    int prc[5];                                          // patch row | column << 16 (row 0xFFFF: past the patch, never valid)
    uint32_t prel[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        int i = wave + 8 * j;
        if (i > PP_INSTR - 1) i = PP_INSTR - 1;          // the last waves repeat the final instruction (uniform DMA counts)
        const int p = 16 * i + sub;
        const int pr = p / PPC, pc = p - pr * PPC;
        const int gs = slot4 ^ ((pc & 4) >> 1);
        prc[j] = (p < PNPX ? pr : 0xFFFF) | (pc << 16);
        prel[j] = (uint32_t)(((pr * a.W + pc) * a.ldi + gs * 8) * 2);
    }
    // kernel tile: LDS row (dy * 128 + nl); within the 64 channels of a wave column (nl >> 6) the 16-row MFMA tile t = (nl >> 4) & 3,
    // row r = nl & 15 holds channel 32 (t >> 1) + 8 (r >> 2) + 4 (t & 1) + (r & 3): a lane then owns 8 consecutive channels
    uint32_t wp[3];
    const int dxs = ((a.flip & 1) ? -C : C) * 2;         // kernel-tap step per dx, bytes
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int row = 16 * (wave + 8 * j) + sub;
        const int dy = row >> 7, nl = row & 127;
        const int gs = slot4 ^ ((nl & 4) >> 1);
        const int t = (nl >> 4) & 3, r = nl & 15;
        const int n = n0 + (nl & 64) + 32 * (t >> 1) + 8 * (r >> 2) + 4 * (t & 1) + (r & 3);
        const int tap0 = (a.flip & 1) ? 8 - 3 * dy : 3 * dy;
        wp[j] = n < a.N ? (uint32_t)((n * ldw + tap0 * C + gs * 8) * 2) : OOB;
    }
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, (int)((size_t)a.N * ldw * 2), 0x00020000);
    const int in_rec = (int)((((size_t)a.H * a.W - 1) * a.ldi + C) * 2);
    const size_t img_elems = (size_t)a.H * a.W * a.ldi;

    // (the per-lane offsets of a tile's patch are recomputed at every issue: 5 x 6 vector instructions per chunk against
    //  10 registers held through the K loop - the loop sits at the register limit)
    bool started = false;
    auto issue_p = [&](const __amdgpu_buffer_rsrc_t& rs, const Tile& t, int ch, int buf) {
        if (UNETRIR_ABL(abl, 1) && started) return;
        unsigned char* dst = smem + buf * PP_BYTES;
        const int base = (((t.y0 - 1) * a.W + t.x0 - 1) * a.ldi) * 2 + ch * 64;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            int i = wave + 8 * j;
            if (i > PP_INSTR - 1) i = PP_INSTR - 1;
            const int iy = t.y0 - 1 + (prc[j] & 0xFFFF), ix = t.x0 - 1 + (prc[j] >> 16);
            const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + i * 1024), 16, ok ? (uint32_t)(base + (int)prel[j]) : OOB, 0, 0, 0);
        }
    };
    auto issue_w = [&](int ch, int dx, int buf) {
        if (UNETRIR_ABL(abl, 8) && started) return;
        unsigned char* dst = smem + 2 * PP_BYTES + buf * PW_BYTES;
        const uint32_t off = ch * 64 + dx * dxs;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(dst + (wave + 8 * j) * 1024), 16, wp[j] + off, 0, 0, 0);
    };
    auto image_rsrc = [&](int img) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(in + img * img_elems), (short)0, in_rec, 0x00020000);
    };

    __bf16* __restrict__ out = (__bf16*)a.out;
    const __bf16* __restrict__ addend = (const __bf16*)a.addend;
    // ---- LDS scratch behind the rings: column statistics [4 row groups][128 channels][2] (each word has ONE owner lane, which adds
    // to it tile after tile: deterministic without atomics across lanes) and the bias of the workgroup's 128 channels
    {
        float* red = reinterpret_cast<float*>(smem + PRED);
        if (tid == 0) {                                   // the first two tickets
            unsigned* tk = reinterpret_cast<unsigned*>(smem + PSCHED);
            if (ctr) { const unsigned t = __hip_atomic_fetch_add(ctr, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); tk[0] = t; tk[1] = t + 1; }   // one round trip
            else { tk[0] = take(); tk[1] = take(); }
        }
        (void)red;
        if (tid < 128) reinterpret_cast<float*>(smem + PBIAS)[tid] = (a.bias && n0 + tid < a.N) ? a.bias[n0 + tid] : 0.f;
    }
    __syncthreads();
    const uint32_t bias_a = lds0 + PBIAS + (uint32_t)((wn * 64 + 8 * lq) * 4);
    const uint32_t red_a = lds0 + PRED + (uint32_t)(((wm * 128) + wn * 64 + 8 * lq) * 8);        // + buffer * PRED_BYTES
    // column statistics of a finished tile: its 4 row groups leave their sums in LDS buffer (tile parity); behind the next barrier
    // 256 threads add the four in a fixed order and write the tile's row - one row per pixel tile whoever served it
    int cs_pending = -1, cs_par = 0;                     // pixel tile whose sums wait in buffer cs_par ^ 1
    auto flush_row = [&](int pix_tile, int buf) {
        if (tid < 256) {
            const int ch = tid >> 1, st = tid & 1;
            const uint32_t ra = lds0 + PRED + buf * PRED_BYTES + (uint32_t)((ch * 2 + st) * 4);
            float r0, r1, r2, r3;
            asm volatile("ds_read_b32 %0, %1 offset:0" : "=v"(r0) : "v"(ra));
            asm volatile("ds_read_b32 %0, %1 offset:1024" : "=v"(r1) : "v"(ra));
            asm volatile("ds_read_b32 %0, %1 offset:2048" : "=v"(r2) : "v"(ra));
            asm volatile("ds_read_b32 %0, %1 offset:3072" : "=v"(r3) : "v"(ra));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const float t = ((r0 + r1) + r2) + r3;
            if (n0 + ch < a.N) a.colstat[((size_t)pix_tile * a.N + n0 + ch) * 2 + st] = t;
        }
    };

    const uint32_t sched_a = lds0 + PSCHED;
    const unsigned tk0 = reinterpret_cast<const unsigned*>(smem + PSCHED)[0], tk1 = reinterpret_cast<const unsigned*>(smem + PSCHED)[1];
    __syncthreads();                                      // (thread 0 reuses word 2 below)
    if (tk0 < (unsigned)cnt) {
        Tile cur = tile_of(tk0);
        bool have_next = tk1 < (unsigned)cnt;           // a next tile is known to exist
        bool pending = false;                             // its ticket is still on its way (drawn in the last epilogue)
        Tile nxt = cur;
        __amdgpu_buffer_rsrc_t rs_nxt = image_rsrc(cur.img);
        if (have_next) { nxt = tile_of(tk1); rs_nxt = image_rsrc(nxt.img); }
        const uint64_t ctr_addr = (uint64_t)(uintptr_t)ctr;
        __amdgpu_buffer_rsrc_t rs_cur = image_rsrc(cur.img);
        // ---- prologue of the first tile only: patch 0, kernel steps 0 and 1
        issue_p(rs_cur, cur, 0, 0);
        issue_w(0, 0, 0);
        issue_w(0, 1, 1);
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        started = true;

        int pbuf = 0;                                     // patch buffer of the current chunk
        int nst_prev = 0;                                 // output stores issued by this wave just before the current step
        for (;;) {

            f32x4 acc[4][2][4];                           // [image row][16-pixel half][16-channel tile]
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[i][h][t] = f32x4{0.f, 0.f, 0.f, 0.f};

            for (int ch = 0; ch < nch; ++ch) {
                const bool last = ch + 1 == nch;
                const bool more = !last || have_next;     // a next chunk exists (this tile's or the next tile's first)
                const int nch_ = last ? 0 : ch + 1;       // its chunk index
#pragma unroll 1
                for (int dx = 0; dx < 3; ++dx) {
                    // ---- the next tile's ticket (drawn during the previous tile's epilogue, handed over through LDS behind the
                    //      barrier of this tile's first step)
                    if (ch == 0 && dx == 1 && cs_pending >= 0) { flush_row(cs_pending, cs_par ^ 1); cs_pending = -1; }
                    if (ch == 0 && dx == 1 && pending) {
                        unsigned v;
                        asm volatile("ds_read_b32 %0, %1 offset:8" : "=v"(v) : "v"(sched_a));
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        const unsigned tk = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
                        have_next = tk < (unsigned)cnt;
                        if (have_next) { nxt = tile_of(tk); rs_nxt = image_rsrc(nxt.img); }
                        pending = false;
                    }
                    // ---- prefetch: kernel tile of step s+2 (ring slot (dx+2)%3), patch of the next chunk
                    if (dx == 0) {
                        issue_w(ch, 2, 2);
                        if (more) { if (last) issue_p(rs_nxt, nxt, 0, pbuf ^ 1); else issue_p(rs_cur, cur, nch_, pbuf ^ 1); }
                    } else if (more) {
                        issue_w(nch_, dx - 1, dx - 1);
                    }
                    const uint32_t ba = lds0 + 2 * PP_BYTES + dx * PW_BYTES + (wn * 64 + l15) * 64 + ((lq << 4) ^ ((l15 & 4) << 3));
                    const uint32_t aa = lds0 + pbuf * PP_BYTES + (4 * wm * PPC + l15 + dx) * 64 + ((lq << 4) ^ (((l15 + dx) & 4) << 3));
                    u32x4 wf[3][4], pf[6][2];
#define RDW(dy) DSR128(wf[dy][0], ba, dy * 8192 + 0); DSR128(wf[dy][1], ba, dy * 8192 + 1024); \
                DSR128(wf[dy][2], ba, dy * 8192 + 2048); DSR128(wf[dy][3], ba, dy * 8192 + 3072)
#define RDP(r) DSR128(pf[r][0], aa, r * PRP + 0); DSR128(pf[r][1], aa, r * PRP + 1024)
#define ROWS16(r)                                                                              \
    _Pragma("unroll") for (int dy = 0; dy < 3; ++dy) {                                         \
        if (r - dy < 0 || r - dy > 3) continue;                                                \
        _Pragma("unroll") for (int h = 0; h < 2; ++h)                                          \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) MMA16(acc[r - dy][h][t], wf[dy][t], pf[r][h]); \
    }
                    if (!UNETRIR_ABL(abl, 4)) {
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
                    __builtin_amdgcn_s_setprio(0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#undef RDW
#undef RDP
#undef ROWS16
                    // ---- retire what the next step reads; younger DMAs (and, in a tile's first steps, the previous tile's
                    //      output stores, which are older than this step's DMAs only) stay in flight across the barrier
                    if (dx == 0) {
                        if (more) { if (nst_prev == 16) asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
                        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    } else if (dx == 1) {
                        if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    } else {
                        if (more) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    nst_prev = 0;
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                pbuf ^= 1;
            }

            // ---- epilogue straight from the accumulators: acc[i][h][t][e] = D[row 4 lq + e of tile t][pixel 16 h + l15] of image
            //      row y0 + 4 wm + i; with the permuted kernel rows, tiles (2m, 2m+1) of a lane are channels 32 m + 8 lq + 0..7
            // ---- the ticket of the tile after next: thread 0 draws it now (a returning atomic: 1-2 us) and hands it over at the end of
            //      the epilogue, when only the output stores issued in between are younger (inline asm on both ends: the compiler's
            //      own atomic sequence waits vmcnt(0) on the spot, i.e. for every DMA and store in flight)
            const bool draw = have_next;                 // no further draw after the first ticket past the end
            unsigned tk_mine = 0xFFFFFFFFu;
            if (draw && tid == 0) {
                if (ctr) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(tk_mine) : "v"(ctr_addr), "v"(1u) : "memory");
                else tk_mine = (unsigned)(slot / ntN + (kstat++) * (32 / ntN));
            }
            int nst = 0;
            const int nb = n0 + wn * 64 + 8 * lq;
            float bias_[2][8];
            {
                u32x4 bq[4];
                DSR128(bq[0], bias_a, 0); DSR128(bq[1], bias_a, 16); DSR128(bq[2], bias_a, 128); DSR128(bq[3], bias_a, 144);
                LGKM_WAIT(0);
#pragma unroll
                for (int e = 0; e < 16; ++e) { const uint32_t u = bq[e >> 2][e & 3]; bias_[e >> 3][e & 7] = __uint_as_float(u); }
            }
            float cs_s[2][8], cs_q[2][8];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int e = 0; e < 8; ++e) { cs_s[m][e] = 0.f; cs_q[m][e] = 0.f; }
            // interior tiles without an addend (all but the image's ragged edges): no lane masks, one buffer store per (row, half, m)
            // with the tile-constant part of the address in a scalar register
            const bool fast = !addend && cur.y0 + PTR <= a.H && cur.x0 + 32 <= a.W && n0 + PBN <= a.N;
            if (fast) {
                const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(out + (size_t)cur.img * a.H * a.W * a.ldo), (short)0, (int)((((size_t)a.H * a.W - 1) * a.ldo + a.N) * 2), 0x00020000);
                const int voff = (((cur.y0 + 4 * wm) * a.W + cur.x0 + l15) * a.ldo + nb) * 2;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int soff = ((i * a.W + 16 * h) * a.ldo) * 2;
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            bf16x8 ov;
#pragma unroll
                            for (int e = 0; e < 8; ++e) ov[e] = (__bf16)(acc[i][h][2 * m + (e >> 2)][e & 3] + bias_[m][e]);
                            if (a.colstat) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) { const float s_ = (float)ov[e]; cs_s[m][e] += s_; cs_q[m][e] += s_ * s_; }
                            }
                            if (!UNETRIR_ABL(abl, 2))
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ov), rs_out, voff + 64 * m, soff, 0);
                        }
                    }
                nst = 16;
            } else
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int y = cur.y0 + 4 * wm + i;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int x = cur.x0 + 16 * h + l15;
                    if (!(y < a.H && cur.x0 + 16 * h < a.W)) continue;          // wave-uniform
                    const bool okp = x < a.W;
                    const size_t pix = ((size_t)cur.img * a.H + y) * a.W + x;
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const bool ok = okp && nb + 32 * m < a.N;               // N % 8 == 0
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = acc[i][h][2 * m + (e >> 2)][e & 3] + bias_[m][e];
                        if (addend && ok) {
                            const bf16x8 ad = *reinterpret_cast<const bf16x8*>(addend + pix * a.ldadd + nb + 32 * m);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = (float)(__bf16)v[e] + (float)ad[e];
                        }
                        bf16x8 ov;
#pragma unroll
                        for (int e = 0; e < 8; ++e) ov[e] = (__bf16)v[e];
                        if (ok && !UNETRIR_ABL(abl, 2)) {
                            if (a.colstat) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) { const float s = (float)ov[e]; cs_s[m][e] += s; cs_q[m][e] += s * s; }
                            }
                            *reinterpret_cast<bf16x8*>(out + pix * a.ldo + nb + 32 * m) = ov;
                        }
                        ++nst;
                    }
                }
            }
            nst_prev = nst;
            if (draw && tid == 0) {
                if (nst == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("ds_write_b32 %0, %1 offset:8" :: "v"(sched_a), "v"(tk_mine) : "memory");
            }
            if (a.colstat) {
                // over the 16 pixel columns of the lane group; lane l15 == 0 then adds to the words it owns (inline asm: an LDS
                // access the compiler can see would make it drain the DMA queue and the stores first)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                    { cs_s[m][e] = row_sum16(cs_s[m][e]); cs_q[m][e] = row_sum16(cs_q[m][e]); }
                if (l15 == 0) {
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(red_a + cs_par * PRED_BYTES), "v"(cs_s[m][e]), "n"((32 * m + e) * 8) : "memory");
                            asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(red_a + cs_par * PRED_BYTES), "v"(cs_q[m][e]), "n"((32 * m + e) * 8 + 4) : "memory");
                        }
                }
                cs_pending = cur.pix;
                cs_par ^= 1;
            }
            if (!have_next) break;
            cur = nxt;
            rs_cur = rs_nxt;
            pending = draw;                               // whether a tile follows `cur` is known after its first step
            have_next = false;
        }
    }

    // ---- column statistics of this workgroup's channel tile: over the 16 pixel columns of a lane group, then the 4 row groups
    if (a.colstat && cs_pending >= 0) {                   // the last tile's row
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        flush_row(cs_pending, cs_par ^ 1);
    }
#ifdef UNETRIR_ABLATIONS
    if (UNETRIR_ABL(abl, 2048) && tid == 0) {
        g_stamps_conv3x3p[blockIdx.x & 255][0] = __builtin_amdgcn_s_memtime() - st0;
        g_stamps_conv3x3p[blockIdx.x & 255][1] = __builtin_amdgcn_s_memrealtime() - sr0;
    }
#endif
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
inline long long pixel_tiles(const Conv3Args& a) { return (long long)a.B * ((a.H + PTR - 1) / PTR) * ((a.W + 31) / 32); }
}

bool conv3x3p_applies(const Conv3Args& a) {
    const int ntN = (a.N + PBN - 1) / PBN;
    return unetrir_cfg().conv3x3p && conv3x3g_applies(a) && a.N > 64 && !conv3x3g_pair_applies(a) && (a.N & 7) == 0 && a.C >= 64 &&
           (ntN == 1 || ntN == 2 || ntN == 4 || ntN == 8) && pixel_tiles(a) * ntN >= 512;
}

long long conv3x3p_colstat_rows(const Conv3Args& a) { return pixel_tiles(a); }

int launch_conv3x3p_bf16(const Conv3Args& a, hipStream_t s) {
    const long long pt = pixel_tiles(a);
    const int per_xcd = (int)((pt + 7) / 8);
    hipLaunchKernelGGL(conv3x3p_bf16_kernel, dim3(256), dim3(512), 0, s, a, (int)pt, per_xcd, sched_slot(s), UNETRIR_ABL_HOST());
    return (int)hipGetLastError();
}
