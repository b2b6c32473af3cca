// upconv3x3q.hip - the PERSISTENT form of upconv3x3g.hip (bf16 2x up-sampling direction of the 3x3 stride-2 layer pair:
// Conv2DTranspose forward, dl_models/u_net.py:297-304, and the data gradient of the strided Conv2D, :269-276).
//
// upconv3x3g runs one workgroup per 8 x 32 coarse-pixel x 64-channel tile: at the full-resolution end of the U-Net a tile has
// FOUR K steps (128 input channels) between a cold prologue and an epilogue that sends 128 accumulators per lane through
// an LDS staging tile - 150 us per launch where the output write alone (268 MB) bounds it near 60.  Here one workgroup
// per CU walks through its tiles with the K loop running on (conv3x3p.hip is the same step for the stride-1 kernel):
//   * the step stream (tile, 32-channel chunk) is continuous: the kernel slice of the next step and the patch two steps ahead
//     are requested across tile boundaries exactly as inside a tile (patch ring of 3, kernel ring of 2, counted vmcnt);
//   * the epilogue needs no LDS: the kernel rows of a wave's 32 channels are permuted in LDS so that a lane's accumulators
//     are 8 consecutive channels of its output pixels - 16-byte stores straight from the accumulators.  They are issued
//     AFTER the next step's DMA requests, so that the counted wait at the end of that step leaves them in flight;
//   * an addend (the skip-connection gradient the strided convolution's data gradient accumulates into, in place) is loaded
//     before those requests, into the registers the K loop's fragments have just left;
//   * tiles are dealt to the XCDs in contiguous ranges, channel tile fastest; a workgroup keeps one channel tile.
// K loop, LDS images, swizzles as upconv3x3g.hip.  Requires C % 32 == 0, C >= 96 (three K steps per tile), N % 8 == 0, N / 64 tiles in {1, 2, 4, 8},
// at least 1024 tiles (four per workgroup: with two the ticket draws at start-up cost what the persistence gains); otherwise upconv3x3g.
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
constexpr int QR = 8, QC = 32;                 // coarse tile
constexpr int QPR = QR + 1, QPC = QC + 1;      // 9 x 33 patch
constexpr int QNP = QPR * QPC;                 // 297 pixels
constexpr int QP_INSTR = (QNP + 15) / 16;      // 19 wave-instructions of 16 pixels x 64 B
constexpr int QP_BYTES = QP_INSTR * 1024;      // 19456
constexpr int QBN = 64;
constexpr int QW_INSTR = 9 * QBN / 16;         // 36: [9 taps][64 channels] rows of 64 B
constexpr int QW_BYTES = QW_INSTR * 1024;      // 36864
constexpr int QRING = 3 * QP_BYTES + 2 * QW_BYTES;   // 132096
constexpr int QBIAS = QRING;                   // bias of the workgroup's 64 channels
constexpr int QSCHED = QBIAS + QBN * 4;          // tile tickets handed from thread 0 to the workgroup
constexpr int QSMEM = QSCHED + 16;
constexpr uint32_t QOOB = 0xF0000000u;

struct Tile { int img, y0, x0; };
}  // namespace

__global__ __launch_bounds__(512) void upconv3x3q_bf16_kernel(const Conv3Args a, int pix_tiles, int per_xcd, unsigned* sched) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[QSMEM];
    const __bf16* __restrict__ in = (const __bf16*)a.in;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lq = lane >> 4;
    // a.H, a.W: coarse (input) grid; the output grid is 2H x 2W
    const int tiles_x = (a.W + QC - 1) / QC, tiles_y = (a.H + QR - 1) / QR;
    const int ntN = (a.N + QBN - 1) / QBN;
    const int C = a.C, nch = C / 32, ldw = 9 * C;
    const int OH = 2 * a.H, OW = 2 * a.W;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;

    // ---- tiles of this workgroup (see conv3x3p.hip): job = pixel tile * ntN + channel tile, XCD-contiguous ranges
    //      which pixel tile of its group (XCD, channel tile) a workgroup takes next is decided at run time by tickets (kernels.h,
    //      sched_slot; conv3x3p.hip has the why)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nt = slot % ntN, n0 = nt * QBN;
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
        t.img = id / tiles_y; t.y0 = ty * QR; t.x0 = tx * QC;
        return t;
    };
    if (tid < QBN) reinterpret_cast<float*>(smem + QBIAS)[tid] = (a.bias && n0 + tid < a.N) ? a.bias[n0 + tid] : 0.f;
    if (tid == 0) {                                       // the first two tickets
        unsigned* tk = reinterpret_cast<unsigned*>(smem + QSCHED);
        if (ctr) { const unsigned t = __hip_atomic_fetch_add(ctr, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); tk[0] = t; tk[1] = t + 1; }   // one round trip
        else { tk[0] = take(); tk[1] = take(); }
    }
    __syncthreads();
    const unsigned tk0 = reinterpret_cast<const unsigned*>(smem + QSCHED)[0], tk1 = reinterpret_cast<const unsigned*>(smem + QSCHED)[1];
    // the last workgroup to leave clears the launch's counters (every workgroup has drawn its last - failing - ticket by then)
    auto leave = [&]() {
        if (sched && tid == 0) {
            const unsigned d = __hip_atomic_fetch_add(sched + 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == gridDim.x - 1)
                for (int i = 0; i < 65; ++i) __hip_atomic_store(sched + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    if (tk0 >= (unsigned)cnt) { leave(); return; }
    const uint32_t sched_a = lds0 + QSCHED;

    // ---- per-lane DMA sources
    const int slot4 = lane & 3, sub = lane >> 2;
    int prc[3];                                          // patch row | column << 16 (row 0xFFFF: past the patch)
    uint32_t prel[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int i = wave + 8 * j;
        if (i > QP_INSTR - 1) i = QP_INSTR - 1;
        const int p = 16 * i + sub;
        const int pr = p / QPC, pc = p - pr * QPC;
        const int gs = slot4 ^ ((pc & 4) >> 1);
        prc[j] = (p < QNP ? pr : 0xFFFF) | (pc << 16);
        prel[j] = (uint32_t)(((pr * a.W + pc) * a.ldi + gs * 8) * 2);
    }
    // kernel slice: LDS row (tap * 64 + nl); within the 32 channels of a wave column (nl >> 5) the 16-row MFMA tile t = (nl >> 4) & 1,
    // row r = nl & 15 holds channel 8 (r >> 2) + 4 t + (r & 3): a lane (quarter lq) then owns channels 8 lq .. 8 lq + 7
    uint32_t wp[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        int i = wave + 8 * j;
        if (i > QW_INSTR - 1) i = QW_INSTR - 1;
        const int row = 16 * i + sub;
        const int t9 = row >> 6, nl = row & 63;
        const int gs = slot4 ^ ((nl & 4) >> 1);
        const int t = (nl >> 4) & 1, r = nl & 15;
        const int n = n0 + (nl & 32) + 8 * (r >> 2) + 4 * t + (r & 3);
        wp[j] = n < a.N ? (uint32_t)((n * ldw + t9 * C + gs * 8) * 2) : QOOB;
    }
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, (int)((size_t)a.N * ldw * 2), 0x00020000);
    const int in_rec = (int)((((size_t)a.H * a.W - 1) * a.ldi + C) * 2);
    const size_t img_elems = (size_t)a.H * a.W * a.ldi;
    auto issue_p = [&](const Tile& t, int ch, int buf) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(in + t.img * img_elems), (short)0, in_rec, 0x00020000);
        unsigned char* dst = smem + buf * QP_BYTES;
        const int base = (((t.y0 - 1) * a.W + t.x0 - 1) * a.ldi) * 2 + ch * 64;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int i = wave + 8 * j;
            if (i > QP_INSTR - 1) i = QP_INSTR - 1;
            const int iy = t.y0 - 1 + (prc[j] & 0xFFFF), ix = t.x0 - 1 + (prc[j] >> 16);
            const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + i * 1024), 16, ok ? (uint32_t)(base + (int)prel[j]) : QOOB, 0, 0, 0);
        }
    };
    auto issue_w = [&](int ch, int buf) {
        unsigned char* dst = smem + 3 * QP_BYTES + buf * QW_BYTES;
        const uint32_t c0b = ch * 64;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            int i = wave + 8 * j;
            if (i > QW_INSTR - 1) i = QW_INSTR - 1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(dst + i * 1024), 16, wp[j] + c0b, 0, 0, 0);
        }
    };

    // fragment addresses.  kernel: row (t9*64 + wn*32 + 16t + l15) * 64 + (lq*16 ^ swz(l15));  patch: ((2wm + r)*33 + 16h + l15 + 1 - dc) * 64 + ...
    const uint32_t b_lane = lds0 + 3 * QP_BYTES + (wn * 32 + l15) * 64 + ((lq << 4) ^ ((l15 & 4) << 3));
    uint32_t a_lane[2];
#pragma unroll
    for (int dc = 0; dc < 2; ++dc) {
        const int col = l15 + 1 - dc;
        a_lane[dc] = lds0 + (2 * wm * QPC + col) * 64 + ((lq << 4) ^ ((col & 4) << 3));
    }
    const uint32_t bias_a = lds0 + QBIAS + (uint32_t)((wn * 32 + 8 * lq) * 4);
    __bf16* __restrict__ out = (__bf16*)a.out;
    const __bf16* __restrict__ addend = (const __bf16*)a.addend;
    const int nb = n0 + wn * 32 + 8 * lq;              // this lane's 8 channels

    Tile cur = tile_of(tk0);
    bool have_next = tk1 < (unsigned)cnt;
    Tile nxt = have_next ? tile_of(tk1) : cur;
    bool drawing = have_next;                            // tickets are drawn until the first one past the end
    bool have_nn = false;                                // the tile after `nxt` (drawn in this tile's first step, known from its second)
    unsigned tk_nn = 0;
    // (tile, chunk) of the step k ahead of (cur, ch), k = 1, 2 (nch >= 2): this tile's or the next one's
    // ---- prologue of the first tile: patch 0, kernel 0, patch 1
    issue_p(cur, 0, 0);
    issue_w(0, 0);
    issue_p(cur, 1, 1);
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    f32x4 acc[4][2][2][2];                             // [parity class][coarse row][16-pixel half][16-channel tile]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[c][i][h][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    int g = 0;                                         // running step: patch ring slot g % 3, kernel ring slot g & 1
    int pslot = 0;
    bool epi = false;                                  // the previous tile's accumulators are still to be stored
    Tile prev = cur;
    for (;;) {
        for (int ch = 0; ch < nch; ++ch) {
            // ---- this step's DMA requests and, in a tile's first step, the previous tile's epilogue.  Without an addend the
            //      requests go first: the output stores are then the youngest entries of the queue and the counted wait at the end
            //      of the step leaves them in flight.  With an addend its loads would wait for the requests ahead of them, so the
            //      epilogue runs first (its loads wait for nothing but the patch the next step needs anyway).
            const bool do_epi = epi && ch == 0;
            // ---- tickets: thread 0 draws the one after next in a tile's first step (returning atomic as inline asm: the compiler's own
            //      sequence would wait vmcnt(0) on the spot), hands it over behind that step's counted wait, everybody reads it in the second
            unsigned tk_mine = 0xFFFFFFFFu;
            const bool draw = drawing && ch == 0;
            if (draw && tid == 0) {
                if (ctr) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(tk_mine) : "v"((uint64_t)(uintptr_t)ctr), "v"(1u) : "memory");
                else tk_mine = (unsigned)(slot / ntN + (kstat++) * (32 / ntN));
            }
            if (drawing && ch == 1) {
                unsigned v;
                asm volatile("ds_read_b32 %0, %1 offset:8" : "=v"(v) : "v"(sched_a));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                const unsigned tk = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
                have_nn = tk < (unsigned)cnt;
                tk_nn = tk;
                drawing = have_nn;
            }
            const bool more1 = ch + 1 < nch || have_next, more2 = ch + 2 < nch || have_next;
            auto prefetch = [&]() {
                if (more1) issue_w(ch + 1 < nch ? ch + 1 : 0, (g + 1) & 1);
                if (more2) {
                    const int ps2 = pslot >= 1 ? pslot - 1 : 2;           // (g + 2) % 3
                    if (ch + 2 < nch) issue_p(cur, ch + 2, ps2); else issue_p(nxt, ch + 2 - nch, ps2);
                }
            };
            int nst = 0;                                   // output stores issued AFTER this step's requests
            if (!do_epi) {
                prefetch();
            } else {
                // the previous tile, straight from the accumulators: acc[c][i][h][t][e] = D[row 4 lq + e of tile t][coarse column
                // 16 h + l15]; with the permuted kernel rows tiles (0, 1) of a lane are channels 8 lq + 0..7 of class c = (ay, ax),
                // coarse row y0 + 2 wm + i -> output pixel (2 cy + ay, 2 cx + ax)
                if (!addend) prefetch();
                const bool fast = prev.y0 + QR <= a.H && prev.x0 + QC <= a.W && n0 + QBN <= a.N;
                float bias_[8];
                {
                    u32x4 bq[2];
                    DSR128(bq[0], bias_a, 0); DSR128(bq[1], bias_a, 16);
                    LGKM_WAIT(0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const uint32_t u = bq[e >> 2][e & 3]; bias_[e] = __uint_as_float(u); }
                }
                const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(out + (size_t)prev.img * OH * OW * a.ldo), (short)0, (int)((((size_t)OH * OW - 1) * a.ldo + a.N) * 2), 0x00020000);
                const int vbase = (((2 * (prev.y0 + 2 * wm)) * OW + 2 * (prev.x0 + l15)) * a.ldo + nb) * 2;
                const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(addend ? addend + (size_t)prev.img * OH * OW * a.ldadd : out), (short)0, (int)((((size_t)OH * OW - 1) * a.ldadd + a.N) * 2), 0x00020000);
                const int vadd = (((2 * (prev.y0 + 2 * wm)) * OW + 2 * (prev.x0 + l15)) * a.ldadd + nb) * 2;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (addend) __builtin_amdgcn_sched_barrier(0);      // four addend loads at a time (the epilogue sits at the register limit)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int cy = prev.y0 + 2 * wm + i, ox = 2 * (prev.x0 + 16 * h + l15) + (c & 1);
                            if (!fast && !(cy < a.H && 2 * (prev.x0 + 16 * h) < OW)) continue;          // wave-uniform
                            const bool ok = fast || (ox < OW && nb < a.N);
                            float v[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = acc[c][i][h][e >> 2][e & 3] + bias_[e];
                            if (addend) {              // (lanes past the image read zeros and store nowhere)
                                const bf16x8 adv = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                    rs_add, vadd, (((2 * i + (c >> 1)) * OW + 32 * h + (c & 1)) * a.ldadd) * 2, 0));
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = (float)(__bf16)v[e] + (float)adv[e];
                            }
                            bf16x8 ov;
#pragma unroll
                            for (int e = 0; e < 8; ++e) ov[e] = (__bf16)v[e];
                            const int soff = (((2 * i + (c >> 1)) * OW + 32 * h + (c & 1)) * a.ldo) * 2;
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ov), rs_out, ok ? vbase : (int)QOOB, soff, 0);
                            ++nst;
                        }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int t = 0; t < 2; ++t) acc[c][i][h][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                epi = false;
                if (addend) { prefetch(); nst = 0; }       // the stores are older than the requests: the end-of-step wait retires them
            }

            const uint32_t ab0 = a_lane[0] + pslot * QP_BYTES, ab1 = a_lane[1] + pslot * QP_BYTES;
            const uint32_t bb = b_lane + (g & 1) * QW_BYTES;
            // patch rows: taps kh = 0, 1 pair output row i with patch row i + 1 (rows 1, 2), taps kh = 2 with patch row i (rows 0, 1):
            // row 2 is dead after the kh = 1 taps and row 0 is read into its registers - 8 fragments live instead of 12
            u32x4 pf[2][2][2];                             // [0]: patch row 1;  [1]: patch row 2, then row 0
#define RDP(slot, r) DSR128(pf[slot][0][0], ab0, (r) * (QPC * 64)); DSR128(pf[slot][0][1], ab0, (r) * (QPC * 64) + 1024); \
                     DSR128(pf[slot][1][0], ab1, (r) * (QPC * 64)); DSR128(pf[slot][1][1], ab1, (r) * (QPC * 64) + 1024)
            RDP(0, 1); RDP(1, 2);
            __builtin_amdgcn_s_setprio(1);
#define RDW(W0, W1, T9) do { DSR128(W0, bb, (T9) * 4096); DSR128(W1, bb, (T9) * 4096 + 1024); } while (0)
            // tap (KH, KW), output row i: kh < 2 -> slot i (rows 1, 2); kh = 2 -> slot 1 - i (row 0 sits in slot 1, row 1 in slot 0)
#define MMT(KH, KW, W0, W1)                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
            _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                      \
                MMA16(acc[((KH) & 1) * 2 + ((KW) & 1)][i][h][0], W0, pf[(KH) < 2 ? i : 1 - i][(KW) >> 1][h]);     \
                MMA16(acc[((KH) & 1) * 2 + ((KW) & 1)][i][h][1], W1, pf[(KH) < 2 ? i : 1 - i][(KW) >> 1][h]);     \
            }
            u32x4 wa0, wa1, wb0, wb1;
            RDW(wa0, wa1, 0);
            RDW(wb0, wb1, 1); LGKM_WAIT(2); MMT(0, 0, wa0, wa1);
            RDW(wa0, wa1, 2); LGKM_WAIT(2); MMT(0, 1, wb0, wb1);
            RDW(wb0, wb1, 3); LGKM_WAIT(2); MMT(0, 2, wa0, wa1);
            RDW(wa0, wa1, 4); LGKM_WAIT(2); MMT(1, 0, wb0, wb1);
            RDW(wb0, wb1, 5); LGKM_WAIT(2); MMT(1, 1, wa0, wa1);
            RDW(wa0, wa1, 6); LGKM_WAIT(2); MMT(1, 2, wb0, wb1);
            __builtin_amdgcn_sched_barrier(0);             // the reads below overwrite row 2: not above its last MFMAs
            RDP(1, 0);
            RDW(wb0, wb1, 7); LGKM_WAIT(2); MMT(2, 0, wa0, wa1);
            RDW(wa0, wa1, 8); LGKM_WAIT(2); MMT(2, 1, wb0, wb1);
            LGKM_WAIT(0); MMT(2, 2, wa0, wa1);
#undef RDP
#undef MMT
#undef RDW
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- retire the kernel slice of the next step and the (older) patch of the next step; the patch two steps ahead and
            //      the output stores issued after it stay in flight
            if (more2) { if (nst == 16) asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (draw && tid == 0) asm volatile("ds_write_b32 %0, %1 offset:8" :: "v"(sched_a), "v"(tk_mine) : "memory");   // (older than every request of the step)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ++g;
            pslot = pslot == 2 ? 0 : pslot + 1;
        }
        // ---- tile done: its accumulators are stored during the first step of the next tile, or right here after the last one
        prev = cur;
        epi = true;
        if (!have_next) break;
        cur = nxt;
        if (have_nn) nxt = tile_of(tk_nn);
        have_next = have_nn;
        have_nn = false;
    }
    // ---- the last tile's epilogue (no DMA in flight any more)
    {
        float bias_[8];
        {
            u32x4 bq[2];
            DSR128(bq[0], bias_a, 0); DSR128(bq[1], bias_a, 16);
            LGKM_WAIT(0);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const uint32_t u = bq[e >> 2][e & 3]; bias_[e] = __uint_as_float(u); }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int cy = prev.y0 + 2 * wm + i, oy = 2 * cy + (c >> 1), ox = 2 * (prev.x0 + 16 * h + l15) + (c & 1);
                    if (!(cy < a.H && ox < OW && nb < a.N)) continue;
                    const size_t pix = ((size_t)prev.img * OH + oy) * OW + ox;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = acc[c][i][h][e >> 2][e & 3] + bias_[e];
                    if (addend) {
                        const bf16x8 adv = *reinterpret_cast<const bf16x8*>(addend + pix * a.ldadd + nb);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (float)(__bf16)v[e] + (float)adv[e];
                    }
                    bf16x8 ov;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ov[e] = (__bf16)v[e];
                    *reinterpret_cast<bf16x8*>(out + pix * a.ldo + nb) = ov;
                }
    }
    leave();
}

namespace {
inline long long q_pixel_tiles(const Conv3Args& a) { return (long long)a.B * ((a.H + QR - 1) / QR) * ((a.W + QC - 1) / QC); }
}

bool upconv3x3q_applies(const Conv3Args& a) {
    const int ntN = (a.N + QBN - 1) / QBN;
    const size_t out_bytes = (((size_t)4 * a.H * a.W - 1) * a.ldo + a.N) * 2;
    return unetrir_cfg().upconv3x3q && upconv3x3g_applies(a) && a.C >= 96 && (a.N & 7) == 0 && (a.ldo & 7) == 0 &&
           (!a.addend || (a.ldadd & 7) == 0) && (ntN == 1 || ntN == 2 || ntN == 4 || ntN == 8) && q_pixel_tiles(a) * ntN >= 1024 &&
           out_bytes < 0x70000000u;
}

int launch_upconv3x3q_bf16(const Conv3Args& a, hipStream_t s) {
    const long long pt = q_pixel_tiles(a);
    const int per_xcd = (int)((pt + 7) / 8);
    hipLaunchKernelGGL(upconv3x3q_bf16_kernel, dim3(256), dim3(512), 0, s, a, (int)pt, per_xcd, sched_slot(s));
    return (int)hipGetLastError();
}
