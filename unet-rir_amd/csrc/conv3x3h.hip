// conv3x3h.hip - bf16 3x3 stride-1 'same' convolution (forward / data gradient) for layers with at most 64 output
// channels (the full-resolution layers of the U-Net): the LDS-DMA pipeline of conv3x3g.hip at 2 workgroups per CU.
//
// These layers move the most bytes per flop (64 channels in, 64 out at 256 x 256), so a workgroup's prologue (patch load)
// and epilogue (output store) must overlap another workgroup's MFMAs: 4 waves own 16 rows x 32 columns x 64 channels and
// keep LDS under 80 KB so two workgroups share a CU.  To double-buffer the patch without doubling it, a 32-channel chunk
// lives in LDS as two 16-channel halves and the K loop walks (chunk, half, dx): while the three dx steps of half B run,
// half A of the next chunk streams in, and vice versa.
//   step = (chunk, half kk, horizontal tap dx): 24 v_mfma_f32_32x32x16_bf16 per wave, 12 fragment reads (row reuse as in
//   conv3x3g.hip), one raw s_barrier with a counted vmcnt; weight tile [3 dy][64 channels][16 input channels] in a ring of 3.
//   LDS images have 32-byte rows; the two 16-byte granules of a row are swapped where bit 3 of the row index is set
//   (conflict-free ds_read_b128), applied on the DMA source side and on the read side.
// Workgroups are persistent (at most 2 per CU) and walk tiles id, id + grid, ...: the first patch half and the first two
// weight tiles of the NEXT tile are requested during the last three steps of the current one, so only the very first tile
// of a workgroup waits on HBM before its first MFMA; the epilogue stages through the patch half / ring slot that the
// prefetch does not occupy.
// Requires C % 32 == 0 and N <= 64.  Fused column statistics as in conv3x3g.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "kernels.h"

// The epilogue staging tile is private to a wave: LDS operations of one wave execute in issue order, so its reads see its
// own earlier writes without a workgroup barrier; this only stops the compiler from moving LDS accesses across the point.
#define WAVE_LDS_FENCE() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

#define DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LGKM_WAIT(n) do { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define MMA(accv, wfrag, pfrag) \
    accv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wfrag), __builtin_bit_cast(bf16x8, pfrag), accv, 0, 0, 0)

namespace {
constexpr int HPC = 34;                        // patch columns
constexpr int HTR = 16;                        // tile rows
constexpr int HNPX = (HTR + 2) * HPC;          // 612 patch pixels
constexpr int HP_INSTR = 20;                   // wave-instructions (32 pixels x 32 B) per patch half: ceil(612 / 32)
constexpr int HP_BYTES = HP_INSTR * 1024;      // 20480
constexpr int HW_BYTES = 3 * 64 * 32;          // 6144: [3 dy][64 channels][32 B]
constexpr int HSROW = 64 * 2 + 16;
constexpr int HSMEM = 2 * HP_BYTES + 3 * HW_BYTES;        // 59392
static_assert(HP_BYTES >= 4 * 32 * HSROW && HW_BYTES >= 4 * 64 * 2 * 4, "epilogue staging / statistics regions");
}  // namespace

template <int ABL>      // ABL: timing ablations only (1: no DMA in the K loop, 2: no vmcnt wait / barrier); results invalid
__global__ __launch_bounds__(256, 2) void conv3x3h_bf16_kernel(const Conv3Args a) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[HSMEM];
    const __bf16* __restrict__ in = (const __bf16*)a.in;
    const __bf16* __restrict__ w = (const __bf16*)a.w;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave;
    const int l31 = lane & 31, hi = lane >> 5;

    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + HTR - 1) / HTR;
    const int ntiles = a.B * tiles_y * tiles_x;
    int wg = blockIdx.x;
    if ((gridDim.x & 7) == 0) wg = (wg & 7) * (gridDim.x >> 3) + (wg >> 3);   // neighbouring tiles on one XCD
    const int C = a.C;
    const int nch = C / 32;
    const int ldw = 9 * C;

    constexpr uint32_t OOB = 0xF0000000u;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)w, (short)0, (int)((size_t)a.N * ldw * 2), 0x00020000);
    const int slot = lane & 1, sub = lane >> 1;
    // patch half: 5 instructions per wave, instruction i covers patch pixels 32 i .. 32 i + 31 of tile t
    auto tile_origin = [&](int t, int& img, int& y0, int& x0) {
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        img = t / tiles_y; y0 = ty * HTR; x0 = tx * 32;
    };
    auto patch_offsets = [&](int y0, int x0, uint32_t (&pa)[5]) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int p = 32 * (wave + 4 * j) + sub;
            const int pr = p / HPC, pc = p - pr * HPC;
            const int gs = slot ^ ((pc >> 3) & 1);
            const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
            const bool ok = p < HNPX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            pa[j] = ok ? (uint32_t)(((iy * a.W + ix) * a.ldi + gs * 8) * 2) : OOB;
        }
    };
    auto image_rsrc = [&](int img) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(in + (size_t)img * a.H * a.W * a.ldi), (short)0,
                                                 (int)((((size_t)a.H * a.W - 1) * a.ldi + C) * 2), 0x00020000);
    };
    // weight tile: 6 instructions of 32 rows; wave w issues instruction w and min(w + 4, 5) (5 twice: uniform DMA counts)
    uint32_t wp[2];
    int wi[2];
    const int dxs = ((a.flip & 1) ? -C : C) * 2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int i = wave + 4 * j;
        if (i > 5) i = 5;
        wi[j] = i;
        const int row = 32 * i + sub;                  // dy * 64 + channel
        const int dy = row >> 6, nl = row & 63;
        const int gs = slot ^ ((nl >> 3) & 1);
        const int tap0 = (a.flip & 1) ? 8 - 3 * dy : 3 * dy;
        wp[j] = nl < a.N ? (uint32_t)((nl * ldw + tap0 * C + gs * 8) * 2) : OOB;
    }
    auto issue_p = [&](const __amdgpu_buffer_rsrc_t& rs, const uint32_t (&pa)[5], int ch, int kk) {
        unsigned char* dst = smem + kk * HP_BYTES;      // half kk (channels 16 kk .. 16 kk + 15) of chunk ch
        const uint32_t off = ch * 64 + kk * 32;
#pragma unroll
        for (int j = 0; j < 5; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + (wave + 4 * j) * 1024), 16, pa[j] + off, 0, 0, 0);
    };
    auto issue_w = [&](int s) {                        // step s = (chunk, half, dx) -> ring slot s % 3 = dx
        const int ch = s / 6, r6 = s - ch * 6;
        const int kk = r6 / 3, dx = r6 - kk * 3;
        unsigned char* dst = smem + 2 * HP_BYTES + dx * HW_BYTES;
        const uint32_t off = ch * 64 + kk * 32 + dx * dxs;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(dst + wi[j] * 1024), 16, wp[j] + off, 0, 0, 0);
    };

    int tile = wg;
    int img, y0, x0;
    tile_origin(tile, img, y0, x0);
    uint32_t pa[5], pan[5];
    patch_offsets(y0, x0, pa);
    __amdgpu_buffer_rsrc_t rs_in = image_rsrc(img), rs_nx = rs_in;
    int img_n = img, y0_n = y0, x0_n = x0;

    f32x16 acc[4][2];
    const int nsteps = nch * 6;
    issue_p(rs_in, pa, 0, 0);
    issue_w(0);
    issue_w(1);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    const uint32_t b_lane = lds0 + 2 * HP_BYTES + l31 * 32 + ((hi ^ ((l31 >> 3) & 1)) << 4);     // + dx*HW_BYTES + dy*2048 + j*1024
    const uint32_t a_lane = lds0 + (4 * wm * HPC + l31) * 32;

  for (;;) {
    const int next = tile + (int)gridDim.x;
    const bool has_next = next < ntiles;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int s = 0;
    for (int ch = 0; ch < nch; ++ch) {
        const bool more = ch + 1 < nch;
#pragma unroll 1
        for (int r6 = 0; r6 < 6; ++r6, ++s) {
            const int kk = r6 >= 3 ? 1 : 0, dx = r6 - 3 * kk;
            // ---- prefetch: weights of step s+2; at dx == 0 the patch half that the step three ahead starts on
            const bool w_issued = (s + 2 < nsteps || has_next) && !(ABL & 1);   // the next tile starts on the same weight tiles
            if (w_issued) issue_w(s + 2 < nsteps ? s + 2 : s + 2 - nsteps);
            bool p_issued = false;
            if (dx == 0 && !(ABL & 1)) {
                if (kk == 0) { issue_p(rs_in, pa, ch, 1); p_issued = true; }
                else if (more) { issue_p(rs_in, pa, ch + 1, 0); p_issued = true; }
                else if (has_next) {                                   // first patch half of the next tile
                    tile_origin(next, img_n, y0_n, x0_n);
                    patch_offsets(y0_n, x0_n, pan);
                    rs_nx = image_rsrc(img_n);
                    issue_p(rs_nx, pan, 0, 0);
                    p_issued = true;
                }
            }
            const int col = l31 + dx;
            const uint32_t aa = a_lane + kk * HP_BYTES + dx * 32 + ((hi ^ ((col >> 3) & 1)) << 4);
            const uint32_t ba = b_lane + dx * HW_BYTES;
            u32x4 w00, w01, w10, w11, w20, w21, p0, p1, p2, p3, p4, p5;
            DSR128(w00, ba, 0 * 2048 + 0);
            DSR128(w01, ba, 0 * 2048 + 1024);
            DSR128(p0, aa, 0 * 1088);
            DSR128(w10, ba, 1 * 2048 + 0);
            DSR128(w11, ba, 1 * 2048 + 1024);
            DSR128(p1, aa, 1 * 1088);
            DSR128(w20, ba, 2 * 2048 + 0);
            DSR128(w21, ba, 2 * 2048 + 1024);
            DSR128(p2, aa, 2 * 1088);
            DSR128(p3, aa, 3 * 1088);
            DSR128(p4, aa, 4 * 1088);
            DSR128(p5, aa, 5 * 1088);
            __builtin_amdgcn_s_setprio(1);
            LGKM_WAIT(9);
            MMA(acc[0][0], w00, p0); MMA(acc[0][1], w01, p0);
            LGKM_WAIT(6);
            MMA(acc[1][0], w00, p1); MMA(acc[1][1], w01, p1);
            MMA(acc[0][0], w10, p1); MMA(acc[0][1], w11, p1);
            LGKM_WAIT(3);
            MMA(acc[2][0], w00, p2); MMA(acc[2][1], w01, p2);
            MMA(acc[1][0], w10, p2); MMA(acc[1][1], w11, p2);
            MMA(acc[0][0], w20, p2); MMA(acc[0][1], w21, p2);
            LGKM_WAIT(2);
            MMA(acc[3][0], w00, p3); MMA(acc[3][1], w01, p3);
            MMA(acc[2][0], w10, p3); MMA(acc[2][1], w11, p3);
            MMA(acc[1][0], w20, p3); MMA(acc[1][1], w21, p3);
            LGKM_WAIT(1);
            MMA(acc[3][0], w10, p4); MMA(acc[3][1], w11, p4);
            MMA(acc[2][0], w20, p4); MMA(acc[2][1], w21, p4);
            LGKM_WAIT(0);
            MMA(acc[3][0], w20, p5); MMA(acc[3][1], w21, p5);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- retire W(s+1) (and with it every older DMA); what was issued after it may stay in flight:
            //      dx == 0: W(s+2) + this step's patch half;  dx == 1: the patch half of the previous step + W(s+2);  dx == 2: W(s+2)
            int allow = w_issued ? 2 : 0;
            if (dx == 0) allow += p_issued ? 5 : 0;
            else if (dx == 1) allow += (kk == 0 || more || has_next) ? 5 : 0;
            if constexpr (ABL & 2) { asm volatile("" ::: "memory"); continue; }
            if (allow == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else if (allow == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if (allow == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }

    // ---- epilogue through LDS, one image row of the wave at a time.  Staging lives in patch half B and the statistics in
    // ring slot 2: half A and slots 0, 1 may already hold the next tile's first DMA.
    unsigned char* stage = smem + HP_BYTES + wave * (32 * HSROW);
    const int cq = lane & 7, pl = lane >> 3;
    const int nq = cq * 8;
    __bf16* __restrict__ out = (__bf16*)a.out;
    const __bf16* __restrict__ addend = (const __bf16*)a.addend;
    float cs_s[8], cs_q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { cs_s[e] = 0.f; cs_q[e] = 0.f; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i) WAVE_LDS_FENCE();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int n = 32 * j + 8 * qd + 4 * hi;
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.bias && n + 3 < a.N) bv = *reinterpret_cast<const float4*>(a.bias + n);
                else if (a.bias) { float* bp = &bv.x; for (int e = 0; e < 4; ++e) if (n + e < a.N) bp[e] = a.bias[n + e]; }
                const f32x16& c = acc[i][j];
                bf16x4 o;
                o[0] = (__bf16)(c[4 * qd + 0] + bv.x); o[1] = (__bf16)(c[4 * qd + 1] + bv.y);
                o[2] = (__bf16)(c[4 * qd + 2] + bv.z); o[3] = (__bf16)(c[4 * qd + 3] + bv.w);
                *reinterpret_cast<bf16x4*>(stage + l31 * HSROW + n * 2) = o;
            }
        }
        WAVE_LDS_FENCE();
        const int y = y0 + 4 * wm + i;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int p = ps * 8 + pl;
            const int x = x0 + p;
            if (y >= a.H || x >= a.W || nq >= a.N) continue;
            uint4 v = *reinterpret_cast<const uint4*>(stage + p * HSROW + cq * 16);
            const size_t pix = ((size_t)img * a.H + y) * a.W + x;
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
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) { cs_s[e] += __shfl_xor(cs_s[e], off); cs_q[e] += __shfl_xor(cs_q[e], off); }
        }
        float* red = reinterpret_cast<float*>(smem + 2 * HP_BYTES + 2 * HW_BYTES);       // [4 wm][64 ch][2] in ring slot 2
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(wm * 64 + lane * 8 + e) * 2 + 0] = cs_s[e];
                red[(wm * 64 + lane * 8 + e) * 2 + 1] = cs_q[e];
            }
        }
        __syncthreads();
        if (tid < 128) {
            const int ch = tid >> 1, st = tid & 1;
            const float t = ((red[(0 * 64 + ch) * 2 + st] + red[(1 * 64 + ch) * 2 + st]) + red[(2 * 64 + ch) * 2 + st]) +
                            red[(3 * 64 + ch) * 2 + st];
            if (ch < a.N) a.colstat[((size_t)tile * a.N + ch) * 2 + st] = t;
        }
    }
    if (!has_next) break;
    tile = next; img = img_n; y0 = y0_n; x0 = x0_n; rs_in = rs_nx;
#pragma unroll
    for (int j = 0; j < 5; ++j) pa[j] = pan[j];
    __syncthreads();                                  // staging / statistics reads are done before the next tile's DMA lands there
  }
}

bool conv3x3h_applies(const Conv3Args& a) {
    const bool on = unetrir_cfg().conv3x3h != 0;
    const size_t img_bytes = (((size_t)a.H * a.W - 1) * a.ldi + a.C) * 2, w_bytes = (size_t)a.N * 9 * a.C * 2;
    return on && a.C % 32 == 0 && a.N <= 64 && !(a.flip & 2) && img_bytes < 0x70000000u && w_bytes < 0x70000000u;
}

int launch_conv3x3h_bf16(const Conv3Args& a, hipStream_t s) {
    const long long tiles = (long long)a.B * ((a.H + HTR - 1) / HTR) * ((a.W + 31) / 32);
    const int cap = 512;                                  // 2 persistent workgroups per CU
    const dim3 grid((unsigned)(tiles < cap ? tiles : cap));
#ifdef UNETRIR_ABLATIONS
    const int abl = (UNETRIR_ABL_HOST() >> 4) & 3;        // ablation build: bits 4, 5 = no DMA in the K loop / no vmcnt wait and barrier
    if (abl == 1) { hipLaunchKernelGGL(conv3x3h_bf16_kernel<1>, grid, dim3(256), 0, s, a); return (int)hipGetLastError(); }
    if (abl == 3) { hipLaunchKernelGGL(conv3x3h_bf16_kernel<3>, grid, dim3(256), 0, s, a); return (int)hipGetLastError(); }
#endif
    hipLaunchKernelGGL(conv3x3h_bf16_kernel<0>, grid, dim3(256), 0, s, a);
    return (int)hipGetLastError();
}
