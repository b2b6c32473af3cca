// upconv3x3g.hip - bf16 2x up-sampling direction of the 3x3 stride-2 layer pair (Conv2DTranspose forward, dl_models/
// u_net.py:297-304, and the data gradient of the strided Conv2D, :269-276) with the LDS-DMA pipeline of conv3x3g.hip.
//
//   out[2p + (kh, kw)] += in[p] . w[kh, kw]   (pad_before = 0, output cropped to 2h x 2w): tap (kh, kw) feeds output parity
//   class (kh & 1, kw & 1) from the coarse pixel at offset (-(kh >> 1), -(kw >> 1)).
// A workgroup (8 waves) owns 8 coarse rows x 32 coarse columns x 64 output channels, i.e. a 16 x 64 output tile; a wave owns
// 2 coarse rows x 32 columns x 32 channels for all four classes (128 accumulator registers, v_mfma_f32_16x16x32_bf16).
// K advances in 32-channel chunks, one chunk per step: the 9 x 33 coarse patch (ring of 3 buffers, requested two steps
// ahead) and the weights of all 9 taps (ring of 2) are staged by buffer_load ... lds; one raw s_barrier per step with a
// counted vmcnt.  A patch fragment (3 rows x 2 column offsets x 2 halves) is read once for the taps that use it and a
// weight fragment serves both coarse rows: 30 fragment reads per 72 MFMAs (upconv3x3.hip: 12 per 8, a barrier per tap).
// Same swizzle / out-of-range conventions as conv3x3g.hip.  Requires C % 32 == 0.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
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
constexpr int QSMEM = 3 * QP_BYTES + 2 * QW_BYTES;   // 132096
constexpr int QSROW = 32 * 2 + 16;             // epilogue staging: 32 channels + pad per output pixel
constexpr uint32_t QOOB = 0xF0000000u;
}  // namespace

__global__ __launch_bounds__(512) void upconv3x3g_bf16_kernel(const Conv3Args a) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[QSMEM];
    const __bf16* __restrict__ in = (const __bf16*)a.in;
    const __bf16* __restrict__ w = (const __bf16*)a.w;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lq = lane >> 4;

    // a.H, a.W: coarse (input) grid; the output grid is 2H x 2W
    const int tiles_x = (a.W + QC - 1) / QC, tiles_y = (a.H + QR - 1) / QR;
    const int ntN = (a.N + QBN - 1) / QBN;
    int id = blockIdx.x;
    if ((gridDim.x & 7) == 0) id = (id & 7) * (gridDim.x >> 3) + (id >> 3);
    const int nt = id % ntN; id /= ntN;
    const int tx = id % tiles_x; id /= tiles_x;
    const int ty = id % tiles_y;
    const int img = id / tiles_y;
    const int y0 = ty * QR, x0 = tx * QC, n0 = nt * QBN;
    const int C = a.C;
    const int nch = C / 32;
    const int ldw = 9 * C;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(in + (size_t)img * a.H * a.W * a.ldi), (short)0, (int)((((size_t)a.H * a.W - 1) * a.ldi + C) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)w, (short)0, (int)((size_t)a.N * ldw * 2), 0x00020000);
    const int slot = lane & 3, sub = lane >> 2;
    uint32_t pa[3];
    int pi[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int i = wave + 8 * j;
        if (i > QP_INSTR - 1) i = QP_INSTR - 1;
        pi[j] = i;
        const int p = 16 * i + sub;
        const int pr = p / QPC, pc = p - pr * QPC;
        const int gs = slot ^ ((pc & 4) >> 1);
        const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
        const bool ok = p < QNP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        pa[j] = ok ? (uint32_t)(((iy * a.W + ix) * a.ldi + gs * 8) * 2) : QOOB;
    }
    uint32_t wp[5];
    int wi[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        int i = wave + 8 * j;
        if (i > QW_INSTR - 1) i = QW_INSTR - 1;
        wi[j] = i;
        const int row = 16 * i + sub;                  // tap * 64 + local channel
        const int t9 = row >> 6, nl = row & 63;
        const int gs = slot ^ ((nl & 4) >> 1);
        const int n = n0 + nl;
        wp[j] = n < a.N ? (uint32_t)((n * ldw + t9 * C + gs * 8) * 2) : QOOB;
    }
    auto issue_p = [&](int ch) {
        unsigned char* dst = smem + (ch % 3) * QP_BYTES;
        const uint32_t c0b = ch * 64;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lptr_t)(dst + pi[j] * 1024), 16, pa[j] + c0b, 0, 0, 0);
    };
    auto issue_w = [&](int ch) {
        unsigned char* dst = smem + 3 * QP_BYTES + (ch & 1) * QW_BYTES;
        const uint32_t c0b = ch * 64;
#pragma unroll
        for (int j = 0; j < 5; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(dst + wi[j] * 1024), 16, wp[j] + c0b, 0, 0, 0);
    };

    f32x4 acc[4][2][2][2];                             // [parity class][coarse row][16-pixel half][16-channel tile]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[c][i][h][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: patch 0, weights 0, patch 1
    issue_p(0);
    issue_w(0);
    if (nch > 1) issue_p(1);
    if (nch > 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    // weights: row (t9*64 + wn*32 + 16t + l15) * 64 + (lq*16 ^ swz(l15));  patch: ((2wm + r)*33 + 16h + l15 + 1 - dc) * 64 + (lq*16 ^ swz(col))
    const uint32_t b_lane = lds0 + 3 * QP_BYTES + (wn * 32 + l15) * 64 + ((lq << 4) ^ ((l15 & 4) << 3));
    uint32_t a_lane[2];
#pragma unroll
    for (int dc = 0; dc < 2; ++dc) {
        const int col = l15 + 1 - dc;                  // + 16h: bit 2 of the column is unchanged by the half offset
        a_lane[dc] = lds0 + (2 * wm * QPC + col) * 64 + ((lq << 4) ^ ((col & 4) << 3));
    }

    for (int ch = 0; ch < nch; ++ch) {
        // ---- prefetch: weights of the next chunk, patch two chunks ahead
        const bool w_issued = ch + 1 < nch, p_issued = ch + 2 < nch;
        if (w_issued) issue_w(ch + 1);
        if (p_issued) issue_p(ch + 2);
        const uint32_t ab0 = a_lane[0] + (ch % 3) * QP_BYTES, ab1 = a_lane[1] + (ch % 3) * QP_BYTES;
        const uint32_t bb = b_lane + (ch & 1) * QW_BYTES;
        // patch fragments pf[r][dc][h]: rows 2wm + r (r = 0 is the row above the wave's first coarse row)
        u32x4 pf[3][2][2];
#define RDP(r) DSR128(pf[r][0][0], ab0, (r) * (QPC * 64)); DSR128(pf[r][0][1], ab0, (r) * (QPC * 64) + 1024); \
               DSR128(pf[r][1][0], ab1, (r) * (QPC * 64)); DSR128(pf[r][1][1], ab1, (r) * (QPC * 64) + 1024)
        RDP(0); RDP(1); RDP(2);
#undef RDP
        __builtin_amdgcn_s_setprio(1);
        // taps in (kh, kw) order; tap (kh, kw) -> class (kh&1, kw&1), patch row r = i + 1 - (kh>>1), column offset dc = kw>>1
#define RDW(W0, W1, T9) do { DSR128(W0, bb, (T9) * 4096); DSR128(W1, bb, (T9) * 4096 + 1024); } while (0)
#define MMT(KH, KW, W0, W1)                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
            _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                      \
                MMA16(acc[((KH) & 1) * 2 + ((KW) & 1)][i][h][0], W0, pf[i + 1 - ((KH) >> 1)][(KW) >> 1][h]);     \
                MMA16(acc[((KH) & 1) * 2 + ((KW) & 1)][i][h][1], W1, pf[i + 1 - ((KH) >> 1)][(KW) >> 1][h]);     \
            }
        // the weight fragments of tap t+1 are requested before the MFMAs of tap t (two register pairs, in-order LDS returns)
        u32x4 wa0, wa1, wb0, wb1;
        RDW(wa0, wa1, 0);
        RDW(wb0, wb1, 1); LGKM_WAIT(2); MMT(0, 0, wa0, wa1);
        RDW(wa0, wa1, 2); LGKM_WAIT(2); MMT(0, 1, wb0, wb1);
        RDW(wb0, wb1, 3); LGKM_WAIT(2); MMT(0, 2, wa0, wa1);
        RDW(wa0, wa1, 4); LGKM_WAIT(2); MMT(1, 0, wb0, wb1);
        RDW(wb0, wb1, 5); LGKM_WAIT(2); MMT(1, 1, wa0, wa1);
        RDW(wa0, wa1, 6); LGKM_WAIT(2); MMT(1, 2, wb0, wb1);
        RDW(wb0, wb1, 7); LGKM_WAIT(2); MMT(2, 0, wa0, wa1);
        RDW(wa0, wa1, 8); LGKM_WAIT(2); MMT(2, 1, wb0, wb1);
        LGKM_WAIT(0); MMT(2, 2, wa0, wa1);
#undef MMT
#undef RDW
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- retire W(ch+1) and the older P(ch+1); P(ch+2), issued after W(ch+1), may stay in flight
        if (p_issued) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    // ---- epilogue: four rounds, one output image row each (coarse row i, row parity ay): stage [64 output columns][32
    // channels] per wave, then 16-byte channel runs -> coalesced NHWC stores.
    // acc[c][i][h][t][j] = D[n = 16t + 4*lq + j][coarse column 16h + l15] of class c, coarse row y0 + 2wm + i.
    const int OH = 2 * a.H, OW = 2 * a.W;
    __bf16* __restrict__ out = (__bf16*)a.out;
    const __bf16* __restrict__ addend = (const __bf16*)a.addend;
    unsigned char* stage = smem + wave * (64 * QSROW);
    const int cq = lane & 3, pl = lane >> 2;           // readback: 4 lanes per pixel, 16 pixels per pass
    const int nrd = n0 + wn * 32 + cq * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int ay = 0; ay < 2; ++ay) {
#pragma unroll
            for (int ax = 0; ax < 2; ++ax)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int nl = 16 * t + 4 * lq;
                    const int n = n0 + wn * 32 + nl;
                    float bv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (a.bias) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (n + e < a.N) bv[e] = a.bias[n + e];
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x4& c = acc[ay * 2 + ax][i][h][t];
                        bf16x4 o;
                        o[0] = (__bf16)(c[0] + bv[0]); o[1] = (__bf16)(c[1] + bv[1]); o[2] = (__bf16)(c[2] + bv[2]); o[3] = (__bf16)(c[3] + bv[3]);
                        *reinterpret_cast<bf16x4*>(stage + (2 * (16 * h + l15) + ax) * QSROW + nl * 2) = o;
                    }
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
            const int cy = y0 + 2 * wm + i;
            const int oy = 2 * cy + ay;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int oc = ps * 16 + pl;              // output column inside the wave's 64
                const int ox = 2 * x0 + oc;
                if (cy < a.H && ox < OW && nrd < a.N) {
                    uint4 v = *reinterpret_cast<const uint4*>(stage + oc * QSROW + cq * 16);
                    const size_t pix = ((size_t)img * OH + oy) * OW + ox;
                    if (addend) {
                        const bf16x8 ad = *reinterpret_cast<const bf16x8*>(addend + pix * a.ldadd + nrd);
                        bf16x8 vv = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                        for (int e = 0; e < 8; ++e) vv[e] = (__bf16)((float)vv[e] + (float)ad[e]);
                        v = __builtin_bit_cast(uint4, vv);
                    }
                    *reinterpret_cast<uint4*>(out + pix * a.ldo + nrd) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
        }
    }
}

bool upconv3x3g_applies(const Conv3Args& a) {
    const bool on = unetrir_cfg().upconv3x3g != 0;
    const size_t img_bytes = (((size_t)a.H * a.W - 1) * a.ldi + a.C) * 2, w_bytes = (size_t)a.N * 9 * a.C * 2;
    return on && a.C % 32 == 0 && img_bytes < 0x70000000u && w_bytes < 0x70000000u;
}

int launch_upconv3x3g_bf16(const Conv3Args& a, hipStream_t s) {
    const long long tiles = (long long)a.B * ((a.H + QR - 1) / QR) * ((a.W + QC - 1) / QC) * ((a.N + QBN - 1) / QBN);
    hipLaunchKernelGGL(upconv3x3g_bf16_kernel, dim3((unsigned)tiles), dim3(512), 0, s, a);
    return (int)hipGetLastError();
}
