// pw1x1.hip - the 1x1 ("pointwise") convolutions of the residual graphs (dl_models/res_ae.py:453-514, :310-371; gfx950, bf16).
//
// Two of the three (res_identity) or three of the four (res_conv) convolutions of every residual block are 1x1: per pixel a
// C -> N matrix product with C, N <= 256, i.e. 16 ... 128 flop per byte moved - HBM bound at every level.  The tap-table
// implicit-GEMM kernel served them with one 128-pixel workgroup per tile: a cold prologue (64-bit index arithmetic, tap masks),
// operands through LDS, an epilogue through LDS and nothing in flight across tiles - 21 ... 30 us per launch for tensors the
// memory system moves in 2 ... 13 us.  This kernel keeps nothing but the KERNEL in LDS and streams pixels through registers:
//
//   * a wave owns strips of 32 consecutive pixels of the iteration grid; its MFMA pixel operand comes straight from global
//     memory in fragment layout (lane l: pixel l & 31, channels 16 kk + 8 (l >> 5) .. + 7 - one 16-byte load per k-block), the
//     loads of the NEXT strip are issued before the current one is computed;
//   * the kernel tile [<= 64 output channels][C] sits in LDS for the whole launch (<= 33 KB), v_mfma_f32_32x32x16_bf16 with the
//     kernel as the row operand gives D[channel][pixel]: a lane holds 16 channels of ITS pixel; one v_permlane32_swap pair moves
//     them so that a lane owns 16 CONSECUTIVE channels: two 16-byte stores per 32-channel block, no LDS on the way out;
//   * persistent workgroups (4 waves) walk through their strips; per-channel (sum, sum of squares) of the stored bf16 values
//     accumulate in registers over the whole walk and leave ONE row of column statistics per workgroup (BatchNormalization
//     statistics and bias gradients without another pass; <= 1024 rows whatever the tensor size);
//   * the strided forms in the same launch: stride-2 gather on the input side (Conv2D(strides=2) forward, Conv2DTranspose 'valid'
//     data gradient), stride-2 scatter on the output side with the three other pixels of every 2 x 2 cell filled with bias
//     (+ addend) (Conv2DTranspose(1, strides=2, 'valid') forward: dl_models/res_ae.py:345, Conv2D(strides=2) data gradient).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ unsigned pack2(float a, float b) {
    const __bf16 x = (__bf16)a, y = (__bf16)b;
    return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ float lo16(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float hi16(unsigned v) { return __builtin_bit_cast(float, v & 0xffff0000u); }

// row-of-16 total in every lane of the row (DPP row rotations: full-rate VALU, no LDS)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
    return v;
}

}  // namespace

// (kernel outside the anonymous namespace: profilers print its name)
// KB = k-blocks of 16 channels (C = 16 KB; C == 8 runs as KB = 1 with the upper half zero), NB = 32-channel output blocks of the
// workgroup's channel tile (NT = 32 NB <= 64), PD = strips a wave keeps in flight (its pixel fragments live in PD x KB x 4
// registers: a strip's registers are re-loaded for the strip PD further on as soon as its MFMAs are issued)
template <int KB, int NB, int PD>
__global__ __launch_bounds__(256) void pw1x1_bf16_kernel(const PwArgs a) {
    constexpr int C16 = KB * 16;
    constexpr int LDW = C16 + 8;                          // LDS row stride of the kernel tile (16-byte pad: rows 16 bytes apart mod 128)
    constexpr int NT = NB * 32;
    __shared__ __attribute__((aligned(16))) __bf16 Ws[NT * LDW];
    __shared__ __attribute__((aligned(16))) float s_bias[NB][2][16];       // as the accumulators want it: [block][h][4 q + e] = channel 8 q + 4 h + e
    __shared__ __attribute__((aligned(16))) unsigned s_bfill[NB][2][8];    // as the exchanged layout stores it: pair i = channels 16 h + 2 i, + 1 (rounded)
    __shared__ float s_stat[4][2][NT][2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int ntN = a.N / NT;
    // workgroups that read the same pixels (the channel tiles of one strip group) sit on one XCD (blockIdx.x % 8), so the second
    // to the (N / NT)-th read of a strip is an L2 hit
    const int xcd = blockIdx.x & 7, q8 = blockIdx.x >> 3;
    const int nt = q8 % ntN, wg = (q8 / ntN) * 8 + xcd, nwg = gridDim.x / ntN;          // gridDim.x = nwg * ntN, nwg a multiple of 8
    const int n0 = nt * NT;

    {   // ---- the kernel tile: [NT][C] -> LDS (zero beyond C for the 8-channel input layer); bias in its two layouts
        constexpr int PIECES = NT * (C16 / 8);            // 16-byte pieces
        for (int i = tid; i < PIECES; i += 256) {
            const int r = i / (C16 / 8), q = i - r * (C16 / 8);
            u32x4 v = {0u, 0u, 0u, 0u};
            if (q * 8 < a.C) v = *reinterpret_cast<const u32x4*>(a.w + (size_t)(n0 + r) * a.C + q * 8);
            *reinterpret_cast<u32x4*>(&Ws[r * LDW + q * 8]) = v;
        }
        if (tid < NB * 32) {
            const int nb = tid >> 5, hh = (tid >> 4) & 1, r = tid & 15;
            s_bias[nb][hh][r] = a.bias ? a.bias[n0 + 32 * nb + 8 * (r >> 2) + 4 * hh + (r & 3)] : 0.f;
            if (r < 8) s_bfill[nb][hh][r] = a.bias ? pack2(a.bias[n0 + 32 * nb + 16 * hh + 2 * r], a.bias[n0 + 32 * nb + 16 * hh + 2 * r + 1]) : 0u;
        }
    }
    __syncthreads();

    const unsigned M = (unsigned)a.B * a.PH * a.PW;       // < 2^31 (checked by the launcher)
    const unsigned nstrip = (M + 31u) / 32u;
    const unsigned plane = (unsigned)a.PH * a.PW;
    const bool flat = (a.SI == 1 && a.SO == 1);           // input pixel = output pixel = p

    float cs[NB][16], css[NB][16];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) { cs[nb][i] = 0.f; css[nb][i] = 0.f; }
    unsigned nfill = 0;                                    // filled pixels this lane wrote (their statistics are analytic)

    bf16x8 fa[PD][KB];
    unsigned opix[PD], edge[PD];
    bool ok[PD];
    // pixel p of the iteration grid: loads its channel fragments; opix = output pixel index; edge bit 0 / 1: the cell's second
    // row / column exists (odd output sizes)
    auto load_strip = [&](unsigned st, bf16x8 (&f)[KB], unsigned& op, unsigned& ed, bool& k) {
        const unsigned p = st * 32u + (unsigned)l31;
        k = st < nstrip && p < M;
        size_t ip = 0;
        op = 0; ed = 0;
        if (k) {
            if (flat) { ip = p; op = p; }
            else {
                const unsigned b = p / plane, rem = p - b * plane;
                const unsigned py = rem / (unsigned)a.PW, px = rem - py * (unsigned)a.PW;
                op = (b * (unsigned)a.OH + py * (unsigned)a.SO) * (unsigned)a.OW + px * (unsigned)a.SO;
                ip = (size_t)((b * (unsigned)a.IH + py * (unsigned)a.SI) * (unsigned)a.IW + px * (unsigned)a.SI);
                ed = ((py * (unsigned)a.SO + 1u < (unsigned)a.OH) ? 1u : 0u) | ((px * (unsigned)a.SO + 1u < (unsigned)a.OW) ? 2u : 0u);
            }
        }
        const __bf16* src = a.in + ip * (size_t)a.ldi + 8 * h;
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (k && (kk * 16 + 8 * h) < a.C) v = *reinterpret_cast<const u32x4*>(src + kk * 16);
            f[kk] = __builtin_bit_cast(bf16x8, v);
        }
    };

    const unsigned sstep = (unsigned)nwg * 4u;
    unsigned st = (unsigned)wg * 4u + (unsigned)wave;
#pragma unroll
    for (int j = 0; j < PD; ++j) load_strip(st + (unsigned)j * sstep, fa[j], opix[j], edge[j], ok[j]);
    for (; st < nstrip; st += (unsigned)PD * sstep) {
#pragma unroll
        for (int j = 0; j < PD; ++j) {
            const bool okj = ok[j];
            const unsigned opj = opix[j], edj = edge[j];
            unsigned o[NB][8];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int kk = 0; kk < KB; ++kk) {
                    const bf16x8 fb = *reinterpret_cast<const bf16x8*>(&Ws[(nb * 32 + l31) * LDW + kk * 16 + 8 * h]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa[j][kk], acc, 0, 0, 0);
                }
                // + bias (after the chain, as the tap-table kernel adds it: the two kernels give the same bits), round, pack:
                // P[q] = channels 8 q + 4 h + {0..3} of pixel l31
                unsigned P[4][2];
                const float4* bp = reinterpret_cast<const float4*>(&s_bias[nb][h][0]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b4 = bp[q];
                    P[q][0] = pack2(acc[4 * q + 0] + b4.x, acc[4 * q + 1] + b4.y);
                    P[q][1] = pack2(acc[4 * q + 2] + b4.z, acc[4 * q + 3] + b4.w);
                }
                // lanes l and l + 32 hold the two halves of every 8-channel group: after the swaps lane h = 0 owns channels 0..15 of
                // the block, lane h = 1 channels 16..31 (v_permlane32_swap: lanes 32..63 of the first <-> lanes 0..31 of the second)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const auto r = __builtin_amdgcn_permlane32_swap(P[jj][e], P[jj + 2][e], false, false);
                        o[nb][4 * jj + e] = r[0];          // channels 16 h + 8 jj + {2 e, 2 e + 1}
                        o[nb][4 * jj + 2 + e] = r[1];      // channels 16 h + 8 jj + 4 + {2 e, 2 e + 1}
                    }
            }
            // this strip's fragments are consumed: its registers take the strip PD further on
            load_strip(st + (unsigned)(j + PD) * sstep, fa[j], opix[j], edge[j], ok[j]);
            if (!okj) continue;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int cbase = n0 + 32 * nb + 16 * h;
                __bf16* dst = a.out + (size_t)opj * a.ldo + cbase;
                if (a.addend) {
                    const __bf16* ad = a.addend + (size_t)opj * a.ldadd + cbase;
                    const u32x4 a0 = *reinterpret_cast<const u32x4*>(ad), a1 = *reinterpret_cast<const u32x4*>(ad + 8);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        o[nb][i] = pack2(lo16(o[nb][i]) + lo16(a0[i]), hi16(o[nb][i]) + hi16(a0[i]));
                        o[nb][4 + i] = pack2(lo16(o[nb][4 + i]) + lo16(a1[i]), hi16(o[nb][4 + i]) + hi16(a1[i]));
                    }
                }
                const u32x4 v0 = {o[nb][0], o[nb][1], o[nb][2], o[nb][3]}, v1 = {o[nb][4], o[nb][5], o[nb][6], o[nb][7]};
                *reinterpret_cast<u32x4*>(dst) = v0;
                *reinterpret_cast<u32x4*>(dst + 8) = v1;
                if (a.colstat) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float x0 = lo16(o[nb][i]), x1 = hi16(o[nb][i]);
                        cs[nb][2 * i] += x0; css[nb][2 * i] += x0 * x0;
                        cs[nb][2 * i + 1] += x1; css[nb][2 * i + 1] += x1 * x1;
                    }
                }
                if (a.fill) {     // the other pixels of the 2 x 2 output cell hold no product: bias (+ what the addend holds there)
                    const bool keep = (a.addend == a.out) && !a.bias;          // in-place accumulation of nothing: leave them alone
#pragma unroll
                    for (int f = 1; f < 4; ++f) {
                        if (((f >> 1) && !(edj & 1u)) || ((f & 1) && !(edj & 2u)) || keep) continue;
                        const size_t off = (size_t)opj + (size_t)((f >> 1) * a.OW + (f & 1));
                        u32x4 w0 = *reinterpret_cast<const u32x4*>(&s_bfill[nb][h][0]), w1 = *reinterpret_cast<const u32x4*>(&s_bfill[nb][h][4]);
                        if (a.addend) {
                            const __bf16* ad = a.addend + off * a.ldadd + cbase;
                            const u32x4 a0 = *reinterpret_cast<const u32x4*>(ad), a1 = *reinterpret_cast<const u32x4*>(ad + 8);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                w0[i] = pack2(lo16(w0[i]) + lo16(a0[i]), hi16(w0[i]) + hi16(a0[i]));
                                w1[i] = pack2(lo16(w1[i]) + lo16(a1[i]), hi16(w1[i]) + hi16(a1[i]));
                            }
                        }
                        __bf16* df = a.out + off * a.ldo + cbase;
                        *reinterpret_cast<u32x4*>(df) = w0;
                        *reinterpret_cast<u32x4*>(df + 8) = w1;
                        if (nb == 0) ++nfill;
                    }
                }
            }
        }
    }

    if (a.colstat == nullptr) return;
    // ---- one row of column statistics per workgroup: the filled pixels hold exactly the rounded bias (statistics are only asked
    // for without an addend), lanes -> rows of 16 by DPP, the 2 rows x 4 waves through LDS in a fixed order
    const float nf = (float)nfill;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned bw = s_bfill[nb][h][i >> 1];
            const float b = (i & 1) ? hi16(bw) : lo16(bw);
            const float v0 = row16_sum(cs[nb][i] + nf * b), v1 = row16_sum(css[nb][i] + nf * b * b);
            if ((lane & 15) == 0) {
                s_stat[wave][(lane >> 4) & 1][32 * nb + 16 * h + i][0] = v0;
                s_stat[wave][(lane >> 4) & 1][32 * nb + 16 * h + i][1] = v1;
            }
        }
    __syncthreads();
    if (tid < NT) {
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int r = 0; r < 2; ++r) { t0 += s_stat[w][r][tid][0]; t1 += s_stat[w][r][tid][1]; }
        float* row = a.colstat + ((size_t)wg * a.N + n0 + tid) * 2;
        row[0] = t0; row[1] = t1;
    }
}

namespace {
template <int KB, int PD>
int launch_kb(const PwArgs& a, int nb, unsigned grid, hipStream_t s) {
    if (nb == 2) hipLaunchKernelGGL((pw1x1_bf16_kernel<KB, 2, PD>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((pw1x1_bf16_kernel<KB, 1, PD>), dim3(grid), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

}  // namespace

bool pw1x1_applies(const PwArgs& a) {
    if (!unetrir_cfg().pw1x1) return false;
    const long long M = (long long)a.B * a.PH * a.PW;
    if (M <= 0 || M >= (1ll << 31) || (long long)a.B * a.OH * a.OW >= (1ll << 31) || (long long)a.B * a.IH * a.IW >= (1ll << 31)) return false;
    if (a.N % 32 != 0 || a.N > 1024) return false;
    if (!(a.C == 8 || a.C == 16 || a.C == 32 || a.C == 64 || a.C == 128 || a.C == 256)) return false;     // the KB instantiations of launch_pw1x1_bf16; other
                                                                                                            // channel counts stay on the tap-table kernel
    if ((a.ldi & 7) || (a.ldo & 7) || (a.addend && (a.ldadd & 7))) return false;
    if ((((uintptr_t)a.in | (uintptr_t)a.out | (uintptr_t)a.w | (uintptr_t)a.addend) & 15) != 0) return false;
    if (a.SI != 1 && a.SI != 2) return false;
    if (a.SO != 1 && a.SO != 2) return false;
    if (a.fill && a.SO != 2) return false;
    if (a.colstat && a.addend) return false;
    return true;
}

// channel tile and workgroup count of a launch: NT = 64 where N is a multiple of 64 and the input has at most 64 channels, else 32;
// one workgroup = 4 waves x their strips of 32 pixels; the workgroup count per channel tile is a multiple of 8 (XCD mapping).
// One or two workgroups per CU in total (measured, batch 32, scripts/micro_pw1x1.py: 32 -> 32 @ 128 x 128 with 512 / 2048 / 4096
// workgroups: 15 / 26 / 45 us; 64 -> 64 @ 64 x 64 with 256 / 1024: 14 / 29 us): every workgroup pays for its kernel tile, its bias
// and its row of column statistics, a wave keeps its strips in flight by itself.
static void pw1x1_plan(const PwArgs& a, int* nb, unsigned* per_tile) {
    *nb = (a.N % 64 == 0 && a.C <= 64) ? 2 : 1;
    const int ntN = a.N / (32 * *nb);
    const long long M = (long long)a.B * a.PH * a.PW;
    const long long groups = (M + 127) / 128;              // 4 strips
    long long g = ((*nb == 1 && a.C <= 32) ? 512 : 256) / ntN;
    if (g < 8) g = 8;
    if (g > groups) g = groups;
    g = (g + 7) / 8 * 8;
    *per_tile = (unsigned)g;
}

long long pw1x1_colstat_rows(const PwArgs& a) {
    int nb; unsigned g;
    pw1x1_plan(a, &nb, &g);
    return (long long)g;
}

int launch_pw1x1_bf16(const PwArgs& a, hipStream_t s) {
    int nb; unsigned g;
    pw1x1_plan(a, &nb, &g);
    const unsigned grid = g * (unsigned)(a.N / (32 * nb));
    const int kb = a.C <= 16 ? 1 : a.C / 16;
    switch (kb) {
        case 1: return launch_kb<1, 4>(a, nb, grid, s);
        case 2: return launch_kb<2, 4>(a, nb, grid, s);
        case 4: return launch_kb<4, 3>(a, nb, grid, s);
        case 8: return launch_kb<8, 2>(a, 1, grid, s);
        case 16: return launch_kb<16, 2>(a, 1, grid, s);
        default: return UNETRIR_EINVAL;
    }
}
