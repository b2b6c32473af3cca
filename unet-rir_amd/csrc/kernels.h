// kernels.h - internal declarations shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/unetrir.h"

#define UNETRIR_MAX_TAPS 36

// Kernel-selection switches (include/unetrir.h: unetrir_config): read once from the environment, replaceable through
// unetrir_set_config (tests, A/B scripts).  No other environment variable is read by the library.
const unetrir_config& unetrir_cfg();

// Timing ablations (results invalid: a kernel with a stage switched off) exist only in the separate build that
// scripts/ use (build.py --ablations -> libunetrir_abl.so, -DUNETRIR_ABLATIONS); in the product library UNETRIR_ABL(...) is
// a compile-time false and the code behind it is not emitted.
#ifdef UNETRIR_ABLATIONS
extern int g_unetrir_abl;                 // set through unetrir_abl_set()
#define UNETRIR_ABL_HOST() g_unetrir_abl
#define UNETRIR_ABL(v, bit) (((v) & (bit)) != 0)
#else
#define UNETRIR_ABL_HOST() 0
#define UNETRIR_ABL(v, bit) false
#endif

// Geometry of one implicit-GEMM launch.  The iteration grid is B x PH x PW "tile pixels" p;
// the input pixel of tap t is (py*SI + dy_t, px*SI + dx_t) and the output pixel is
// (py*SO + ooy, px*SO + oox).  tap[t] packs dy (int8) | dx (int8) << 8 | weight-tap index << 16.
struct IgemmGeom {
    int B, PH, PW;
    int IH, IW, C, ldi;
    int OH, OW, N, ldo;
    int SI, SO, ooy, oox;
    int ntaps, wtaps;
    uint32_t tap[UNETRIR_MAX_TAPS];
};

struct IgemmArgs {
    IgemmGeom g;
    const float* in;
    const float* w;        // [N][wtaps][C]
    const float* bias;     // nullable
    const float* addend;   // nullable, indexed like out with ldadd
    int ldadd;
    float* out;
    int ksplit;            // > 1: blockIdx.y owns a K slice and writes raw partial sums to part[slice][pixel][N]
    float* part;
};

struct WgradArgs {
    IgemmGeom g;           // uses B,PH,PW (= dy grid), IH,IW,C,ldi (x), N, SI, taps
    const float* x;
    const float* dy;
    int lddy;
    float* part;           // [nsplit][N][wtaps*C]
    long long chunks_per_split;
};

int launch_igemm_fwd(const IgemmArgs& a, hipStream_t s);
int wgrad_plan(const IgemmGeom& g, int* nsplit, long long* chunks_per_split);
int launch_igemm_wgrad(WgradArgs a, float* dw, float reg, const float* w, void* ws, size_t ws_bytes, hipStream_t s, int bf16_operands = 0);
int launch_transpose_weight(const float* w, float* wt, int N, int T, int C, hipStream_t s);
int launch_splitk_reduce(const float* part, int nsplit, size_t n, float* out, float reg, const float* w, hipStream_t s);
// While a sink is set (this host thread), launch_splitk_reduce records its arguments there instead of launching: the
// *_wgrad_partials_* entry points (api.hip) run the ordinary weight-gradient launchers under it.
void set_reduce_sink(unetrir_reduce_desc* d);

// 3x3 weight gradient with a halo-staged x patch (wgrad3x3.hip)
struct Wgrad3Args {
    const float* x; int ldx; int IH, IW;      // layer input  [B,IH,IW,C]
    const float* dy; int lddy; int OH, OW;    // output grad  [B,OH,OW,N]
    int B, C, N;
    int pad_t, pad_l;                         // TF 'same' pad_before per axis
    float* part;                              // [nsplit][N][9][C]
    int patches_per_split, npy, npx;
};
size_t wgrad3x3_ws_bytes(int stride, int B, int OH, int OW, int N, int C);
int launch_wgrad3x3(Wgrad3Args a, int stride, float* dw, float reg, const float* w, void* ws, size_t ws_bytes, hipStream_t s);

// ---- bf16-storage variants (igemm_bf16.hip): activations / weight work copies bf16, accumulate fp32 ----
struct IgemmArgsH {
    IgemmGeom g;
    const __bf16* in;
    const __bf16* w;       // [N][wtaps][C] bf16
    const float* bias;     // nullable, fp32
    const __bf16* addend;  // nullable
    int ldadd;
    __bf16* out;
    float* colstat;        // nullable: [pixel tile][N][2] per-channel (sum, sum of squares) of the stored bf16 output, one row per
                           // 128-pixel tile of the launch (row = tile index); requires addend == nullptr
};
// The tap-table launch for small problems (igemm2_bf16.hip): pixel-tile height it would use for an iteration grid of M pixels, N
// output channels and ncls launches sharing one grid (1, or the 4 parity classes of a stride-2 transposed layer); 0 = not taken
// (switch off / too large): the general kernel with 128-pixel tiles.
int igemm_bf16_tile_m(long long M, int N, int ncls);
int launch_igemm2_fwd_bf16(const IgemmArgsH* a, int ncls, hipStream_t s);
struct IgemmArgsH4 { IgemmArgsH a[4]; };
// rows of column statistics one tap-table launch writes (= its pixel tiles)
inline long long igemm_colstat_rows(long long M, int N, int ncls = 1) {
    const int bm = igemm_bf16_tile_m(M, N, ncls);
    return (M + (bm ? bm : 128) - 1) / (bm ? bm : 128);
}
struct Wgrad3ArgsH {
    const __bf16* x; int ldx; int IH, IW;
    const __bf16* dy; int lddy; int OH, OW;
    int B, C, N;
    int pad_t, pad_l;
    float* part;                              // fp32 partial slabs [nsplit][N][9][C]
    int patches_per_split, npy, npx;
    int xcd_remap;                            // wgrad3x3g: workgroups that share x / dy patches (one split, all tiles) on one XCD
};
int launch_igemm_fwd_bf16(const IgemmArgsH& a, hipStream_t s);
int launch_igemm_fwd_bf16_x4(const IgemmArgsH* a, hipStream_t s);
int launch_wgrad3x3_bf16(Wgrad3ArgsH a, int stride, float* dw, float reg, const float* w, void* ws, size_t ws_bytes, hipStream_t s);
bool wgrad3x3d_applies(const Wgrad3ArgsH& a);       // stride 2, even sizes: LDS-DMA kernel with the de-interleaved x patch
size_t wgrad3x3d_ws_bytes(int B, int OH, int OW, int N, int C);
int launch_wgrad3x3d_bf16(Wgrad3ArgsH a, float* dw, float reg, const float* w, void* ws, size_t ws_bytes, hipStream_t s);
size_t wgrad1x1_bf16_ws_bytes(int B, int OH, int OW, int N, int C);
int launch_wgrad1x1_bf16(Wgrad3ArgsH a, int stride, float* dw, float reg, const float* w, void* ws, size_t ws_bytes, hipStream_t s);
#define WGRAD3X3R_NOT_TAKEN (-12345)
int launch_wgrad3x3g_bf16(Wgrad3ArgsH a, float* dw, float reg, const float* w, void* ws, size_t ws_bytes, hipStream_t s);
int launch_wgrad3x3r_bf16(Wgrad3ArgsH a, float* dw, float reg, const float* w, void* ws, size_t ws_bytes, hipStream_t s);
int launch_cast_weight(const float* w, void* o, int N, int T, int C, int Cp, hipStream_t s);
int launch_cast_weights_batched(const unetrir_cast_desc* desc_dev, int n_layers, hipStream_t s);
int launch_transpose_cast_weight(const float* w, void* wt, int N, int T, int C, int Np, hipStream_t s);

// 3x3 stride-1 conv / data gradient with the halo patch staged once in LDS (conv3x3.hip); T = float or __bf16
struct Conv3Args {
    const void* in; int ldi;
    const void* w;            // [N][9][C], element type T
    const float* bias;        // nullable
    const void* addend; int ldadd;
    void* out; int ldo;
    int B, H, W, C, N;
    int flip;                 // 0: forward (weight tap t at offset t); 1: data gradient (weight tap 8-t at offset t)
    const void* wpk;          // nullable; conv3x3d only: the packed copy of w (unetrir_conv3x3s2_packed_elems)
    float* colstat;           // nullable; conv3x3g / conv3x3r<4,1> only: [pixel tile][N][2] per-channel (sum, sum of squares) of the
                              // stored output, one row per 16 x 32 pixel tile (row = (img * tiles_y + ty) * tiles_x + tx)
};
int launch_conv3x3(const Conv3Args& a, int bf16, hipStream_t s);
int launch_conv3x3r_bf16(const Conv3Args& a, hipStream_t s);
bool head_mfma_applies(int W, int C);
int launch_head_fwd_mfma(const void* x, int ldx, int B, int H, int W, int C, const float* w, const float* bias, float* y, int ldy, hipStream_t s);
int launch_head_wgrad_mfma(const void* x, int ldx, int B, int H, int W, int C, const void* dy, int lddy, float* part, int max_blocks,
                           int* nblk_out, hipStream_t s);
bool head_dgrad_mfma_applies(int W, int C);
int launch_head_dgrad_mfma(const void* dy, int lddy, int B, int H, int W, const float* w, int C, void* dx, int lddx, hipStream_t s);
bool upconv3x3g_applies(const Conv3Args& a);
bool upconv3x3q_applies(const Conv3Args& a);        // upconv3x3g's layers with >= 512 tiles: persistent form
int launch_upconv3x3q_bf16(const Conv3Args& a, hipStream_t s);
int launch_upconv3x3g_bf16(const Conv3Args& a, hipStream_t s);
bool conv3x3g_applies(const Conv3Args& a);
bool conv3x3g_pair_applies(const Conv3Args& a);     // images <= 16 pixels wide: two images per tile
long long conv3x3g_colstat_rows(const Conv3Args& a);
bool stem3x3_applies(const Conv3Args& a);
int launch_stem3x3_bf16(const Conv3Args& a, hipStream_t s);
bool conv3x3_has_colstat(const Conv3Args& a);
bool conv3x3s_applies(const Conv3Args& a);          // 64 -> 64 channels: strip kernel with the whole 3x3 kernel resident in LDS
long long conv3x3s_colstat_rows(const Conv3Args& a);
int launch_conv3x3s_bf16(const Conv3Args& a, hipStream_t s);
// Run-time tile tickets of the persistent kernels: 64 group counters + 1 count of finished workgroups, zero between launches.
// One slot per stream (launches on a stream run in order; the last workgroup of a launch clears the slot); nullptr when more
// than 128 streams are in use - the kernels then keep their fixed assignment.
unsigned* sched_slot(hipStream_t s);
bool conv3x3p_applies(const Conv3Args& a);          // conv3x3g's layers with >= 512 tiles: persistent form, continuous K loop across tiles
long long conv3x3p_colstat_rows(const Conv3Args& a);
int launch_conv3x3p_bf16(const Conv3Args& a, hipStream_t s);
bool conv3x3d_applies(const Conv3Args& a);          // 3x3 stride 2, forward form (H, W = input size): persistent LDS-DMA kernel
int launch_conv3x3d_bf16(const Conv3Args& a, hipStream_t s);
bool conv3x3h_applies(const Conv3Args& a);
int launch_conv3x3h_bf16(const Conv3Args& a, hipStream_t s);
int launch_conv3x3g_bf16(const Conv3Args& a, hipStream_t s);
int launch_dense_fwd(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int B, int K, int N,
                     void* ws, size_t ws_bytes, hipStream_t s);
size_t dense_fwd_ws_bytes(int B, int K, int N);
int launch_splitk_rows_reduce(const float* part, int nsplit, long long M, int N, const float* bias, float* y, int ldy, hipStream_t s);
bool dense_dgrad_applies(int B, int K, int N);
size_t dense_dgrad_ws_bytes(int B, int K, int N);
int launch_dense_dgrad(const float* dy, int lddy, const float* w, float* dx, int lddx, int B, int K, int N, void* ws, size_t ws_bytes,
                       hipStream_t s);
int launch_upconv3x3(const Conv3Args& a, int bf16, hipStream_t s);

// 1x1 convolutions (pw1x1.hip): out[opix(p)][n] = sum_c in[ipix(p)][c] w[n][c] + bias[n] (+ addend) over the iteration grid
// B x PH x PW; ipix = (b, py SI, px SI) of the IH x IW input grid, opix = (b, py SO, px SO) of the OH x OW output grid; fill
// (SO = 2): the three other pixels of each 2 x 2 output cell are written with bias (+ addend).
struct PwArgs {
    const __bf16* in; int ldi; int IH, IW;
    const __bf16* w;                 // [N][C], C contiguous
    const float* bias;               // nullable
    const __bf16* addend; int ldadd; // nullable; indexed like out (may be out itself)
    __bf16* out; int ldo; int OH, OW;
    int B, PH, PW, SI, SO, fill;
    int C, N;
    float* colstat;                  // nullable: [pw1x1_colstat_rows][N][2]; requires addend == nullptr
};
bool pw1x1_applies(const PwArgs& a);
long long pw1x1_colstat_rows(const PwArgs& a);
int launch_pw1x1_bf16(const PwArgs& a, hipStream_t s);   // H, W = coarse (input) grid; output 2H x 2W
