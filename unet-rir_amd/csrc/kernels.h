// kernels.h - internal declarations shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/unetrir.h"

#define UNETRIR_MAX_TAPS 36

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
int launch_igemm_wgrad(WgradArgs a, float* dw, float reg, const float* w, void* ws, size_t ws_bytes, hipStream_t s);
int launch_transpose_weight(const float* w, float* wt, int N, int T, int C, hipStream_t s);
int launch_splitk_reduce(const float* part, int nsplit, size_t n, float* out, float reg, const float* w, hipStream_t s);
