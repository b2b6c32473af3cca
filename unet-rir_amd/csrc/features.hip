// features.hip - waveform <-> network feature transforms on the device (gfx950): the two ends of the reference's data path.
//
//   analysis  (preprocess.py:13-18, :26-32, :65-70; Loader's mean removal :56):
//       wav [B][T] fp32 -> centred Hann STFT (librosa.stft semantics: the win_length window centred in n_fft zeros, n_fft/2
//       samples of 'reflect' or zero padding, hop_length stride) -> (|S|, angle S) -> Normalizer.normalize -> zero padding
//       to the network's [B][2][H][W] planes, in ONE kernel: one workgroup per (frame column, waveform).
//   synthesis (postprocess.py:68-73, :127-136 'ph'; Normalizer.denormalize preprocess.py:34-41; un_pad :107-113):
//       feature [B][2][H][W] fp32 -> first n_bins x n_frames block -> amp (cos p + i sin p) -> per-frame inverse real DFT ->
//       windowed overlap-add / squared-window envelope -> waveform [B][hop (frames-1)], as a GATHER: one workgroup per 64
//       output samples sums the (at most win/hop + 1) frames that reach them in a fixed order - no atomics.
//
// Both are direct DFTs in fp64 (n_fft = 256, 128 non-zero taps: 2 x 33 k MAC per frame, 0.3 GFLOP per batch of 32 - far below
// anything worth an FFT's data movement); twiddles come from sincospi in fp64 and live in LDS, indexed (k n) mod n_fft, so the
// result agrees with a double-precision FFT to ~1e-13 and the fp32 outputs are rounded once.  HBM traffic is the waveform
// read (n_fft/hop = 4 times, L2-resident) plus one write of the feature planes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "kernels.h"

namespace {

constexpr int FEAT_MAX_NFFT = 1024;
constexpr double FEAT_MD = 100.0;               // Normalizer.md (preprocess.py:23)
constexpr double FEAT_EP = 1e-5;                // 10^(-md/20)    (preprocess.py:24)
constexpr double FEAT_REF = 128.0;              // amp / 128      (preprocess.py:27)
constexpr double FEAT_PI = 3.14159265358979323846;
constexpr double FEAT_TINY32 = 1.1754943508222875e-38;   // numpy.finfo(float32).tiny: librosa's envelope threshold

// periodic Hann of win samples centred in n_fft zeros, tap n
__device__ __forceinline__ double hann_tap(int n, int n_fft, int win) {
    const int m = n - (n_fft - win) / 2;
    if (m < 0 || m >= win) return 0.0;
    return 0.5 - 0.5 * cospi(2.0 * (double)m / (double)win);
}

__global__ __launch_bounds__(256) void stft_feature_kernel(const float* __restrict__ wav, int T, int n_fft, int win, int hop,
                                                           int pad_mode, int remove_mean, int normalize, int n_frames,
                                                           float* __restrict__ out, int H, int W) {
    __shared__ double tw_c[FEAT_MAX_NFFT], tw_s[FEAT_MAX_NFFT], xw[FEAT_MAX_NFFT], red[256];
    const int f = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    float* __restrict__ o_amp = out + ((size_t)b * 2 + 0) * H * W;
    float* __restrict__ o_ph = out + ((size_t)b * 2 + 1) * H * W;
    if (f >= n_frames) {                               // TensorPadder.col_transform: appended zero columns
        for (int k = tid; k < H; k += 256) { o_amp[(size_t)k * W + f] = 0.f; o_ph[(size_t)k * W + f] = 0.f; }
        return;
    }
    const float* __restrict__ x = wav + (size_t)b * T;
    double mean = 0.0;
    if (remove_mean) {                                 // every workgroup of a waveform sums it in the same fixed order
        double s = 0.0;
        for (int i = tid; i < T; i += 256) s += (double)x[i];
        red[tid] = s;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) red[tid] += red[tid + st];
            __syncthreads();
        }
        mean = red[0] / (double)T;
    }
    for (int n = tid; n < n_fft; n += 256) {
        double sn, cs;
        sincospi(2.0 * (double)n / (double)n_fft, &sn, &cs);
        tw_c[n] = cs; tw_s[n] = sn;
        const int i = f * hop + n - n_fft / 2;         // sample index in the unpadded signal
        double v;
        if (i < 0) v = pad_mode == 0 ? (double)x[-i] - mean : 0.0;                  // numpy 'reflect' (edge not repeated)
        else if (i >= T) v = pad_mode == 0 ? (double)x[2 * (T - 1) - i] - mean : 0.0;
        else v = (double)x[i] - mean;
        xw[n] = v * hann_tap(n, n_fft, win);
    }
    __syncthreads();
    const int nb = n_fft / 2 + 1, n_lo = (n_fft - win) / 2, n_hi = n_lo + win, mask = n_fft - 1;
    for (int k = tid; k < H; k += 256) {
        float a = 0.f, p = 0.f;                        // TensorPadder.row_transform: appended zero rows
        if (k < nb) {
            double re = 0.0, im = 0.0;
            int idx = (k * n_lo) & mask;
            for (int n = n_lo; n < n_hi; ++n) {
                const double v = xw[n];
                re += v * tw_c[idx];
                im -= v * tw_s[idx];
                idx = (idx + k) & mask;
            }
            double amp = sqrt(re * re + im * im), ph = atan2(im, re);
            if (normalize) {
                amp = (20.0 * log10(amp / FEAT_REF + FEAT_EP) + FEAT_MD) / FEAT_MD;
                ph = (ph + FEAT_PI) / (2.0 * FEAT_PI);
            }
            a = (float)amp; p = (float)ph;
        }
        o_amp[(size_t)k * W + f] = a;
        o_ph[(size_t)k * W + f] = p;
    }
}

constexpr int ISEG = 64;                                // output samples per workgroup

__global__ __launch_bounds__(256) void istft_feature_kernel(const float* __restrict__ feat, int H, int W, int n_frames,
                                                            int n_fft, int win, int hop, int denormalize,
                                                            float* __restrict__ wav, int T_out) {
    __shared__ double tw_c[FEAT_MAX_NFFT], tw_s[FEAT_MAX_NFFT], xr[FEAT_MAX_NFFT / 2 + 1], xi[FEAT_MAX_NFFT / 2 + 1];
    __shared__ double red[4][ISEG];
    const int tid = threadIdx.x, s = tid & (ISEG - 1), kg = tid >> 6;
    const int b = blockIdx.y, t0 = blockIdx.x * ISEG;
    const float* __restrict__ f_amp = feat + ((size_t)b * 2 + 0) * H * W;
    const float* __restrict__ f_ph = feat + ((size_t)b * 2 + 1) * H * W;
    const int nb = n_fft / 2 + 1, n_lo = (n_fft - win) / 2, n_hi = n_lo + win, mask = n_fft - 1;
    for (int n = tid; n < n_fft; n += 256) {
        double sn, cs;
        sincospi(2.0 * (double)n / (double)n_fft, &sn, &cs);
        tw_c[n] = cs; tw_s[n] = sn;
    }
    // frames whose non-zero window taps reach [p0, p0 + ISEG) in padded coordinates: n = p - hop f in [n_lo, n_hi)
    const int p0 = t0 + n_fft / 2, p = p0 + s;
    int f_lo = (p0 - n_hi + 1 + hop - 1) / hop;          // ceil; the numerator may be negative only when f_lo clamps to 0 anyway
    if (p0 - n_hi + 1 <= 0) f_lo = 0;
    int f_hi = (p0 + ISEG - 1 - n_lo) / hop;
    if (f_hi > n_frames - 1) f_hi = n_frames - 1;
    double acc = 0.0, wss = 0.0;
    for (int f = f_lo; f <= f_hi; ++f) {
        __syncthreads();                                 // the previous frame's spectrum is no longer read (first pass: twiddles)
        for (int k = tid; k < nb; k += 256) {
            double amp = (double)f_amp[(size_t)k * W + f], ph = (double)f_ph[(size_t)k * W + f];
            if (denormalize) {
                amp = (pow(10.0, (amp * FEAT_MD - FEAT_MD) / 20.0) - FEAT_EP) * FEAT_REF;
                ph = ph * 2.0 * FEAT_PI - FEAT_PI;
                const double v = ph + FEAT_PI;           // python's (phase + pi) % (2 pi) - pi
                ph = v - floor(v / (2.0 * FEAT_PI)) * (2.0 * FEAT_PI) - FEAT_PI;
            }
            double sn, cs;
            sincos(ph, &sn, &cs);
            xr[k] = amp * cs; xi[k] = amp * sn;
        }
        __syncthreads();
        const int n = p - hop * f;
        if (n < n_lo || n >= n_hi) continue;             // zero window tap (the barriers above are reached by every thread
                                                         // before this point in each iteration)
        const double w = hann_tap(n, n_fft, win);
        // irfft sample n: (1/N) [X0 + (-1)^n X_{N/2} + 2 sum_{0<k<N/2} (Re X_k cos(2 pi k n / N) - Im X_k sin(2 pi k n / N))];
        // the imaginary parts of bins 0 and N/2 are ignored, as numpy.fft.irfft does
        double part = 0.0;
        for (int k = kg; k < nb; k += 4) {
            const int idx = (k * n) & mask;
            if (k == 0 || k == nb - 1) part += xr[k] * tw_c[idx];
            else part += 2.0 * (xr[k] * tw_c[idx] - xi[k] * tw_s[idx]);
        }
        acc += w * part / (double)n_fft;
        if (kg == 0) wss += w * w;
    }
    red[kg][s] = acc;
    __syncthreads();
    if (kg == 0 && t0 + s < T_out) {
        double y = ((red[0][s] + red[1][s]) + red[2][s]) + red[3][s];
        if (wss > FEAT_TINY32) y /= wss;
        wav[(size_t)b * T_out + t0 + s] = (float)y;
    }
}

bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

extern "C" {

int unetrir_stft_frames(int T, int hop_length) { return hop_length > 0 && T >= 0 ? 1 + T / hop_length : 0; }

int unetrir_stft_features_f32(const float* wav, int B, int T, int n_fft, int win_length, int hop_length, int pad_mode,
                              int remove_mean, int normalize, float* out, int H, int W, unetrir_stream_t stream) {
    if (!wav || !out || B <= 0 || T <= 0 || !pow2(n_fft) || n_fft < 4 || n_fft > FEAT_MAX_NFFT || win_length <= 0 ||
        win_length > n_fft || hop_length <= 0 || (pad_mode != 0 && pad_mode != 1) || T <= n_fft / 2 ||
        (long long)T + n_fft > 0x7fffffffLL)
        return UNETRIR_EINVAL;
    const int n_frames = 1 + T / hop_length;
    // TensorPadder.get_needed_transform (preprocess.py:82-92) pads only when neither dimension exceeds the target
    if (H < n_fft / 2 + 1 || W < n_frames || B > 65535) return UNETRIR_EINVAL;
    hipLaunchKernelGGL(stft_feature_kernel, dim3(W, B), dim3(256), 0, (hipStream_t)stream, wav, T, n_fft, win_length,
                       hop_length, pad_mode, remove_mean, normalize, n_frames, out, H, W);
    return (int)hipGetLastError();
}

int unetrir_istft_features_f32(const float* feat, int B, int H, int W, int n_bins, int n_frames, int n_fft, int win_length,
                               int hop_length, int denormalize, float* wav, unetrir_stream_t stream) {
    if (!feat || !wav || B <= 0 || B > 65535 || !pow2(n_fft) || n_fft < 4 || n_fft > FEAT_MAX_NFFT || win_length <= 0 ||
        win_length > n_fft || hop_length <= 0 || n_bins != n_fft / 2 + 1 || n_bins > H || n_frames < 2 || n_frames > W ||
        (long long)hop_length * n_frames + n_fft > 0x7fffffffLL)
        return UNETRIR_EINVAL;
    const int T_out = hop_length * (n_frames - 1);
    hipLaunchKernelGGL(istft_feature_kernel, dim3((T_out + ISEG - 1) / ISEG, B), dim3(256), 0, (hipStream_t)stream, feat, H, W,
                       n_frames, n_fft, win_length, hop_length, denormalize, wav, T_out);
    return (int)hipGetLastError();
}

}  // extern "C"
