/*
 * unetrir.h - C ABI of the MI355X (gfx950) U-Net train-step hot path.
 *
 * The reference (igmsalinas/unet-rir) has no FFI: its hot path is a Keras graph
 * (dl_models/u_net.py:201-251) that TensorFlow lowers to cuDNN / NCCL.  Each
 * entry point below replaces the kernel TensorFlow would dispatch for one Keras
 * call site; the call site is cited next to it.  INTEGRATION.md shows the
 * ctypes binding a maintainer would add.
 *
 * Conventions (all entry points):
 *   - device pointers only, caller owns every buffer, 16-byte aligned;
 *   - activations are NHWC fp32 (the reference's own layout) addressed as
 *     base + pixel * ld + channel, `ld` (pixel stride in elements) >= channels and a
 *     multiple of 4, so a channel-concat is two writers into one buffer (no copy);
 *   - Conv2D weights are [Cout][kh][kw][Cin] ("OHWI"), Conv2DTranspose weights
 *     are [Cin][kh][kw][Cout] ("IHWO"): the layout the weight-gradient kernel
 *     produces; unetrir_transpose_weight_f32 makes the other one;
 *   - Cin and Cout multiples of 4 unless stated; TF padding='same' geometry;
 *   - asynchronous on `stream`, never allocate, never synchronise; thread-safe for
 *     distinct streams.  The library holds three pieces of process-global state, documented
 *     where they are declared: the kernel-selection switches (unetrir_config), the profiling
 *     brackets (unetrir_prof_*), and the tile-ticket slots of the persistent convolution
 *     kernels - a static device array of counters PER DEVICE (resolved for the device that is
 *     current at the launch; up to 16 devices per process), one slot per (device, stream) that
 *     has launched such a kernel (at most 128 streams per device; beyond that the kernels fall
 *     back to a fixed tile assignment), zero between launches (unetrir_reset_tile_tickets
 *     clears them from the host after a failed launch); no memory is allocated for it;
 *   - return 0 on success, a hipError_t value or UNETRIR_EINVAL otherwise.
 */
#ifndef UNETRIR_H
#define UNETRIR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNETRIR_EINVAL 10001
#define UNETRIR_ABI_VERSION 1

typedef void* unetrir_stream_t; /* hipStream_t */
typedef uint16_t unetrir_bf16;  /* bfloat16 storage (upper 16 bits of an IEEE fp32) */

/* Geometry of one Conv2D / Conv2DTranspose(padding='same') call.
 * H, W: spatial size of the layer INPUT (for the transpose: the low-res input). */
typedef struct {
    int B, H, W;
    int Cin, Cout;
    int k;      /* square kernel size: 1, 3 or 6 in the reference graph */
    int stride; /* 1 or 2 */
} unetrir_conv_geom;

int unetrir_abi_version(void);

/* ---- kernel-selection switches.  Several hand-written kernels can serve the same layer (e.g. a 3x3 stride-1 convolution:
 *      tap-table implicit GEMM -> register-staged patch kernel -> LDS-DMA kernels).  The dispatch picks the most
 *      specialised one that takes the shape; a switch set to 0 removes that kernel from the dispatch, so the layer runs on
 *      the next more general one - for A/B measurements and for parity cross-checks between kernels.  The switches are
 *      read ONCE, at first use, from the environment variables named below (the only variables the library reads) into
 *      an immutable snapshot of this struct; unetrir_set_config publishes a new snapshot atomically (tests, A/B scripts:
 *      process-global; a launch issued concurrently on another thread sees the old values or the new ones, never a mixture
 *      within one read - but a dispatch reads the switches more than once, so flip them between launches).  Defaults: all 1.
 *      (Rounds 3's measured refusals - one-launch BatchNorm `bn_fused`, the LDS-DMA tap-table kernel `igemm3`, the head with
 *      BatchNorm on its load path - were removed from the library in round 4; DESIGN.md keeps their numbers.)
 *        conv3x3        UNETRIR_CONV3X3        3x3 stride-1: patch-staged kernels at all (0: tap-table implicit GEMM)
 *        conv3x3g       UNETRIR_CONV3X3G       bf16 LDS-DMA kernel, > 64 output channels, and 64 output channels from > 64 input channels (64-channel tiles; conv3x3g.hip)
 *        conv3x3g_pair  UNETRIR_CONV3X3G_PAIR  its two-images-per-tile form for images <= 16 wide: 0 off, 1 when it yields
 *                                              >= 32 workgroups (64-channel tiles), 2 whenever the shape allows
 *        conv3x3h       UNETRIR_CONV3X3H       bf16 LDS-DMA kernel, <= 64 output channels where conv3x3g / conv3x3s do not take the layer (conv3x3h.hip)
 *        conv3x3s       UNETRIR_CONV3X3S       bf16 strip kernel, 64 -> 64 channels, kernel resident in LDS (conv3x3s.hip)
 *        conv3x3r       UNETRIR_CONV3X3R       bf16 register-staged row-reuse kernel (conv3x3r.hip)
 *        stem           UNETRIR_STEM           bf16 first layer, 8 stored input channels (stem3x3.hip)
 *        upconv3x3g     UNETRIR_UPCONV3X3G     bf16 LDS-DMA transposed / strided-dgrad kernel (upconv3x3g.hip)
 *        wgrad3x3g      UNETRIR_WGRAD3X3G      bf16 LDS-DMA 3x3 weight gradient (wgrad3x3g.hip)
 *        wgrad3x3r      UNETRIR_WGRAD3X3R      bf16 register-staged 3x3 weight gradient (wgrad3x3r.hip)
 *        head_mfma      UNETRIR_HEAD_MFMA      bf16 6x6 head on the matrix cores (head_mfma.hip)
 *        wgrad3x3d      UNETRIR_WGRAD3X3D      bf16 LDS-DMA 3x3 stride-2 weight gradient (wgrad3x3d.hip)
 *        conv3x3d       UNETRIR_CONV3X3D       bf16 LDS-DMA 3x3 stride-2 forward / transposed data gradient (conv3x3d.hip)
 *        conv3x3p       UNETRIR_CONV3X3P       bf16 persistent form of conv3x3g for layers with >= 512 tiles (conv3x3p.hip)
 *        upconv3x3q     UNETRIR_UPCONV3X3Q     bf16 persistent form of upconv3x3g for layers with >= 512 tiles (upconv3x3q.hip)
 *        dyn_tiles      UNETRIR_DYN_TILES      bf16 persistent kernels draw their tiles at run time (0: fixed assignment per workgroup)
 *        pw1x1          UNETRIR_PW1X1          bf16 register-streaming kernel of the 1x1 layers, forward and data gradient (pw1x1.hip)
 *        igemm2         UNETRIR_IGEMM2         bf16 tap-table kernel for small problems: 64-pixel tiles, two K chunks in flight (igemm2_bf16.hip) */
typedef struct {
    int conv3x3, conv3x3g, conv3x3g_pair, conv3x3h, conv3x3s, conv3x3r, stem, upconv3x3g, wgrad3x3g, wgrad3x3r, head_mfma,
        wgrad3x3d, conv3x3d, conv3x3p, upconv3x3q, dyn_tiles, pw1x1, igemm2;
} unetrir_config;
int unetrir_get_config(unetrir_config* out);
int unetrir_set_config(const unetrir_config* in);
/* Zero every tile-ticket slot of the CURRENT device from the host (synchronous; call with the device idle): after a launch that
 * failed or was aborted, or when dyn_tiles is toggled between launches.  A healthy run never needs it - the last workgroup of
 * every persistent launch clears its own slot. */
int unetrir_reset_tile_tickets(void);

/* ---- Conv2D(padding='same'): dl_models/u_net.py:269-276 (strided, stride 1|2),
 *      :366 (3x3 block conv), :248 (6x6 head), :262 (1x1 on the information vector).
 *      y = conv(x, w) + bias (+ addend): `addend` (nullable) is the Add() of
 *      dl_models/u_net.py:229 fused into the epilogue.                            */
int unetrir_conv2d_fwd_f32(const unetrir_conv_geom* g, const float* x, int ldx, const float* w,
                           const float* bias, const float* addend, int ldadd, float* y, int ldy,
                           unetrir_stream_t stream);
/* dL/dx of the above (TensorFlow: Conv2DBackpropInput via tape.gradient, main_training.py:267).
 * wt = weights transposed to [Cin][kh][kw][Cout]; dx = conv^T(dy) (+ addend). */
int unetrir_conv2d_dgrad_f32(const unetrir_conv_geom* g, const float* dy, int lddy, const float* wt,
                             const float* addend, int ldadd, float* dx, int lddx,
                             unetrir_stream_t stream);
/* dL/dw (Conv2DBackpropFilter): dw[Cout][k][k][Cin] = sum_pixels dy * x  + reg_coef * w.
 * Deterministic split-K: partials go to `ws` (>= unetrir_conv2d_wgrad_ws_bytes), then a
 * fixed-order reduction.  `w` may be NULL when reg_coef == 0. */
size_t unetrir_conv2d_wgrad_ws_bytes(const unetrir_conv_geom* g);
int unetrir_conv2d_wgrad_f32(const unetrir_conv_geom* g, const float* x, int ldx, const float* dy,
                             int lddy, float* dw, float reg_coef, const float* w, void* ws,
                             size_t ws_bytes, unetrir_stream_t stream);

/* ---- Conv2DTranspose(strides=2, padding='same'): dl_models/u_net.py:297-304.
 *      Defined as the adjoint of the SAME stride-2 conv on the 2H x 2W grid.
 *      fwd takes wt = [Cout][k][k][Cin]; dgrad and wgrad take/produce the primary
 *      [Cin][k][k][Cout].  g->H, g->W are the low-res input size, g->stride = 2. */
int unetrir_conv2d_transpose_fwd_f32(const unetrir_conv_geom* g, const float* x, int ldx,
                                     const float* wt, const float* bias, float* y, int ldy,
                                     unetrir_stream_t stream);
int unetrir_conv2d_transpose_dgrad_f32(const unetrir_conv_geom* g, const float* dy, int lddy,
                                       const float* w, const float* addend, int ldadd, float* dx,
                                       int lddx, unetrir_stream_t stream);
size_t unetrir_conv2d_transpose_wgrad_ws_bytes(const unetrir_conv_geom* g);
int unetrir_conv2d_transpose_wgrad_f32(const unetrir_conv_geom* g, const float* x, int ldx,
                                       const float* dy, int lddy, float* dw, float reg_coef,
                                       const float* w, void* ws, size_t ws_bytes,
                                       unetrir_stream_t stream);

/* Dense(N) (dl_models/u_net.py:259, Flatten -> Dense) on a small batch: y[B][N] = x[B][K] . w[N][K]^T + bias (bias nullable),
 * split-K over the workgroups so the weight matrix streams from every CU; ws >= unetrir_dense_fwd_ws_bytes.  The data
 * gradient is the same call with the [K][N] weight copy and no bias. */
size_t unetrir_dense_fwd_ws_bytes(int B, int K, int N);
int unetrir_dense_fwd_f32(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int B, int K,
                          int N, void* ws, size_t ws_bytes, unetrir_stream_t stream);
/* Data gradient of the same layer from the SAME [N][K] kernel (no transposed copy): dx[b][k] = sum_n dy[b][n] w[n][k] for
 * B <= 32 batch rows, K % 4 == 0 (unetrir_dense_dgrad_supported).  The kernel streams once with 16-byte loads; n is split
 * over the workgroups and reduced in a fixed order; ws >= unetrir_dense_dgrad_ws_bytes. */
int unetrir_dense_dgrad_supported(int B, int K, int N);
size_t unetrir_dense_dgrad_ws_bytes(int B, int K, int N);
int unetrir_dense_dgrad_f32(const float* dy, int lddy, const float* w, float* dx, int lddx, int B, int K, int N, void* ws,
                            size_t ws_bytes, unetrir_stream_t stream);

/* [N][T][C] -> [C][T][N] (swap the channel roles of a conv kernel, taps kept). */
int unetrir_transpose_weight_f32(const float* w, float* wt, int N, int T, int C,
                                 unetrir_stream_t stream);

/* ---- BatchNormalization() training mode + Activation('relu'):
 *      dl_models/u_net.py:367-369 (Keras defaults eps=1e-3, momentum=0.99).
 *      x is [P][C] with pixel stride ldx.  stats: per-channel batch mean / biased
 *      variance (fp64 accumulation, two-stage deterministic), writes
 *      scale = gamma*rsqrt(var+eps), shift = beta - mean*scale into `affine[2*C]`,
 *      mean and rstd into `saved[2*C]`, and updates the moving statistics
 *      (moving = momentum*moving + (1-momentum)*batch, unbiased variance) when non-NULL.
 *      ws >= unetrir_bn_ws_bytes(P, C). */
size_t unetrir_bn_ws_bytes(long long P, int C);
int unetrir_bn_stats_f32(const float* x, int ldx, long long P, int C, const float* gamma,
                         const float* beta, float eps, float momentum, float* moving_mean,
                         float* moving_var, float* affine, float* saved, void* ws, size_t ws_bytes,
                         unetrir_stream_t stream);
/* y = max(x*scale + shift, 0) (relu != 0) or x*scale + shift */
int unetrir_bn_apply_f32(const float* x, int ldx, long long P, int C, const float* affine, int relu,
                         float* y, int ldy, unetrir_stream_t stream);
/* Backward of BN(+ReLU): given da = dL/d(output), the saved conv output x and the batch
 * statistics, writes dx = dL/dx and dgamma, dbeta.  (TensorFlow: FusedBatchNormGradV3 + ReluGrad.) */
int unetrir_bn_bwd_f32(const float* da, int ldda, const float* x, int ldx, long long P, int C,
                       const float* gamma, const float* affine, const float* saved, int relu,
                       float* dx, int lddx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                       unetrir_stream_t stream);
/* Per-channel column sum (BiasAddGrad): out[c] = sum_p x[p][c]. */
int unetrir_colsum_f32(const float* x, int ldx, long long P, int C, float* out, void* ws,
                       size_t ws_bytes, unetrir_stream_t stream);
/* ReLU without BN (BatchNorm=False graphs): y = max(x, 0); dx = da * (x > 0). */
int unetrir_relu_fwd_f32(const float* x, int ldx, long long P, int C, float* y, int ldy,
                         unetrir_stream_t stream);
int unetrir_relu_bwd_f32(const float* da, int ldda, const float* x, int ldx, long long P, int C,
                         float* dx, int lddx, unetrir_stream_t stream);

/* ---- residual blocks of the ResAE graph (dl_models/res_ae.py:310-371, :453-514): BatchNormalization -> Add ->
 *      LeakyReLU() (keras default alpha 0.3).  Activation codes: 0 none, 1 ReLU, 2 LeakyReLU(0.3); the `relu` argument of
 *      unetrir_bn_apply / unetrir_bn_bwd accepts the same codes.
 *      bn_act_add: y = act(x*scale + shift + addend) (affine NULL = identity, addend NULL = none).
 *      act_bwd:    g = da * act'(out), derivative chosen by the sign of the activation output.
 *      add:        y = a + b (n % 4 == 0). */
int unetrir_bn_act_add_f32(const float* x, int ldx, long long P, int C, const float* affine, int act,
                           const float* addend, int ldadd, float* y, int ldy, unetrir_stream_t stream);
int unetrir_act_bwd_f32(const float* da, int ldda, const float* out, int ldo, long long P, int C, int act, float* g,
                        int ldg, unetrir_stream_t stream);
int unetrir_add_f32(const float* a, const float* b, float* y, long long n, unetrir_stream_t stream);
/* Backward of the whole junction out = act(BatchNormalization(x) + skip) in three launches (reduce, finalize, apply) instead
 * of seven: with g = da * act'(out) (decided by the sign of the stored OUTPUT `out`), dgamma / dbeta and dx are the BatchNorm
 * backward of g, and gskip = g (+ gskip_add; gskip_add may be gskip itself: in-place accumulation when `skip` has another
 * consumer whose gradient is already there; gskip NULL: the other operand needs no gradient).  ws >= unetrir_bn_ws_bytes(P, C). */
int unetrir_bn_bwd_junction_f32(const float* da, int ldda, const float* x, int ldx, const float* out, int ldo, long long P, int C,
                                const float* affine, const float* saved, int act, float* dx, int lddx, float* gskip, int ldgs,
                                const float* gskip_add, int ldga, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                                unetrir_stream_t stream);

/* ---- boundary layout: the model input is NCHW [B,2,H,W] (north_star); the first conv reads
 *      NHWC padded to 4 channels.  nchw -> [B,H,W,Cpad] with zero fill, and back. */
int unetrir_nchw_to_nhwc_pad_f32(const float* x, int B, int C, int H, int W, float* y, int Cpad,
                                 unetrir_stream_t stream);

/* ---- head: Activation('sigmoid') (dl_models/u_net.py:249) fused with the loss of
 *      main_training.py:184-235.  logits is NHWC [B*H*W][ldl>=2]; pred/target are NCHW
 *      [B,2,H,W].  Writes pred = sigmoid(logits); partial sums of the data loss into ws and
 *      the final scalar loss_out[0] = sum(alpha*(a-a^)^2 + (1-alpha)*(1-cos(2pi(p-p^)))) /
 *      (2*H*W*global_batch), loss_out[1] = sum of the amplitude term, loss_out[2] = sum of
 *      the phase term; dlogits [B*H*W][4] = dL/dlogits (channels 2,3 zero). */
size_t unetrir_loss_ws_bytes(long long npix);
int unetrir_sigmoid_loss_f32(const float* logits, int ldl, const float* target, int B, int H, int W,
                             float alpha, float inv_norm, float* pred, float* dlogits,
                             float* loss_out, void* ws, size_t ws_bytes, unetrir_stream_t stream);
/* The same with the two switches of compute_loss (main_training.py:38-39): `phase_ref` (nullable; diff_loss, :214-217) = the
 * network INPUT, NCHW [B,2,H,W]: the phase target becomes target[:,1] - phase_ref[:,1]; `phase_weight` (nullable; sigmoid_loss,
 * :221-222 with preprocess.py:116-121) = [W] column weights multiplied into the phase term of the loss and its gradient.
 * loss_out[2] stays the UNWEIGHTED phase sum (the train_loss_phase metric, main_training.py:278-284). */
int unetrir_sigmoid_loss_ex_f32(const float* logits, int ldl, const float* target, const float* phase_ref,
                                const float* phase_weight, int B, int H, int W, float alpha, float inv_norm, float* pred,
                                float* dlogits, float* loss_out, void* ws, size_t ws_bytes, unetrir_stream_t stream);
/* sigmoid only (inference / forward without a target) */
int unetrir_sigmoid_nchw_f32(const float* logits, int ldl, int B, int H, int W, float* pred,
                             unetrir_stream_t stream);
/* dlogits from an upstream gradient on the NCHW prediction: dl = dpred * p * (1 - p) */
int unetrir_sigmoid_bwd_f32(const float* pred, const float* dpred, int B, int H, int W,
                            float* dlogits, unetrir_stream_t stream);

/* ---- output head Conv2D(2, (6,6), padding='same') (dl_models/u_net.py:248), direct (non-MFMA) kernels: with 2
 *      output channels an implicit GEMM would waste 15/16 of every MFMA tile.  C % 8 == 0.
 *      fwd: w is [>=2][6][6][C] (rows 0,1 used); y is [B*H*W][ldy], ldy >= 2; with ldy >= 4 channels 2,3 are zeroed.
 *      wgrad: dw[0..1][6][6][C] = sum_pixels dy[p][0..1] * x[p+off][c]; dy is [B*H*W][lddy>=2]; further rows of a
 *      padded kernel gradient are left untouched.  ws >= unetrir_head6x6_wgrad_ws_bytes(C). */
int unetrir_head6x6_supported(int C);
int unetrir_head6x6_fwd_f32(const float* x, int ldx, int B, int H, int W, int C, const float* w, const float* bias,
                            float* y, int ldy, unetrir_stream_t stream);
size_t unetrir_head6x6_wgrad_ws_bytes(int C);
int unetrir_head6x6_wgrad_f32(const float* x, int ldx, int B, int H, int W, int C, const float* dy, int lddy,
                              float* dw, void* ws, size_t ws_bytes, unetrir_stream_t stream);

/* ---- information vector branch: Embedding(2000,256) -> Flatten (dl_models/u_net.py:257-258).
 *      idx int32 [n_idx]; out [n_idx][dim].  Backward is a deterministic gather-by-row:
 *      dtable[v] = sum over positions with idx == v, in position order. */
int unetrir_embedding_fwd_f32(const int* idx, int n_idx, const float* table, int vocab, int dim,
                              float* out, unetrir_stream_t stream);
int unetrir_embedding_bwd_f32(const int* idx, int n_idx, const float* dout, int vocab, int dim,
                              float* dtable, unetrir_stream_t stream);
/* y = x * mask (Dropout(.3), dl_models/u_net.py:260; mask already scaled by 1/(1-p)) */
int unetrir_mul_f32(const float* x, const float* m, float* y, long long n, unetrir_stream_t stream);
/* sum_i x_i^2 over n elements into out[0] (+= when accumulate != 0): l2 regulariser value. */
int unetrir_sumsq_f32(const float* x, long long n, float coef, float* out, int accumulate,
                      void* ws, size_t ws_bytes, unetrir_stream_t stream);

/* ---- tf.keras.optimizers.Adam (main_training.py:168-169, :268): beta1 .9, beta2 .999,
 *      epsilon 1e-7 outside the sqrt, lr_t = lr*sqrt(1-b2^t)/(1-b1^t).  One launch over a flat
 *      parameter buffer.  grad_scale multiplies g first (1/world_size style rescaling). */
int unetrir_adam_f32(float* theta, const float* g, float* m, float* v, long long n, float lr_t,
                     float beta1, float beta2, float eps, float grad_scale,
                     unetrir_stream_t stream);

/* ---- the other optimizers main_training.py:164-169 selects by name.  SGD(learning_rate): theta -= lr * grad_scale * g.
 *      Nadam(learning_rate) (tf.keras optimizer_v2: Nesterov Adam with the momentum schedule mu_t = beta1 (1 - 0.5 * 0.96^(0.004 t))):
 *      m, v as Adam; theta -= lr (c_g g + c_m m) / (sqrt(c_v v) + eps) with c_g = (1 - mu_t) / (1 - prod_{i<=t} mu_i),
 *      c_m = mu_{t+1} / (1 - prod_{i<=t+1} mu_i), c_v = 1 / (1 - beta2^t), computed by the caller per step. */
int unetrir_sgd_f32(float* theta, const float* g, long long n, float lr, float grad_scale, unetrir_stream_t stream);
int unetrir_nadam_f32(float* theta, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                      float c_g, float c_m, float c_v, float grad_scale, unetrir_stream_t stream);

/* ---- step counters in DEVICE memory, so that a whole train step can be captured once into a HIP graph and replayed: the values
 *      that change from step to step (Adam's bias-corrected rate, the dropout draw number) are then read by the kernels instead
 *      of being launch arguments.
 *      state: 3 x uint64 {Adam step count t, dropout draws made so far, first draw of the current step};
 *      cfg:   5 x float  {lr, beta1, beta2, eps, grad_scale} as the host last set them;
 *      hyper: 5 x float  written by step_advance: {lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t), beta1, beta2, eps, grad_scale}.
 *      step_advance:      t += 1 (advance_t != 0), hyper from cfg, state[2] = state[1], state[1] += n_draws (one launch at the
 *                         start of a step; advance_t = 0 for a forward-only pass that draws dropout masks).
 *      adam_dev:          unetrir_adam_f32 with its five scalars read from `hyper`.
 *      dropout_mask_dev:  unetrir_dropout_mask_f32 with draw number *step_base + step_offset (step_base = state + 2). */
int unetrir_step_advance(unsigned long long* state, const float* cfg, float* hyper, int n_draws, int advance_t,
                         unetrir_stream_t stream);
int unetrir_adam_dev_f32(float* theta, const float* g, float* m, float* v, long long n, const float* hyper,
                         unetrir_stream_t stream);
int unetrir_dropout_mask_dev_f32(float* mask, long long n, float p, unsigned long long seed,
                                 const unsigned long long* step_base, unsigned long long step_offset, unetrir_stream_t stream);

/* ---- bf16-storage variants (BASELINE.json configs[1] names bf16): activations, their gradients and the weight work
 *      copies are bfloat16 (unetrir_bf16), accumulation is fp32 (v_mfma_f32_32x32x16_bf16), bias / BatchNorm parameters /
 *      statistics / weight gradients / master weights stay fp32.  Channel counts and pixel strides are multiples of 8
 *      (16-byte rows).  Same call sites as the _f32 entry points above.  The weight gradient has dedicated bf16 kernels
 *      for 3x3 and 1x1 layers; other kernel sizes (kernels = 6, the reference's constructor default) run on the tap-table
 *      weight-gradient kernel with bf16 operand loads (fp32 MFMA arithmetic); the Dense layers stay fp32. */
int unetrir_conv2d_fwd_bf16(const unetrir_conv_geom* g, const unetrir_bf16* x, int ldx, const unetrir_bf16* w,
                            const float* bias, const unetrir_bf16* addend, int ldadd, unetrir_bf16* y, int ldy,
                            unetrir_stream_t stream);
int unetrir_conv2d_dgrad_bf16(const unetrir_conv_geom* g, const unetrir_bf16* dy, int lddy, const unetrir_bf16* wt,
                              const unetrir_bf16* addend, int ldadd, unetrir_bf16* dx, int lddx,
                              unetrir_stream_t stream);
int unetrir_conv2d_wgrad_bf16(const unetrir_conv_geom* g, const unetrir_bf16* x, int ldx, const unetrir_bf16* dy,
                              int lddy, float* dw, float reg_coef, const float* w, void* ws, size_t ws_bytes,
                              unetrir_stream_t stream);
int unetrir_conv2d_transpose_fwd_bf16(const unetrir_conv_geom* g, const unetrir_bf16* x, int ldx,
                                      const unetrir_bf16* wt, const float* bias, unetrir_bf16* y, int ldy,
                                      unetrir_stream_t stream);
int unetrir_conv2d_transpose_dgrad_bf16(const unetrir_conv_geom* g, const unetrir_bf16* dy, int lddy,
                                        const unetrir_bf16* w, const unetrir_bf16* addend, int ldadd,
                                        unetrir_bf16* dx, int lddx, unetrir_stream_t stream);
int unetrir_conv2d_transpose_wgrad_bf16(const unetrir_conv_geom* g, const unetrir_bf16* x, int ldx,
                                        const unetrir_bf16* dy, int lddy, float* dw, float reg_coef, const float* w,
                                        void* ws, size_t ws_bytes, unetrir_stream_t stream);
/* Deferred split-K reduction.  Every weight gradient above ends in a fixed-order reduction of its fp32 partial slabs - a launch of
 * 5-20 us of which most is the launch itself, 23 times per step at configs[1], 57 times at configs[4].  The *_partials_* forms run
 * the same weight-gradient kernel, LEAVE the slabs in `ws` (which the caller must keep untouched until the reduction has run) and
 * fill `desc`; unetrir_splitk_reduce_batched then performs up to 16 such reductions per launch (more: several launches), each
 * output element summed by the same code in the same slab order as the single launch: bit-identical results.  desc->nsplit == 0
 * after the call: the kernel wrote dw directly, nothing is left to do (the batched call skips such descriptors). */
typedef struct unetrir_reduce_desc {
    const float* part; /* [nsplit][n] partial slabs */
    int nsplit;
    size_t n;          /* outputs */
    float* out;        /* [n]: sum over slabs + reg * w */
    float reg;
    const float* w;    /* nullable when reg == 0 */
} unetrir_reduce_desc;
int unetrir_conv2d_wgrad_partials_bf16(const unetrir_conv_geom* g, const unetrir_bf16* x, int ldx, const unetrir_bf16* dy,
                                       int lddy, float* dw, float reg_coef, const float* w, void* ws, size_t ws_bytes,
                                       unetrir_reduce_desc* desc, unetrir_stream_t stream);
int unetrir_conv2d_transpose_wgrad_partials_bf16(const unetrir_conv_geom* g, const unetrir_bf16* x, int ldx,
                                                 const unetrir_bf16* dy, int lddy, float* dw, float reg_coef, const float* w,
                                                 void* ws, size_t ws_bytes, unetrir_reduce_desc* desc, unetrir_stream_t stream);
int unetrir_splitk_reduce_batched(const unetrir_reduce_desc* desc, int n, unetrir_stream_t stream);
/* The stride-2 3x3 forward kernel (conv3x3d.hip: strided Conv2D forward, Conv2DTranspose data gradient) streams its kernel
 * 16 input channels at a time; from the [N][9][C] copy that is a gather of 32-byte pieces.  A PACKED copy holds the same values
 * in the order the kernel's LDS-DMA reads them - [N / 128][C / 16][9 taps][4 blocks of 32 channels][64 lanes][8 values], lane =
 * (row of the block, permuted as the MFMA accumulator layout wants it; 8-channel half) - so every DMA piece is 1 KB of
 * contiguous memory (measured: 22-26 us of 100-160 per launch).  unetrir_cast_weights_batched_bf16 writes it when the
 * descriptor carries a destination; the _packed entry points take it beside the plain copy (NULL: plain copy only).
 * unetrir_conv3x3s2_packed_elems returns the element count of the packed copy, 0 where none is defined (N or C not a
 * multiple of 64). */
size_t unetrir_conv3x3s2_packed_elems(int N, int C);
int unetrir_conv2d_fwd_packed_bf16(const unetrir_conv_geom* g, const unetrir_bf16* x, int ldx, const unetrir_bf16* w,
                                   const unetrir_bf16* w_packed, const float* bias, const unetrir_bf16* addend, int ldadd,
                                   unetrir_bf16* y, int ldy, unetrir_stream_t stream);
int unetrir_conv2d_transpose_dgrad_packed_bf16(const unetrir_conv_geom* g, const unetrir_bf16* dy, int lddy,
                                               const unetrir_bf16* w, const unetrir_bf16* w_packed,
                                               const unetrir_bf16* addend, int ldadd, unetrir_bf16* dx, int lddx,
                                               unetrir_stream_t stream);
/* fp32 master weights [N][T][C] -> bf16 work copies: same orientation with channels zero-padded to Cp; transposed
 * [C][T][Np] with the row dimension zero-padded to Np. */
int unetrir_cast_weight_bf16(const float* w, unetrir_bf16* o, int N, int T, int C, int Cp, unetrir_stream_t stream);
int unetrir_transpose_cast_weight_bf16(const float* w, unetrir_bf16* wt, int N, int T, int C, int Np,
                                       unetrir_stream_t stream);
/* every layer's work copies in two launches: `desc` is an array of n_layers descriptors IN DEVICE MEMORY (same
 * semantics per layer as the two calls above; a NULL destination skips that copy). */
typedef struct unetrir_cast_desc {
    const float* w;            /* fp32 master [N][T][C] */
    unetrir_bf16* same;        /* bf16 [N][T][Cp] or NULL */
    unetrir_bf16* transposed;  /* bf16 [C][T][Np] or NULL */
    int N, T, C, Cp, Np, reserved;
    unetrir_bf16* packed_s2;   /* 3x3 kernels served by the stride-2 forward kernel: a third copy in the order its LDS-DMA reads it
                                  (unetrir_conv3x3s2_packed_elems(N, C) elements, zero-initialised by the caller), or NULL */
} unetrir_cast_desc;
int unetrir_cast_weights_batched_bf16(const unetrir_cast_desc* desc, int n_layers, unetrir_stream_t stream);
int unetrir_bn_stats_bf16(const unetrir_bf16* x, int ldx, long long P, int C, const float* gamma, const float* beta,
                          float eps, float momentum, float* moving_mean, float* moving_var, float* affine,
                          float* saved, void* ws, size_t ws_bytes, unetrir_stream_t stream);
int unetrir_bn_apply_bf16(const unetrir_bf16* x, int ldx, long long P, int C, const float* affine, int relu,
                          unetrir_bf16* y, int ldy, unetrir_stream_t stream);
int unetrir_bn_bwd_bf16(const unetrir_bf16* da, int ldda, const unetrir_bf16* x, int ldx, long long P, int C,
                        const float* affine, const float* saved, int relu, unetrir_bf16* dx, int lddx, float* dgamma,
                        float* dbeta, void* ws, size_t ws_bytes, unetrir_stream_t stream);
int unetrir_colsum_bf16(const unetrir_bf16* x, int ldx, long long P, int C, float* out, void* ws, size_t ws_bytes,
                        unetrir_stream_t stream);
int unetrir_relu_bwd_bf16(const unetrir_bf16* da, int ldda, const unetrir_bf16* x, int ldx, long long P, int C,
                          unetrir_bf16* dx, int lddx, unetrir_stream_t stream);
int unetrir_nchw_to_nhwc_pad_bf16(const float* x, int B, int C, int H, int W, unetrir_bf16* y, int Cpad,
                                  unetrir_stream_t stream);
/* head: bf16 activations in, fp32 logits [P][ldy] out; weight gradient from bf16 dlogits [P][lddy] */
int unetrir_head6x6_fwd_bf16(const unetrir_bf16* x, int ldx, int B, int H, int W, int C, const float* w,
                             const float* bias, float* y, int ldy, unetrir_stream_t stream);
int unetrir_head6x6_wgrad_bf16(const unetrir_bf16* x, int ldx, int B, int H, int W, int C, const unetrir_bf16* dy,
                               int lddy, float* dw, void* ws, size_t ws_bytes, unetrir_stream_t stream);
/* ---- fused column statistics (bf16).  The 3x3 stride-1 kernels that serve most layers can emit, per 16 x 32 pixel tile,
 *      the per-channel (sum, sum of squares) of the bf16 output they store: colstat is [rows][N][2] floats with N the
 *      convolution's output channels (forward: Cout, data gradient: Cin).  BatchNormalization statistics (dl_models/
 *      u_net.py:204-206) and bias gradients then come from these rows instead of re-reading the tensor.
 *      unetrir_conv2d_colstat_rows_bf16 returns 0 when the kernel serving this layer (geometry, dgrad flag, pixel stride
 *      ld_in of its input) has no fused statistics; the *_colstat entry points then return UNETRIR_EINVAL. */
long long unetrir_conv2d_colstat_rows_bf16(const unetrir_conv_geom* g, int dgrad, int ld_in);
/* which hand-written kernel serves a 3x3 stride-1 layer (forward or data gradient) under the switches in effect: for tests and
 * measurement scripts that must know what they are looking at */
enum { UNETRIR_K3_TAPTABLE = 0, UNETRIR_K3_CONV3X3R = 1, UNETRIR_K3_CONV3X3G = 2, UNETRIR_K3_CONV3X3G_PAIR = 3, UNETRIR_K3_CONV3X3H = 4,
       UNETRIR_K3_CONV3X3S = 5, UNETRIR_K3_CONV3X3P = 6, UNETRIR_K3_STEM = 7 };
int unetrir_conv3x3_kernel_id_bf16(const unetrir_conv_geom* g, int dgrad, int ld_in);
int unetrir_conv2d_fwd_colstat_bf16(const unetrir_conv_geom* g, const unetrir_bf16* x, int ldx, const unetrir_bf16* w,
                                    const float* bias, const unetrir_bf16* addend, int ldadd, unetrir_bf16* y, int ldy,
                                    float* colstat, unetrir_stream_t stream);
int unetrir_conv2d_dgrad_colstat_bf16(const unetrir_conv_geom* g, const unetrir_bf16* dy, int lddy, const unetrir_bf16* wt,
                                      const unetrir_bf16* addend, int ldadd, unetrir_bf16* dx, int lddx, float* colstat,
                                      unetrir_stream_t stream);
/* The same for every other convolution of the graphs (1x1 layers, strided layers, the tap-table path of small images): the
 * implicit-GEMM kernel emits one row per pixel tile (128 or 64 pixels) of its launch.  unetrir_conv2d_colstat_rows_bf16 answers for these
 * layers too (0 only where the serving kernel has none: 3x3 stride-2 forward, strided data gradients).  Conv2DTranspose
 * forward (dl_models/res_ae.py:310-371: every decoder convolution of the residual graph sits in front of a BatchNormalization):
 * stride 1 = one row per 128-pixel tile; stride 2 (1x1 'valid', 6x6) = the four output-parity classes, each with its own row
 * range (4 x tiles rows); 3x3 stride 2: none (0). */
long long unetrir_conv2d_transpose_colstat_rows_bf16(const unetrir_conv_geom* g, int ld_in);
int unetrir_conv2d_transpose_fwd_colstat_bf16(const unetrir_conv_geom* g, const unetrir_bf16* x, int ldx, const unetrir_bf16* wt,
                                              const float* bias, unetrir_bf16* y, int ldy, float* colstat,
                                              unetrir_stream_t stream);
/* as unetrir_bn_stats_f32 / unetrir_colsum_f32 but from colstat rows; fixed-order fp64 reduction over the rows.
 * colsum: out[0..C) = sum over rows of colstat[row][c0 + c][0], ldc = channels per row. */
int unetrir_bn_stats_colstat(const float* colstat, long long rows, long long P, int C, const float* gamma, const float* beta,
                             float eps, float momentum, float* moving_mean, float* moving_var, float* affine, float* saved,
                             unetrir_stream_t stream);
int unetrir_colsum_colstat(const float* colstat, long long rows, int ldc, int c0, int C, float* out, unetrir_stream_t stream);
/* unetrir_bn_stats_colstat followed by unetrir_bn_act_add_* in one call: statistics rows -> affine / saved / moving statistics, then
 * y = act(x * scale + shift (+ addend)): the finalize and the apply launch. */
int unetrir_bn_colstat_act_add_f32(const float* colstat, long long rows, const float* x, int ldx, long long P, int C, const float* gamma,
                                   const float* beta, float eps, float momentum, float* moving_mean, float* moving_var, float* affine,
                                   float* saved, int act, const float* addend, int ldadd, float* y, int ldy, unetrir_stream_t stream);
int unetrir_bn_colstat_act_add_bf16(const float* colstat, long long rows, const unetrir_bf16* x, int ldx, long long P, int C,
                                    const float* gamma, const float* beta, float eps, float momentum, float* moving_mean,
                                    float* moving_var, float* affine, float* saved, int act, const unetrir_bf16* addend, int ldadd,
                                    unetrir_bf16* y, int ldy, unetrir_stream_t stream);
/* head data gradient on the matrix cores: dx[p][c] = sum_{n<2,kh,kw} dy[p - off][n] * w[n][kh][kw][c] (the adjoint of
 * unetrir_head6x6_fwd_bf16; w is the fp32 [>=2][6][6][C] kernel, rounded to bf16 on load).  Only channels 0,1 of dy are
 * read.  Supported for C a multiple of 32 up to 512 (unetrir_head6x6_dgrad_supported; 64 channels per workgroup, images wider
 * than 256 pixels in column blocks); otherwise UNETRIR_EINVAL and the caller uses unetrir_conv2d_dgrad_bf16 on the padded kernel. */
int unetrir_head6x6_dgrad_supported(int W, int C);
int unetrir_head6x6_dgrad_bf16(const unetrir_bf16* dy, int lddy, int B, int H, int W, const float* w, int C,
                               unetrir_bf16* dx, int lddx, unetrir_stream_t stream);
/* as unetrir_sigmoid_loss_f32 / unetrir_sigmoid_bwd_f32 but dlogits is bf16 [B*H*W][8] (channels 2..7 zero) */
int unetrir_sigmoid_loss_bf16(const float* logits, int ldl, const float* target, int B, int H, int W, float alpha,
                              float inv_norm, float* pred, unetrir_bf16* dlogits, float* loss_out, void* ws,
                              size_t ws_bytes, unetrir_stream_t stream);
int unetrir_sigmoid_loss_ex_bf16(const float* logits, int ldl, const float* target, const float* phase_ref,
                                 const float* phase_weight, int B, int H, int W, float alpha, float inv_norm, float* pred,
                                 unetrir_bf16* dlogits, float* loss_out, void* ws, size_t ws_bytes, unetrir_stream_t stream);
int unetrir_sigmoid_bwd_bf16(const float* pred, const float* dpred, int B, int H, int W, unetrir_bf16* dlogits,
                             unetrir_stream_t stream);
/* glue between the bf16 trunk and the fp32 information-vector branch: y = a + b (Add(), dl_models/u_net.py:229);
 * fp32 copy of a bf16 gradient.  n is a multiple of 4. */
int unetrir_add_f32_to_bf16(const unetrir_bf16* a, const float* b, unetrir_bf16* y, long long n,
                            unetrir_stream_t stream);
int unetrir_cast_bf16_to_f32(const unetrir_bf16* a, float* y, long long n, unetrir_stream_t stream);
int unetrir_cast_f32_to_bf16(const float* a, unetrir_bf16* y, long long n, unetrir_stream_t stream);
/* residual-block glue in bf16 storage (as unetrir_bn_act_add_f32 / unetrir_act_bwd_f32; C % 8 == 0): the graphs of
 * dl_models/res_ae.py and the feature-block modes 1-3 of dl_models/u_net.py on the bf16 convolution kernels */
int unetrir_bn_act_add_bf16(const unetrir_bf16* x, int ldx, long long P, int C, const float* affine, int act,
                            const unetrir_bf16* addend, int ldadd, unetrir_bf16* y, int ldy, unetrir_stream_t stream);
int unetrir_act_bwd_bf16(const unetrir_bf16* da, int ldda, const unetrir_bf16* out, int ldo, long long P, int C, int act,
                         unetrir_bf16* g, int ldg, unetrir_stream_t stream);
int unetrir_bn_bwd_junction_bf16(const unetrir_bf16* da, int ldda, const unetrir_bf16* x, int ldx, const unetrir_bf16* out, int ldo,
                                 long long P, int C, const float* affine, const float* saved, int act, unetrir_bf16* dx, int lddx,
                                 unetrir_bf16* gskip, int ldgs, const unetrir_bf16* gskip_add, int ldga, float* dgamma, float* dbeta,
                                 void* ws, size_t ws_bytes, unetrir_stream_t stream);

/* ---- waveform <-> feature transforms at the two ends of the data path; all fp32 in HBM,
 *      fp64 direct DFT inside.  n_fft is a power of two <= 1024, win_length <= n_fft, window = periodic Hann centred in
 *      n_fft (librosa's default), centred frames (n_fft/2 samples of padding each side), frames = 1 + T / hop_length.
 *
 * unetrir_stft_features_f32 replaces Loader.load's mean removal (preprocess.py:56, remove_mean), FeatureExtractor.extract
 * (preprocess.py:13-18: librosa.stft -> abs, angle), Normalizer.normalize (preprocess.py:26-32, normalize) and
 * TensorPadder.pad_amp_phase (preprocess.py:65-70): wav [B][T] -> out [B][2][H][W] (plane 0 amplitude, plane 1 phase; row =
 * frequency bin, column = frame; rows >= n_fft/2+1 and columns >= frames are written as zeros).  pad_mode 0 = 'reflect'
 * (librosa < 0.10, the reference's era), 1 = 'constant'.  UNETRIR_EINVAL when H < n_fft/2+1, W < frames or T <= n_fft/2.
 *
 * unetrir_istft_features_f32 replaces PostProcess.post_process's arithmetic (postprocess.py:68-71, 'ph' branch :127-134):
 * TensorPadder.un_pad to n_bins x n_frames (preprocess.py:107-113), Normalizer.denormalize (preprocess.py:34-41,
 * denormalize) and librosa.istft: feat [B][2][H][W] -> wav [B][hop_length * (n_frames - 1)].  n_bins must be n_fft/2+1. */
int unetrir_stft_frames(int T, int hop_length);
int unetrir_stft_features_f32(const float* wav, int B, int T, int n_fft, int win_length, int hop_length, int pad_mode,
                              int remove_mean, int normalize, float* out, int H, int W, unetrir_stream_t stream);
int unetrir_istft_features_f32(const float* feat, int B, int H, int W, int n_bins, int n_frames, int n_fft, int win_length,
                               int hop_length, int denormalize, float* wav, unetrir_stream_t stream);

/* ---- glue that would otherwise be framework kernels inside the step.
 * bn_inference_affine: BatchNormalization with training=False (rir_generation.py:165): scale = gamma * rsqrt(moving_var + eps),
 *   shift = beta - moving_mean * scale into affine[2*C] (gamma / beta NULL = 1 / 0), consumed by unetrir_bn_apply_*.
 * dropout_mask: the keep mask of Dropout(p) (dl_models/u_net.py:260), already scaled by 1/(1-p): element i of draw
 *   (seed, step) is a fixed function of (seed, step, i) (counter-based generator), so a step is reproducible.
 * index_to_i32: the int32 / int64 information-vector indices as the int32 array the embedding kernels read. */
int unetrir_bn_inference_affine_f32(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                                    float eps, int C, float* affine, unetrir_stream_t stream);
int unetrir_dropout_mask_f32(float* mask, long long n, float p, unsigned long long seed, unsigned long long step,
                             unetrir_stream_t stream);
int unetrir_index_to_i32(const void* idx, int elem_bytes, long long n, int* out, unetrir_stream_t stream);

/* ---- input staging for a host-fed train step (DataGenerator.__getitem__, datageneratorv2.py:64-102, hands over host arrays):
 *      for k < n: memcpy src[k] -> pinned[k] (page-locked staging owned by the caller), then an asynchronous host -> device copy
 *      pinned[k] -> dev[k] on `stream`.  One call per batch from the producer thread: the foreign call runs without the
 *      interpreter lock, so staging a batch does not hold the lock the thread that launches the step takes ~300 times per step. */
int unetrir_stage_h2d(int n, const void* const* src, void* const* pinned, void* const* dev, const size_t* bytes,
                      unetrir_stream_t stream);

/* ---- profiling hooks used by bench.py: when enabled every conv launch is bracketed by HIP
 *      events on its own stream; collect() synchronises those events and returns, per kernel
 *      family, launch count, total milliseconds and total algorithmic FLOPs.  Process-global
 *      (one mutex-protected record list); off by default; the second piece of global state. */
#define UNETRIR_PROF_FAMILIES 8
enum { UNETRIR_FAM_CONV_FWD = 0, UNETRIR_FAM_CONV_DGRAD = 1, UNETRIR_FAM_CONV_WGRAD = 2,
       UNETRIR_FAM_BN = 3, UNETRIR_FAM_OTHER = 4, UNETRIR_FAM_SMALL_CHANNEL = 5,
       UNETRIR_FAM_DOMINANT = 6 /* launches served by the dominant kernel of the bf16 step (conv3x3p), counted here IN ADDITION to their
                                   forward / data-gradient family */ };
int unetrir_prof_enable(int on);      /* 0: off, 1: every family, 2: forward convolutions only (fewer events in the stream) */
int unetrir_prof_collect(int* counts, double* ms, double* flops);

#ifdef __cplusplus
}
#endif
#endif /* UNETRIR_H */
