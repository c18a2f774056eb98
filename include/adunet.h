/*
 * adunet.h -- C ABI of the MI355X (gfx950) adaptive-depth U-Net hot path.
 *
 * This is the drop-in boundary.  The reference
 * (KunalNN/Adaptive-Depth-U-Net-for-Image-Super-Resolution-Segmentation) has no
 * FFI of its own: its seam is the Keras Layer/Model call surface, below which sit
 * TensorFlow ops (cuDNN / Eigen kernels).  Each entry point below replaces one of
 * those TensorFlow ops at the reference call site cited next to it.
 *
 * Conventions
 *  - All tensor pointers are DEVICE pointers owned by the caller, NHWC, dense.
 *  - `dtype` selects the activation element type: AD_F32 (parity path) or
 *    AD_BF16 (throughput path; fp32 accumulation, fp32 statistics) or AD_F16 (the same kernels on IEEE
 *    half storage: the reference's mixed_float16 policy).
 *    Parameters, gradients, statistics and reduction outputs are always fp32.
 *  - `stream` is a hipStream_t passed as void*; every call is asynchronous on it.
 *  - No allocation, no synchronisation: scratch memory is passed in as (`ws`, `ws_bytes`);
 *    ad_*_ws_bytes() says how much a call needs.  Process-wide state is limited to what is
 *    declared here: the last-error string, the device's CU count (read once, ad_device_cus)
 *    and the options set through ad_set_option.  The library never reads the environment.
 *  - Return value: 0 = ok, negative = error (AD_ERR_*); ad_last_error() returns a
 *    thread-local message.  No exceptions cross the boundary.
 *  - One caller thread per device (the reference drives the model from a single
 *    Python thread: SURVEY 8b).
 */
#ifndef ADUNET_H
#define ADUNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AD_F32 0
#define AD_BF16 1
#define AD_F16 2             /* IEEE half storage, fp32 accumulation: the reference's mixed_float16 policy
                                (Super_resolution/code/train_adaptive_unet.py:471-477); needs loss scaling (ad_loss_scale_*) */

#define AD_OK 0
#define AD_ERR_ARG (-1)      /* bad shape / unsupported configuration */
#define AD_ERR_WS (-2)       /* workspace too small */
#define AD_ERR_LAUNCH (-3)   /* HIP launch error */

#define AD_EPI_NONE 0
#define AD_EPI_RELU 1

int ad_version(void);
const char* ad_last_error(void);

/* Compute units of the current device (hipDeviceAttributeMultiprocessorCount, queried at the first call and rounded down
 * to a multiple of 64; 256 on MI355X): the persistent kernels launch one workgroup per CU and deal tiles XCD by XCD. */
int ad_device_cus(void);

/* A/B switches for measurements, all 0 by default; nothing else changes the kernel a call selects:
 *   "no_map1"      1x1 feature maps take the generic kernel instead of conv3x3_map1_kernel
 *   "no_map4"      4x4 feature maps take the generic kernel instead of conv3x3_map4_kernel
 *   "no_dgrad_ln"  ad_conv3x3_dgrad_ln_bwd_is_fused() answers 0 (dgrad and LayerNorm backward as two launches)
 *   "no_mosaic"    the wave-specialised conv kernels tile every image by itself even where the image mosaic
 *                  (one virtual map of all images, a single zero line between neighbours) needs fewer 16 x 16 tiles
 *   "no_pw_wide"   ad_pw_gemm as until the first half of r05: 192- / 128-channel tiles only (no 256-channel tiles on
 *                  pw_gemm_pp_kernel) and every row group of the XCD-aware tile order one XCD's
 * ad_get_option returns the value, -1 for an unknown name. */
int ad_set_option(const char* name, int value);
int ad_get_option(const char* name);

/* Channel granularity of the conv kernels for `dtype`: Cin of every conv input
 * tensor must be a multiple of this (32 for bf16, 16 for f32); the 3-channel network
 * input is zero-padded to it by ad_pad_channels(). */
int ad_cin_granule(int dtype);

/* ---------------------------------------------------------------- packing -- */

/* x[npix, c] fp32 -> y[npix, cpad] (dtype), zero channel padding.
 * Feeds the first Conv2D (train_adaptive_unet.py:202 on `low_res_input`). */
int ad_pad_channels(const float* x, void* y, int64_t npix, int c, int cpad, int dtype, void* stream);

/* Keras HWIO fp32 kernel [3,3,cin,cout] -> MFMA operand layouts (dtype):
 *   w_fwd  : [9][cin_pad/KV][pad64(cout)][KV]            (KV = 16 B / sizeof(dtype))
 *   w_dgrad: [9 (taps rotated 180)][cout/KV][pad64(cin_pad)][KV]  (may be NULL)
 * The output-channel dimension of each pack is zero-padded to whole 64-channel blocks (the kernels' block size), so
 * layers with 16 / 32 / 96 ... output channels work (Segmenation/code/unet_vinillia.py:72 defaults to base_channels=32);
 * ad_conv3x3_pack_elems() gives the element count of either pack.  cout must be a multiple of the dtype's channel
 * granule for w_dgrad (it becomes the contraction axis). */
size_t ad_conv3x3_pack_elems(int cin_pad, int cout, int dgrad);
int ad_conv3x3_pack(const float* w_hwio, int cin, int cout, int cin_pad,
                    void* w_fwd, void* w_dgrad, int dtype, void* stream);

/* The same for up to 64 layers of a model in ONE launch.  jobs_dev: device array of njobs records
 *   struct { const float* w_hwio; void* w_fwd; void* w_dgrad (or NULL); int cin, cout, cin_pad, first_block; }
 * (ad_conv3x3_pack_job_bytes() = 40).  A job owns ad_conv3x3_pack_job_blocks(cin_pad, cout) consecutive blocks (one per
 * 64 x 64 channel tile of each tap); first_block is the running sum, nblocks the total.  All layers share dtype.  The table is caller-owned and read at run time, so a captured hipGraph keeps using it. */
size_t ad_conv3x3_pack_job_bytes(void);
int ad_conv3x3_pack_job_blocks(int cin_pad, int cout);
int ad_conv3x3_pack_batch(const void* jobs_dev, int njobs, int nblocks, int dtype, void* stream);

/* ------------------------------------------------------------ convolution -- */

/* Images per mosaic row when a bias / bias + ReLU forward or dgrad launch (wgrad = 0; outputs split on a 64-channel block)
 * or a weight-gradient launch (wgrad = 1) of this shape walks the image mosaic -- all n images as one virtual map with a
 * single zero line between neighbours, for maps whose extent is not a multiple of the 16 x 16 tile -- and 0 when it tiles
 * image by image.  Only addresses differ: forward results are bitwise those of the per-image tiling (option "no_mosaic").
 * A batch whose tensors reach 2 GiB is launched as runs of images, each planned on its own: the answer then describes the
 * full-size runs (the last, shorter run may plan differently). */
int ad_conv3x3_mosaic(int n, int h, int w, int c1, int c2, int cout, int dtype, int wgrad);

/* Conv2D 3x3, stride 1, padding "same" (+bias, +optional ReLU):
 *   L.Conv2D(nf, 3, padding="same")            train_adaptive_unet.py:202,207
 *   L.Conv2D(nf, 3, ..., activation="relu")    train_adaptive_unet.py:259
 * Input is the *virtual concatenation* of x1[n,h,w,c1] and x2[n,h,w,c2] along
 * channels (L.Concatenate, :261); pass x2 = NULL, c2 = 0 for a single input.
 * Output channels [0,cy1) go to y1[n,h,w,cy1], the rest to y2[n,h,w,cout-cy1]
 * (used when the same kernel runs as dgrad of a concatenated input); pass
 * y2 = NULL, cy1 = cout normally.  cout % 16 == 0 (a ragged last 64-channel block computes the padding and does not
 * store it), c1 % granule == 0, c2 % granule == 0, cy1 % 16 == 0.
 * Launches with very few spatial tiles (the 4x4 / 1x1 bottleneck maps) split the channel contraction over
 * several workgroups and sum fp32 partial slabs in a fixed order; that path needs ad_conv3x3_fwd_ws_bytes()
 * of workspace (0 for all other shapes; with ws = NULL the unsplit kernel is used). */
size_t ad_conv3x3_fwd_ws_bytes(int n, int h, int w, int cin, int cout, int dtype);
int ad_conv3x3_fwd(const void* x1, int c1, const void* x2, int c2,
                   const void* w_packed, const float* bias,
                   void* y1, int cy1, void* y2,
                   int n, int h, int w, int cout, int epilogue,
                   void* ws, size_t ws_bytes, int dtype, void* stream);

/* Weight gradient of the same convolution (TF Conv2DBackpropFilter):
 *   dw_hwio[3,3,c1+c2,cout] (fp32, Keras layout) = sum_pixels x(+tap) * dz
 * Only the first `cin_real` input channels are written (3 for the padded first layer).
 * Deterministic: per-workgroup partial slabs in `ws`, summed in a fixed order. */
size_t ad_conv3x3_wgrad_ws_bytes(int n, int h, int w, int cin, int cout, int dtype);
int ad_conv3x3_wgrad(const void* x1, int c1, const void* x2, int c2, const void* dz,
                     float* dw_hwio, int cin_real,
                     int n, int h, int w, int cout,
                     void* ws, size_t ws_bytes, int dtype, void* stream);

/* conv_block's Conv2D -> LayerNormalization -> ReLU (train_adaptive_unet.py:202-204, 207-209) in one call:
 * z = conv(x) + bias (saved for the backward pass), act = relu(gamma * (z - mean) * rstd + beta), mean / rstd [npix]
 * as ad_layernorm_relu_fwd writes them.  For cout == 64 on bf16 launches large enough for the wave-specialised
 * kernels the LayerNorm runs in the convolution's epilogue, on the fp32 accumulators (one launch, z is never
 * re-read); every other shape runs ad_conv3x3_fwd followed by ad_layernorm_relu_fwd.  Workspace as ad_conv3x3_fwd. */
int ad_conv3x3_ln_relu_is_fused(int n, int h, int w, int c1, int c2, int cout, int dtype);   /* 1: one launch */
/* act == NULL: z, mean and rstd only (the caller re-derives the activation where it is consumed: ad_head_ln_bwd with
 * xh == NULL).  Exists for the weights-resident kernel (c1 + c2 = 64 -> cout = 64); ad_conv3x3_ln_stats_is_fused tells. */
int ad_conv3x3_ln_stats_is_fused(int n, int h, int w, int c1, int c2, int cout, int dtype);
/* z == NULL (r05): the activation only -- inference (model(x), evaluate_model.py:94-137) reads neither the conv output nor the
 * statistics, so the fused launch writes one tensor instead of two and two vectors; mean / rstd may be NULL.  The same arithmetic
 * as the full form: act is bitwise what that writes.  Exists wherever ad_conv3x3_ln_relu_is_fused says 1. */
int ad_conv3x3_ln_relu_fwd(const void* x1, int c1, const void* x2, int c2,
                           const void* w_packed, const float* bias,
                           const float* gamma, const float* beta, float eps,
                           void* z, void* act, float* mean, float* rstd,
                           int n, int h, int w, int cout,
                           void* ws, size_t ws_bytes, int dtype, void* stream);

/* The network's first conv_block step -- L.Conv2D(base_channels, 3, padding="same") on the 3-channel input followed
 * by LayerNormalization and ReLU (train_adaptive_unet.py:202-204 with inputs of :225) -- and its weight gradient,
 * without the zero-padded copy of the input: x is the raw [n, h, w, 3] fp32 batch, w_hwio the fp32 master kernel
 * [3, 3, 3, 64]; K = 27 is padded to one 32-deep MFMA step inside the kernels.  bf16 outputs, cout = 64 only
 * (ad_conv3x3_c3_supported tells); other first layers use ad_pad_channels + the general entry points.
 * z == NULL: the activation only (inference, as ad_conv3x3_ln_relu_fwd); mean / rstd may then be NULL. */
int ad_conv3x3_c3_supported(int n, int h, int w, int cout, int dtype);
int ad_conv3x3_c3_ln_relu_fwd(const float* x, const float* w_hwio, const float* bias,
                              const float* gamma, const float* beta, float eps,
                              void* z, void* act, float* mean, float* rstd,
                              int n, int h, int w, int dtype, void* stream);
/* The same first layer without the normalisation: z = conv(x) + bias (the first Conv2D of the BatchNorm segmentation model,
 * Segmenation/code/train_adaptive_unet.py:326-327, whose BatchNormalization needs the statistics of the whole batch first). */
int ad_conv3x3_c3_fwd(const float* x, const float* w_hwio, const float* bias, void* z,
                      int n, int h, int w, int dtype, void* stream);
size_t ad_conv3x3_c3_wgrad_ws_bytes(int n, int h, int w);
int ad_conv3x3_c3_wgrad(const float* x, const void* dz, float* dw_hwio,
                        int n, int h, int w, void* ws, size_t ws_bytes, int dtype, void* stream);

/* ---------------------------------------------------- LayerNorm (+ ReLU) -- */

/* L.LayerNormalization(axis=-1) (eps 1e-3) followed by L.Activation("relu"):
 * train_adaptive_unet.py:203-204,208-209.  Statistics in fp32; mean/rstd [npix] are
 * saved for the backward pass.  relu = 0 gives plain LayerNorm. */
int ad_layernorm_relu_fwd(const void* z, const float* gamma, const float* beta,
                          void* y, float* mean, float* rstd,
                          int64_t npix, int c, float eps, int relu, int dtype, void* stream);

/* Backward of the pair above.  dy = gradient w.r.t. the ReLU output.
 * Produces dz (same dtype) and overwrites fp32 dgamma[c], dbeta[c] and dbias[c]
 * (= column sums of dz: the gradient of the preceding conv's bias); deterministic. */
size_t ad_layernorm_bwd_ws_bytes(int64_t npix, int c);
int ad_layernorm_relu_bwd(const void* dy, const void* z, const float* mean, const float* rstd,
                          const float* gamma, const float* beta,
                          void* dz, float* dgamma, float* dbeta, float* dbias,
                          int64_t npix, int c, int relu,
                          void* ws, size_t ws_bytes, int dtype, void* stream);

/* ReLU backward for the conv+ReLU up-convolution (train_adaptive_unet.py:259):
 * dz = dy * (y > 0); dbias[c] = column sums of dz. */
int ad_relu_bwd(const void* dy, const void* y, void* dz, float* dbias,
                int64_t npix, int c, void* ws, size_t ws_bytes, int dtype, void* stream);

/* ----------------------------------------------------------------- resize -- */

/* tf.image.resize(..., "bilinear", antialias=True) and its gradient
 * (shared/custom_layers.py:102 ResizeByScale, :124 ResizeToMatch), as a separable
 * banded linear map with host-precomputed per-axis tap tables:
 *   y[n,oy,ox,c] (+)= sum_{a<ky, b<kx} wy[oy*ky+a] * wx[ox*kx+b] * x[n, sy[oy]+a, sx[ox]+b, c]
 * Taps beyond the band carry weight 0 and a clamped index, so the same entry point
 * computes the forward map and (with transposed tables) the backward map.
 * Arithmetic is fp32 regardless of dtype, as in the reference. */
int ad_resample(const void* x, void* y,
                const int* sy, const float* wy, int ky,
                const int* sx, const float* wx, int kx,
                int n, int h, int w, int oh, int ow, int c,
                int accumulate, int dtype, void* stream);

/* -------------------------------------------------------------- head+loss -- */

/* residual_rgb 1x1 Conv2D (train_adaptive_unet.py:267-274) + ClippedResidualAdd
 * (shared/custom_layers.py:136-139) [+ Charbonnier / L1 loss and squared error
 * (:308-334) when `target` != NULL], fused in one pass over xh[npix, ch]:
 *   out = clip(inp + xh @ w[ch,3] + b, 0, 1)           (fp32, [npix,3])
 *   stats[0] = sum sqrt((t-out)^2 + eps^2)   (loss_kind 0, Charbonnier)
 *            | sum |t-out|                   (loss_kind 1, L1)
 *   stats[1] = mean over the n images of tf.image.psnr(target, out, max_val=1)
 *            = mean(-10 log10(sqerr[img] / (pix_per_img*3)))   (psnr_metric, :308-311;
 *              +inf at MSE 0 as in TensorFlow)
 *   stats[2] = stats[0] / (n * pix_per_img * 3)          (the Keras loss value: mean over all elements)
 *   sqerr[img] = sum over the image of (t-out)^2   (PSNR / MSE numerator)
 * stats (3 floats) / sqerr are overwritten (deterministic two-stage reduction through ws). */
size_t ad_head_ws_bytes(int n, int ch);
int ad_head_fwd(const void* xh, const float* w, const float* b, const float* inp,
                const float* target, float* out, float* stats, float* sqerr,
                int n, int64_t pix_per_img, int ch, int loss_kind, float eps,
                void* ws, size_t ws_bytes, int dtype, void* stream);

/* Backward of the fused head: with g = dLoss/dout * [0 <= inp+r <= 1] * grad_scale
 *   dxh[npix,ch] = g @ w^T ; dw[ch,3] = xh^T @ g ; db[3] = sum g          */
int ad_head_bwd(const void* xh, const float* w, const float* b, const float* inp,
                const float* target, void* dxh, float* dw, float* db,
                int n, int64_t pix_per_img, int ch, int loss_kind, float eps, float grad_scale,
                const float* loss_scale /* NULL, or device float: g is multiplied by loss_scale[0] (ad_loss_scale_*) */,
                void* ws, size_t ws_bytes, int dtype, void* stream);

/* ad_head_bwd and the LayerNormalization + ReLU backward of the conv_block layer that feeds the head
 * (train_adaptive_unet.py:265-276: head_conv's second Conv2D -> LayerNormalization -> ReLU -> residual_rgb) in one
 * pass over the pixels: the gradient of the head activations stays in fp32 registers instead of being stored and read
 * back.  Inputs as ad_head_bwd plus that layer's saved conv output z, per-pixel mean / rstd and gamma / beta;
 *   dz[npix,ch] = LayerNorm/ReLU backward of (g @ w^T)        (what ad_layernorm_relu_bwd would return)
 *   dw, db: as ad_head_bwd;  dgamma, dbeta: the LayerNorm's;  dbias_conv[ch] = column sums of dz as stored.
 * stats / sqerr (both may be NULL): what ad_head_fwd reports for the same operands -- stats[3] = (loss sum, mean
 * tf.image.psnr, loss mean), sqerr[n] per-image squared error.  The pass re-derives the head's output for the gradient
 * anyway, so a TRAIN step (model.fit, :622-632: only loss and metric leave the step) needs no forward launch over the head.
 * xh == NULL: the head's input relu(gamma * (z - mean) * rstd + beta) is re-derived from z (rounded to the storage type, as
 * a stored activation would be) -- the layer's forward pass then need not write its activation at all
 * (ad_conv3x3_ln_relu_fwd with act == NULL). */
size_t ad_head_ln_bwd_ws_bytes(int n, int ch);
int ad_head_ln_bwd(const void* xh, const float* w, const float* b, const float* inp, const float* target,
                   const void* z, const float* mean, const float* rstd, const float* gamma, const float* beta,
                   void* dz, float* dw, float* db, float* dgamma, float* dbeta, float* dbias_conv,
                   int n, int64_t pix_per_img, int ch, int loss_kind, float eps, float grad_scale,
                   const float* loss_scale, float* stats, float* sqerr, void* ws, size_t ws_bytes, int dtype, void* stream);

/* -------------------------------------------------------------- optimizer -- */

/* Keras-form Adam (tf.keras.optimizers.Adam, train_adaptive_unet.py:489-494):
 *   m += (g-m)(1-b1); v += (g*g-v)(1-b2); p -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps)
 * over flat fp32 buffers; `gscale` multiplies g first (1/world_size after all-reduce). */
int ad_adam_step(float* p, const float* g, float* m, float* v, int64_t count,
                 float lr, float b1, float b2, float eps, int step, float gscale, void* stream);

/* ------------------------------------------------- tier 2: segmentation ops -- */

/* L.BatchNormalization() (eps 1e-3, momentum 0.99) + L.Activation("relu"):
 * Segmenation/code/train_adaptive_unet.py:325-332.  Training mode: batch statistics over (N,H,W) (biased variance; ONE pass
 * over z: sums of the deviations from the first pixel's row and of their squares, so the variance has no E[x^2] - mean^2
 * cancellation), saved mean/rstd/var [c] for the backward pass, and the Keras moving-average update
 * moving = moving*momentum + batch*(1-momentum) (moving_* may be NULL).  Inference mode uses the moving stats.
 * ad_batchnorm_relu_pool_fwd_train additionally writes pooled[n, h/2, w/2, c] = MaxPooling2D(2) of y in the pass that writes
 * y (the encoder's conv_block -> MaxPooling2D, :349-351); h, w even, c at most 256 16-byte vectors.
 * ad_batchnorm_relu_bwd_dbias additionally returns dbias[c] = column sums of dz AS STORED (the BiasAddGrad of the convolution
 * in front: what ad_colsum(dz) returns) from the pass that writes dz; dbias == NULL is ad_batchnorm_relu_bwd. */
size_t ad_batchnorm_ws_bytes(int c);
int ad_batchnorm_relu_fwd_train(const void* z, const float* gamma, const float* beta, void* y,
                                float* save_mean, float* save_rstd, float* save_var,
                                float* moving_mean, float* moving_var, float momentum,
                                int64_t npix, int c, float eps, int relu,
                                void* ws, size_t ws_bytes, int dtype, void* stream);
int ad_batchnorm_relu_pool_fwd_train(const void* z, const float* gamma, const float* beta, void* y, void* pooled,
                                     float* save_mean, float* save_rstd, float* save_var,
                                     float* moving_mean, float* moving_var, float momentum,
                                     int n, int h, int w, int c, float eps, int relu,
                                     void* ws, size_t ws_bytes, int dtype, void* stream);
int ad_batchnorm_relu_fwd_infer(const void* z, const float* gamma, const float* beta,
                                const float* moving_mean, const float* moving_var, void* y, float* rstd_tmp,
                                int64_t npix, int c, float eps, int relu, int dtype, void* stream);
int ad_batchnorm_relu_bwd(const void* dy, const void* z, const float* save_mean, const float* save_rstd,
                          const float* gamma, const float* beta, void* dz, float* dgamma, float* dbeta,
                          int64_t npix, int c, int relu, void* ws, size_t ws_bytes, int dtype, void* stream);

int ad_batchnorm_relu_bwd_dbias(const void* dy, const void* z, const float* save_mean, const float* save_rstd,
                                const float* gamma, const float* beta, void* dz, float* dgamma, float* dbeta, float* dbias,
                                int64_t npix, int c, int relu, void* ws, size_t ws_bytes, int dtype, void* stream);

/* out[c] = sum over pixels of x[npix, c] (BiasAddGrad of layers without a fused producer); ws as batchnorm. */
int ad_colsum(const void* x, float* out, int64_t npix, int c, void* ws, size_t ws_bytes, int dtype, void* stream);

/* L.MaxPooling2D(pool_size=(2,2)) (train_adaptive_unet.py:351 of Segmenation/code) and its gradient
 * (first maximal element of each window, as TF MaxPoolGrad). y: [n, h/2, w/2, c]. */
int ad_maxpool2_fwd(const void* x, void* y, int n, int h, int w, int c, int dtype, void* stream);
int ad_maxpool2_bwd(const void* dy, const void* x, void* dx, int n, int h, int w, int c, int dtype, void* stream);
/* dx = pooling gradient + add[n,h,w,c] (fp32 sum, one rounding): the encoder junction of the U-Net's backward pass, where the
 * gradient through MaxPooling2D meets the skip connection's (:349-351,358). */
int ad_maxpool2_bwd_add(const void* dy, const void* x, const void* add, void* dx, int n, int h, int w, int c, int dtype,
                        void* stream);

/* Pixel shuffle behind L.Conv2DTranspose(nf, 2, strides=2) (Segmenation/code/unet_vinillia.py:67): the
 * transposed conv is one pointwise GEMM to 4*nf channels (ad_conv3x3_fwd on a 1x1 geometry) followed by
 * depth-to-space; to_space=1: x[n,h,w,4c] (block a*2+b) -> y[n,2h,2w,c]; to_space=0: the inverse gather. */
int ad_pixel_shuffle2(const void* x, void* y, int n, int h, int w, int c, int to_space, int dtype, void* stream);

/* Conv2D(1, 1, activation="sigmoid") head (:361) + the per-sample sums behind BinaryCrossentropy, dice and
 * IoU (:258-318): prob[npix] = sigmoid(xh @ w[ch] + b); sums[n][3] = {sum BCE, sum y*pc, sum (y+pc)} with
 * pc = clip(p, 1e-7, 1-1e-7).  Backward of loss = bce_weight*mean(BCE) + dice_weight*(1 - mean_n dice_n). */
size_t ad_seg_head_ws_bytes(int n, int ch);
int ad_seg_head_fwd(const void* xh, const float* w, const float* b, const float* target, float* prob, float* sums,
                    int n, int64_t pix_per_img, int ch, void* ws, size_t ws_bytes, int dtype, void* stream);
/* ad_seg_head_fwd plus counts[n][6] for the vanilla baseline's Keras metrics (Segmenation/code/unet_vinillia.py:266-271:
 * BinaryAccuracy / Precision / Recall at threshold 0.5, and its dice_coefficient on the unclipped probability, :94-99):
 *   counts[img] = { sum [p > .5] y, sum [p > .5], sum y, sum [(p > .5) == (y > .5)], sum y p, sum (y + p) }.
 * counts == NULL is ad_seg_head_fwd. */
int ad_seg_head_fwd_counts(const void* xh, const float* w, const float* b, const float* target, float* prob,
                           float* sums, float* counts, int n, int64_t pix_per_img, int ch,
                           void* ws, size_t ws_bytes, int dtype, void* stream);
/* The batch values Keras logs from those sums (:258-304, :307-314): out3 = { loss = bce_weight * sum_n sums[n][0] / count +
 * dice_weight * (1 - dice), dice = mean_n (2 I_n + smooth) / (U_n + smooth), iou = mean_n (I_n + smooth) / (U_n - I_n + smooth) }
 * with I_n = sums[n][1], U_n = sums[n][2]; count = elements of the mask batch.  One launch, stays on the device (graph replay). */
int ad_seg_metrics(const float* sums, int n, float count, float bce_weight, float dice_weight, float smooth, float* out3,
                   void* stream);
int ad_seg_head_bwd(const void* xh, const float* w, const float* target, const float* prob, const float* sums,
                    void* dxh, float* dw, float* db, int n, int64_t pix_per_img, int ch,
                    float bce_weight, float dice_weight, float smooth,
                    const float* loss_scale /* NULL or device float, as ad_head_bwd */,
                    void* ws, size_t ws_bytes, int dtype, void* stream);

/* Conv2D(num_classes, 1, activation="softmax") head of build_unet(num_classes > 1)
 * (Segmenation/code/unet_vinillia.py:89-90).  w: [ch][num_classes] fp32 (Keras kernel (1,1,ch,K)), b: [num_classes];
 * prob: [npix][num_classes] fp32.  Forward only: the reference defines no loss or metric for this head. */
int ad_softmax_head_fwd(const void* xh, const float* w, const float* b, float* prob, int64_t npix, int ch,
                        int num_classes, int dtype, void* stream);

/* Same update with the step-dependent factor alpha = lr*sqrt(1-b2^t)/(1-b1^t) read from DEVICE memory, so that a
 * train step captured in a hipGraph can be replayed with a new learning rate / step index (the host writes
 * ad_adam_alpha(lr, b1, b2, t) into alpha_dev before each replay). */
float ad_adam_alpha(float lr, float b1, float b2, int step);
int ad_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t count, const float* alpha_dev,
                     float b1, float b2, float eps, float gscale, void* stream);

/* Dynamic loss scaling for AD_F16 (Keras LossScaleOptimizer under the reference's mixed_float16 policy,
 * Super_resolution/code/train_adaptive_unet.py:471-477; Segmenation/code/train_adaptive_unet.py:471-476): the loss
 * gradient is multiplied by `scale` before the backward pass (ad_head_bwd / ad_seg_head_bwd read it from device memory),
 * the optimizer divides it out, a step whose gradients contain inf / NaN is skipped and halves the scale, and
 * `growth_steps` consecutive finite steps double it.  state (device, 8 floats, initialised by the host to
 * {scale, 1/scale, 0, 0, 0, 0, 0, 0}): [0] scale, [1] 1/scale, [2] finite steps since the last change, [3] overflow
 * flag of the current step, [4] optimizer applications so far, [5] skipped steps so far.  One step is
 *   backward -> ad_loss_scale_check(G) -> ad_adam_step_scaled -> ad_loss_scale_update
 * with no host round trip, so the whole sequence can sit in a hipGraph. */
int ad_loss_scale_check(const float* grads, int64_t count, float* state, void* stream);
int ad_loss_scale_update(float* state, int growth_steps, void* stream);
/* Keras-form Adam as ad_adam_step_dev, but: nothing happens when state[3] != 0; gradients are multiplied by
 * gscale * state[1]; the bias-correction step index is state[4] + 1 (applied updates); lr_dev[0] = learning rate. */
int ad_adam_step_scaled(float* p, const float* g, float* m, float* v, int64_t count, const float* lr_dev,
                        float b1, float b2, float eps, float gscale, const float* state, void* stream);

/* -------------------------------------------------------------- utilities -- */

int ad_cast(const void* x, int dtype_in, void* y, int dtype_out, int64_t count, void* stream);

/* dgrad of a conv whose (first) input is the ReLU output of a preceding Conv2D(..., activation="relu") -- the decoder's
 * up-conv (train_adaptive_unet.py:259) feeding conv_block -- with that ReLU's gradient fused into the epilogue:
 *   y1 = (dz (*) W^T)[..., :cy1] * [relu_out > 0]      (= the gradient w.r.t. the up-conv's PRE-activation output)
 *   y2 = (dz (*) W^T)[..., cy1:]                        (skip half of the virtual concat, untouched; NULL if cy1 == cout)
 *   dbias[cy1] = column sums of y1 as stored            (the up-conv's bias gradient, BiasAddGrad)
 * replacing ad_conv3x3_fwd on the dgrad pack followed by ad_relu_bwd (one full read-modify-write pass less).  Only
 * launches that ad_conv3x3_dgrad_relu_is_fused() accepts (16-bit dtypes, 64-channel contraction, >= 4 tiles per CU);
 * ws: ad_conv3x3_dgrad_relu_ws_bytes(). */
int ad_conv3x3_dgrad_relu_is_fused(int n, int h, int w, int c1, int cout, int cy1, int dtype);
size_t ad_conv3x3_dgrad_relu_ws_bytes(void);
int ad_conv3x3_dgrad_relu(const void* dz, int c1, const void* w_dgrad, const void* relu_out, void* y1, int cy1, void* y2,
                          float* dbias, int n, int h, int w, int cout, void* ws, size_t ws_bytes, int dtype, void* stream);

/* dgrad of a 64 -> 64 conv whose INPUT was the activation of a Conv2D -> LayerNormalization -> ReLU layer (conv_block's
 * second conv, Super_resolution/code/train_adaptive_unet.py:200-210; replaces Conv2DBackpropInput followed by ReluGrad,
 * the LayerNormalization gradient and BiasAddGrad of the layer below).  dz [n,h,w,c1]: gradient of this conv's output;
 * w_dgrad: its dgrad pack (cout = 64 input channels); z_prev / mean / rstd / gamma / beta: what the layer below stored
 * in the forward pass.  Writes dz_prev [n,h,w,64] (gradient of the lower conv's output) and dgamma / dbeta / dbias [64]
 * of the lower layer; the gradient of the activation itself never goes to memory (it is NOT rounded to the storage
 * type on the way, unlike the two-launch path).  Only where _is_fused() says so (half types, weights-resident kernel);
 * ws: ad_conv3x3_dgrad_ln_bwd_ws_bytes(). */
int ad_conv3x3_dgrad_ln_bwd_is_fused(int n, int h, int w, int c1, int cout, int dtype);
size_t ad_conv3x3_dgrad_ln_bwd_ws_bytes(void);
int ad_conv3x3_dgrad_ln_bwd(const void* dz, int c1, const void* w_dgrad, const void* z_prev, const float* mean,
                            const float* rstd, const float* gamma, const float* beta, void* dz_prev, float* dgamma,
                            float* dbeta, float* dbias, int n, int h, int w, int cout, void* ws, size_t ws_bytes, int dtype,
                            void* stream);

/* The skip connection's gradient junction (train_adaptive_unet.py:247-250: `skips.append(x); x = enc_down(x)`) fused
 * with the LayerNorm + ReLU backward of the conv_block that produced the skip:
 *   d(act) = dskip + ResizeByScale^T d_low       (= ad_resample(..., accumulate = 1) into dskip)
 *   dz, dgamma, dbeta, dbias = LayerNorm/ReLU backward of d(act)      (= ad_layernorm_relu_bwd)
 * in one pass, d(act) never stored.  Tables as ad_resample (transposed spans).  Shapes ad_resample_ln_bwd_supported()
 * accepts: c / (16 B of dtype) a power of two <= 64, ow * that a multiple of 256, kx <= 8. */
int ad_resample_ln_bwd_supported(int n, int oh, int ow, int c, int kx, int dtype);
size_t ad_resample_ln_bwd_ws_bytes(int n, int oh, int ow, int c, int dtype);
int ad_resample_ln_bwd(const void* d_low, const void* dskip, const void* z, const float* mean, const float* rstd,
                       const float* gamma, const float* beta, void* dz, float* dgamma, float* dbeta, float* dbias,
                       const int* sy, const float* wy, int ky, const int* sx, const float* wx, int kx,
                       int n, int h, int w, int oh, int ow, int c, void* ws, size_t ws_bytes, int dtype, void* stream);

/* ------------------------------------------- decoder step without the up-resized tensor -- */

/* `x = dec_up([x, skip]); x = L.Conv2D(nf, 3, padding="same", activation="relu")(x)`
 * (Super_resolution/code/train_adaptive_unet.py:258-259; ResizeToMatch: shared/custom_layers.py:121-125).
 * The resize acts on pixels, the convolution's contraction on channels, so they commute:
 *     conv3x3(U x)[p] = b + sum_tap (U (x W_tap))[p + tap]            (zero outside the high-resolution image)
 * and the step runs as a bank of nine 1x1 convolutions on the LOW-resolution map (one GEMM, 1 / ratio^2 of the FLOPs)
 * followed by an interpolating / shifting gather; backwards as the transposes.  The up-resized activation and its
 * gradient never exist in memory.  tests/test_oracle_factored_upconv.py proves the identity on the oracle.
 *
 *   ad_pw_bank_pack   fp32 Keras kernel W[3][3][Cin][Cout] -> the two GEMM operands (dtype): bank_fwd = B[ci][tap*Cout+co]
 *                     (for Y = x B), bank_bwd = B^T (for dx = dY B^T), ad_pw_bank_elems() elements each.
 *   ad_pw_gemm        y[m][n] = sum_k x[m][k] bank[k][n] on the matrix cores; x, y NHWC with the pixels flattened;
 *                     shapes ad_pw_supported() accepts: k % 64 == 0, n % 64 == 0, every tensor below 2 GiB.
 *   ad_pw_bank_grad   dW[3][3][Cin][Cout] (Keras layout, fp32) from the [3][3][Cin][9 Cout] tensor that
 *                     ad_conv3x3_wgrad(x as [m,1,1,Cin], dY as [m,1,1,9 Cout]) writes (dBank = x^T dY is its centre tap).
 *   ad_upconv_gather_fwd   out[n,oy,ox,c] = act(bias[c] + sum_{dy,dx in -1..1} [oy+dy, ox+dx inside] sum_{a,b<2}
 *                              wy[2(oy+dy)+a] wx[2(ox+dx)+b] ybank[n, sy[oy+dy]+a, sx[ox+dx]+b, (dy+1)*3+(dx+1), c])
 *                     two-tap tables of the up-resize (second index clamped where its weight is 0), fp32 arithmetic;
 *                     sy non-decreasing with sy[r+1] - sy[r-1] <= window - 2, window in {3, 4} low-resolution rows;
 *                     slab_cols = the most low-resolution columns any group of 256 / (c / V) adjacent output columns (and
 *                     their +-1 neighbours) reads, V = channels per 8 bytes: the piece of a bank row a workgroup stages in LDS.
 *   ad_upconv_gather_bwd   dybank = gather^T(g): tables of the TRANSPOSED resize (first reading row / column and kyt / kxt
 *                     weights per low-resolution index); kxt as ad_upconv_gather_bwd_supported() accepts. */
int ad_pw_supported(int64_t m, int k, int n, int dtype);
size_t ad_pw_bank_elems(int cin, int cout);
int ad_pw_bank_pack(const float* w_hwio, int cin, int cout, void* bank_fwd, void* bank_bwd, int dtype, void* stream);
int ad_pw_gemm(const void* x, const void* bank, void* y, int64_t m, int k, int n, int dtype, void* stream);
/* Introspection of the LDS-tiled ad_pw_gemm launch (host only, no GPU work): channels per tile it would use for a shape (256 /
 * 192 / 128; 0: the shape runs on the kernel without LDS tiles), and the XCD-aware tile order of a launch of `grid` workgroups
 * over tiles_m x tiles_n tiles: tiles[b * max_rounds + r] = tile index (row-major) workgroup b computes in its round r, or -1;
 * returns the number of rounds of the slowest workgroup (-1: bad arguments / more than max_rounds).  tests/test_host_logic.py
 * checks that every tile appears exactly once. */
int ad_pw_gemm_tile_channels(int64_t m, int k, int n, int dtype);
int ad_pw_gemm_tile_order(int tiles_m, int tiles_n, int grid, int* tiles, int max_rounds);
/* the kernel ad_pw_gemm launches for a shape: 0 fragments from L2, 1 / 2 LDS-tiled with one / two k-stages of loads in flight
 * (-1: unsupported shape); for tests that must reach a given variant */
int ad_pw_gemm_variant(int64_t m, int k, int n, int dtype);
int ad_pw_bank_grad(const float* dw9, int cin, int cout, float* dw_hwio, void* stream);
/* dW[3][3][Cin][Cout] (Keras layout, fp32) = re-ordered x^T dybank over the m pixels, in one pass over x and dybank
 * (16-bit types, Cin % 128 == 0, Cout % 64 == 0: ad_pw_wgrad_supported; elsewhere ad_conv3x3_wgrad + ad_pw_bank_grad).
 * Per-workgroup partial sums go to ws (ad_pw_wgrad_ws_bytes) and are added in a fixed order: bitwise deterministic. */
int ad_pw_wgrad_supported(int64_t m, int cin, int cout, int dtype);
size_t ad_pw_wgrad_ws_bytes(int64_t m, int cin, int cout);
int ad_pw_wgrad(const void* x, const void* dybank, float* dw_hwio, int64_t m, int cin, int cout, void* ws, size_t ws_bytes,
                int dtype, void* stream);
int ad_upconv_gather_fwd_supported(int c, int slab_cols, int dtype);
/* slab_cols for a horizontal table (HOST copy of sx: ow entries, non-decreasing, inside [0, w)); -1 if the table or c is not
 * acceptable.  Callers take the value from here instead of restating the kernel's staging rule. */
int ad_upconv_slab_cols(const int* sx_host, int w, int ow, int c, int dtype);
int ad_upconv_gather_fwd(const void* ybank, const float* bias, void* out, const int* sy, const float* wy,
                         const int* sx, const float* wx, int window, int slab_cols, int n, int h, int w, int oh, int ow,
                         int c, int relu, int dtype, void* stream);
int ad_upconv_gather_bwd_supported(int kxt);
int ad_upconv_gather_bwd(const void* g, void* dybank, const int* ryt, const float* wyt, int kyt, const int* cxt,
                         const float* wxt, int kxt, int n, int h, int w, int oh, int ow, int c, int dtype,
                         void* stream);

/* ------------------------------------------------------- gradient exchange -- */

/* Data parallelism over patches (the reference is single-GPU): one process per GPU, the flat fp32 gradient
 * buffer is summed across ranks bucket by bucket with RCCL over xGMI, in place, on a communication stream the caller
 * orders against its compute stream with events; the optimizer applies 1/world.  The communicator is created from a
 * 128-byte unique id that rank 0 obtains and the host distributes (any channel: torch.distributed, a file, MPI).
 * RCCL is loaded with dlopen at the first call. */
int ad_comm_unique_id(void* id128);
int ad_comm_create(const void* id128, int rank, int world, void** comm);
int ad_comm_destroy(void* comm);
int ad_allreduce_bucket(void* comm, float* grads, int64_t count, void* stream);

/* ------------------------------------------------------------- feed path -- */

/* LR synthesis on the device (shared/pipeline.py:79-94 degrade_image, applied to a whole HR batch in HBM by two
 * ad_resample launches with INTER_AREA / INTER_CUBIC tables): layout helpers around the 16-byte channel vectors the
 * resample kernel works on.  All fp32.
 *   ad_u8_to_float_pad  x_u8[npix,3] -> hr3[npix,3] = x/255 and hr4[npix,4] = (r,g,b,0)
 *   ad_pad_clip_f32     y[npix,cpad] = clip(x[npix,c], 0, 1), zero channel padding
 *   ad_take_channels    y[npix,cout] = x[npix, 0..cout) */
int ad_u8_to_float_pad(const void* x_u8, float* hr3, float* hr4, int64_t npix, void* stream);
int ad_pad_clip_f32(const float* x, float* y, int64_t npix, int c, int cpad, void* stream);
int ad_take_channels(const float* x, float* y, int64_t npix, int cin, int cout, void* stream);

/* ------------------------------------------------------- evaluation metrics -- */

/* The reference's eval loops (Super_resolution/code/train_adaptive_unet.py:673-694, evaluate_model.py:94-163) on the
 * device.  Planes are single-channel fp32 images addressed as plane[img * image_stride + y * row_stride + x], so a
 * shaved window of a larger plane needs no copy (pass the window's first pixel, its extent and the parent's strides).
 *   ad_luma_bt601     y = clip((65.481 r + 128.553 g + 24.966 b + 16) / 255, 0, 1) of clip(rgb, 0, 1)     (:144-157)
 *   ad_mse_per_image  mse[img] = mean (a - b)^2            (tf.reduce_mean(tf.square(..)), :691; PSNR = -10 log10)
 *   ad_ssim_per_image ssim_cs[img][0] = tf.image.ssim(a, b, max_val) (11x11 Gaussian sigma 1.5, VALID, K1 .01, K2 .03),
 *                     ssim_cs[img][1] = the mean contrast-structure term that tf.image.ssim_multiscale multiplies
 *   ad_avgpool2_plane 2x2 average pooling between MS-SSIM scales; odd extents repeat their last row / column
 * ws: ad_metrics_ws_bytes(n, h, w). */
int ad_luma_bt601(const float* rgb, float* y, int64_t npix, void* stream);
size_t ad_metrics_ws_bytes(int n, int h, int w);
int ad_mse_per_image(const float* a, const float* b, int n, int h, int w, int64_t image_stride, int row_stride,
                     float* mse, void* ws, size_t ws_bytes, void* stream);
int ad_ssim_per_image(const float* a, const float* b, int n, int h, int w, int64_t image_stride, int row_stride,
                      float max_val, float* ssim_cs, void* ws, size_t ws_bytes, void* stream);
int ad_avgpool2_plane(const float* x, int n, int h, int w, int64_t image_stride, int row_stride, float* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ADUNET_H */
