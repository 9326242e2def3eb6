/*
 * nc_hip.h -- C ABI of libnc_hip.so: the MI355X (gfx950) hot path of Neuroclear.
 *
 * The reference (peterhpark/neuroclear) has NO native / FFI interface: its hot path is torch.nn modules executed by
 * cuDNN/ATen.  Each entry point below therefore cites the reference *call site* it replaces
 * (paths relative to the reference checkout), and INTEGRATION.md shows the ctypes stub a maintainer adds.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch tensors on the Python side), dense, fp32 unless
 *    the name says otherwise; activations are NCDHW (2-D data: D = 1, kd = 1);
 *  - `stream` is a hipStream_t passed as void*; nothing here allocates or frees, and nothing synchronises -- with ONE exception:
 *    the image-staged PatchGAN kernels (nc_conv_* on 2-D 4x4 layers at large batches, conv2d_img.hip) time their candidate tile
 *    shapes on the FIRST call of a problem shape: a hipDeviceSynchronize plus a few timed launches, skipped when the stream is
 *    being captured.  nc_sconv_set_tune(0) (or NC_SCONV_TUNE=0) switches that off for the whole process: then the rule holds
 *    without exception (a fixed heuristic picks the shape; results are bit-identical either way);
 *  - process-wide switches (nc_set_conv_split, nc_set_c8x_mode, nc_set_s3_fusion, nc_sconv_set_tune, nc_sconv_set_cfg, nc_set_force_direct) are
 *    atomics read once per call: flip them between calls, not while other host threads are inside the library;
 *  - scratch memory comes from the caller: `ws` / `ws_bytes`; query the size with the matching *_ws_bytes();
 *  - return value: 0 = ok, negative = NC_ERR_*; nc_last_error() gives a thread-local message.
 */
#ifndef NC_HIP_H
#define NC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NC_OK 0
#define NC_ERR_SHAPE (-1)  /* unsupported / inconsistent dimensions                       */
#define NC_ERR_WS (-2)     /* workspace missing or too small                              */
#define NC_ERR_HIP (-3)    /* a HIP runtime call failed (message has hipGetErrorString)   */
#define NC_ERR_ARG (-4)    /* null pointer / bad enum                                     */

const char* nc_last_error(void);
int nc_version(void);
/* Which implementation nc_conv_* would pick for a shape: 0 = direct (VALU), 1 = MFMA brick implicit GEMM (3^3/5^3,
 * stride 1), 2 = MFMA gather implicit GEMM (any kernel / stride), 3 = flat-voxel MFMA kernel of the 1x1 weight
 * gradient (<= 64 channels, planes of >= 16384 voxels), 4 = tap-axis MFMA kernel of the 1 -> 64 channel weight
 * gradient (3^3 / 7^3, W % 4 == 0).  The query assumes a 256^2 plane for pointwise kernels, a 32^3 volume otherwise. */
int nc_conv_fwd_path(int C, int K, int kd, int kh, int kw, int stride, int pad);
int nc_conv_wgrad_path(int C, int K, int kd, int kh, int kw, int stride, int pad);
/* Live launch profiler (bench.py's `roofline`): between nc_prof_begin and nc_prof_end every convolution entry point of
 * >= min_flop algorithmic FLOP is bracketed by HIP events on the stream it is launched on.  nc_prof_end returns the number
 * of recorded calls and fills the first `max`: cls = op (0 fwd, 1 dgrad, 2 wgrad) | path << 4 | kernel edge << 8 |
 * (16-bit kernel ? 1 << 16 : 0), flop = 2 C K kd kh kw x output voxels, ms = event-to-event duration.  Synchronise first. */
void nc_prof_begin(double min_flop);
int nc_prof_end(int max, int* cls, double* flop, float* ms);
int nc_prof_end2(int max, int* cls, double* flop, float* ms, double* abytes); /* + the launch's algorithmic bytes: operands once (4 B; 2 B on the
                                   * 16-bit path), result, weights */
/* Force the direct path everywhere (tests cross-check MFMA vs direct on the GPU). 0 = auto (default), 1 = force. */
void nc_set_force_direct(int on);
/* The image-staged PatchGAN kernels (conv2d_img.hip) pick their tile shape per problem by timing the candidates on the
 * first call (NC_SCONV_TUNE=0: a fixed heuristic).  Every shape gives the same bits; this pins shape `cfg` (0-based, -1 =
 * back to automatic) so that a test can check exactly that. */
void nc_sconv_set_cfg(int cfg);
/* 1 (default; NC_SCONV_TUNE): time the candidate shapes on the first call of a problem shape (synchronises once per shape, see
 * Conventions); 0: never time, never synchronise -- the heuristic shape; -1: back to the environment's choice. */
void nc_sconv_set_tune(int on);

/* ---- Convolution: nn.Conv3d / nn.Conv2d (models/networks.py:361-369; used at :420-425,:442,:460-469 (U-Net 3^3),
 *      :899-911 (deep_linear 7^3/5^3/3^3/1^3), :507-508 (1x1 tail), :1030-1057 (PatchGAN 4x4 s2/s1)).
 *      x[N,C,D,H,W], w[K,C,kd,kh,kw], bias[K] or NULL, y[N,K,Do,Ho,Wo]; Do = (D+2p-kd)/s+1 (p,s apply to depth only
 *      when kd > 1).  dgrad / wgrad are the autograd backward of the same call (loss.backward(), apollo:283).      */
size_t nc_conv_ws_bytes(int N, int C, int D, int H, int W, int K, int kd, int kh, int kw, int stride, int pad);
int nc_conv_fwd(const float* x, const float* w, const float* bias, float* y, int N, int C, int D, int H, int W,
                int K, int kd, int kh, int kw, int stride, int pad, void* ws, size_t ws_bytes, void* stream);
int nc_conv_dgrad(const float* dy, const float* w, float* dx, int N, int C, int D, int H, int W, int K, int kd,
                  int kh, int kw, int stride, int pad, void* ws, size_t ws_bytes, void* stream);
int nc_conv_wgrad(const float* x, const float* dy, float* dw, float* dbias /* or NULL */, int N, int C, int D, int H,
                  int W, int K, int kd, int kh, int kw, int stride, int pad, void* ws, size_t ws_bytes, void* stream);
/* Backward of one layer in one call: dx (or NULL) and dw (+ dbias or NULL) -- nc_conv_dgrad followed by nc_conv_wgrad with
 * the conversions of dy shared where the kernels have one (nn.Conv3d backward, same call sites). */
int nc_conv_bwd(const float* x, const float* dy, const float* w, float* dx, float* dw, float* dbias, int N, int C, int D, int H, int W,
                int K, int kd, int kh, int kw, int stride, int pad, void* ws, size_t ws_bytes, void* stream);

/* ---- The same convolution on the 16-bit matrix cores (BASELINE.json configs[3]: "fp16 MFMA path with fp32
 *      InstanceNorm accumulate").  Tensors and master weights stay fp32 at this boundary; inside, operands are rounded
 *      (round-to-nearest-even) to `dtype` (NC_DT_BF16: v_mfma_f32_32x32x16_bf16, NC_DT_F16: ..._f16), products are
 *      accumulated in fp32 and the result is written in fp32.  Covers odd cubic kernels 3^3 / 5^3, stride 1, "same"
 *      padding with C % 16 == 0 and K % 64 == 0 (fwd), K % 16 == 0 and C % 64 == 0 (dgrad) -- every 3^3 layer of
 *      unet_deconv except the first, G_B's feature block; nc_conv_lp_supported tells (what: 0 fwd, 1 dgrad, 2 wgrad);
 *      unsupported shapes return NC_ERR_SHAPE (callers use nc_conv_* for those -- there is no silent fallback).       */
#define NC_DT_F32 0
#define NC_DT_F16 1
#define NC_DT_BF16 2
int nc_conv_lp_supported(int what, int N, int C, int D, int H, int W, int K, int kd, int kh, int kw, int stride, int pad);
size_t nc_conv_lp_ws_bytes(int N, int C, int D, int H, int W, int K, int kd, int kh, int kw, int stride, int pad);
/* Operand layout of the 16-bit kernels, "C8": [N][C/8][S voxels][8 channels] 16-bit (S = D*H*W), C % 8 == 0.  Each
 * operand of the *_lp calls is given EITHER as the fp32 NCDHW tensor (x / dy; converted into the workspace on every
 * call) OR already converted (xh / dyh non-NULL, then the fp32 pointer may be NULL) -- a caller that needs the same
 * tensor twice (x: forward and weight gradient; dy: data and weight gradient) converts it once with nc_to_c8.        */
size_t nc_c8_bytes(int N, int C, long S);
int nc_to_c8(const float* x, void* xh, int N, int C, long S, int dtype, void* stream);
int nc_conv_fwd_lp(const float* x, const void* xh, const float* w, const float* bias, float* y, int N, int C, int D, int H,
                   int W, int K, int kd, int kh, int kw, int stride, int pad, int dtype, void* ws, size_t ws_bytes,
                   void* stream);
int nc_conv_dgrad_lp(const float* dy, const void* dyh, const float* w, float* dx, int N, int C, int D, int H, int W, int K,
                     int kd, int kh, int kw, int stride, int pad, int dtype, void* ws, size_t ws_bytes, void* stream);
/* weight gradient: 3^3 / 5^3, C % 32 == 0, K % 64 == 0; dbias (nullable) is summed in fp32 from the fp32 dy */
int nc_conv_wgrad_lp(const float* x, const void* xh, const float* dy, const void* dyh, float* dw,
                     float* dbias /* or NULL */, int N, int C, int D, int H, int W, int K, int kd, int kh, int kw,
                     int stride, int pad, int dtype, void* ws, size_t ws_bytes, void* stream);

/* ---- ConvTranspose3d(k=2, s=2) (networks.py:500,503): x[N,C,D,H,W], w[C,K,2,2,2], bias[K], y[N,K,2D,2H,2W].   */
size_t nc_convT_ws_bytes(int N, int C, int D, int H, int W, int K);
int nc_convT_k2s2_fwd(const float* x, const float* w, const float* bias, float* y, int N, int C, int D, int H, int W,
                      int K, void* stream);
int nc_convT_k2s2_dgrad(const float* dy, const float* w, float* dx, int N, int C, int D, int H, int W, int K,
                        void* ws, size_t ws_bytes, void* stream);
int nc_convT_k2s2_wgrad(const float* x, const float* dy, float* dw, float* dbias, int N, int C, int D, int H, int W,
                        int K, void* ws, size_t ws_bytes, void* stream);

/* ---- InstanceNorm{2,3}d(affine=False, track_running_stats=False) + ReLU / LeakyReLU(slope)
 *      (networks.py:33-34 with :422-423 (slope 0) and :1042-1046 (slope 0.2)).  One instance = one (n, c) plane of S
 *      elements; biased variance, eps inside the sqrt.  `stats` computes mean / rstd (fp64 accumulation), `fwd`
 *      applies y = act((x-mean)*rstd), `bwd` is the backward of the pair given the pre-norm x.                   */
size_t nc_instnorm_ws_bytes(int NC, long S);
int nc_instnorm_stats(const float* x, int NC, long S, float eps, float* mean, float* rstd, void* ws, size_t ws_bytes,
                      void* stream);
/* stats + fwd in one call (mean / rstd are outputs).  Instances of S <= 2048 elements -- the 2-D PatchGAN layers --
 * run as ONE kernel with a group of 16 or 64 lanes per instance; the three entry points agree bit for bit. */
int nc_instnorm_fwd(const float* x, float eps, float slope, float* mean, float* rstd, float* y, int NC, long S, void* ws,
                    size_t ws_bytes, void* stream);
int nc_instnorm_act_fwd(const float* x, const float* mean, const float* rstd, float slope, float* y, int NC, long S,
                        void* stream);
int nc_instnorm_act_bwd(const float* dy, const float* x, const float* mean, const float* rstd, float slope, float* dx,
                        int NC, long S, void* ws, size_t ws_bytes, void* stream);
/* --norm batch: nn.BatchNorm{2,3}d(affine=True, track_running_stats=True) (+ the ReLU / LeakyReLU behind it) of get_norm_layer
 * (models/networks.py:30-31) -- statistics over (N, spatial) per channel; training: running statistics updated in place (momentum, unbiased
 * variance), evaluation (training = 0): the running statistics are used.  mean / rstd: [C]; ws: nc_instnorm_ws_bytes(N * C, S); N * C <= 65535. */
int nc_batchnorm_stats(const float* x, int N, int C, long S, float eps, float momentum, int training, float* mean, float* rstd,
                       float* running_mean, float* running_var, void* ws, size_t ws_bytes, void* stream);
int nc_batchnorm_act_fwd(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, float slope, float* y,
                         int N, int C, long S, void* stream);
int nc_batchnorm_act_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                         float slope, int training, float* dx, float* dgamma, float* dbeta, float* coef /* 2 C floats of scratch */, int N, int C,
                         long S, void* ws, size_t ws_bytes, void* stream);
/* Same backward, and dbias[C] = the per-channel sum of dx over samples and voxels: dx is the gradient at the output of
 * the convolution in front of the norm (networks.py:420-423), so this IS that convolution's bias gradient -- taken
 * while dx is in registers instead of by a second pass over dx (nc_conv_wgrad with dbias = NULL then). */
size_t nc_instnorm_bwd_dbias_ws_bytes(int NC, long S);
int nc_instnorm_act_bwd_dbias(const float* dy, const float* x, const float* mean, const float* rstd, float slope, float* dx,
                              float* dbias, int N, int C, long S, void* ws, size_t ws_bytes, void* stream);
/* The same backward with dx delivered ONLY as an S3 tensor [N][C/8][3][S][8] bf16 (the three-term operand form of the split-operand
 * convolutions, nc_to_s3): dx is the dY of the convolution in front of the norm, and when that convolution's gradients run on the
 * split-operand kernels nothing else reads it (nc_unet_deconv_bwd writes it straight into the convolution's workspace).  Values are bit
 * for bit those nc_instnorm_act_bwd_dbias stores; C % 8 == 0, S > 2048 (longer than the short-instance path), N * C <= 65535. */
int nc_instnorm_act_bwd_dbias_s3(const float* dy, const float* x, const float* mean, const float* rstd, float slope, void* dxs, float* dbias,
                                 int N, int C, long S, void* ws, size_t ws_bytes, void* stream);
/* The normalise + activate pass and its backward for the 16-bit convolution path: next to the fp32 result they emit it
 * in the C8 operand layout (nc_to_c8) of the convolution that consumes it -- yh for the next layer's forward, dxh (the
 * gradient at the previous convolution's output) for its data / weight gradient -- which saves those conversion passes.
 * C % 8 == 0; dbias nullable (see nc_instnorm_act_bwd_dbias); workspace: nc_instnorm_bwd_dbias_ws_bytes. */
int nc_instnorm_act_fwd_c8(const float* x, const float* mean, const float* rstd, float slope, float* y /* or NULL */, void* yh,
                           int N, int C, long S, int dtype, void* stream);
int nc_instnorm_act_bwd_c8(const float* dy, const float* x, const float* mean, const float* rstd, float slope, float* dx,
                           void* dxh, float* dbias /* or NULL */, int N, int C, long S, int dtype, void* ws, size_t ws_bytes,
                           void* stream);
/* LeakyReLU alone (PatchGAN first block, networks.py:1030) */
int nc_leaky_relu_fwd(const float* x, float slope, float* y, long n, void* stream);
int nc_leaky_relu_bwd(const float* dy, const float* x, float slope, float* dx, long n, void* stream);

/* ---- MaxPool3d(2) (networks.py:491,494): floor mode; first maximum wins on ties (scan order d,h,w).             */
int nc_maxpool2_fwd(const float* x, float* y, int NC, int D, int H, int W, void* stream);
int nc_maxpool2_bwd(const float* dy, const float* x, float* dx, int NC, int D, int H, int W, void* stream);
/* dx = skip + pool-backward(dy): the pooled tensor of Unet_deconv also feeds the skip concat (networks.py:526,531), so its
 * gradient is the sum of both paths (autograd adds them in a separate pass); even D (or D == 1), H, W only.        */
int nc_maxpool2_bwd_add(const float* dy, const float* x, const float* skip, float* dx, int NC, int D, int H, int W,
                        void* stream);

/* ---- Sigmoid (networks.py:510,536) */
int nc_sigmoid_fwd(const float* x, float* y, long n, void* stream);
int nc_sigmoid_bwd(const float* dy, const float* y, float* dx, long n, void* stream);

/* ---- Volume.get_slice / get_projection (apollo_model.py:328-351): vol[N,C,D,H,W], axis in {0,1,2} = (D,H,W).
 *      slice: out = vol[..., index, ...]; mip: out = max over [start, start+depth) along axis (+ int32 argmax).
 *      The *_bwd kernels write the FULL dvol (zeros elsewhere).                                                   */
int nc_slice_fwd(const float* vol, float* out, int NC, int D, int H, int W, int axis, int index, void* stream);
int nc_slice_bwd(const float* dout, float* dvol, int NC, int D, int H, int W, int axis, int index, void* stream);
int nc_mip_fwd(const float* vol, float* out, int32_t* arg, int NC, int D, int H, int W, int axis, int start, int depth,
               void* stream);
int nc_mip_bwd(const float* dout, const int32_t* arg, float* dvol, int NC, int D, int H, int W, int axis,
               void* stream);

/* ---- Athena's iter_f (axial_to_lateral_gan_athena_model.py:286-296): every slice along `axis` as one batch.
 *      to_volume = 0: slices[(n*L+s), c, a, b] = vol[n, c, ...s...]; to_volume = 1: the inverse permutation (backward). */
int nc_volume_slices(const float* src, float* dst, int N, int C, int D, int H, int W, int axis, int to_volume,
                     void* stream);

/* ---- GANLoss('lsgan') = MSELoss against a constant (networks.py:276,299-313) and L1Loss (apollo:128,279).
 *      out[0] = mean; backward scales by the device scalar gscale[0] (autograd's incoming gradient).              */
size_t nc_loss_ws_bytes(long n);
int nc_mse_const_fwd(const float* pred, long n, float target, float* out, void* ws, size_t ws_bytes, void* stream);
int nc_mse_const_bwd(const float* pred, long n, float target, const float* gscale, float* dpred, void* stream);
/* GANLoss('vanilla') = nn.BCEWithLogitsLoss against the constant label (networks.py:278, 308-313): mean(max(p,0) - p t + log(1 + exp(-|p|)));
 * GANLoss('wgangp') = -+mean(p) (networks.py:314-318): nc_mean_fwd gives mean(p), the caller applies the sign (gscale carries it backward). */
int nc_bce_logits_const_fwd(const float* pred, long n, float target, float* out, void* ws, size_t ws_bytes, void* stream);
int nc_bce_logits_const_bwd(const float* pred, long n, float target, const float* gscale, float* dpred, void* stream);
int nc_mean_fwd(const float* pred, long n, float* out, void* ws, size_t ws_bytes, void* stream);
int nc_mean_bwd(long n, const float* gscale, float* dpred, void* stream);
int nc_l1_fwd(const float* a, const float* b, long n, float* out, void* ws, size_t ws_bytes, void* stream);
int nc_l1_bwd(const float* a, const float* b, long n, const float* gscale, float* da, void* stream);

/* ---- torch.optim.Adam step over one flat parameter buffer (apollo:131-136, step at :295,:307).                  */
int nc_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                 int step, void* stream);

/* ---- Dice / assemble (data/diceImage_dataset.py:95-120 + base_dataset.py:134-143; util/assemble_dice.py:130-213).
 *      vol: ORIGINAL (unpadded) uint16 / uint8 volume [L0,L1,L2] resident in HBM.  cut_cube writes cube `index` of
 *      edge E = roi + 2*border, float32 in [0,1]: zero dicing pad (util/util.py:196-215) then reflect border.
 *      scatter_add: acc[P0,P1,P2] += cube[border:-border]^3 / 8 at the cube's origin; finalize: (acc/cnt)*8*scale,
 *      truncating cast, crop to the original size; cnt is computed analytically from the grid.                    */
int nc_dice_cut_cube(const void* vol, int is_u16, int L0, int L1, int L2, int roi, int overlap, int border, int index,
                     float* cube, void* stream);
/* ---- Training augmentation "clean rotation" (data/base_dataset.py:306-460: rotate every z-slice with cv2.warpAffine,
 *      crop to the inscribed rectangle) fused with the random crop (:187-206) and __normalize (:134-143): produces only
 *      the crop.  inv_affine = 2x3 row-major map from rotated-image pixel (x, y) to source pixel; (x0, y0) = crop origin
 *      in the rotated image (inscribed-rectangle offset + random crop offset); z0 = first source slice.           */
int nc_rotate_crop(const void* vol, int is_u16, int D, int H, int W, int z0, int y0, int x0, int cz, int cy, int cx,
                   const double* inv_affine, float* out, void* stream);
int nc_assemble_scatter_add(const float* cube, float* acc, int P0, int P1, int P2, int roi, int overlap, int border,
                            int index, void* stream);
int nc_assemble_finalize(const float* acc, void* out, int out_is_u16, int P0, int P1, int P2, int L0, int L1, int L2,
                         int roi, int overlap, void* stream);
/* The same for planes [z0, z0 + nz) of the volume: acc_slab holds the PADDED planes from za (<= z0) on, out [nz][L1][L2] (sharded
 * inference: every rank finalises the z-slab it owns, neuroclear_amd/test_dice.py assemble='slab').  nc_assemble_scatter_add adds into
 * such a slab when it is given the slab's address minus za * P1 * P2 floats (it only touches the planes of its cube). */
int nc_assemble_finalize_slab(const float* acc_slab, void* out, int out_is_u16, int P0, int P1, int P2, int L0, int L1, int L2, int roi,
                              int overlap, int z0, int nz, int za, void* stream);

/* ---- --normalize_intensity (util/assemble_dice.py:188-192): merged = (acc / count) * 8 on the padded volume;
 *      nc_radix_hist = one pass of an exact radix select (order-preserving 32-bit key of a float: pass 0 bins key >> 20,
 *      pass 1 bins (key >> 8) & 0xfff among key >> 20 == prefix, pass 2 bins key & 0xff among key >> 8 == prefix;
 *      hist4096 is zeroed by the call) from which the host takes the order statistics np.percentile interpolates;
 *      rescale_finalize = skimage.exposure.rescale_intensity(in_range=(lo, hi)) on a float32 image (clip to [lo, hi],
 *      (v - lo) / range with range = float32(hi64 - lo64), output range (0, 1) or (-1, 1) if lo < 0) + cast + crop.     */
int nc_assemble_merge(const float* acc, float* merged, int L0, int L1, int L2, int roi, int overlap, void* stream);
int nc_radix_hist(const float* x, long n, int pass, unsigned prefix, unsigned* hist4096, void* stream);
int nc_assemble_rescale_finalize(const float* merged, void* out, int out_is_u16, int L0, int L1, int L2, int roi,
                                 int overlap, float lo, float hi, float range, void* stream);

/* ---- --histogram_match (util/assemble_dice.py:149-151): skimage.exposure.match_histograms(fake_cube, real_cube) on
 *      the border-cropped cubes, restated from scikit-image 0.18.3 (_match_cumulative_cdf: np.unique + np.cumsum +
 *      np.interp, float64).  source / tmpl / out: n floats each (out = the float64 result rounded to float32).      */
size_t nc_match_histograms_ws_bytes(long n);
int nc_match_histograms(const float* source, const float* tmpl, float* out, long n, void* ws, size_t ws_bytes,
                        void* stream);

/* ---- torch.nn.utils.spectral_norm on a convolution weight (NLayerDiscriminatorSN, networks.py:1069-1111; --netD
 *      basic_SN / n_layers_SN).  w_orig viewed as [K][M]; power_iteration = 1 (training): v <- normalize(W^T u),
 *      u <- normalize(W v) in place (eps 1e-12), then sigma = u . W v and w_out = W / sigma; 0 (eval): u, v as stored.
 *      scratch_k: K floats.  bwd: dw_orig = (g - <g, w_sn> u v^T) / sigma (u, v are constants of the forward).          */
int nc_spectral_norm_fwd(const float* w_orig, float* u, float* v, float* w_out, float* sigma, float* scratch_k, int K, int M,
                         int power_iteration, float eps, void* stream);
int nc_spectral_norm_bwd(const float* g, const float* w_sn, const float* u, const float* v, const float* sigma, float* dw_orig, int K,
                         int M, void* stream);

/* ---- Whole-network PatchGAN (NLayerDiscriminator with InstanceNorm, networks.py:1009-1067; called from
 *      apollo_model.py:195-283 through netD_*): forward and backward as ONE call each (the op-by-op path is
 *      host-enqueue-bound on these ~25-kernel chains).  params = the 2 * (n_layers + 2) tensors in state-dict order,
 *      packed; saved = activations / raw conv outputs / InstanceNorm statistics for the backward
 *      (nc_patchgan_saved_floats); nd = 2 (x is [B,1,H,W], pass D = 1) or 3.  bwd: dx and dparams may be NULL;
 *      dparams is overwritten (not accumulated).                                                                    */
size_t nc_patchgan_param_floats(int n_layers, int ndf, int nd);
size_t nc_patchgan_saved_floats(int B, int D, int H, int W, int n_layers, int ndf, int nd);
size_t nc_patchgan_ws_bytes(int B, int D, int H, int W, int n_layers, int ndf, int nd);
int nc_patchgan_out_shape(int B, int D, int H, int W, int n_layers, int ndf, int nd, int* oD, int* oH, int* oW);
int nc_patchgan_fwd(const float* params, const float* x, float* y, float* saved, int B, int D, int H, int W,
                    int n_layers, int ndf, int nd, void* ws, size_t ws_bytes, void* stream);
int nc_patchgan_bwd(const float* params, const float* x, const float* saved, const float* dy, float* dx, float* dparams,
                    int B, int D, int H, int W, int n_layers, int ndf, int nd, void* ws, size_t ws_bytes, void* stream);
/* Planes b0 .. b0 + B - 1 of a batch of Btot (x / y / dy / dx hold the B planes; saved and the workspace are the whole
 * batch's).  Athena sends the slices of `fake` through every discriminator twice with the same weights -- in the
 * generator loss (athena_model.py:240-260) and, detached, next to the slices of `real` in the discriminator loss
 * (:190-238, before optimizer_D.step()): the first pass is run as the second half of that batch (fwd_part + bwd_part, input
 * gradient only), the discriminator loss then only runs the `real` half and nc_patchgan_bwd over the whole batch. */
int nc_patchgan_fwd_part(const float* params, const float* x, float* y, float* saved, int Btot, int b0, int B, int D, int H,
                         int W, int n_layers, int ndf, int nd, void* ws, size_t ws_bytes, void* stream);
int nc_patchgan_bwd_part(const float* params, const float* x, const float* saved, const float* dy, float* dx, int Btot, int b0,
                         int B, int D, int H, int W, int n_layers, int ndf, int nd, void* ws, size_t ws_bytes, void* stream);

/* ---- Whole-network forward of Unet_deconv (networks.py:512-538; called from TestModel.forward test_model.py:60-62):
 *      params = the 28 tensors in state-dict order, packed back to back (see neuroclear_amd.models.networks).       */
size_t nc_unet_deconv_fwd_ws_bytes(int N, int S0, int S1, int S2);
int nc_unet_deconv_fwd(const float* params, const float* x, float* y, int N, int S0, int S1, int S2, void* ws,
                       size_t ws_bytes, void* stream);


/* ---- Whole-network TRAINING entry points of the generators (one C call per direction, like nc_patchgan_*):
 *      Unet_deconv.forward (networks.py:512-538) with everything its backward needs kept in `saved`, and the autograd
 *      backward of it (driven by loss_G.backward(), apollo_model.py:283); DeepLinearGenerator.forward (networks.py:
 *      913-917) and its backward.  params / dparams: the tensors in state-dict order, packed (28 for unet_deconv, 6 for
 *      deep_linear_gen); dparams is OVERWRITTEN.  dx may be NULL (the data gradient of the first layer is then skipped).
 *      Same kernels in the same order as the layer-by-layer path: bit-identical results.  The skip concats cost no copy
 *      (producers write into halves of the concat buffers) and the two gradients of a skip tensor are merged inside the
 *      max-pool backward (nc_maxpool2_bwd_add).  nc_deep_linear_fwd with saved == NULL is the inference form.          */
size_t nc_unet_deconv_param_floats(void);
size_t nc_unet_deconv_saved_floats(int N, int S0, int S1, int S2);
size_t nc_unet_deconv_train_ws_bytes(int N, int S0, int S1, int S2);
/* `kept` (HOST word, out, may be NULL): bit i set = the forward left the three-term (S3) copy of layer i's input in `saved`; hand the
 * same word to the backward that reads this `saved` buffer (0 is always valid: the backward then converts the inputs again).  The
 * word travels with the caller's record of the forward (the autograd context), there is no library-side table. */
int nc_unet_deconv_train_fwd(const float* params, const float* x, float* y, float* saved, int N, int S0, int S1, int S2,
                             void* ws, size_t ws_bytes, void* stream, unsigned* kept);
int nc_unet_deconv_bwd(const float* params, const float* x, const float* y, const float* saved, const float* dy, float* dx,
                       float* dparams, int N, int S0, int S1, int S2, void* ws, size_t ws_bytes, void* stream, unsigned kept);
size_t nc_deep_linear_param_floats(void);
size_t nc_deep_linear_saved_floats(int N, int S0, int S1, int S2);
size_t nc_deep_linear_ws_bytes(int N, int S0, int S1, int S2);
int nc_deep_linear_fwd(const float* params, const float* x, float* y, float* saved /* or NULL */, int N, int S0, int S1,
                       int S2, void* ws, size_t ws_bytes, void* stream, unsigned* kept /* as above */);
/* deep_linear_gen's layers 1 .. 5 (5^3, 3^3, then three 1 x 1; reference networks.py:900-911) are bias-free with nothing in between.  Level of
 * nc_set_dl_collapse (NC_DL_COLLAPSE at load time):
 *   0  layer by layer, as the reference's autograd does;
 *   1  (round 5) layers 2 .. 5 in collapsed form -- ONE 64 -> 1 convolution forward, and backward the six parameter gradients and dL/dact1 from
 *      dy, act1 and the weights alone (csrc/gen_nets.hip, "the collapsed tail") -- and the 5^3 layer's two gradients from 27 shifted copies of the
 *      one-channel dy, its forward as a 64 -> 27 convolution plus a shifted sum ("the forward without act1");
 *   2  (default, round 6) layers 1 .. 5 as ONE position-typed 7^3 convolution 64 -> 1 of act0 (csrc/dl_typed.hip): the zero padding of
 *      act1 only matters ON the faces, where it removes the taps of the 3^3 kernel that point outside -- 27 composed kernels, one per position
 *      type; the interior one runs everywhere on the two-term matrix kernels, the voxels on a face are recomputed / corrected with their own.  The
 *      5^3 convolution is not executed at all.  Where the shape does not admit it (an extent below 8, W % 4 != 0, nc_set_split_terms(3)): as 1.
 * Exact algebra in every level (weight-space products in fp64; tests/test_collapse_algebra.py in fp64 against autograd), the same outputs and
 * gradients to fp32 rounding, each level closer to an fp64 evaluation than the one before.  nc_deep_linear_lp_* follows the switch as 0 / non-zero.
 * Both pairs carry the forward's choice in `kept`, so the switch may move between a forward and its backward. */
void nc_set_dl_collapse(int on);
int nc_get_dl_collapse(void);
int nc_deep_linear_bwd(const float* params, const float* x, const float* saved, const float* dy, float* dx, float* dparams,
                       int N, int S0, int S1, int S2, void* ws, size_t ws_bytes, void* stream, unsigned kept);

/* ---- The 16-bit END-TO-END path (BASELINE.json configs[3]): operators between which activations and gradients exist in
 *      HBM only as 16-bit "C8" tensors [N][C/8][S][8] (nc_to_c8's layout).  A tensor argument (ptr, ctot, c0) means
 *      channels [c0, c0 + C) of a ctot-channel C8 buffer -- halves of the skip concat buffers (networks.py:526,531) are
 *      read / written in place.  Gradient tensors are bf16.  Statistics: fp32 per-thread partials, fp64 reduction.
 *      nc_conv_fwd_c8 / nc_conv_dgrad_c8: nc_conv_fwd_lp / nc_conv_dgrad_lp with a C8 result (rounded to dtype);
 *      nc_c8_instnorm_*: InstanceNorm3d(affine=False) + ReLU / LeakyReLU (networks.py:33-34,422-423) and its backward
 *      (g = gradient at the activation output, xh = raw convolution output; dbias nullable = sum of dx per channel);
 *      nc_c8_maxpool2_*: MaxPool3d(2) (networks.py:491,494), backward fused with the add of the skip gradient;
 *      nc_convT_k2s2_*_c8: ConvTranspose3d(k 2, s 2) (networks.py:500,503) on the 16-bit matrix cores (C, K % 32 == 0);
 *      nc_from_c8: C8 -> fp32 NCDHW.                                                                                   */
int nc_conv_fwd_c8(const void* xh, const float* w, const float* bias, void* yh, int out_ctot, int out_c0, int N, int C, int D, int H,
                   int W, int K, int kd, int kh, int kw, int stride, int pad, int dtype, void* ws, size_t ws_bytes, void* stream);
int nc_conv_dgrad_c8(const void* dyh, const float* w, void* dxh, int N, int C, int D, int H, int W, int K, int kd, int kh, int kw,
                     int stride, int pad, int dtype, void* ws, size_t ws_bytes, void* stream);
/* One-channel layers (Conv3d(1, 64, ks, padding ks/2), ks = 3: networks.py:420 first U-Net layer; ks = 7: :899 first
 * deep_linear layer) on the 16-bit cores in "pseudo-channel" form: the ks taps along x become 8 input channels
 * X8[v][j] = x[v + j - ks/2], the layer a ks x ks x 1 convolution of that C8 tensor.  fwd: x fp32 [N,1,D,H,W] -> C8 result
 * in channels [out_c0, out_c0 + 64); dgrad: dyh C8 [N][8][S][8] (bf16) -> dx fp32 [N,1,D,H,W].                        */
size_t nc_conv_c1_c8_ws_bytes(int N, int D, int H, int W, int ks);
int nc_conv_c1_fwd_c8(const float* x, const float* w, const float* bias, void* yh, int out_ctot, int out_c0, int N, int D, int H, int W,
                      int ks, int dtype, void* ws, size_t ws_bytes, void* stream);
int nc_conv_c1_dgrad_c8(const void* dyh, const float* w, float* dx, int N, int D, int H, int W, int ks, void* ws, size_t ws_bytes,
                        void* stream);
/* Weight gradient of the same layers from the C8 (bf16) gradient: dw fp32 [64,1,ks,ks,ks] = sum dyh * shifted x.  The K-dim of
 * the 16-bit MFMA is 16 consecutive voxels of an x row (planar 16-bit copies of dy and of the 8 x-shifted copies of x are built
 * in the workspace); W <= 192. */
size_t nc_conv_c1_wgrad_c8_ws_bytes(int N, int D, int H, int W, int ks);
int nc_conv_c1_wgrad_c8(const float* x, const void* dyh, float* dw, int N, int D, int H, int W, int ks, void* ws, size_t ws_bytes,
                        void* stream);
size_t nc_c8_instnorm_ws_bytes(int N, int C, long S);
int nc_c8_instnorm_stats(const void* xh, int N, int C, long S, float eps, float* mean, float* rstd, int dtype, void* ws,
                         size_t ws_bytes, void* stream);
int nc_c8_instnorm_act_fwd(const void* xh, const float* mean, const float* rstd, float slope, void* yh, int out_ctot, int out_c0,
                           int N, int C, long S, int dtype, void* stream);
int nc_c8_instnorm_act_bwd(const void* gh, int g_ctot, int g_c0, const void* xh, const float* mean, const float* rstd, float slope,
                           void* dxh, float* dbias /* or NULL */, int N, int C, long S, int dtype, void* ws, size_t ws_bytes,
                           void* stream);
int nc_c8_maxpool2_fwd(const void* xh, int x_ctot, int x_c0, void* yh, int N, int C, int D, int H, int W, int dtype, void* stream);
int nc_c8_maxpool2_bwd_add(const void* dph, const void* xh, int x_ctot, int x_c0, const void* skiph, int s_ctot, int s_c0, void* dxh,
                           int N, int C, int D, int H, int W, int dtype, void* stream);
int nc_from_c8(const void* xh, int x_ctot, int x_c0, float* y, int N, int C, long S, int dtype, void* stream);
size_t nc_convT_c8_ws_bytes(int N, int C, int D, int H, int W, int K);
int nc_convT_k2s2_fwd_c8(const void* xh, const float* w, const float* bias, void* yh, int out_ctot, int out_c0, int N, int C, int D,
                         int H, int W, int K, int dtype, void* ws, size_t ws_bytes, void* stream);
int nc_convT_k2s2_dgrad_c8(const void* dyh, int dy_ctot, int dy_c0, const float* w, void* dxh, int N, int C, int D, int H, int W, int K,
                           void* ws, size_t ws_bytes, void* stream);
int nc_convT_k2s2_wgrad_c8(const void* xh, const void* dyh, int dy_ctot, int dy_c0, float* dw, float* dbias /* or NULL */, int N,
                           int C, int D, int H, int W, int K, void* ws, size_t ws_bytes, void* stream);

/* ---- Whole-network training passes of the generators on that path (dtype = NC_DT_BF16): same contract as
 *      nc_unet_deconv_train_fwd / _bwd and nc_deep_linear_fwd / _bwd, `saved` in bytes.  The one-channel first layers
 *      (3^3 of the U-Net, 7^3 of deep_linear_gen), statistics, weight gradients and the pointwise heads' sums stay fp32;
 *      deep_linear_gen's bias-free linear tail 64 -> 32 -> 16 -> 1 (networks.py:903-911) is evaluated as the single
 *      64-vector W6 W5 W4 and its weight gradients as the rank-1 products they are.  *_supported: every layer's shape is
 *      covered by the 16-bit kernels (otherwise callers use the layer-by-layer 16-bit path).                         */
int nc_unet_deconv_lp_supported(int N, int S0, int S1, int S2, int dtype);
size_t nc_unet_deconv_lp_saved_bytes(int N, int S0, int S1, int S2);
size_t nc_unet_deconv_lp_ws_bytes(int N, int S0, int S1, int S2);
int nc_unet_deconv_lp_fwd(const float* params, const float* x, float* y, void* saved, int N, int S0, int S1, int S2, int dtype,
                          void* ws, size_t ws_bytes, void* stream);
int nc_unet_deconv_lp_bwd(const float* params, const float* x, const float* y, const void* saved, const float* dy, float* dx,
                          float* dparams, int N, int S0, int S1, int S2, int dtype, void* ws, size_t ws_bytes, void* stream);
int nc_deep_linear_lp_supported(int N, int S0, int S1, int S2, int dtype);
size_t nc_deep_linear_lp_saved_bytes(int N, int S0, int S1, int S2);
size_t nc_deep_linear_lp_ws_bytes(int N, int S0, int S1, int S2);
/* `kept` (round 6, as nc_deep_linear_fwd / _bwd): the form the forward took (layered / collapsed tail / without f2 -- it depends on
 * nc_set_dl_collapse at forward time and decides what `saved` holds); the caller hands it to the backward, which follows it whatever the
 * switches say by then (a value the forward cannot return: NC_ERR_ARG). */
int nc_deep_linear_lp_fwd(const float* params, const float* x, float* y, void* saved, int N, int S0, int S1, int S2, int dtype,
                          void* ws, size_t ws_bytes, void* stream, unsigned* kept);
int nc_deep_linear_lp_bwd(const float* params, const float* x, const void* saved, const float* dy, float* dx, float* dparams,
                          int N, int S0, int S1, int S2, int dtype, void* ws, size_t ws_bytes, void* stream, unsigned kept);

/* ---- fp32 3^3 / 5^3 convolutions on the 16-bit matrix cores (csrc/conv_split.hip): an fp32 value is exactly the sum of three
 *      bf16 terms; the six products a_i b_j with i + j <= 2 are bf16 MFMAs on one fp32 accumulator (what is dropped is below
 *      2^-23 of a product).  Same operands and results as nc_conv_fwd / nc_conv_dgrad (networks.py:420-425, 460-469) to
 *      fp32 rounding.  "S3" tensor: [N][C/8][3 terms][D][H][W][8] bf16 (nc_to_s3); xs / dys: the operand already in that
 *      form, or NULL (then converted into the workspace).  what: 0 forward, 1 data gradient, 2 weight gradient.                             */
void nc_set_conv_split(int on); /* 1 (default; or the value of NC_CONV_SPLIT at load time): nc_conv_fwd / nc_conv_dgrad and the
                                  * whole-network calls built on them take this path for the shapes it covers; 0: the fp32
                                  * MFMA kernels (v_mfma_f32_32x32x2_f32) serve those shapes */
int nc_get_conv_split(void);
/* The TWO-TERM form (round 4; csrc/conv_s3x.hip NT = 2, csrc/h2.hip, s3_common.hpp; the default): each operand as two fp16 terms of the tensor
 * times a power of two (chosen per tensor from its largest finite magnitude, or from a bound known by construction; results scaled back
 * exactly) and THREE fp16 MFMA products per fp32 product -- half the matrix work of the three-term bf16 form at the same error against fp64
 * (tests/test_gpu_h2.py).  The price is fp16's exponent range WITHIN one tensor: elements below 2^-17 of the tensor's largest magnitude keep
 * fewer than 22 bits, below 2^-39 they vanish (fp32: 2^-126) -- immaterial for normalised activations and their gradients, wrong for a tensor
 * that mixes magnitudes 1e12 apart; nc_set_split_terms(3) is for those.  terms = 2 (default; NC_SPLIT_TERMS at load time): two-term wherever
 * it exists -- nc_conv_fwd / _dgrad / _wgrad / _bwd of the covered shapes (channels % 64) and the whole-network training and inference calls
 * built on them; 3: three-term everywhere; 0: two-term in the inference forward nc_unet_deconv_fwd only.  The explicit S3 entry points below
 * (nc_to_s3, nc_conv_*_split) are three-term by definition.  In the whole-network calls InstanceNorm outputs are converted with the power of
 * two their bound sqrt(voxels) allows, the norm's backward with a bound from its own first pass, everything else with a measured one; the
 * ratio between the halves of a concatenation is folded into the consuming layer's weights.  Do not change the setting between a training
 * forward and its backward (the backward then re-converts what the forward kept). */
void nc_set_split_terms(int terms);
int nc_get_split_terms(void);
/* The PatchGAN's 4 x 4 layers at Athena's batches (csrc/conv_p2d.hip; reference networks.py:1030-1057): 1 (default; NC_P2D_TERMS at load time): the
 * stride-1 256 -> 512 layer's forward and data gradient on the two-term form whenever nc_get_split_terms() == 2 (a measured power of two per
 * call: results depend in the last bit on which planes share a call); 3: three-term everywhere (bit-identical however the planes are batched);
 * 2: two-term for the stride-2 layers too (slower: their conversion dominates). */
void nc_set_p2d_terms(int mode);
int nc_get_p2d_terms(void);
/* The RANGE GUARD of the two-term form (round 5; csrc/h2.hip, csrc/common.hpp): wherever a call converts an fp32 operand ITSELF with a measured
 * power of two -- nc_conv_fwd / _dgrad / _wgrad / _bwd and every dY of the whole-network backward calls that arrives as fp32 -- the conversion
 * pass also counts the CHUNKS (64 voxels x 8 channels) whose largest magnitude lies below 2^-17 of the tensor's; when more than 1/64 of the
 * non-zero chunks do (a region of the volume or a block of channels far below the rest: the one case the two-term form degrades, see above),
 * THAT CALL runs on the exact three-term kernels instead.  The decision is taken on the device (no host synchronisation): both kernel
 * families are launched and the one whose turn it is not leaves at its first instruction.  Modes (nc_set_h2_guard; NC_H2_GUARD at load
 * time): 0 off (round-4 behaviour); 1 (default): the in-call fallback in the per-layer entry points (nc_conv_fwd / _dgrad / _wgrad / _bwd);
 * INSIDE the whole-network calls (nc_unet_deconv_*, nc_deep_linear_*) every data-derived power of two is checked the same way -- fp32 dY
 * tensors, the dY tensors the InstanceNorm backward writes in two-term form itself (power of two from the tensor's own per-instance maxima),
 * the activations deep_linear_gen keeps -- but a flagged tensor is only COUNTED (the fallback would cost the training step ~65 near-empty
 * launches, 0.5 ms of 35, for an event InstanceNorm networks do not produce); 2: in-call fallback everywhere it exists (everything but the
 * kept activations of deep_linear_gen).  Operands whose power of two is a bound by construction (InstanceNorm
 * outputs, ConvTranspose outputs) and WEIGHTS (one measured power of two per weight tensor: assumed well-conditioned, as any initialisation
 * and training of this path leaves them) are not guarded.  nc_h2_guard_stats(out, reset): out[0] = tensors measured, out[1] = calls that fell
 * back to three terms, out[2] = flagged tensors that could not switch (the Python models go to nc_set_split_terms(3) when they see one,
 * models/base_model.py), out[3] = the largest share of low chunks seen, in parts per million; host-visible counters, readable at any time
 * without synchronising (they trail the device by whatever is still queued). */
void nc_set_h2_guard(int on);
int nc_get_h2_guard(void);
int nc_h2_guard_stats(unsigned long long* out4, int reset);
/* Epilogue statistics of the inference forward (round 5; csrc/conv_s3x.hip, template parameter ST): in nc_unet_deconv_fwd's two-term mode every
 * 3^3 convolution leaves, per (tile, wave), the sum and the sum of squares of its bias-free outputs, and a small pass adds them up in fp64 in a
 * fixed order -- InstanceNorm (reference networks.py:513-515, nn.InstanceNorm3d after every Conv3d) needs no pass of its own over the raw
 * output.  mean / rstd agree with the separate pass to 4e-8 relative.  1 (default; NC_EPI_STATS at load time); 0: the separate pass; 2: also in
 * nc_unet_deconv_train_fwd's blocks whose input arrives converted (tested; no measurable gain in the training step, hence not the default). */
void nc_set_epi_stats(int on);
int nc_get_epi_stats(void);
/* Which kernel serves the whole 512-position tiles of a two-term 3^3 launch (forward and data gradient of every 3^3 layer of the U-Net,
 * reference networks.py:496-561, whose input channels are a multiple of 64; csrc/conv_s3x.hip): 1 (default; NC_S3X_W64 at load time) = k_conv_s3w,
 * the 64-channel x 64-position wave tile with the weights staged through LDS (round 6; -4 % on the 140^3 / 108^3 layers, profiles/r06_ab_w64.txt);
 * 0 = k_conv_s3x everywhere.  Same tiles, same records of the epilogue statistics; results differ by the order of the fp32 sums only (one running
 * accumulator per tile instead of restarts every four k-steps). */
void nc_set_s3x_w64(int on);
int nc_get_s3x_w64(void);
int nc_unet_deconv_fwd_terms(int S0, int S1, int S2); /* 2: nc_unet_deconv_fwd runs its 3^3 layers on the two-term form at this size under the
                                                       * current switches; 3: on the three-term form (or the fp32 kernels); 0: bad size */
void nc_set_c8x_mode(int mode); /* which kernel serves the 16-bit 3^3 / 5^3 forward / data-gradient calls (nc_conv_fwd_lp, nc_conv_*_c8, the
                                  * *_lp whole-network calls; csrc/conv_c8x.hip): 1 (default; NC_C8X at load time) = for 3^3 layers the tap-stream kernel
                                  * k_conv_c8x where its 512-position tiles fill the launch's rounds of 512 workgroups to >= 60 %, k_conv_h elsewhere (a few planes); 2 = k_conv_c8x
                                  * wherever the shape fits (channels read and written % 64 == 0); 0 = k_conv_h everywhere.  Both kernels multiply
                                  * the same 16-bit operands and accumulate in fp32; only the summation order differs */
int nc_get_c8x_mode(void);
int nc_conv_lp_uses_c8x(int what /* 0 forward, 1 data gradient */, int fp32_out, int N, int C, int D, int H, int W, int K, int ks); /* 1: under
                                  * the current mode this call runs on k_conv_c8x (layer C -> K channels; fp32_out: nc_conv_*_lp, else the C8 forms) */
void nc_set_s3_fusion(int on); /* 1 (default; NC_S3_FUSE): nc_unet_deconv_fwd has InstanceNorm + ReLU write the three-term form of a
                                * layer that only feeds a split-operand convolution; 0: separate conversion passes (bit-identical) */
int nc_conv_split_supported(int what, int N, int C, int D, int H, int W, int K, int kd, int kh, int kw, int stride, int pad);
size_t nc_conv_split_ws_bytes(int N, int C, int D, int H, int W, int K, int ks);
size_t nc_s3_bytes(int N, int C, long S);
int nc_to_s3(const float* x, void* xs, int N, int C, long S, void* stream);
int nc_conv_fwd_split(const float* x, const void* xs, const float* w, const float* bias, float* y, int N, int C, int D, int H, int W,
                      int K, int ks /* 3 or 5: cubic kernel, stride 1, padding ks/2 */, void* ws, size_t ws_bytes, void* stream);
int nc_conv_dgrad_split(const float* dy, const void* dys, const float* w, float* dx, int N, int C, int D, int H, int W, int K, int ks,
                        void* ws, size_t ws_bytes, void* stream);
int nc_conv_wgrad_split(const float* x, const void* xs, const float* dy, const void* dys, float* dw, int N, int C, int D, int H, int W,
                        int K, int ks, void* ws, size_t ws_bytes, void* stream); /* nc_conv_wgrad (weights only) */

/* ConvTranspose3d(kernel 2, stride 2) forward -- nn.ConvTranspose3d at networks.py:471-478 -- on the same arithmetic (csrc/convt_s3.hip):
 * C % 32 == 0 (<= 256), K % 16 == 0.  xs: the input in S3 form, or NULL (x is converted into the workspace); y (nullable): fp32 output
 * [N][K][2D][2H][2W]; ys (nullable): channels [ys_c0, ys_c0 + K) of a ys_ctot-channel S3 tensor of the output volume.  At least one of y / ys. */
/* Conv2d 4 x 4, padding 1, stride 1 or 2 of the PatchGAN (networks.py:1037-1055: the 64 -> 128, 128 -> 256 and 256 -> 512 layers) at batches of
 * >= 8192 output pixels on the same arithmetic (csrc/conv_p2d.hip; forward and data gradient; channels % 64 == 0 on the written side, % 16 /
 * % 64 on the read side) and the stride-1 layer's weight gradient (csrc/wgrad_p2d.hip; C % 32 == 0, K % 64 == 0): 1 when nc_conv_fwd (what 0) /
 * nc_conv_dgrad (what 1) / nc_conv_wgrad (what 2) -- and with them nc_patchgan_fwd / _bwd -- take it for this call under the current switches
 * (nc_set_conv_split; NC_P2D at load time: bit 0 the stride-1 layer forward / data gradient, bit 1 the stride-2 layers, bit 2 the stride-1 weight
 * gradient, 0 = never).  Smaller batches, the one-channel first layer and head, and the stride-2 weight gradients run on the kernels of
 * conv2d_img.hip / patchgan_edge.hip / the gather GEMM. */
int nc_conv2d_split_active(int what, int N, int C, int H, int W, int K, int k, int stride, int pad);

int nc_convT_k2s2_split_supported(int N, int C, int D, int H, int W, int K);
int nc_convT_k2s2_split_active(int N, int C, int D, int H, int W, int K); /* supported AND nc_get_conv_split(): what the whole-network calls do */
size_t nc_convT_k2s2_split_ws_bytes(int N, int C, int D, int H, int W, int K);
int nc_convT_k2s2_fwd_split(const float* x, const void* xs, const float* w, const float* bias, float* y, void* ys, int ys_ctot, int ys_c0,
                            int N, int C, int D, int H, int W, int K, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NC_HIP_H */
