/*
 * stereo_hip.h -- C ABI of libstereo_hip.so: the MI355X (gfx950) stereo
 * cost-volume + disparity-regression hot path.
 *
 * The reference (Magicboomliu/StereoFormer) has no FFI for this path: every
 * function below replaces a plain PyTorch nn.Module.forward / function, cited
 * as file:line relative to the reference root.  The binding a maintainer
 * would add on the reference side is in INTEGRATION.md.
 *
 * Conventions
 *   - all tensors fp32, contiguous NCHW, DEVICE pointers owned by the caller (the `_f16` entry points at the end take
 *     their FEATURE operands in fp16; outputs are fp32 everywhere);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*;
 *     NULL = the default stream); no internal sync, no allocation, no state
 *     carried between calls (the only process-wide data are the test hooks of
 *     stereo_test_option and a once-per-device opt-in flag for large LDS
 *     tiles, both atomics): safe to capture into a hipGraph, re-entrant and
 *     callable from several host threads;
 *   - return value: STEREO_OK or a negative STEREO_ERR_* code; on error no
 *     kernel has been launched;
 *   - zero-sized problems (any extent == 0) return STEREO_OK and launch nothing.
 */
#ifndef STEREO_HIP_H
#define STEREO_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define STEREO_OK 0
#define STEREO_ERR_NULL_POINTER (-1)
#define STEREO_ERR_BAD_SHAPE (-2)   /* negative extent, C % G != 0, D > limit ... */
#define STEREO_ERR_UNSUPPORTED (-3) /* size outside what the kernels index safely */
#define STEREO_ERR_LAUNCH (-4)      /* hipGetLastError() after the launch */

#define STEREO_MASK_ZERO 0   /* out-of-image band logits are 0.0 (A1+A4 semantics) */
#define STEREO_MASK_NEGINF 1 /* out-of-image band entries are excluded */
#define STEREO_SCALE_SQRT_C 0 /* 1/sqrt(C): corr.py:111 */
#define STEREO_SCALE_MEAN 1   /* 1/C: build_cost_volume.py:44-45 */

#define STEREO_MAX_LEVELS 8

int stereo_abi_version(void);
const char *stereo_strerror(int code);

/* TEST HOOK, not part of the product contract: steers which kernel serves the following calls so that the test-suite
 * can run a fallback kernel on a shape the fast path would take ("corr_mfma" 0/1, "gwc_fused" 0/1, "att_kernel" 0/2/4/5/6/7,
 * "att_waves" 0/4/5/8, "prob_single_pass" 0/1) or make small inputs walk the persistent multi-tile loops
 * ("tiles_per_wg" n).  Process-wide atomics; value < 0 restores the default.  The library reads no environment
 * variable.  Returns STEREO_ERR_BAD_SHAPE for an unknown name. */
int stereo_test_option(const char *name, int value);

/* A1  CostVolume.forward 'correlation'  models/utils/build_cost_volume.py:39-47
 *     left,right [B,C,H,W] -> vol [B,D,H,W];  vol[b,d,h,w] = mean_c L*R(w-d), 0 for w<d */
int stereo_corr_volume_f32(const float *left, const float *right, float *vol,
                           int B, int C, int H, int W, int D, void *stream);

/* A2  'difference' (:21-28) -> [B,C,D,H,W];  'concat' (:30-37) -> [B,2C,D,H,W] */
int stereo_diff_volume_f32(const float *left, const float *right, float *vol,
                           int B, int C, int H, int W, int D, void *stream);
int stereo_concat_volume_f32(const float *left, const float *right, float *vol,
                             int B, int C, int H, int W, int D, void *stream);

/* A3  build_gwc_volume  models/CrossAttentionCostVolume/GWC_Cost_Volume.py:14-24
 *     -> vol [B,G,D,H,W]; STEREO_ERR_BAD_SHAPE when C % G != 0 (the reference asserts, :7) */
int stereo_gwc_volume_f32(const float *left, const float *right, float *vol,
                          int B, int C, int H, int W, int D, int G, void *stream);

/* A4  DisparityEstimation(.WithProb).forward  models/utils/estimation.py:14-33,44-62
 *     vol [B,D,H,W] -> disp [B,H,W] (+ prob [B,D,H,W] when prob != NULL) */
int stereo_soft_argmin_f32(const float *vol, float *disp, float *prob,
                           int B, int D, int H, int W, int match_similarity, void *stream);

/* disparity_regression  models/utils/submodule.py:23-27:  x [B,D,H,W] -> out[b,h,w] = sum_d x*d */
int stereo_disparity_regression_f32(const float *x, float *out, int B, int D, int H, int W, void *stream);

/* cfg-2 chain reader: volg [B,G,D,H,W] -> group mean (torch .mean(1)) -> A4.
 * vol_mean [B,D,H,W] is written when non-NULL. */
int stereo_group_mean_soft_argmin_f32(const float *volg, float *vol_mean, float *disp,
                                      int B, int G, int D, int H, int W,
                                      int match_similarity, void *stream);

/* A1+A4 fused, no volume materialised: disp = soft_argmin(corr_volume(L,R,D)) */
int stereo_corr_soft_argmin_f32(const float *left, const float *right, float *disp,
                                int B, int C, int H, int W, int D, int match_similarity,
                                void *stream);

/* A1 + A4 in one pass: vol [B,D,H,W] = correlation volume, disp [B,H,W] = its soft-argmin
 *     (build_cost_volume.py:18-52 + estimation.py:14-34); the volume is not read back. */
int stereo_corr_volume_soft_argmin_f32(const float *left, const float *right, float *vol,
                                       float *disp, int B, int C, int H, int W, int D,
                                       int match_similarity, void *stream);

/* A3 + group mean + A4 in one call: vol [B,G,D,H,W] = build_gwc_volume (GWC_Cost_Volume.py:14-24),
 *     mean [B,D,H,W] = vol.mean(1) (may be NULL), disp [B,H,W] = soft-argmin over D of mean
 *     (estimation.py:14-34).  With equal group sizes vol.mean(1) equals the all-channel correlation
 *     volume, so mean/disp come from the feature tiles and the 4*B*G*D*H*W-byte volume is written
 *     once and never read back. */
int stereo_gwc_volume_soft_argmin_f32(const float *left, const float *right, float *vol,
                                      float *mean, float *disp, int B, int C, int H, int W, int D,
                                      int G, int match_similarity, void *stream);

/* A5  build_cost_volume_from_volume  .../Attempts/fixed_local_cost_volume.py:61-80
 *     vol [B,D,H,W], cand [B,K,H,W] -> out [B,K,H,W] */
int stereo_sample_volume_f32(const float *vol, const float *cand, float *out,
                             int B, int D, int H, int W, int K, void *stream);

/* A5  LocalCostVolume.forward  fixed_local_cost_volume.py:18-58
 *     vol [B,D,H,W], disp [B,1,H,W] -> out [B,1,H,W]; S = sample_points (S+1 candidates) */
int stereo_local_lookup_f32(const float *vol, const float *disp, float *out,
                            int B, int D, int H, int W, float radius, int S,
                            int consider_valid, void *stream);

/* A6  Variance_Based_Local_Cost_Volume.forward  .../Attempts/variance_local_cost_volume.py:34-78 */
int stereo_variance_lookup_f32(const float *vol, const float *disp, float *out,
                               int B, int D, int H, int W, float gamma, int S,
                               int consider_valid, void *stream);

/* A11 sampler with learned offsets  .../Attempts/update.py:279-313
 *     bounds [B,2,H,W] (channel 0 = lower offset, 1 = upper offset; torch.chunk order) */
int stereo_bounds_lookup_f32(const float *vol, const float *disp, const float *bounds, float *out,
                             int B, int D, int H, int W, int S, int consider_valid, void *stream);

/* A11 uncertainty volume  .../Attempts/update.py:71-72,86:  softmax_d(vol)*d*(d-disp)^2 */
int stereo_uncertainty_volume_f32(const float *vol, const float *disp, float *out,
                                  int B, int D, int H, int W, void *stream);

/* A11 the inputs of BasicGuidanceNet's two encoders (.../Attempts/update.py:75-90) in one pass, CHANNELS-LAST and zero-padded
 * to the channel counts stereo_conv_nhwc_* takes: err [B][H][W][err_channels] = disp_warp(right, disp) - left (border padding,
 * utils/disparity_warper.py:83-106) in its first Ci channels, unc [B][H][W][unc_channels] = softmax_d(vol)*d*(d-disp)^2 in
 * its first D channels, 0 elsewhere.  left / right [B,Ci,H,W] (already at the disparity's resolution), vol [B,D,H,W]. */
int stereo_guidance_inputs_nhwc_f32(const float *vol, const float *disp, const float *left, const float *right, float *err,
                                    float *unc, int B, int D, int Ci, int H, int W, int err_channels, int unc_channels,
                                    void *stream);

/* A8  CorrBlock1D.__init__ + .corr  models/IterativeLocalCostVolume/corr/corr.py:55-76,103-111
 *     fmap1 [B,C,H,W1], fmap2 [B,C,H,W2] -> levels[i] [B*H*W1, W2>>i], i = 0..num_levels
 *     (levels = HOST array of num_levels+1 device pointers; the reference builds
 *     num_levels+1 pyramid entries).  fp32-faithful on split fp16 matrix-core operands on the aligned path (W1, W2 % 4 == 0):
 *     |feature| < 4 094 there (inf / NaN beyond, never silently wrong). */
int stereo_allpairs_pyramid_f32(const float *fmap1, const float *fmap2, float *const *levels,
                                int B, int C, int H, int W1, int W2, int num_levels, void *stream);

/* A9  CorrBlock1D.__call__ + bilinear_sampler  corr.py:78-101, .../utils/utils.py:59-76
 *     coords [B,2,H,W1] (x = channel 0) -> out [B, num_levels*(2r+1), H, W1] */
int stereo_corr_lookup_f32(const float *const *levels, const float *coords, float *out,
                           int B, int H, int W1, int W2, int num_levels, int radius, void *stream);

/* The same, written CHANNELS-LAST for the motion encoder's first convolution (update/update.py:77,85): out [B][H][W1][out_channels],
 * out_channels >= num_levels*(2r+1) (a multiple of 16 for stereo_conv_nhwc_*), the padding channels zeroed. */
int stereo_corr_lookup_nhwc_f32(const float *const *levels, const float *coords, float *out,
                                int B, int H, int W1, int W2, int num_levels, int radius, int out_channels, void *stream);

/* A10 PytorchAlternateCorrBlock1D.__call__  corr.py:8-51 (no all-pairs volume).
 *     workspace: (B*C*H*(W/2 + W/4 + ...)) floats for the pooled fmap2 copies */
int stereo_alt_corr_lookup_f32(const float *fmap1, const float *fmap2, const float *coords,
                               float *out, float *workspace,
                               int B, int C, int H, int W, int num_levels, int radius, void *stream);
long long stereo_alt_corr_workspace_floats(int B, int C, int H, int W, int num_levels);

/* bilinear_sampler(img, coords, mask)  .../utils/utils.py:59-76 (general 2-D form)
 *     img [N,C,H,W], coords [N,Ho,Wo,2] pixel units -> out [N,C,Ho,Wo], mask [N,Ho,Wo,1] or NULL */
int stereo_bilinear_sampler_f32(const float *img, const float *coords, float *out, float *mask,
                                int N, int C, int H, int W, int Ho, int Wo, void *stream);

/* A12 disp_warp  utils/disparity_warper.py:83-106;  img [B,Ci,H,W], disp [B,1,H,W]
 *     -> warped [B,Ci,H,W], mask [B,Ci,H,W] (or NULL); padding_border: 1 'border', 0 'zeros' */
int stereo_disp_warp_f32(const float *img, const float *disp, float *warped, float *mask,
                         int B, int Ci, int H, int W, int padding_border, void *stream);

/* A13 upsample_convex8  models/utils/upsample.py:22-38 (f=8) and RAFT_Stereo.upsample_flow
 *     models/IterativeLocalCostVolume/raft_stereo.py:110-122 (f = 2^downsample)
 *     disp [N,Cd,H,W], mask [N,9*f*f,H,W] -> out [N,Cd,f*H,f*W] */
int stereo_upsample_convex_f32(const float *disp, const float *mask, float *out,
                               int N, int Cd, int H, int W, int f, void *stream);

/* The same with the mask CHANNELS-LAST, mask [N][H*W][9*f*f] (the memory of a torch.channels_last [N,9*f*f,H,W] tensor: what
 * the update blocks' mask heads produce on the channels-last convolution kernels); f = 8 or 4, else STEREO_ERR_UNSUPPORTED. */
int stereo_upsample_convex_nhwc_mask_f32(const float *disp, const float *mask, float *out,
                                         int N, int Cd, int H, int W, int f, void *stream);

/* F.interpolate(bilinear) to (Ho,Wo): upsample_simple8 / upflow8 / interp / the 1/8 image
 * resize (upsample.py:41-43, utils.py:85-87, update/update.py:125-127, Attempts/update.py:73-77);
 * out = scale * interp(in) */
int stereo_resize_bilinear_f32(const float *in, float *out, int NC, int Hi, int Wi, int Ho, int Wo,
                               int align_corners, float scale, void *stream);
/* The same on channels-last tensors: in [N][Hi][Wi][C] -> out [N][Ho][Wo][C], C % 4 == 0 (the update block's `interp` when the
 * block runs in torch.channels_last, update.py:125-127). */
int stereo_resize_bilinear_nhwc_f32(const float *in, float *out, int N, int C, int Hi, int Wi, int Ho, int Wo,
                                    int align_corners, float scale, void *stream);

/* A7  epipolar cross-attention (build-defined; reference file is empty:
 *     models/CrossAttentionCostVolume/cross_attention_cost_volume.py)
 *     left,right [B,C,H,W], value [B,Cv,H,W] -> out [B,Cv,H,W] (or NULL), disp [B,H,W] (or NULL)
 *     Numerics: fp32-faithful (split operands on the fp16 / bf16 matrix cores, fp32 accumulation).  Feature range of the
 *     default kernels: |x| < 4 094 (fp32 features; 65 504 for fp16 ones) at C <= 32 with value == right and at C > 32
 *     without a value tensor (bands <= 128), fp32's whole range otherwise; beyond it inf / NaN, never silently wrong. */
int stereo_cross_attention_f32(const float *left, const float *right, const float *value,
                               float *out, float *disp, int B, int C, int Cv, int H, int W, int D,
                               int mask_mode, int scale_mode, void *stream);

/* A11/A14 ConvGRU gate math around the (host-framework) convolutions
 *     Attempts/update.py:41-48, update/update.py:28-32; bias pointers may be NULL
 *     reset: out = sigmoid(rc + cr) * h;  blend: out = (1-z)*h + z*act(qc+cq), z = sigmoid(zc+cz),
 *     act = tanh, or x/(1+|x|) when softsign != 0 (ConvGRU(softsign=True), Attempts/update.py:31-34) */
int stereo_gru_reset_f32(const float *h, const float *rc, const float *cr, float *out,
                         long long n, void *stream);
int stereo_gru_blend_f32(const float *h, const float *zc, const float *qc, const float *cz,
                         const float *cq, float *out, long long n, int softsign, void *stream);

/* The same two gates with fp16 tensors in and out (gate math in fp32 registers): the update block under the
 * reference's optional mixed_precision flag (raft_stereo.py:176) hands over fp16 convolution outputs. */
int stereo_gru_reset_f16(const void *h, const void *rc, const void *cr, void *out, long long n,
                         void *stream);
int stereo_gru_blend_f16(const void *h, const void *zc, const void *qc, const void *cz,
                         const void *cq, void *out, long long n, int softsign, void *stream);

/* ---- backward: the cfg-2 chain is differentiable -------------------------------- */

/* backward of A4 (estimation.py:14-33,44-62): grad_disp [B,H,W], optional grad_prob [B,D,H,W] (the WithProb
 * output's cotangent, may be NULL) -> grad_vol [B,D,H,W] */
int stereo_soft_argmin_bwd_f32(const float *vol, const float *grad_disp, const float *grad_prob,
                               float *grad_vol, int B, int D, int H, int W, int match_similarity, void *stream);

/* the same, with a second cotangent of the SAME volume added on the way out: grad_vol = addend + d disp / d vol * grad_disp
 * (the volume of estimation.py:14-33 also feeds fixed_local_cost_volume.py:18-80 in the reference's chain, so its gradient
 * is a sum of two terms; autograd would form it with an extra read-read-write pass).  addend, grad_vol [B,D,H,W] (may alias:
 * every element is read before it is written by the same lane).  HW % 4 == 0, D <= 192, else STEREO_ERR_UNSUPPORTED and
 * the caller adds the cotangents itself. */
int stereo_soft_argmin_bwd_add_f32(const float *vol, const float *grad_disp, const float *addend, float *grad_vol,
                                   int B, int D, int H, int W, int match_similarity, void *stream);

/* backward of A1/A3 (build_cost_volume.py:39-47, GWC_Cost_Volume.py:14-24): grad_vol [B,G,D,H,W] ->
 * grad_left, grad_right [B,C,H,W] (either may be NULL); G = 1 for the correlation volume.  With both gradients, C / G <= 32,
 * D <= 192 and W % 4 == 0 the volume's cotangent is read ONCE (split-fp16 matrix cores; gradients of any magnitude, features
 * |x| < 4 094 -- inf / NaN beyond, never silently wrong); otherwise two launches on the f32 matrix cores / vector units. */
int stereo_gwc_volume_bwd_f32(const float *left, const float *right, const float *grad_vol,
                              float *grad_left, float *grad_right, int B, int C, int H, int W, int D, int G,
                              void *stream);

/* backward of A5 (fixed_local_cost_volume.py:18-80, STE semantics :5-9): grad_out [B,1,H,W] ->
 * grad_vol [B,D,H,W] (MUST be zero-filled by the caller; taps are accumulated into it) and grad_disp [B,1,H,W] */
int stereo_local_lookup_bwd_f32(const float *vol, const float *disp, const float *grad_out,
                                float *grad_vol_zeroed, float *grad_disp, int B, int D, int H, int W,
                                float radius, int S, int consider_valid, void *stream);
/* The same, writing grad_vol DENSELY (every element, zeros included) in the one launch: no zero-fill by the caller, no
 * read-modify-write.  2 * radius + 3 <= 16 and H*W % 4 == 0, else STEREO_ERR_UNSUPPORTED (use the entry above). */
int stereo_local_lookup_bwd_dense_f32(const float *vol, const float *disp, const float *grad_out, float *grad_vol,
                                      float *grad_disp, int B, int D, int H, int W, float radius, int S, int consider_valid,
                                      void *stream);

/* backward of A13 (upsample.py:22-38, raft_stereo.py:110-122), f in {4, 8}: grad_out [N,Cd,fH,fW] ->
 * grad_disp [N,Cd,H,W] (MUST be zero-filled by the caller) and grad_mask [N,9*f*f,H,W] */
int stereo_upsample_convex_bwd_f32(const float *disp, const float *mask, const float *grad_out,
                                   float *grad_disp_zeroed, float *grad_mask, int N, int Cd, int H, int W,
                                   int f, void *stream);

/* backward of the A11 / A14 pointwise pieces (Attempts/update.py:41-48,71-86,279-313; update/update.py:28-32;
 * utils/disparity_warper.py:83-106).  The bias cotangents equal the pre-activation cotangents. */
int stereo_gru_reset_bwd_f32(const float *h, const float *rc, const float *cr, const float *grad_out,
                             float *grad_h, float *grad_rc, long long n, void *stream);
int stereo_gru_blend_bwd_f32(const float *h, const float *zc, const float *qc, const float *cz, const float *cq,
                             const float *grad_out, float *grad_h, float *grad_zc, float *grad_qc, long long n,
                             int softsign, void *stream);
int stereo_uncertainty_volume_bwd_f32(const float *vol, const float *disp, const float *grad_u,
                                      float *grad_vol, float *grad_disp, int B, int D, int H, int W, void *stream);
/* gradient of the warped image w.r.t. the disparity only (the image is an input of the model) */
int stereo_disp_warp_bwd_disp_f32(const float *img, const float *disp, const float *grad_warped,
                                  float *grad_disp, int B, int Ci, int H, int W, int padding_border, void *stream);
/* grad_vol must be zero-filled; grad_bounds [B,2,H,W] */
int stereo_bounds_lookup_bwd_f32(const float *vol, const float *disp, const float *bounds, const float *grad_out,
                                 float *grad_vol_zeroed, float *grad_disp, float *grad_bounds,
                                 int B, int D, int H, int W, int S, int consider_valid, void *stream);

/* backward of A9 (corr.py:78-101): grad_out [B, L*(2r+1), H, W1] scattered onto the two taps of each level;
 * grad_levels_zeroed[i] [B*H*W1, W2>>i] must be zero-filled.  No gradient w.r.t. coords (detached in the
 * reference, raft_stereo.py:171). */
int stereo_corr_lookup_bwd_f32(float *const *grad_levels_zeroed, const float *coords, const float *grad_out,
                               int B, int H, int W1, int W2, int num_levels, int radius, void *stream);

/* backward of A8 (corr.py:55-76,103-111): the level cotangents grad_levels[0..num_levels] are folded coarse to
 * fine INTO grad_levels[0] (modified), then grad_fmap1 = G0 F2^T / sqrt(C), grad_fmap2 = G0^T F1 / sqrt(C)
 * on the f32 matrix cores; either output may be NULL */
int stereo_allpairs_pyramid_bwd_f32(const float *fmap1, const float *fmap2, float *const *grad_levels,
                                    float *grad_fmap1, float *grad_fmap2, int B, int C, int H, int W1, int W2,
                                    int num_levels, void *stream);

/* backward of the bilinear resize; grad_in must be zero-filled */
int stereo_resize_bilinear_bwd_f32(const float *grad_out, float *grad_in_zeroed, int NC, int Hi, int Wi, int Ho, int Wo,
                                   int align_corners, float scale, void *stream);

/* N3 completed: backward of A6 (variance_local_cost_volume.py:34-78 as autograd differentiates it; grad_vol must be
 * zero-filled by the caller) and of A2 ('difference' / 'concat', build_cost_volume.py:21-37; either output may be NULL). */
int stereo_variance_lookup_bwd_f32(const float *vol, const float *disp, const float *grad_out, float *grad_vol_zeroed,
                                   float *grad_disp, int B, int D, int H, int W, float gamma, int S, int consider_valid,
                                   void *stream);
/* The same in one dense launch: grad_vol need NOT be zero-filled (every element is written once: the variance path makes the
 * gradient dense in d).  D <= 192, H*W % 4 == 0, sample_points <= 20, 16-byte aligned tensors, else STEREO_ERR_UNSUPPORTED
 * (the caller zero-fills and takes the entry above). */
int stereo_variance_lookup_bwd_dense_f32(const float *vol, const float *disp, const float *grad_out, float *grad_vol,
                                         float *grad_disp, int B, int D, int H, int W, float gamma, int sample_points,
                                         int consider_valid, void *stream);
int stereo_diff_volume_bwd_f32(const float *grad_vol, float *grad_left, float *grad_right,
                               int B, int C, int H, int W, int D, void *stream);
int stereo_concat_volume_bwd_f32(const float *grad_vol, float *grad_left, float *grad_right,
                                 int B, int C, int H, int W, int D, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * N2: feature hand-over in fp16 + the extractor's fused norm epilogue.
 * Reference sites: the fnet output under autocast is fp16 and is widened by `.float()` before the correlation
 * block (models/IterativeLocalCostVolume/raft_stereo.py:155-156); InstanceNorm2d + ReLU close the extractor stages
 * (models/IterativeLocalCostVolume/extractor/extractor.py:135,160-162,187-188); LowCNN hands [B,C,H/8,W/8]
 * features to CostVolume (models/LocalCostVolume/baseline.py:89-114).
 *
 * `_f16` entry points: feature operands are fp16 [B,C,H,W] (const void *), every output is fp32.  The matrix-core
 * kernels widen the features on the way into LDS: same fp32 arithmetic on the fp16-rounded values, half the read
 * bytes.  They cover the matrix-core envelope only (W % 4 == 0, 8-byte aligned bases; volumes: C <= 32, D <= 256
 * (192 for the group-wise forms), channels per group in {2,4,8,16}; attention: D <= 192 for C > 32) and return
 * STEREO_ERR_UNSUPPORTED outside it -- there is no slow fp16 path; the caller widens and uses the fp32 entry point. */
int stereo_corr_volume_f16(const void *left, const void *right, float *vol,
                           int B, int C, int H, int W, int D, void *stream);
int stereo_gwc_volume_f16(const void *left, const void *right, float *vol,
                          int B, int C, int H, int W, int D, int G, void *stream);
int stereo_corr_soft_argmin_f16(const void *left, const void *right, float *disp,
                                int B, int C, int H, int W, int D, int match_similarity, void *stream);
int stereo_corr_volume_soft_argmin_f16(const void *left, const void *right, float *vol, float *disp,
                                       int B, int C, int H, int W, int D, int match_similarity, void *stream);
int stereo_gwc_volume_soft_argmin_f16(const void *left, const void *right, float *vol, float *mean, float *disp,
                                      int B, int C, int H, int W, int D, int G, int match_similarity, void *stream);
int stereo_cross_attention_f16(const void *left, const void *right, const void *value, float *out, float *disp,
                               int B, int C, int Cv, int H, int W, int D, int mask_mode, int scale_mode, void *stream);
int stereo_allpairs_pyramid_f16(const void *fmap1, const void *fmap2, float *const *levels,
                                int B, int C, int H, int W1, int W2, int num_levels, void *stream);

/* nn.InstanceNorm2d(C, eps, affine) [+ ReLU] in one call (extractor.py:135 + :187-188): x [B,C,H,W] fp32 ->
 * out [B,C,H,W] fp32 (out_is_f16 = 0) or fp16 (1), the layout the `_f16` loaders above take.  gamma / beta [C] or NULL
 * (affine=False, the nn.InstanceNorm2d default).  Biased variance over H*W, eps inside the square root.
 * workspace: stereo_instance_norm_workspace_bytes(B, C) bytes of device memory, 8-byte aligned, caller-owned (it is
 * zeroed on the stream by the call).  H*W % 4 == 0 and aligned bases, else STEREO_ERR_UNSUPPORTED. */
long long stereo_instance_norm_workspace_bytes(int B, int C);
int stereo_instance_norm_relu_f32(const float *x, const float *gamma, const float *beta, void *out, void *workspace,
                                  int out_is_f16, int B, int C, int H, int W, float eps, int relu, void *stream);
/* The same with CHANNELS-LAST input: x is [B][H*W][C] (the memory of a torch.channels_last [B,C,H,W] tensor, which is how
 * MIOpen's convolutions like to hand an extractor's activations over), fp32 (x_is_f16 = 0) or fp16 (1); out is NCHW as above,
 * so the transpose in front of the cost-volume loaders rides on the normalisation's second read instead of being a pass of
 * its own.  C % 4 == 0, C <= 256, 256 % (C / 4) == 0, H*W % 4 == 0, else STEREO_ERR_UNSUPPORTED. */
int stereo_instance_norm_relu_nhwc(const void *x, int x_is_f16, const float *gamma, const float *beta, void *out,
                                   void *workspace, int out_is_f16, int B, int C, int H, int W, float eps, int relu,
                                   void *stream);

/* Channels-last -> NCHW relayout of a feature tensor, x [B][H*W][C] fp32 / fp16 -> out [B,C,H,W] fp32 / fp16 (widening or
 * rounding once if the two differ): for an extractor that ends in a plain convolution (raft_stereo.py's fnet, extractor.py:201)
 * and runs in torch.channels_last.  C % 4 == 0, C <= 256, H*W % 4 == 0, else STEREO_ERR_UNSUPPORTED. */
int stereo_relayout_nhwc_to_nchw(const void *x, int x_is_f16, void *out, int out_is_f16, int B, int C, int H, int W,
                                 void *stream);

/* The opposite direction, fp32: x [B,C,H,W] -> out [B][H*W][C] (the memory of a torch.channels_last tensor): the update loops
 * (raft_stereo.py:150-199) take the hidden states and context tensors the extractors hand over NCHW and run channels-last.
 * C % 32 == 0, H*W % 4 == 0, else STEREO_ERR_UNSUPPORTED. */
int stereo_relayout_nchw_to_nhwc_f32(const float *x, float *out, int B, int C, int H, int W, void *stream);

/* A14's convolutions (update/update.py:6-33,66-91,130-178: nn.Conv2d, stride 1, "same" padding, 3x3 or 1x1) over
 * channels-last fp32 activations, on the bf16 matrix cores with split operands (fp32-faithful: every operand is hi + mid + lo
 * exactly, six piece products per fp32 product, fp32 accumulation).
 *   stereo_conv_pack_weights: weight [Cout][Cin][k][k] fp32 (nn.Conv2d.weight, contiguous) -> `packed`
 *       (stereo_conv_packed_bytes(Cout, Cin, k) bytes, 16-byte aligned), once per weight tensor.  Cin % 16 == 0.
 *   stereo_conv_nhwc_f32: the K dimension runs over `nsrc` (<= 4) source tensors [B][H][W][src_channels[i]] in concatenation
 *       order (= conv(torch.cat(srcs, 1)) without the concatenation; every src_channels[i] % 16 == 0); output channels go, in
 *       order, to `nout` destination tensors [B][H][W][out_channels[j]] (every out_channels[j] but the last a multiple of 32;
 *       at most 512 output channels in all): out = conv + bias (bias [sum out_channels] or NULL), ReLU if `relu`.
 *       out_strides[j] (NULL: = out_channels[j]) is the channel count of destination j's pixels when the convolution fills
 *       only its first out_channels[j] channels (the motion encoder's 126 + 2 layout, update.py:89-91).
 *   A source image must stay below 2 GB (H*W*C*4 < 2^31), else STEREO_ERR_UNSUPPORTED. */
long long stereo_conv_packed_bytes(int Cout, int Cin, int ksize);
int stereo_conv_pack_weights(const float *weight, void *packed, int Cout, int Cin, int ksize, void *stream);
int stereo_conv_nhwc_f32(const float *const *srcs, const int *src_channels, int nsrc, const void *packed, const float *bias,
                         float *const *outs, const int *out_channels, const int *out_strides, int nout, int ksize, int relu,
                         int B, int H, int W, void *stream);
/* A reduced-precision form for fp32 activations ("bf16x3", the counterpart of torch.set_float32_matmul_precision('high')): every
 * operand as hi + mid bf16 pieces, the three leading piece products -- ~2^-16 relative error per product (100x finer than
 * TF32, 250x coarser than the default form), half the matrix work.  NOT the parity configuration; opt-in on the host side
 * (`conv_precision="high"`).  `packed` from stereo_conv_pack_weights_bf16x3 / stereo_conv_packed_bytes_bf16x3. */
long long stereo_conv_packed_bytes_bf16x3(int Cout, int Cin, int ksize);
int stereo_conv_pack_weights_bf16x3(const float *weight, void *packed, int Cout, int Cin, int ksize, void *stream);
int stereo_conv_nhwc_f32_bf16x3(const float *const *srcs, const int *src_channels, int nsrc, const void *packed,
                                const float *bias, float *const *outs, const int *out_channels, const int *out_strides, int nout,
                                int ksize, int relu, int B, int H, int W, void *stream);
/* The same for fp16 activations (the update block under the reference's mixed_precision flag, raft_stereo.py:176: fp16 autocast
 * around the block): sources and destinations fp16 channels-last, weights rounded to fp16 by stereo_conv_pack_weights_f16
 * (what autocast hands the library), v_mfma_f32_32x32x16_f16 with fp32 accumulation, bias (fp32) added before the one rounding
 * to fp16.  `packed` from stereo_conv_pack_weights_f16 / stereo_conv_packed_bytes_f16. */
long long stereo_conv_packed_bytes_f16(int Cout, int Cin, int ksize);
int stereo_conv_pack_weights_f16(const float *weight, void *packed, int Cout, int Cin, int ksize, void *stream);
int stereo_conv_nhwc_f16(const void *const *srcs, const int *src_channels, int nsrc, const void *packed, const float *bias,
                         void *const *outs, const int *out_channels, const int *out_strides, int nout, int ksize, int relu,
                         int B, int H, int W, void *stream);

/* The general form of the three entry points above: arithmetic form selected by `arith`, and a per-destination EPILOGUE that
 * replaces the pointwise launches around the ConvGRU's convolutions (update/update.py:28-32: z = sigmoid(convz(hx) + cz),
 * r = sigmoid(convr(hx) + cr), q = tanh(convq([r h, x]) + cq), h' = (1 - z) h + z q).
 *   arith   STEREO_CONV_F32_F16X3   fp32 activations, every operand as two fp16 pieces x = hi + 2^-11 lo' (22-23 significant
 *                                   bits), the three leading piece products (each exact in fp32) into two fp32 accumulators:
 *                                   fp32-faithful for |x| < 65 504 (inf / NaN beyond, never silently wrong), HALF the matrix
 *                                   work of the bf16 form.  The host layer's default for fp32 activations.
 *           STEREO_CONV_F32_F16X3S  the same three products into ONE accumulator: every output channel's weights are pre-scaled
 *                                   by a power of two (max into [2^13, 2^14)) and packed as three fp16 pieces (hi, unscaled
 *                                   lo, 2^-11 hi), the epilogue undoes the scale; wave tile 4 x 2 blocks (8-row pixel tiles):
 *                                   half the LDS operand traffic per MFMA (opt-in on the host side: 7 % faster on the widest launch, equal over
 *                                   a whole loop)
 *           STEREO_CONV_F32_BF16X6  = stereo_conv_nhwc_f32 (three bf16 pieces, six products: fp32's full exponent range)
 *           STEREO_CONV_F32_BF16X3  = stereo_conv_nhwc_f32_bf16x3        STEREO_CONV_F16  = stereo_conv_nhwc_f16
 *           (`packed` must come from stereo_conv_pack_weights_ex with the same `arith`.)
 *   epilogue[j] (NULL: all linear), with t = conv + bias + addend[j] (addend[j], aux0[j], aux1[j]: tensors of destination j's
 *   geometry [B][H][W][out_strides[j]] and element type, or NULL; `relu` only with linear epilogues):
 *           STEREO_EPI_LINEAR       out = t (ReLU if `relu`)
 *           STEREO_EPI_SIGMOID      out = sigmoid(t)                                  (z; addend = cz)
 *           STEREO_EPI_SIGMOID_MUL  out = sigmoid(t) * aux0                           (r h; addend = cr, aux0 = h)
 *           STEREO_EPI_GRU_BLEND    out = (1 - aux1) * aux0 + aux1 * tanh(t)          (h'; addend = cq, aux0 = h, aux1 = z)
 *           STEREO_EPI_GRU_BLEND_SOFTSIGN   the same with t / (1 + |t|) in place of tanh(t)   (Attempts/update.py:33-34)
 *   out may alias aux0 / aux1 / addend element for element (every output element is read and written by the same thread). */
#define STEREO_CONV_F32_BF16X6 0
#define STEREO_CONV_F32_BF16X3 1
#define STEREO_CONV_F16 2
#define STEREO_CONV_F32_F16X3 3
#define STEREO_CONV_F32_F16X3S 4
#define STEREO_EPI_LINEAR 0
#define STEREO_EPI_SIGMOID 1
#define STEREO_EPI_SIGMOID_MUL 2
#define STEREO_EPI_GRU_BLEND 3
#define STEREO_EPI_GRU_BLEND_SOFTSIGN 4
long long stereo_conv_packed_bytes_ex(int arith, int Cout, int Cin, int ksize);
int stereo_conv_pack_weights_ex(int arith, const float *weight, void *packed, int Cout, int Cin, int ksize, void *stream);
int stereo_conv_nhwc_ex(int arith, const void *const *srcs, const int *src_channels, int nsrc, const void *packed,
                        const float *bias, void *const *outs, const int *out_channels, const int *out_strides, int nout,
                        const int *epilogue, const void *const *addend, const void *const *aux0, const void *const *aux1,
                        int ksize, int relu, int B, int H, int W, void *stream);

/* Content fingerprints of `n` device tensors (fp32 parameters: nbytes[k] % 4 == 0) in one launch: out[k] (device, n x 64 bit)
 * = a position-dependent 64-bit hash sum of tensor k's words.  The host layer compares them once per forward of a loop to
 * rebuild packed weights after an in-place update through `tensor.data`, which leaves no host-visible trace. */
int stereo_fingerprint(const void *const *tensors, const long long *nbytes, int n, unsigned long long *out, void *stream);

/* 3x3 stride-1 "same" convolution with 1..4 output channels (flow_head.conv2, update.py:11-14: 256 -> 2) on the fp32 vector
 * units: x [B][H][W][Cin] channels-last (Cin % 32 == 0), weight_tco = weight.permute(2, 3, 1, 0) contiguous, i.e.
 * [3][3][Cin][Cout]; bias [Cout] or NULL; out_nchw [B][Cout][H][W]. */
int stereo_conv3x3_small_nhwc_f32(const float *x, const float *weight_tco, const float *bias, float *out_nchw, int B, int Cin,
                                  int Cout, int H, int W, void *stream);
/* fp16 channels-last activations (mixed_precision), widened on the way into LDS; weights / bias / output fp32 as above (the
 * caller rounds the weights to fp16 first if it wants the library's autocast numerics). */
int stereo_conv3x3_small_nhwc_f16(const void *x, const float *weight_tco, const float *bias, float *out_nchw, int B, int Cin,
                                  int Cout, int H, int W, void *stream);

/* pool2x / pool4x (update/update.py:119-123: F.avg_pool2d(x, k, stride, padding), zero padding, divisor k*k) on channels-last
 * fp32 tensors: in [N][Hi][Wi][C] -> out [N][Ho][Wo][C], Ho = (Hi + 2 pad - k) / stride + 1; C % 4 == 0. */
int stereo_avg_pool_nhwc_f32(const float *in, float *out, int N, int C, int Hi, int Wi, int ksize, int stride, int pad,
                             void *stream);

/* KxK stride-1 "same" convolution with 1..4 INPUT channels (the motion encoder's convf1, update/update.py:79,87: 7x7 over the
 * 2-channel flow) on the fp32 vector units: x [B,Cin,H,W] NCHW fp32, weight_tco = weight.permute(2, 3, 1, 0) contiguous
 * ([K][K][Cin][Cout], Cout % 16 == 0, Cout <= 64), bias [Cout] or NULL, ReLU if `relu`; out [B][H][W][out_stride] channels-last
 * (its first Cout channels are written).  K odd, K <= 7. */
int stereo_conv_small_cin_nhwc_f32(const float *x, const float *weight_tco, const float *bias, float *out, int B, int Cin, int Cout,
                                   int H, int W, int ksize, int out_stride, int relu, void *stream);

/* The two entries above in one, with an optional ReLU on the output (OffsetPredictionHead's relu2, Attempts/update.py:103-105). */
int stereo_conv3x3_small_nhwc_ex(const void *x, int x_is_f16, const float *weight_tco, const float *bias, float *out_nchw, int B,
                                 int Cin, int Cout, int H, int W, int relu, void *stream);

/* HBM arena of separately mapped physical chunks (HIP virtual-memory management), for placing the volume stage's 25.5-GB output
 * where the memory system takes its stores fastest (the reference leaves the placement of
 * build_gwc_volume's output, utils/build_cost_volume.py:5-54, to torch's allocator).  `nchunks` allocations of `chunk_bytes` (a
 * multiple of stereo_hbm_arena_granularity) are mapped back to back at *base_out; after the caller has timed its windows,
 * stereo_hbm_arena_keep unmaps and releases every chunk outside [first, first + count) -- addresses inside stay valid, any
 * access outside faults.  keep / destroy synchronise the device first.  STEREO_ERR_UNSUPPORTED without VMM support,
 * STEREO_ERR_LAUNCH when the driver refuses (out of memory); nothing stays allocated on error. */
int stereo_hbm_arena_granularity(long long *bytes);
int stereo_hbm_arena_create(long long chunk_bytes, int nchunks, void **arena_out, void **base_out);
int stereo_hbm_arena_keep(void *arena, int first, int count);
int stereo_hbm_arena_destroy(void *arena);

#ifdef __cplusplus
}
#endif
#endif /* STEREO_HIP_H */
