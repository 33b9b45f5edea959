/* ccengine -- C ABI of the MI355X-native Competitive-Collaboration training hot path.
 *
 * libccengine.so (built by hipcc for gfx950 from the .hip sources under cc_amd/csrc) is the drop-in boundary
 * underneath the reference's Python call surface.  The reference itself has
 * no FFI on this path -- it calls ATen/cuDNN ops and one third-party CUDA extension -- so every
 * entry point below names the reference call it replaces (file:line under anuragranj/cc).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 NCHW data owned by the caller
 *     (torch tensors); nothing is allocated or freed inside and no state is kept between calls (the one exception is the
 *     measurement aid cc_timing_*, off by default), all entry points are thread-safe and capturable into a hipGraph;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - return value: 0 = launched; CC_ERR_ARG (-1) bad argument; CC_ERR_LAUNCH (-2) launch failed;
 *   - P is the 3x4 projection K.[R|t] (inverse_warp.py:214,278), Kinv the 3x3 inverse
 *     intrinsics, both row-major per batch item;
 *   - align_corners selects grid_sample semantics (0 = what the reference executes under
 *     torch >= 1.3, 1 = the authors' torch-1.0 behaviour);
 *   - *_ws_* / *_bytes functions size caller-provided scratch.
 */
#ifndef CCENGINE_H
#define CCENGINE_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

size_t cc_version(void);

/* ---------------------------------------------------------------- geometry (inverse_warp.py) */

/* inverse_warp.py:31-45 pixel2cam: cam[B,3,H,W] = (Kinv . (x, y, 1)) * depth, and :48-79 cam2pixel: grid[B,H,W,2] in [-1,1]
 * from cam and P[B,12] = rows (rot | tr) (has_rot / has_tr: the reference's `is not None` switches; rewrite_oob: padding_mode
 * 'zeros' sets out-of-range coordinates to 2, :72-76).  Stand-alone forms of the two halves the fused warp kernels compute
 * in registers (same fmaf chains: their composition gives the fused kernels' coordinates bit for bit). */
int cc_pixel2cam(const float* depth, const float* Kinv, float* cam, int B, int H, int W, void* stream);
int cc_cam2pixel(const float* cam, const float* P, float* grid, int B, int H, int W, int has_rot, int has_tr, int rewrite_oob,
                 void* stream);
/* their backward passes (inverse_warp.py:31-79 under autograd): per-pixel adjoints + a deterministic two-stage reduction of the
 * matrix gradients.  gKinv12 [B,12]: slots 0..8 = d/dKinv row-major; gP [B,12]: rows (d/drot | d/dtr); rewritten out-of-range
 * coordinates and a clamped Z carry no gradient.  ws_partials: cc_warp_partials_bytes(B, H, W). */
int cc_pixel2cam_bwd(const float* g, const float* depth, const float* Kinv, float* gdepth_or_null, float* gKinv12, float* ws_partials,
                     int B, int H, int W, void* stream);
int cc_cam2pixel_bwd(const float* ggrid, const float* cam, const float* P, float* gcam_or_null, float* gP, float* ws_partials, int B,
                     int H, int W, int has_rot, int has_tr, int rewrite_oob, void* stream);

/* scratch for the per-workgroup partial sums of dL/dP: B * ceil(H*W/256) * 12 floats */
size_t cc_warp_partials_bytes(int B, int H, int W);

/* inverse_warp.py:250-283 inverse_warp (pixel2cam :31-45, cam2pixel :48-79, F.grid_sample :281):
 * out[B,C,H,W] = bilinear sample of img[B,C,H,W] at the projection of every target pixel.
 * padding_border: 0 = 'zeros' (OOB coordinates rewritten to 2, :72-76), 1 = 'border'. */
int cc_inverse_warp_fwd(const float* img, const float* depth, const float* P, const float* Kinv, float* out,
                        int B, int C, int H, int W, int padding_border, int align_corners, void* stream);

/* autograd of the above w.r.t. depth[B,H,W] and P[B,12] (and, optionally, img via atomics:
 * gimg must be zero-filled).  ws_partials: cc_warp_partials_bytes(). */
int cc_inverse_warp_bwd(const float* gout, const float* img, const float* depth, const float* P,
                        const float* Kinv, float* gdepth, float* gP, float* gimg_or_null, float* ws_partials,
                        int B, int C, int H, int W, int padding_border, int align_corners, void* stream);

/* inverse_warp.py:195-220 pose2flow: rigid flow [B,2,H,W] in pixels.  rewrite_oob = 1 only for
 * padding_mode='zeros' (the reference's callers all use padding_mode=None -> 0). */
int cc_pose2flow_fwd(const float* depth, const float* P, const float* Kinv, float* flow, int B, int H, int W,
                     int rewrite_oob, void* stream);
int cc_pose2flow_bwd(const float* gflow, const float* depth, const float* P, const float* Kinv, float* gdepth,
                     float* gP, float* ws_partials, int B, int H, int W, int rewrite_oob, void* stream);

/* inverse_warp.py:164-192 flow_warp: sample img at (x+u, y+v). */
int cc_flow_warp_fwd(const float* img, const float* flow, float* out, int B, int C, int H, int W,
                     int padding_border, int align_corners, void* stream);
int cc_flow_warp_bwd(const float* gout, const float* img, const float* flow, float* gflow_or_null,
                     float* gimg_or_null, int B, int C, int H, int W, int padding_border, int align_corners,
                     void* stream);

/* models/back2future.py:287-321 Model.warp: border-padded feature warp, grads to features
 * (atomics, gfeat zero-filled by the caller) and flow.  flow_scale: constant multiplied into the flow first (the callers'
 * `up_flow * 0.625` ... `* 5.0` and their negations, back2future.py:196-285, folded in). */
int cc_feature_warp_fwd(const float* feat, const float* flow, float* out, int B, int C, int H, int W,
                        int align_corners, float flow_scale, void* stream);
int cc_feature_warp_bwd(const float* gout, const float* feat, const float* flow, float* gflow_or_null,
                        float* gfeat_or_null, int B, int C, int H, int W, int align_corners, float flow_scale, void* stream);
/* The same backward with a run-to-run REPRODUCIBLE feature gradient (cc_amd.config.deterministic): the scatter accumulates
 * 64-bit fixed-point integers (integer atomics are order-independent; scale = a power of two from max |gout|) and converts at
 * the end: gfeat = (accumulate ? gfeat : 0) + sum.  ws: cc_feature_warp_bwd_det_ws_bytes().  The reference's grid_sample
 * backward (models/back2future.py:287-321) scatters with float atomics on CUDA and is not reproducible either. */
size_t cc_feature_warp_bwd_det_ws_bytes(int B, int C, int H, int W);
int cc_feature_warp_bwd_det(const float* gout, const float* feat, const float* flow, float* gflow_or_null, float* gfeat,
                            void* ws, int B, int C, int H, int W, int align_corners, float flow_scale, int accumulate,
                            void* stream);

/* inverse_warp.py:82-119,146-162,214,278 (+ loss_functions.py:91): P[n] = K_s[n] . [Rx.Ry.Rz | t] for pose[n] =
 * (tx,ty,tz,rx,ry,rz) at pose + n*pose_stride, K_s = K with rows 0,1 divided by k_div (the pyramid downscale).
 * euler rotation mode only (the only mode train.py reaches). */
int cc_pose_proj_fwd(const float* pose, long pose_stride, const float* K, float* P, int N, float k_div, void* stream);
int cc_pose_proj_bwd(const float* gP, const float* pose, long pose_stride, const float* K, float* gpose, long gpose_stride,
                     int N, float k_div, int accumulate, void* stream);

/* ---------------------------------------------------------------- SSIM / photometric (ssim.py, loss_functions.py)
 * gauss13_host is the ONE host pointer of this ABI: the 13 taps of the normalised 1-D Gaussian
 * (ssim.py:9-11, sigma 1.5), copied into the kernel arguments at launch. */

/* number of 32x32 tiles = length of the partial-sum buffer of cc_ssim_photo_fwd (x4 floats) */
size_t cc_ssim_num_blocks(int B, int H, int W);

/* ssim.py:68-76 ssim(img1, img2, window_size=13): per-pixel, per-channel map [B,3,H,W], zero padding 6 */
int cc_ssim_fwd(const float* img1, const float* img2, float* out, const float* gauss13_host, int B, int H, int W,
                void* stream);
int cc_ssim_bwd(const float* img1, const float* img2, const float* gout, float* adjA, float* adjB, float* adjC,
                float* gimg2, const float* gauss13_host, int B, int H, int W, void* stream);

/* The body of one (scale, reference-frame) term of photometric_reconstruction_loss
 * (loss_functions.py:99-114) / photometric_flow_loss (:44-58), fused:
 *   valid = 1 - prod_c(warped == 0); m = mask_a * mask_b (either may be null; mask_b may be used as 1 - mask_b);
 *   diff = (tgt - warped) * valid * m;  ssim_loss = (1 - ssim(tgt, warped) * valid) * m;
 *   loss_accum[0] += (1-wssim) * N/sum(valid) * (mean (diff^2+0.01)^q + wssim * mean ssim_loss)
 *                    + lambda_oob * robust_l1(1 - valid);
 *   scale_out[0]   = (1-wssim) * N/sum(valid) / (3N)   (common factor of all adjoints of this term);
 *   nan_flag[0]    = 1 if the term is NaN (deferred form of the reference's asserts :60,105,115).
 * With want_grad it also writes the adjoint maps consumed by cc_ssim_photo_bwd (adjA/B/C, g0: [B,3,H,W])
 * and gmask = d(term)/d(mask_b) / scale  (element (b,p) at gmask[b*gmask_bstride + p]).
 * partials: cc_ssim_num_blocks()*4 floats. */
int cc_ssim_photo_fwd(const float* tgt, const float* warped, const float* mask_a, int mask_a_bstride,
                      const float* mask_b, int mask_b_bstride, int mask_b_complement, float* partials, float* adjA,
                      float* adjB, float* adjC, float* g0, float* gmask, int gmask_bstride, int want_grad,
                      float wssim, float q, float lambda_oob, float* loss_accum, float* scale_out, float* nan_flag,
                      const float* gauss13_host, int B, int H, int W, void* stream);

/* gwarped (+)= scale[0] * (g0 + G*adjA + 2*warped*(G*adjB) + tgt*(G*adjC)): d(term)/d(warped) */
int cc_ssim_photo_bwd(const float* adjA, const float* adjB, const float* adjC, const float* g0, const float* tgt,
                      const float* warped, const float* scale, float* gwarped, int accumulate,
                      const float* gauss13_host, int B, int H, int W, void* stream);

/* loss_functions.py:181-188 (consensus_exp_masks error maps): err[B,1,H,W] = (1-wssim)*mean_c sqrt((tgt-w)^2+0.01)
 * + wssim*mean_c(1 - ssim(tgt,w)); valid[B,1,H,W] = 1 - prod_c(w == 0) */
int cc_ssim_err_fwd(const float* tgt, const float* warped, float* err, float* valid, float wssim,
                    const float* gauss13_host, int B, int H, int W, void* stream);

/* loss_functions.py:173-175,189-193: target = (wrig * min(err_cf, err_cb) * OR(valid_cf, valid_cb) <= err_ff + 1e-8) */
int cc_consensus_target(const float* err_cam_fwd, const float* err_cam_bwd, const float* err_flow_fwd,
                        const float* valid_cam_fwd, const float* valid_cam_bwd, float* target, float wrig, int n,
                        void* stream);

/* ---------------------------------------------------------------- pyramid, masks, smoothness, BCE (loss_functions.py)
 * Every *_fwd_bwd entry adds its loss value into loss_accum[0] (device scalar) and, when the gradient
 * pointer is non-null, writes d(value)/d(input) * gscale in the same pass.  partials: scratch of
 * (number of workgroups) floats -- cc_elem_num_blocks(H*W) * planes. */
size_t cc_elem_num_blocks(int n);

/* F.adaptive_avg_pool2d(x, (h, w)) as used at loss_functions.py:36-37,89-90,163-165,315 */
int cc_adaptive_avg_pool(const float* in, float* out, int planes, int H, int W, int h, int w, void* stream);
/* the same for levels 1..nlevels-1 (sizes H>>l x W>>l), packed back to back in out_packed */
int cc_pyramid_build(const float* level0, float* out_packed, int nlevels, int planes, int H, int W, void* stream);
/* ... of nimg (<= 8) same-sized images in one launch: the target frame and the reference frames of a training step (each loss pools
 * them per scale, loss_functions.py:36-37,89-90).  level0_host / out_packed_host: HOST arrays of nimg device addresses; H, W
 * multiples of 32, 2 <= nlevels <= 6.  Bit-identical to cc_pyramid_build per image. */
int cc_pyramid_build_multi(const long* level0_host, const long* out_packed_host, int nimg, int nlevels, int planes, int H, int W,
                           void* stream);

/* loss_functions.py:343-352 occlusion_masks -> (1 - occ) [B,1,H,W] (occ_fw == occ_bw) */
int cc_flow_noocc(const float* flow_bw, const float* flow_fw, float* out, int B, int H, int W, void* stream);
/* loss_functions.py:132-137 depth_occlusion_masks -> (1 - occ) [B,4,H,W]; flows4 = the four rigid flows
 * [4][B,2,H,W] (pose2flow with the full-resolution K) */
int cc_rigid_noocc(const float* flows4, float* out, int B, int H, int W, void* stream);
/* the same from depth [B,H,W], the four projection matrices P4 = [4][B,3,4] (K.[R|t] with the FULL-resolution K) and
 * Kinv [B,3,3]: the four pose2flow results never leave the registers */
int cc_rigid_noocc_fused(const float* depth, const float* P4, const float* Kinv, float* out, int B, int H, int W, void* stream);

/* accum[0] += coef * sum(partials[0..n)) -- deterministic second reduction stage */
int cc_reduce_add(const float* partials, int n, float coef, float* accum, void* stream);

/* loss_functions.py:287-319 edge_aware_smoothness_loss, one scale (img already pooled to H x W) */
int cc_edge_smooth_fwd_bwd(const float* img, const float* pred, float* gpred_or_null, float* partials, float* loss_accum,
                           float gscale, int B, int C, int H, int W, void* stream);
/* loss_functions.py:323-341 smooth_loss, one scale (weight = 1/2.3^scale) */
int cc_smooth2_fwd_bwd(const float* pred, float* gpred_or_null, float* partials, float* loss_accum, float weight,
                       float gscale, int planes, int H, int W, void* stream);
/* loss_functions.py:148-155 explainability_loss, one scale: BCE(mask, 1) */
int cc_bce_ones_fwd_bwd(const float* mask, float* gmask_or_null, float* partials, float* loss_accum, float gscale, int n,
                        void* stream);
/* loss_functions.py:221-261 consensus_depth_flow_mask + weighted_binary_cross_entropy, one scale */
int cc_consensus_bce_fwd_bwd(const float* exp_mask, const float* census_bwd, const float* census_fwd,
                             const float* target_bwd, const float* target_fwd, float* gmask_or_null, float* partials,
                             float* loss_accum, float thresh, float wbce, float gscale, int B, int H, int W,
                             void* stream);
/* out[b, e] (=, +=) sum_k src_k[b, e], e < chw: n <= 8 tensors [B, chw] with their own batch strides (channel slices of
 * larger buffers), summed in a fixed order by one launch; src / src_bs: HOST arrays of device addresses / strides in floats.
 * Replaces: the autograd engine's pairwise accumulation of a multi-consumer tensor's gradient (train.py:567). */
int cc_sum_strided(int n, const long* src, const long* src_bs, float* out, long out_bs, int B, long chw, int accumulate,
                   void* stream);
/* out = a * scalar_dev[0] */
int cc_scale_by_scalar(const float* a, const float* scalar_dev, float* out, int n, void* stream);
/* dst_k (=, +=) src_k * scalar_dev[0] for njobs spans in one launch per 32; jobs_host: njobs x 4 longs {src, dst, n floats,
 * accumulate}.  Replaces: the `* grad_output` of a loss term's backward AND the autograd engine's pairwise accumulation of the
 * gradients several loss terms send to one network output (train.py:509,567): every term scales its stashed gradients straight
 * into the step's per-tensor accumulator (first writer =, later writers +=, in the engine's own fixed order). */
int cc_scale_acc_jobs(const long* jobs_host, int njobs, const float* scalar_dev, void* stream);

/* ---------------------------------------------------------------- cost volume (models/back2future.py:15-25)
 * vol[b, ch(d), y, x] = (1/C) sum_c f1[b,c,y,x] * f2[b,c,y+dy-4,x+dx-4], d = dy*9+dx: the spatial_correlation_sampler
 * call (kernel_size=1, patch_size=9, stride=1) + the division by C (:24) + index_select(idx_fwd/idx_bwd) (:175-177,
 * chan_of_disp = inverse permutation, int32 device array or null) + torch.cat (written at out_channel_offset of a
 * tensor with out_channels_total channels). */
int cc_corr9x9_fwd(const float* f1, const float* f2, float* out, const int* chan_of_disp_or_null, int B, int C, int H,
                   int W, int out_channels_total, int out_channel_offset, void* stream);
int cc_corr9x9_bwd(const float* gout, const float* f1, const float* f2, float* g1, float* g2_or_null,
                   const int* chan_of_disp_or_null, int B, int C, int H, int W, int g_channels_total,
                   int g_channel_offset, int accumulate_g1, void* stream);
/* general odd patch P with dilation D (models/FlowNetC6.py:18-30: P = 21, D = 2): out [B, P*P, H, W] = sample / C */
int cc_corr_patch_fwd(const float* f1, const float* f2, float* out, int B, int C, int H, int W, int patch, int dilation,
                      void* stream);
int cc_corr_patch_bwd(const float* gout, const float* f1, const float* f2, float* g1_or_null, float* g2_or_null, int B, int C,
                      int H, int W, int patch, int dilation, void* stream);

/* ---------------------------------------------------------------- convolutions (nn.Conv2d / nn.ConvTranspose2d of models/ *.py)
 * fp32 implicit GEMM on v_mfma_f32_32x32x2_f32.  act: 0 none, 1 ReLU, 2 LeakyReLU(0.2), 3 act_a*sigmoid+act_b.
 * Batch strides (elements) let inputs/outputs be channel slices of wider NCHW tensors (no torch.cat copies). */
size_t cc_conv2d_fwd_ws_bytes(int B, int Cin, int IH, int IW, int Cout, int R, int S, int stride, int pad, int OH, int OW);
int cc_conv2d_fwd(const float* x, const float* w, const float* bias_or_null, const float* res_or_null, float* y, float* ws,
                  const float* prepacked_or_null, int B, int Cin, int IH, int IW, long x_bs, int Cout, int R, int S, int stride,
                  int pad, int OH, int OW, long y_bs, long res_bs, int act, float act_a, float act_b, void* stream);
/* gx[n,c,iy,ix] = act(bias[c] + sum_{k,r,s} w(k,c,r,s) * gy[n,k,oy,ox]), iy = oy*stride - pad + r: the data-gradient of
 * conv2d (act 0, bias null) and the forward of ConvTranspose2d (weight [Cin=K, Cout=C, R, S]); one launch per
 * output parity class.  w(k,c,r,s) = w[k*w_k_stride + c*w_c_stride + r*S + s].
 * ws (both calls): scratch for the per-call weight repack [tap][c][m], a zero line for the LDS-DMA halo and the
 * split-K partial slabs of deep layers on small maps. */
size_t cc_conv2d_dgrad_ws_bytes(int B, int K, int OH, int OW, int C, int R, int S, int stride, int pad, int IH, int IW);
int cc_conv2d_dgrad(const float* gy, const float* w, const float* bias_or_null, float* gx, float* ws,
                    const float* prepacked_or_null, int B, int K, int OH, int OW, long gy_bs, int C, int R, int S, int stride,
                    int pad, int IH, int IW, long gx_bs, long w_k_stride, long w_c_stride, int act, float act_a, float act_b,
                    void* stream);
/* Optional per-step weight prepack: the [tap][c][m] weight images every conv call otherwise builds itself (one tiny
 * launch each, ~540 per step) are produced for ALL layers by one cc_repack_table launch.  *_pack_floats = size of a
 * layer's image buffer (0: geometry not eligible), which must start with 64 zero floats; *_pack_desc write 16-long
 * descriptors into a HOST array (src_ptr / pack_base_ptr are device addresses) and return their count; the caller sets
 * desc[14] = first block (cumulative sum of desc[15], the descriptor's workgroup count) and uploads the table. */
size_t cc_conv2d_fwd_pack_floats(int B, int Cin, int IH, int IW, int Cout, int R, int S, int stride, int pad, int OH, int OW);
int cc_conv2d_fwd_pack_desc(int B, int Cin, int IH, int IW, int Cout, int R, int S, int stride, int pad, int OH, int OW,
                            long src_ptr, long pack_base_ptr, long* desc_out_host);
size_t cc_conv2d_dgrad_pack_floats(int B, int K, int OH, int OW, int C, int R, int S, int stride, int pad, int IH, int IW,
                                   long w_k_stride, long w_c_stride);
int cc_conv2d_dgrad_pack_desc(int B, int K, int OH, int OW, int C, int R, int S, int stride, int pad, int IH, int IW,
                              long w_k_stride, long w_c_stride, long src_ptr, long pack_base_ptr, long* desc_out_host);
int cc_repack_table(const long* table_dev, int ndesc, long total_blocks, void* stream);
size_t cc_conv2d_wgrad_ws_bytes(int B, int M, int AH, int AW, int Cin, int R, int S, int si);
/* gw[m*o_sm + c*o_sc + r*S + s] = sum_{n,ty,tx} a[n,m,ty,tx] * x[n,c,si*ty-pad+r,si*tx-pad+s] (split over pixels,
 * deterministic second-stage reduction through ws).  accumulate != 0: gw += (gradient accumulation of a weight that is
 * used more than once per step / written straight into the optimizer's flat gradient bucket, train.py:566-567). */
int cc_conv2d_wgrad(const float* a, const float* x, float* gw, float* ws, int B, int M, int AH, int AW, long a_bs, int Cin,
                    int IH, int IW, long x_bs, int R, int S, int si, int pad, long o_sm, long o_sc, int accumulate,
                    void* stream);
/* ---- group forms: G same-shaped problems in ONE launch -- the parallel branches of a network that the reference runs as
 * separate nn.Sequential stacks (models/back2future.py:152-171,186-204: decoder_fwd / decoder_bwd / decoder_occ of a pyramid
 * level; :27-33,159-169: the conv{l}a / conv{l}b / conv{l}c feature streams).  x / w / bias / res / y / gy / gx / mul / a / gw /
 * prepacked are HOST arrays of G device addresses (0 = null, uniformly over the group; array pointer null = all null); ws is
 * G consecutive areas of (*_group_ws_bytes / G) resp. the single-problem *_ws_bytes each.  G <= 12 (fwd, dgrad: G * stride^2
 * <= 12 for the merged launch) resp. G <= 4 (wgrad, act_bwd); without prepacked weight images the problems run one by one.
 * cc_conv2d_dgrad_group additionally takes `mul`: tensors of gx's shape; gx = act'(mul) * sum, i.e. the gradient w.r.t. the
 * PRE-activation of the layer whose output `mul` is (act / act_a / act_b describe that activation): the producer's separate
 * activation-backward pass over (gy, y) -> geff disappears. */
size_t cc_conv2d_fwd_group_ws_bytes(int G, int B, int Cin, int IH, int IW, int Cout, int R, int S, int stride, int pad, int OH,
                                    int OW);
int cc_conv2d_fwd_group(int G, const long* x, const long* w, const long* bias, const long* res, const long* y, float* ws,
                        const long* prepacked, int B, int Cin, int IH, int IW, long x_bs, int Cout, int R, int S, int stride,
                        int pad, int OH, int OW, long y_bs, long res_bs, int act, float act_a, float act_b, void* stream);
size_t cc_conv2d_dgrad_group_ws_bytes(int G, int B, int K, int OH, int OW, int C, int R, int S, int stride, int pad, int IH, int IW);
int cc_conv2d_dgrad_group(int G, const long* gy, const long* w, const long* bias, const long* gx, const long* mul, float* ws,
                          const long* prepacked, int B, int K, int OH, int OW, long gy_bs, int C, int R, int S, int stride,
                          int pad, int IH, int IW, long gx_bs, long mul_bs, long w_k_stride, long w_c_stride, int act, float act_a,
                          float act_b, void* stream);
/* ... with `add` (HOST array like mul): gx = (sum + add) * act'(mul), or act(sum + add) without mul (act 0: plain accumulation) -- the other gradient contributions of a fan-out
 * tensor (the gradient a residual shortcut carries, what earlier data-gradients left in gx: add may alias gx) are summed in the
 * epilogue instead of by accumulation launches (models/DispResNet6.py:31-43: out = relu(conv2(relu(conv1(x))) + x)). */
int cc_conv2d_dgrad_group_add(int G, const long* gy, const long* w, const long* gx, const long* mul, const long* add, float* ws,
                              const long* prepacked, int B, int K, int OH, int OW, long gy_bs, int C, int R, int S, int stride,
                              int pad, int IH, int IW, long gx_bs, long mul_bs, long add_bs, long w_k_stride, long w_c_stride,
                              int act, float act_a, float act_b, void* stream);
int cc_conv2d_wgrad_group(int G, const long* a, const long* x, const long* gw, float* ws, int B, int M, int AH, int AW, long a_bs,
                          int Cin, int IH, int IW, long x_bs, int R, int S, int si, int pad, long o_sm, long o_sc, int accumulate,
                          void* stream);
/* The same launch with the reduction of its partial slabs left to the caller (the weight gradients are not needed before the
 * optimizer step): the reduction descriptors -- 16 longs each, at most G, none when the kernel wrote gw itself -- are written to
 * the HOST array red_host[0 .. *nred_host); ws must stay untouched until cc_wgrad_reduce_table has run on them.
 * zeros64_or_null: 64 zero floats owned by the caller (source of halo pixels; saves a fill launch per call).
 * cc_wgrad_reduce_table: any number of parked reductions in one launch per 32 (fixed summation order, no atomics). */
int cc_conv2d_wgrad_group_defer(int G, const long* a, const long* x, const long* gw, float* ws, int B, int M, int AH, int AW,
                                long a_bs, int Cin, int IH, int IW, long x_bs, int R, int S, int si, int pad, long o_sm, long o_sc,
                                int accumulate, const float* zeros64_or_null, long* red_host, int red_cap, int* nred_host,
                                void* stream);
int cc_wgrad_reduce_table(const long* desc_host, int n, void* stream);
/* n parked groups of DIFFERENT shapes at once (the end of a backward stage): desc_host = n x 32 longs
 *   {G, a[4], x[4], gw[4], ws, B, M, AH, AW, a_bs, Cin, IH, IW, x_bs, R, S, si, pad, o_sm, o_sc, accumulate, 0, 0}
 * Each group is computed exactly as by one cc_conv2d_wgrad_group_defer call (replaces the per-layer calls of
 * torch.autograd's convolution_backward(weights) behind loss.backward(), train.py:567); the groups that take the generic kernel
 * share launches.  red_host: room for the reduce descriptors of all groups (16 longs each, <= G per group). */
int cc_conv2d_wgrad_list(int n, const long* desc_host, const float* zeros64_or_null, long* red_host, int red_cap, int* nred_host,
                         void* stream);
int cc_act_bwd_bias_group(int G, const long* gy, const long* y, const long* geff, const long* gbias, float* ws, int B, int C, int H,
                          int W, long gy_bs, long y_bs, long geff_bs, int act, float act_a, float act_b, int accumulate_bias,
                          void* stream);
/* ---- launch lists (round 3): n INDEPENDENT convolution problems of different shapes -- layers of different networks
 * (train.py:454-463 runs DispResNet6, PoseNetB6, MaskNet6 and Back2Future one after the other although they share nothing but
 * their input), or a layer's forward next to another's data-gradient -- in as few launches as their tile configurations
 * allow: problems whose kernel instance agrees share ONE launch (+ one split-K epilogue launch), split-K is planned for the
 * launch as a whole.  desc_host: n records of 32 longs (HOST memory):
 *   0 kind (0: conv2d forward arithmetic as cc_conv2d_fwd, 1: transposed arithmetic as cc_conv2d_dgrad_group)
 *   1 x (kind 1: gy)  2 w  3 bias  4 res (kind 0: added before act; kind 1: `mul`)  5 y (kind 1: gx)
 *   6 prepacked weight image (required, from cc_repack_table)  7 add (kind 1 with mul: gx = (sum + add) * act'(mul): the other
 *     gradient contributions of a fan-out tensor summed in the epilogue; may alias gx)
 *   8 B  9 Cin (K)  10 IH (OH)  11 IW (OW)  12 x_bs  13 Cout (C)  14 R  15 S  16 stride  17 pad  18 OH (IH)  19 OW (IW)
 *   20 y_bs  21 res_bs  22 add_bs  23 act  24 act_a (float bits)  25 act_b (float bits)  26 w_k_stride  27 w_c_stride
 * ws: cc_conv2d_list_ws_bytes() bytes.  split_target: workgroups a launch should at least have (0 -> 512). */
size_t cc_conv2d_list_ws_bytes(int n, const long* desc_host, int split_target);
int cc_conv2d_list(int n, const long* desc_host, float* ws, int split_target, void* stream);
/* per-kernel timing (measurement aid) -- TOOLS BUILD ONLY (tools/_bin/libccengine_tools.so, cc_amd/build.py build_tools(): the
 * same sources with -DCC_TOOLS; cc_is_tools_build() tells which one is loaded).  The product library reads no environment
 * variable and keeps no state: there cc_timing_enable(1) returns CC_ERR_ARG and cc_timing_collect 0.  In the tools build,
 * between cc_timing_enable(1) and cc_timing_collect the MAIN device kernel of every conv / weight-gradient call is bracketed
 * with HIP events on its stream; collect returns (HOST buffer) one line per device kernel:
 * "name\tlaunches\ttotal_ms\ttotal_gflop\n" and the number of characters written.  The kernel-selection switches listed in
 * tools/README.md (CC_CONV_*, CC_WGRAD_*, ...) exist in the tools build only as well. */
int cc_is_tools_build(void);
int cc_timing_enable(int on);
int cc_timing_collect(void* out_host, int cap);
/* introspection: the name of the device kernel the corresponding entry point dispatches to for this geometry (as it
 * appears in a rocprofv3 kernel trace; "+splitk" = followed by the split-K epilogue).  name_out_host: HOST char buffer. */
int cc_conv2d_fwd_kernel(int B, int Cin, int IH, int IW, int Cout, int R, int S, int stride, int pad, int OH, int OW,
                         void* name_out_host, int cap);
int cc_conv2d_dgrad_kernel(int B, int K, int OH, int OW, int C, int R, int S, int stride, int pad, int IH, int IW,
                           int prepacked, void* name_out_host, int cap);
int cc_conv2d_wgrad_kernel(int B, int M, int AH, int AW, int Cin, int IH, int IW, int R, int S, int si, int pad,
                           void* name_out_host, int cap);
size_t cc_act_bwd_ws_bytes(int C);
/* geff = gy * act'(y);  gbias[c] (+)= sum geff  (either output may be null; geff may alias gy) */
/* Group form with the second stage of the bias gradient parked (see cc_conv2d_wgrad_group_defer / cc_wgrad_reduce_table). */
int cc_act_bwd_bias_group_defer(int G, const long* gy, const long* y, const long* geff, const long* gbias, float* ws, int B, int C,
                                int H, int W, long gy_bs, long y_bs, long geff_bs, int act, float act_a, float act_b,
                                int accumulate_bias, long* red_host, int red_cap, int* nred_host, void* stream);
/* Pure bias gradients (gbias[c] (+)= sum over (n, h, w) of gy) of a whole backward stage in one launch: cc_bias_grad_defer
 * launches nothing and fills job_host[12] (+ red_host[16], *nred_host = 1, when the map is summed in chunks: the second stage
 * goes to cc_wgrad_reduce_table); cc_bias_grad_table runs n parked jobs, 32 per launch, BEFORE that reduce table.  gy and ws
 * (cc_act_bwd_ws_bytes(C)) stay untouched in between.  Replaces: autograd's per-layer bias sum of nn.Conv2d
 * (torch ConvolutionBackward, as run by train.py:567 loss.backward()). */
int cc_bias_grad_defer(const float* gy, float* gbias, float* ws, int B, int C, int H, int W, long gy_bs, int accumulate,
                       long* job_host, long* red_host, int* nred_host);
int cc_bias_grad_table(const long* jobs_host, int n, void* stream);
int cc_act_bwd_bias(const float* gy, const float* y_or_null, float* geff_or_null, float* gbias_or_null, float* ws, int B,
                    int C, int H, int W, long gy_bs, long y_bs, long geff_bs, int act, float act_a, float act_b,
                    int accumulate_bias, void* stream);

/* ---------------------------------------------------------------- input pipeline, device side
 * custom_transforms.py:21-30,47-57 ArrayToTensor + Normalize with the crop of RandomScaleCrop (:93-121) and the mirror of
 * RandomHorizontalFlip (:60-73) folded in.  src: [N,H,W,3] uint8 or float32 (0..255); dst: [N,3,h,w];
 * geo: int32 [N,3] = (flip, off_y, off_x) per frame. */
int cc_frames_to_tensor(const void* src, int src_is_u8, float* dst, const int* geo, int N, int H, int W, int h, int w, float mean0,
                        float mean1, float mean2, float std0, float std1, float std2, void* stream);
/* ... with RandomScaleCrop's resize (custom_transforms.py:93-121: scipy.misc.imresize = byte-scale a float frame to its own
 * min..max, then Pillow's 8-bit bilinear resampler) on the device as well, bit-exact with the host path: RandomHorizontalFlip ->
 * resize to (sh_n, sw_n) -> crop h x w at (off_y_n, off_x_n) -> ArrayToTensor -> Normalize in 2 (uint8) / 4 (float32) launches.
 * geo: int32 [N,8] = (flip, off_y, off_x, sh, sw, htab, vtab, 0); tab: int32 resampling tables, per distinct output size
 * `first_input_index[size]` then `weights[size][KT]` (Pillow's 22-bit fixed-point coefficients, zero padded to KT taps);
 * htab / vtab are offsets into tab.  ws: cc_frames_resize_ws_bytes(N, H, tmp_w) bytes with tmp_w >= max_sw = max_n sw_n. */
size_t cc_frames_resize_ws_bytes(int N, int H, int tmp_w);
int cc_frames_resize_to_tensor(const void* src, int src_is_u8, float* dst, const int* geo, const int* tab, int KT, void* ws, int N,
                               int H, int W, int tmp_w, int max_sw, int h, int w, float mean0, float mean1, float mean2, float std0,
                               float std1, float std2, void* stream);
/* RandomRotate (custom_transforms.py:75-86; first transform of train.py:178-184's pipeline) on the device: scipy.misc.imrotate =
 * byte-scale + Pillow Image.rotate(angle, BILINEAR) restated in double (Geometry.c affine transform + bilinear_filter32RGB),
 * bit-exact with the host path.  dst_u8: uint8 [N,H,W,3]; rot: double [N,8] = (apply, a0..a5, 0), the coefficients Image.rotate
 * builds (apply == 0: byte-scale only, the first thing the resize that follows would do).  ws: cc_frames_rotate_ws_bytes(N). */
size_t cc_frames_rotate_ws_bytes(int N);
int cc_frames_rotate(const void* src, int src_is_u8, void* dst_u8, const double* rot, void* ws, int N, int H, int W, void* stream);

/* ---------------------------------------------------------------- BatchNorm2d, training mode
 * (models/DispResNet6.py:53-56: the Conv1x1 + BatchNorm2d shortcut of every ResNet stage; 13 per forward)
 * y = (x - mean_c) / sqrt(var_c + eps) * w_c + b_c with batch statistics over (B,H,W); running stats updated with
 * `momentum` (unbiased variance); save_mean / save_invstd [C] feed the backward.  ws: cc_bn_ws_bytes(C) bytes. */
size_t cc_bn_ws_bytes(int C);
int cc_bn_train_fwd(const float* x, const float* weight_or_null, const float* bias_or_null, float* running_mean_or_null,
                    float* running_var_or_null, float* y, float* save_mean, float* save_invstd, float* ws, int B, int C,
                    int H, int W, float momentum, float eps, void* stream);
/* gx; gweight[c] (+)= sum gy * xhat, gbias[c] (+)= sum gy (either may be null; accumulate_wb: add into them) */
int cc_bn_train_bwd(const float* gy, const float* x, const float* weight_or_null, const float* save_mean,
                    const float* save_invstd, float* gx, float* gweight_or_null, float* gbias_or_null, float* ws, int B, int C,
                    int H, int W, int accumulate_wb, void* stream);
/* eval mode (running statistics): y = (x - running_mean_c) / sqrt(running_var_c + eps) * w_c + b_c; grad_only != 0: the input
 * gradient of that map, y = x * w_c / sqrt(running_var_c + eps).  ws: 2*C floats. */
int cc_bn_eval_fwd(const float* x, const float* weight_or_null, const float* bias_or_null, const float* running_mean,
                   const float* running_var, float* y, float* ws, int B, int C, int H, int W, float eps, int grad_only,
                   void* stream);

/* ---------------------------------------------------------------- job-table forms of the loss path
 * loss_functions.py runs every photometric / consensus term once per (pyramid scale, reference frame): `for scale ... for ref`
 * loops of 24 (:80-128), 12 (:27-77) and 18 (:160-202) warp + SSIM evaluations per step, five of the six scales being
 * launch-latency sized.  A job-table call evaluates ALL terms of one loss in ONE launch: `jobs` is a HOST array of
 * njobs x 10 longs {slot0 .. slot7, H, W} (device addresses / packed integers, meaning per call below), every job spans B batch
 * items, njobs <= 24; blocks are dealt to jobs through a prefix table carried in the kernel arguments.
 *   cc_inverse_warp_fwd_jobs  slots: img, depth, P, Kinv, out                      (cc_inverse_warp_fwd per job)
 *   cc_inverse_warp_bwd_jobs  slots: gout, img, depth, P, Kinv, gdepth, gP partials [B][ceil(HW/256)][12]
 *   cc_flow_warp_fwd_jobs     slots: img, flow, out;      cc_flow_warp_bwd_jobs    slots: gout, img, flow, gflow
 *   cc_rigid_noocc_jobs       slots: depth, P4 [4,B,12], Kinv, out [B,4,H,W]       (loss_functions.py:132-137 per scale)
 *   cc_pose2flow_fwd_jobs     slots: depth, P, Kinv, flow                          (train.py:470-471 per scale)
 *   cc_ssim_photo_fwd_jobs    slots: tgt, warped, mask_a, mask_b, gmask, adjoint maps (4 x [B,3,H,W]), partials [B*tiles][4],
 *                             batch strides of mask_a | mask_b << 8 | gmask << 16 in units of H*W; + one finalize launch:
 *                             loss_accum += sum_j term_j (job order), scale_out[j], nan_flag
 *   cc_ssim_photo_bwd_jobs    slots: adjoint maps, tgt, warped, scale (1 float), gwarped
 *   cc_ssim_err_fwd_jobs      slots: tgt, warped, err [B,1,H,W], valid [B,1,H,W]   (loss_functions.py:181-188)
 *   cc_pose_proj_levels       pose [B,R,6], K [B,9] -> P_all [L][R][B][12], level l with K rows 0,1 / kdiv_host[l]
 *   cc_pose_grad_jobs         jobs (level-major, njobs = L*R) slot 0 = the gP partials of cc_inverse_warp_bwd_jobs -> gpose [B,R,6] */
int cc_inverse_warp_fwd_jobs(const long* jobs, int njobs, int B, int C, int padding_border, int align_corners, void* stream);
int cc_inverse_warp_bwd_jobs(const long* jobs, int njobs, int B, int C, int padding_border, int align_corners, void* stream);
int cc_flow_warp_fwd_jobs(const long* jobs, int njobs, int B, int C, int padding_border, int align_corners, void* stream);
int cc_flow_warp_bwd_jobs(const long* jobs, int njobs, int B, int C, int padding_border, int align_corners, void* stream);
int cc_rigid_noocc_jobs(const long* jobs, int njobs, int B, void* stream);
int cc_pose2flow_fwd_jobs(const long* jobs, int njobs, int B, int rewrite_oob, void* stream);
int cc_ssim_photo_fwd_jobs(const long* jobs, int njobs, int B, int mask_b_complement, int want_grad, float wssim, float q,
                           float lambda_oob, float* loss_accum, float* scale_out, float* nan_flag, const float* gauss13_host,
                           void* stream);
int cc_ssim_photo_bwd_jobs(const long* jobs, int njobs, int B, const float* gauss13_host, void* stream);
int cc_ssim_err_fwd_jobs(const long* jobs, int njobs, int B, float wssim, const float* gauss13_host, void* stream);
/*   cc_flow_noocc_jobs        slots: flow_bw, flow_fw, out [B,1,H,W]              (loss_functions.py:343-352 per scale)
 *   cc_consensus_target_jobs  slots: err_cam_fwd, err_cam_bwd, err_flow_fwd, valid_cam_fwd, valid_cam_bwd, target (:189-193)
 *   cc_sum_refs_scale_jobs    slots: gd_all [R][B][HW] (or 0), gdepth [B][HW], gmask [B][MC][HW] (or 0), scales (MC floats):
 *                             gdepth = sum_r gd_all[r]; gmask[:, c] *= scales[c]
 *   cc_edge_smooth_fwd_bwd_jobs  slots: img level, pred [B,C,H,W], gpred (or 0), this job's partials  (:287-319, all scales);
 *                             C = 0: the jobs of several terms share the launch (train.py:497-501), slot 4 = the job's channel count
 *   cc_bce_ones_fwd_bwd_jobs     slots: mask, gmask (or 0), partials; planes = B * C                  (:148-155)
 *   cc_consensus_bce_fwd_bwd_jobs slots: exp_mask, census_bwd, census_fwd, tgt_bwd, tgt_fwd, gmask (or 0), partials (:221-261)
 *   The three losses expect the per-job partial areas back to back starting at `partials` (cc_loss_jobs_num_blocks floats in
 *   total for `planes` planes per job) and add the sum to loss_accum in one finalize launch. */
/*   cc_elementwise_jobs       the element-wise glue of train.py:458,475-476,488 over all scales (planes per job):
 *                             op 0 out = 1/a (slots a, out); 1 ga = -g*y*y (g, y, ga); 2 out = |a-b| (a, b, out);
 *                             3 out[b,c] = 1 - m[b,c0+c] (m, out; planes = B*nc); 4 gm[b,c] = -g[b,c-c0] in [c0,c0+nc) else 0
 *                             (g, gm; planes = B*MC); 5 out = ((a*0.5+0.5) - mean_c)/std_c with the ImageNet statistics, c = plane % 3
 *                             (models/back2future.py:118-132 normalize of the three input images) */
int cc_elementwise_jobs(const long* jobs, int njobs, int planes, int op, int c0, int nc, int MC, void* stream);
int cc_flow_noocc_jobs(const long* jobs, int njobs, int B, void* stream);
int cc_consensus_target_jobs(const long* jobs, int njobs, int B, float wrig, void* stream);
int cc_sum_refs_scale_jobs(const long* jobs, int njobs, int B, int R, int MC, void* stream);
size_t cc_loss_jobs_num_blocks(const long* jobs, int njobs, int planes);
int cc_edge_smooth_fwd_bwd_jobs(const long* jobs, int njobs, int B, int C, float* partials, float* loss_accum, float gscale,
                                void* stream);
/* the edge weights exp(-mean_c |dI| ) of loss_functions.py:296-306 once per (image, scale): slots img level [B,3,H,W], out [B,2,H,W]
 * = (weight of the pair (p, p + 1), of (p, p + W)).  A job of cc_edge_smooth_fwd_bwd_jobs with this tensor in slot 5 reads the four
 * weights of a pixel instead of recomputing them per pixel and plane: same expression, bit-identical losses and gradients. */
int cc_edge_weights_jobs(const long* jobs, int njobs, int B, void* stream);
int cc_bce_ones_fwd_bwd_jobs(const long* jobs, int njobs, int planes, float* partials, float* loss_accum, float gscale, void* stream);
int cc_consensus_bce_fwd_bwd_jobs(const long* jobs, int njobs, int B, float* partials, float* loss_accum, float thresh, float wbce,
                                  float gscale, void* stream);
int cc_pose_proj_levels(const float* pose, const float* K, float* P_all, int L, int R, int B, const float* kdiv_host, void* stream);
int cc_pose_grad_jobs(const long* jobs, int njobs, int L, int R, int B, const float* pose, const float* K, float* gpose,
                      const float* kdiv_host, void* stream);

/* ---------------------------------------------------------------- x2 bilinear up-sampling
 * F.interpolate(scale_factor=2, mode='bilinear', align_corners=False) of the prediction maps, with the constant the callers
 * multiply in afterwards fused (models/DispResNet6.py:170-186 disp_up; models/back2future.py:196-285 `up_flow`,
 * `20 * upsample(...)`, `-0.625 * ...`):  y[B,C,2H,2W] = scale * up(x[B,C,H,W]).  x_bs / y_bs: batch strides in floats
 * (y may be a channel slice of a wider concat buffer; W even, y 16-byte aligned).  bwd: gx (+)= scale * up^T(gy), gather form. */
int cc_upsample2x_fwd(const float* x, float* y, int B, int C, int H, int W, long x_bs, long y_bs, float scale, void* stream);
int cc_upsample2x_bwd(const float* gy, float* gx, int B, int C, int H, int W, long gy_bs, long gx_bs, float scale, int accumulate,
                      void* stream);

/* ---------------------------------------------------------------- validation side (train.py:588-777)
 * The rigidity-mask composition of validate_flow_with_gt (train.py:673-687) in one pass; every output may be NULL:
 *   rigidity [B,1,H,W] = (1 - (1-exp[:,1])*(1-exp[:,2]) > 0.5);  census [B,H,W] = (|cam-fwd|_u < thresh)*(|cam-fwd|_v < thresh);
 *   combined [B,1,H,W] = 1 - (1-rigidity)*(1-census);  flow_non_rigid = (combined <= thresh)*flow_fwd;
 *   flow_rigid = (combined > thresh)*flow_cam;  total_flow = flow_rigid + flow_non_rigid  (all [B,2,H,W]);
 *   oob_rigid / oob_non_rigid [B,H,W] = flow2oob(flow_cam) / flow2oob(flow_fwd) (inverse_warp.py:222-238) as 0/1 floats.
 * exp_mask [B,MC,H,W] with MC >= 3.  Per-sample semantics: the reference runs this with batch size 1 only (train.py:236). */
int cc_rigidity_compose(const float* exp_mask, int MC, const float* flow_cam, const float* flow_fwd, float* rigidity,
                        float* census, float* combined, float* flow_non_rigid, float* flow_rigid, float* total_flow,
                        float* oob_rigid, float* oob_non_rigid, float thresh, int B, int H, int W, void* stream);

/* ---------------------------------------------------------------- optimizer (train.py:307-310,568)
 * torch.optim.Adam(betas, eps, weight_decay=0) on the flat fp32 bucket; grads are multiplied by grad_scale first
 * (1/world_size after the RCCL all-reduce).  step_dev: device float, incremented by the call. */
int cc_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* step_dev, long n, float lr,
                 float beta1, float beta2, float eps, float grad_scale, void* stream);
/* ... on a sub-range of the bucket (base pointers of the range; tick = 0: do not advance the step counter -- the second and
 * later segments of one optimizer step), so that a segment can be updated while the next one's all-reduce is in flight. */
int cc_adam_step_segment(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* step_dev, long n, float lr,
                         float beta1, float beta2, float eps, float grad_scale, int tick, void* stream);
/* step_dev += 1 alone: the per-network pipeline (cc_amd/trainer.py, round 6) advances the counter ONCE at the start of the step and
 * then updates the networks' bucket segments with tick = 0 from different streams, as their gradients arrive. */
int cc_adam_tick(float* step_dev, void* stream);
int cc_fill(float* p, long n, float value, void* stream);

#ifdef __cplusplus
}
#endif
#endif
