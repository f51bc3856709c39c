/* libsr3d -- C ABI of the MI355X-native voxel super-resolution training hot path.
 *
 * The reference (YukiYasuda2718/3d-sr-micrometeorology) has no FFI of its own:
 * every operation on its hot path is a call into PyTorch/ATen (SURVEY.md section
 * 2b, 8b).  Each entry point below replaces one such call site; the reference
 * line it stands in for is cited next to it.  A Python/ctypes binding is in
 * 3d-sr-micrometeorology_amd/_lib.py; INTEGRATION.md shows how the reference's
 * modules would call it.
 *
 * Conventions
 *  - all tensors are dense, layout NCDHW = (B, C, z, y, x), x contiguous.  Element type: fp32 everywhere (the
 *    reference's precision, dataset.py:29), or -- `dtype = SR3D_DTYPE_BF16` in the convolution descriptor and in the
 *    activation-backward / bias-gradient calls -- ACTIVATIONS AND THEIR GRADIENTS in bfloat16 (BASELINE configs[4]:
 *    bf16 storage + bf16 MFMA with fp32 accumulation).  Weights, biases, weight / bias gradients, the loss and the
 *    optimizer state are fp32 in both modes (fp32 master weights);
 *  - every pointer is a DEVICE pointer unless stated; the library never
 *    allocates, frees or caches device memory -- workspaces are passed in and
 *    sized by the *_bytes() queries;
 *  - every call is asynchronous on the hipStream_t handed in (passed as void*)
 *    and re-entrant (autograd calls backward from another thread);
 *  - return value 0 = ok, negative = SR3D_E_*; sr3d_last_error() gives the
 *    message of the calling thread's last failure.  Nothing throws.
 */
#ifndef SR3D_H_
#define SR3D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libsr3d.so is linked with -fvisibility=hidden: what this header declares is ALL the library exports */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define SR3D_VERSION 100 /* 0.1.0 */

enum {
  SR3D_OK = 0,
  SR3D_E_ARG = -1,     /* bad argument / unsupported shape */
  SR3D_E_HIP = -2,     /* a HIP runtime call failed */
  SR3D_E_WORKSPACE = -3 /* workspace too small */
};

/* element type of activations and activation gradients (see Conventions) */
enum { SR3D_DTYPE_F32 = 0, SR3D_DTYPE_BF16 = 1 };

/* activations fused into conv epilogues (custom_conv.py:111-126, unet.py:35,84,105) */
enum { SR3D_ACT_NONE = 0, SR3D_ACT_RELU = 1, SR3D_ACT_LRELU = 2 /* slope 0.01 */,
       SR3D_ACT_OUT_F32 = 0x100 /* flag OR-ed to `act` of sr3d_conv3d_fwd: see there */,
       SR3D_ACT_FROM_Y = 0x200 /* flag OR-ed to `act` of sr3d_gated_act_bwd: see there */,
       SR3D_ACT_UNSHUFFLE = 0x400 /* flag OR-ed to `act` of sr3d_conv3d_bwd_data_act / _fuses_act: see there */ };

/* One operand of a virtual channel concatenation (replaces torch.cat at
 * unet.py:255-293): `channels` channels of a (B, channels, Z, Y, X) tensor.
 * As a gradient destination ptr may be NULL (that slice needs no gradient). */
typedef struct {
  void* ptr;
  int32_t channels;
} sr3d_slice_t;

/* 3x3x3 convolution, padding 1 (every nn.Conv3d of the model: unet.py:30,44,76,
 * 89,103,196,243).  (Z,Y,X) is the INPUT grid; the output grid is
 * floor((Z-1)/stride)+1 per dim.  For gated layers Cout is the width of ONE
 * branch. */
typedef struct {
  int32_t B, Cin, Cout;
  int32_t Z, Y, X;
  int32_t stride; /* 1 or 2 */
  int32_t dtype;  /* SR3D_DTYPE_*: element type of x / y / saved activations / dy / dx */
} sr3d_conv_desc_t;

/* kinds of packed (MFMA-tile-ordered) weight images; the two BWD kinds are built inside
 * sr3d_conv3d_bwd_data (in its workspace) and are not accepted by sr3d_pack_weights */
enum {
  SR3D_PACK_FWD = 0,        /* plain conv forward                              */
  SR3D_PACK_FWD_GATED = 1,  /* feature + gate branches interleaved by 32 rows  */
  SR3D_PACK_BWD = 2,        /* transposed image, rows = input channels that need a gradient */
  SR3D_PACK_BWD_GATED = 3,  /* same with K = [d_feat ; d_gate]                 */
  SR3D_PACK_FWD_UNSHUFFLE = 4 /* plain conv forward of a layer that is called with unshuffle != 0 (UpBlock.up): the
                                 image may hold its rows in voxel-unshuffle order so that the two x-neighbours of an
                                 output voxel are stored together; an image packed as SR3D_PACK_FWD works too */
};

int sr3d_version(void);
const char* sr3d_last_error(void);

/* ---- weights ------------------------------------------------------------ */
/* bytes of the packed image of one layer */
size_t sr3d_packed_weight_bytes(const sr3d_conv_desc_t* d, int kind);
/* w_feat / w_gate: (Cout, Cin, 3,3,3) as in the state_dict; w_gate only for *_GATED */
int sr3d_pack_weights(const sr3d_conv_desc_t* d, int kind, const void* w_feat, const void* w_gate,
                      void* w_packed, void* stream);

/* ---- forward ------------------------------------------------------------ */
/* y = act(conv3d(cat(x_srcs); W) + bias)           nn.Conv3d + LeakyReLU: unet.py:196-198, 72-97, 240-246
 * unshuffle != 0: y is written as unshuffle_voxels(., 2) of that (voxel_shuffle.py:26-42, unet.py:99-108),
 *                 i.e. y has Cout/8 channels on the 2x grid.
 * act | SR3D_ACT_OUT_F32 (bf16 storage, stride 1, no unshuffle): y is an fp32 tensor -- the network's prediction (`last`,
 *                 unet.py:240-246, 295) leaves the engine unrounded, the accumulator's value. */
int sr3d_conv3d_fwd(const sr3d_conv_desc_t* d, const sr3d_slice_t* x_srcs, int n_src, const void* w_packed,
                    const void* bias, void* y, int act, int unshuffle, void* x_absmax, void* stream);

/* Operand maxima for the split-f16 weight gradient, without a separate sweep of the tensors.
 * The split kernels scale their operands by powers of two taken from max |x| and max |dY| per slice.  Kernels that see
 * every element anyway can export those maxima: `x_absmax` of the two forward calls (optional; 4 x 64 uint32, ZEROED by the
 * caller; slice i's maximum is the largest of [64 i, 64 i + 64) read as float bits) is filled when
 * sr3d_conv3d_fwd_exports_absmax() says so for this layer (1: the forward runs on the split-f16 kernel; 0: it does not,
 * the buffer stays untouched and must not be handed on); `absmax_out` of the three activation-backward calls (64 uint32
 * per output tensor, zeroed by the caller, fp32 only).  sr3d_conv3d_bwd_weight takes them as `x_absmax` ([n_src][64]) and
 * `dy_absmax` ([n_dy][64]); NULL = compute by a sweep. */
int sr3d_conv3d_fwd_exports_absmax(const sr3d_conv_desc_t* d, int gated);

/* y = sigmoid(conv(x;Wg)+bg) * act(conv(x;Wf)[+bf])   custom_conv.py:119-123, 303-304
 * save_f = act(feat), save_s = sigmoid(gate) are kept for the backward pass. */
int sr3d_gated_conv3d_fwd(const sr3d_conv_desc_t* d, const sr3d_slice_t* x_srcs, int n_src, const void* w_packed,
                          const void* bias_f, const void* bias_g, void* y, void* save_f, void* save_s, int act,
                          void* x_absmax, void* stream);

/* ---- backward (autograd of the above: optim_helper.py:165 loss.backward()) -- */
/* dx_dsts[i] = slice i of d(cat(x)) = conv_transpose(cat(dy_srcs); W)  (aten convolution_backward, input grad)
 * dy_srcs hold the gradient w.r.t. the PRE-activation conv outputs (1 slice, or 2 = [d_feat, d_gate]);
 * w_feat / w_gate are the (Cout, Cin, 3,3,3) state_dict tensors (w_gate only when n_dy == 2).
 * Destination slices with a NULL ptr (building mask, network input) are not computed at all: their rows
 * are left out of the GEMM.  The workspace receives the transposed weight image. */
size_t sr3d_conv3d_bwd_data_workspace_bytes(const sr3d_conv_desc_t* d, int n_dy);
int sr3d_conv3d_bwd_data(const sr3d_conv_desc_t* d, const sr3d_slice_t* dy_srcs, int n_dy, const void* w_feat,
                         const void* w_gate, const sr3d_slice_t* dx_dsts, int n_dst, void* workspace,
                         size_t workspace_bytes, void* stream);

/* The same with the activation backward of ONE destination slice fused into the epilogue: slice `act_slice` of
 * cat(x) is the output y of a LeakyReLU layer (act = SR3D_ACT_LRELU; `act_y` = that tensor, same shape as the slice), and what
 * is stored there is dL/dy * lrelu'(y) = dL/dpre of THAT layer -- its lrelu_bwd pass (autograd's leaky_relu_backward,
 * pytorch/model/unet.py:72-97,192-199) never runs; in fp32 the stored values are bit-identical to sr3d_conv3d_bwd_data followed
 * by sr3d_lrelu_bwd.  act_absmax (optional, [64] zeroed words, fp32 storage): max |stored| for that layer's weight gradient.
 * Only launches for which sr3d_conv3d_bwd_data_fuses_act answers 1 have this epilogue (the split-f16 / bf16 stride-1 kernel).
 * act | SR3D_ACT_UNSHUFFLE: the slice is the voxel-UNSHUFFLED output of a conv + LeakyReLU layer (UpBlock.up, unet.py:99-108):
 * act_y has the slice's own (fine-grid) layout, and dx_dsts[act_slice].ptr -- a buffer with the slice's element count --
 * receives dL/dpre of THAT layer in ITS layout, (B, 8 C, Z/2, Y/2, X/2) for a slice of C channels (what
 * sr3d_unshuffle_lrelu_bwd would have produced).  Needs X % 4 == 0. */
int sr3d_conv3d_bwd_data_fuses_act(const sr3d_conv_desc_t* d, int n_dy, const sr3d_slice_t* dx_dsts, int n_dst, int act_slice, int act);
int sr3d_conv3d_bwd_data_act(const sr3d_conv_desc_t* d, const sr3d_slice_t* dy_srcs, int n_dy, const void* w_feat,
                             const void* w_gate, const sr3d_slice_t* dx_dsts, int n_dst, int act_slice, const void* act_y,
                             int act, void* act_absmax, void* workspace, size_t workspace_bytes, void* stream);

/* dw[(n, c, kz,ky,kx)] over n in cat(dy_srcs) channels (so for gated layers dw = [dWf ; dWg]).
 * deterministic: fixed split of the voxel reduction + ordered second stage. */
size_t sr3d_conv3d_bwd_weight_workspace_bytes(const sr3d_conv_desc_t* d, int n_total);
int sr3d_conv3d_bwd_weight(const sr3d_conv_desc_t* d, const sr3d_slice_t* x_srcs, int n_src,
                           const sr3d_slice_t* dy_srcs, int n_dy, void* dw, void* workspace, size_t workspace_bytes,
                           const void* x_absmax, const void* dy_absmax, void* stream);

/* db[c] = sum_{b,voxels} dy[b,c,:]   (bias gradient, fp32); workspace >= sr3d_bias_grad_workspace_bytes.
 * `dtype` (here and in the three activation-backward calls below): element type of dy / y / saved tensors / outputs */
size_t sr3d_bias_grad_workspace_bytes(int B, int C, long long voxels);
int sr3d_bias_grad(const void* dy, int B, int C, long long voxels, void* db, void* workspace, int dtype, void* stream);

/* d_feat = dy * s * act'(f),  d_gate = dy * f * s * (1 - s)      (autograd of custom_conv.py:119-123); f = act(feat), s = sigmoid(gate)
 * act | SR3D_ACT_FROM_Y: `save_f` is the layer's OUTPUT y = s * f instead (s > 0: y has the sign of f, and f * s = y), so the
 * forward pass need not store act(feat) at all (sr3d_gated_conv3d_fwd with save_f = NULL, save_s given) */
int sr3d_gated_act_bwd(const void* dy, const void* save_f, const void* save_s, void* d_feat, void* d_gate,
                       long long n, int act, int dtype, void* absmax_out /* [2][64]: d_feat, d_gate */, void* stream);
/* the same with dy + dy2 as the incoming gradient (dy2 may be NULL): a gated output that feeds TWO consumers -- the next block and the
 * U-Net's skip connection (unet.py:262-283: f0..f3) -- receives two gradient tensors, which autograd would add in a pass of its own
 * (AccumulateGrad / the engine's sum of a node's incoming gradients) before this one; the sum is formed in fp32 */
int sr3d_gated_act_bwd_sum(const void* dy, const void* dy2, const void* save_f, const void* save_s, void* d_feat, void* d_gate,
                           long long n, int act, int dtype, void* absmax_out /* [2][64] */, void* stream);
/* dpre = dy * (y > 0 ? 1 : 0.01)        (autograd of nn.LeakyReLU, y = post-activation) */
int sr3d_lrelu_bwd(const void* dy, const void* y, void* dpre, long long n, int dtype, void* absmax_out, void* stream);
/* dpre(B, 8C, Z, Y, X) = shuffle_voxels(dy * lrelu'(y)) for y, dy of shape (B, C, 2Z, 2Y, 2X)
 * (autograd of unet.py:99-108 up to the conv) */
int sr3d_unshuffle_lrelu_bwd(const void* dy, const void* y, void* dpre, int B, int C, int Z, int Y, int X, int dtype,
                             void* absmax_out, void* stream);

/* ---- small data-movement ops ------------------------------------------------ */
/* x0 = cat[nearest_upsample(x, scale), b]     unet.py:143,254-255 ; x: (B,C,Z/s,Y/s,X/s), b: (B,1,Z,Y,X) */
int sr3d_upsample_cat(const void* x, const void* b, void* x0, int B, int C, int Z, int Y, int X, int scale,
                      void* stream);
/* out = AvgPool3d(2,2)(in) for a (B,1,Z,Y,X) mask          unet.py:156,261 */
int sr3d_avgpool2(const void* in, void* out, int B, int Z, int Y, int X, void* stream);
/* near = calc_mask_near_build_wall(b)                       loss_maker.py:57-83 */
int sr3d_near_wall(const void* b, void* near, int B, int Z, int Y, int X, void* stream);

/* Sample preprocessing of the reference's Dataset on the device (dataset.py:139-161, 174, 191-195):
 * out = nan_to_num(clamp?((scaling * x - means[c]) / stds[c], 0, 1), nan = nan_value) for x: (B, C, Z, Y, X) in
 * physical units (C <= 8; means / stds are HOST arrays); z levels below discard_z are then set to nan_value
 * (max_discarded_lr_z_index).  Bit-identical to the CPU expressions of the reference. */
int sr3d_preprocess(const void* x, void* out, int B, int C, int Z, int Y, int X, const float* means, const float* stds,
                    float scaling, int clip, float nan_value, int discard_z, void* stream);

/* ---- PartialConv3d (model/custom_conv.py:129-234; not reachable from UNetSR, kept as an op) ---------------- */
/* update_mask = clamp(s, 0, 1), mask_ratio = slide_winsize / (s + 1e-8) * update_mask with s = sum of the mask over the
 * zero-padded 3x3x3 window and its Cm channels; mask: (Bm, Cm, Z, Y, X) -> both outputs (Bm, 1, OZ, OY, OX). */
int sr3d_pconv_mask_update(const void* mask, int Bm, int Cm, int Z, int Y, int X, int stride, float slide_winsize,
                           void* update_mask, void* mask_ratio, void* stream);
/* out = x * mask, the mask broadcast over batch (Bm = 1) and / or channels (Cm = 1) */
int sr3d_mul_mask(const void* x, const void* mask, void* out, int B, int C, long long voxels, int Bm, int Cm,
                  void* stream);
/* forward (backward = 0): out = bias ? ((in - bias[c]) * ratio + bias[c]) * update : in * ratio
 * backward (= 1): out = in * ratio * (bias ? update : 1); bias_terms (optional) = in * update * (1 - ratio), whose
 * per-channel sum (sr3d_bias_grad) is the bias gradient */
int sr3d_pconv_scale(const void* in, const void* bias, const void* update_mask, const void* mask_ratio, void* out,
                     void* bias_terms, int B, int C, long long voxels, int Bm, int backward, void* stream);

/* ---- losses (forward value + dL/dp in one pass) --------------------------- */
/* out[0] = mean|p-t| ; dLdp = sign(p-t)/n * grad_scale           MyL1Loss, loss_maker.py:194-202 */
size_t sr3d_loss_workspace_bytes(int B, int Z, int Y, int X);
int sr3d_l1_fwd_bwd(const void* p, const void* t, long long n, void* loss_out, void* dLdp, void* workspace,
                    void* stream);
/* MixedDivergenceGradientL2Loss, loss_maker.py:358-450.  p,t: (B,4,Z,Y,X); b: (B,1,Z,Y,X).
 * terms_out (device, 4 floats) = {mse, grd_mse, div_mse, total};  dLdp may be NULL. */
int sr3d_mixed_div_grad_l2_fwd_bwd(const void* p, const void* t, const void* b, int B, int Z, int Y, int X,
                                   const float scales[3], float delta_meter, float w_g, float w_d,
                                   void* terms_out, void* dLdp, void* workspace, void* stream);

/* Gradient of WeightedL1Loss / WeightedL2Loss (loss_maker.py:216-255):
 *   dLdp = e'(p - t) * (b * coef[0] + (1 - b) * coef[1]),   e = |.| (power 1) or (.)^2 (power 2),
 * coef: 2 floats in DEVICE memory (the region weights divided by their voxel counts, times the upstream gradient). */
int sr3d_weighted_lp_bwd(const void* p, const void* t, const void* b, int B, int C, long long voxels, int power,
                         const void* coef, void* dLdp, void* stream);

/* dLdp = d(wts[0]*mse + wts[1]*grd_mse + wts[2]*div_mse)/dp with the three weights read from DEVICE memory
 * (autograd / GradNorm, gradnorm.py:95-100, differentiate individual terms); reuses the workspace a previous
 * sr3d_mixed_div_grad_l2_fwd_bwd call with the same arguments filled. */
int sr3d_mixed_div_grad_l2_bwd(const void* p, const void* t, int B, int Z, int Y, int X, const float scales[3],
                               float delta_meter, float w_g, float w_d, const void* term_weights, void* dLdp,
                               void* workspace, void* stream);

/* ---- evaluation metrics (forward-only test pass) -------------------------------- */
/* Every metric of the reference's final evaluation (script/train_model.py:366-379; modules at loss_maker.py:194-213,
 * 522-741) from ONE pass over prediction, target and building mask.  p, t: (B,4,Z,Y,X); b: (B,1,Z,Y,X);
 * stds = config['data']['stds'] (temperature scale, then the three velocity scales); `lev` is the z index of the
 * *_LEV entries (the reference uses lev = 0).  out: SR3D_EVAL_COUNT floats in DEVICE memory. */
enum {
  SR3D_EVAL_L1 = 0,                      /* MyL1Loss                      loss_maker.py:194-202 */
  SR3D_EVAL_L2 = 1,                      /* MyL2Loss                      :205-213 */
  SR3D_EVAL_MASKED_L1 = 2,               /* MaskedL1Loss                  :522-536 */
  SR3D_EVAL_MASKED_L2 = 3,               /* MaskedL2Loss                  :539-553 */
  SR3D_EVAL_MASKED_L1_NEAR_WALL = 4,     /* MaskedL1LossNearWall          :556-574 */
  SR3D_EVAL_MASKED_L2_NEAR_WALL = 5,     /* MaskedL2LossNearWall          :577-595 */
  SR3D_EVAL_RESIDUAL_CONTINUITY = 6,     /* ResidualContinuity.forward    :598-615 */
  SR3D_EVAL_RESIDUAL_CONTINUITY_TARGET = 7, /* ... .calc_both_pred_and_target, target part   :617-633 */
  SR3D_EVAL_ABS_DIFF_TEMPERATURE = 8,    /* AbsDiffTemperature            :674-703 */
  SR3D_EVAL_DIFF_VELOCITY_NORM = 9,      /* DiffVelocityVectorNorm        :636-671 */
  SR3D_EVAL_ABS_DIFF_TEMPERATURE_LEV = 10, /* the same two at z == lev */
  SR3D_EVAL_DIFF_VELOCITY_NORM_LEV = 11,
  SR3D_EVAL_ABS_DIFF_DIVERGENCE = 12,    /* AbsDiffDivergence             :706-727 */
  SR3D_EVAL_DIFF_OMEGA_NORM = 13,        /* DiffOmegaVectorNorm           :730-745 */
  SR3D_EVAL_SUM_ABS = 14,                /* raw sums over the 4 channels: sum |p - t|, sum b |p - t|,            */
  SR3D_EVAL_SUM_MASK_ABS = 15,           /* sum (p - t)^2, sum b (p - t)^2 and sum b (one channel) -- what the      */
  SR3D_EVAL_SUM_SQ = 16,                 /* weighted losses (loss_maker.py:216-255) are made of                     */
  SR3D_EVAL_SUM_MASK_SQ = 17,
  SR3D_EVAL_SUM_MASK = 18,
  SR3D_EVAL_COUNT = 19
};
size_t sr3d_eval_metrics_workspace_bytes(int B, int Z, int Y, int X);
int sr3d_eval_metrics(const void* p, const void* t, const void* b, int B, int Z, int Y, int X, const float stds[4],
                      float delta_meter, int lev, void* out, void* workspace, void* stream);

/* SSIM3D (reference src/ssim.py:52-115: masked structural similarity with a w x w x w Gaussian or uniform window,
 * zero padding).  img1, img2: (B, C, Z, Y, X); mask: (B, Cm, Z, Y, X) with Cm = 1 or C; `window`: the n 1-D taps
 * (HOST array, n odd, <= 15) whose triple outer product is the reference's 3-D window; mean_out: 1 float (device) =
 * mean of the SSIM map; ssim_map (optional): the map itself (size_average = False). */
size_t sr3d_ssim3d_workspace_bytes(int B, int C, int Z, int Y, int X);
int sr3d_ssim3d(const void* img1, const void* img2, const void* mask, int B, int C, int Cm, int Z, int Y, int X,
                const float* window, int n, float max_val, float eps, void* mean_out, void* ssim_map, void* workspace,
                void* stream);

/* ---- optimizer -------------------------------------------------------------- */
/* torch.optim.Adam defaults (train_model.py:183) on one flat fp32 buffer; step is 1-based.  The
 * hyper-parameters are doubles, as in torch (1 - beta2 must be formed in double to match it). */
int sr3d_adam_step(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, long long n, double lr,
                   double beta1, double beta2, double eps, int step, double grad_scale, void* stream);

/* The same update with the step number kept in DEVICE memory: `step_counter` (int32, device) is incremented by the call
 * itself and the bias corrections are derived from it on the device (`scalars`: 2 floats of device scratch), so a
 * training step captured into a hipGraph advances correctly on every replay (the host-side `step` of sr3d_adam_step
 * would be frozen into the graph). */
int sr3d_adam_step_device_counter(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, long long n, double lr,
                                  double beta1, double beta2, double eps, void* step_counter, void* scalars,
                                  double grad_scale, void* stream);

/* ---- measurement hook (bench.py's roofline leg; no reference counterpart) --------
 * When enabled, every launch of the kernels below is bracketed by two HIP events (taken from a pool that
 * sr3d_profile_enable(1) creates, so nothing is allocated inside a timed region) on the stream it is launched on.
 * sr3d_profile_read() returns, for one kernel family, the summed event time, the summed ALGORITHMIC work and the
 * number of launches since the last enable.  Work = FLOPs (2 * 27 * Cin * Cout * output voxels, padding not counted)
 * for the convolution families, BYTES (compulsory reads + writes of the operation) for the HBM-bound families. */
enum {
  SR3D_PROF_IGEMM_S1 = 0,     /* stride-1 conv forward and stride-1 input gradient           [FLOP] */
  SR3D_PROF_IGEMM_S2 = 1,     /* stride-2 conv forward                                       [FLOP] */
  SR3D_PROF_IGEMM_BWD_S2 = 2, /* stride-2 input gradient (8 parity classes)                  [FLOP] */
  SR3D_PROF_WGRAD = 3,        /* weight gradient (main kernel, without the slab reduce)      [FLOP] */
  SR3D_PROF_LOSS = 4,         /* L1 / mixed loss: value + dL/dp (all passes)                 [byte] */
  SR3D_PROF_ACT_BWD = 5,      /* gated / LeakyReLU / unshuffle activation backward           [byte] */
  SR3D_PROF_BIAS_GRAD = 6,    /* bias gradient (both stages)                                 [byte] */
  SR3D_PROF_ADAM = 7,         /* fused Adam                                                  [byte] */
  SR3D_PROF_DATA = 8,         /* upsample+concat, mask pyramid, near-wall mask               [byte] */
  SR3D_PROF_PACK = 9,         /* weight packing / transforms and split-K reductions          [byte] */
  SR3D_PROF_EVAL = 10,        /* fused evaluation metrics                                    [byte] */
  SR3D_PROF_HCONV = 11,       /* stride-1 conv on the split-f16 kernel (SR3D_SPLIT_F16=1): launches of at least 448
                                 workgroups, which fill the chip's 512 slots                                  [FLOP] */
  SR3D_PROF_HCONV_SMALL = 12, /* ... its launches below that (U-Net levels 3-4: small grids, one partial round) [FLOP] */
  SR3D_PROF_FAMILIES = 13,
  SR3D_PROF_DROPPED = 99      /* launches: records lost because the event pool was exhausted */
};
/* on = 1: every family; on = 2: only the stride-1 convolution families (SR3D_PROF_HCONV, SR3D_PROF_HCONV_SMALL, SR3D_PROF_IGEMM_S1): bench.py
 * brackets only the dominant kernel inside its timed region (48 instead of ~1100 event records per training step) and
 * collects the other families in a separate, untimed pass; on = 0: off. */
int sr3d_profile_enable(int on);
int sr3d_profile_read(int kernel_id, double* ms, double* work, long long* launches);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* SR3D_H_ */
