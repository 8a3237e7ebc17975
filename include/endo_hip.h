/*
 * endo_hip.h -- C ABI of libendo_hip.so, the MI355X (gfx950) implementation of the training hot
 * path of EndoscopyDepthEstimation-Pytorch.
 *
 * The reference has no native layer: its "operator interface" for this path is a set of
 * torch.nn.Module classes whose forward() bottoms out in ATen kernels.  Each
 * entry point below replaces the ATen op sequence behind one of those modules; the file:line it
 * replaces is cited on every declaration.  The Python mirror of the reference's module API
 * (endoscopydepthestimation-pytorch_amd/{models,losses}.py) binds these symbols with ctypes --
 * INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - all pointers are DEVICE pointers to contiguous fp32 (or, where said, fp64 / u8 / i32) data on
 *     the current HIP device; images are NCHW; hw = H*W
 *   - the caller owns every buffer; the library keeps no pointer past the call (network handles
 *     excepted: endo_net_create/destroy own their descriptor tables only, never activations)
 *   - every call launches asynchronously on `stream` (a hipStream_t passed as void*), never
 *     synchronises the device, and is re-entrant per stream
 *   - return value: 0 = ok, > 0 = hipError_t, < 0 = argument error (ENDO_E_*)
 *   - `stats` / `work` arguments are small fp64 device scratch arrays that the call zeroes itself;
 *     the forward call's `stats` must be handed unchanged to the matching backward call
 */
#ifndef ENDO_HIP_H
#define ENDO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ENDO_E_BADARG (-1)
#define ENDO_E_UNSUPPORTED (-2)

/* library identification: returns ENDO_ABI_VERSION (bumped whenever the set of entry points or a signature changes).
 * 2: round 2 (adds the tiled warp entries, endo_relative_poses, endo_loss_head, endo_set_option, endo_net_tape_offset,
 * endo_jpeg_*, endo_point_brightness).
 * 3: round 3 -- kernel-form / precision options move from the process to the network handle: endo_net_set_option /
 * endo_net_get_option REPLACE endo_set_option and endo_set_wgrad_overlap (removed; no environment defaults any more);
 * adds endo_warp_consistency and endo_warp_fallback_blocks.
 * 4: round 3 -- the 16-bit-storage family (endo_net16_*, endo_net16h_*, endo_bf16_*, endo_f16_*).
 * 5: round 4 -- the non-finite-loss guard moves onto the device: endo_loss_head writes a FOURTH float (the flag),
 * endo_sgd_clip_step takes a `skip_flag` device pointer; endo_net16_offset what = 7; adds endo_hsv_full. */
#define ENDO_ABI_VERSION 6
int endo_abi_version(void);
/* hipGetErrorString for positive codes, a fixed string for ENDO_E_* */
const char* endo_error_string(int code);

/* ---------------------------------------------------------------------------------------------
 * DepthScalingLayer.forward -- reference models.py:346-363
 * stats: n x 8 fp64  [sum sd*bin, sum bin, sum smap, sum above, sum smap^2, scale, std, -]
 * ratio: 1 fp32 = mean over the (N x N) broadcast of std_j / scale_i (the reference divides a
 *        (N,) tensor by a (N,1,1,1) tensor before torch.mean -- models.py:363)
 * ------------------------------------------------------------------------------------------- */
int endo_depth_scale_fwd(const float* pred, const float* sparse_depth, const float* sparse_mask,
                         float* scaled, float* ratio, double* stats,
                         int n, int hw, float eps, void* stream);
/* grad_scaled / grad_ratio may be NULL (output unused); work: n fp64 */
int endo_depth_scale_bwd(const float* grad_scaled, const float* grad_ratio,
                         const float* pred, const float* sparse_depth, const double* stats,
                         float* grad_pred, double* work,
                         int n, int hw, float eps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * FlowfromDepthLayer.forward -- reference models.py:370-374 -> 433-451 -> 377-429
 * t: n x 3, R: n x 9 (row major), K: n x 9; flow: n x 2 x H x W
 * ------------------------------------------------------------------------------------------- */
int endo_flow_from_depth_fwd(const float* depth, const float* mask, const float* t, const float* R,
                             const float* K, float* flow, int n, int h, int w, void* stream);
int endo_flow_from_depth_bwd(const float* grad_flow, const float* depth, const float* mask,
                             const float* t, const float* R, const float* K, float* grad_depth,
                             int n, int h, int w, void* stream);

/* ---------------------------------------------------------------------------------------------
 * DepthWarpingLayer.forward -- reference models.py:460-465 -> 469-554, sampler models.py:325-336
 * (F.grid_sample bilinear / zeros / align_corners=False on the grid (2u/W-1, 2v/H-1))
 * warped, intersect: n x 1 x H x W.  Backward: grad_d1 written, grad_d2 zeroed then scatter-added.
 * ------------------------------------------------------------------------------------------- */
int endo_depth_warp_fwd(const float* depth_1, const float* depth_2, const float* mask,
                        const float* t, const float* R, const float* K,
                        float* warped, float* intersect,
                        int n, int h, int w, float eps, void* stream);
int endo_depth_warp_bwd(const float* grad_warped, const float* depth_1, const float* depth_2,
                        const float* mask, const float* t, const float* R, const float* K,
                        float* grad_d1, float* grad_d2,
                        int n, int h, int w, float eps, void* stream);
/* The same with an explicit LDS source-tile shape (geometry.hip "LDS-staged depth warp"): tile_h x tile_w in
 * {8x32, 16x32, 16x64, 32x32, 32x64}, or 0 x 0 for the L2-gather kernels; ENDO_E_UNSUPPORTED otherwise.  The entry points
 * above use the shape the sweep under profiles/ settled on.  Results do not depend on the shape (forward bit-identical,
 * the d2 gradient up to the order of its atomic additions). */
int endo_depth_warp_fwd_tiled(const float* depth_1, const float* depth_2, const float* mask,
                              const float* t, const float* R, const float* K,
                              float* warped, float* intersect,
                              int n, int h, int w, float eps, int tile_h, int tile_w, void* stream);
int endo_depth_warp_bwd_tiled(const float* grad_warped, const float* depth_1, const float* depth_2,
                              const float* mask, const float* t, const float* R, const float* K,
                              float* grad_d1, float* grad_d2,
                              int n, int h, int w, float eps, int tile_h, int tile_w, void* stream);
/* Test hook: blocks of the tiled kernels since the last reset whose source box did not fit the LDS staging buffers and that took
 * the gather path instead (large / divergent motion, e.g. the gap-scaled poses of BASELINE configs[4]); per process and device.
 * Copies two counters from the device (synchronises); either pointer may be null. */
int endo_warp_fallback_blocks(long long* forward, long long* backward, int reset);


/* ---------------------------------------------------------------------------------------------
 * SparseMaskedL1Loss.forward -- reference losses.py:62-66
 * flows, flows_hat: n x c x H x W; mask: n x 1 x H x W; stats: n x 2 fp64 [sum m|f-f^|, sum m]
 * ------------------------------------------------------------------------------------------- */
int endo_sparse_l1_fwd(const float* flows, const float* flows_hat, const float* mask,
                       float* loss, double* stats, int n, int c, int hw, float eps, void* stream);
/* grad_flows / grad_hat may be NULL when that input needs no gradient */
int endo_sparse_l1_bwd(const float* grad_loss, const float* flows, const float* flows_hat,
                       const float* mask, const double* stats, float* grad_flows, float* grad_hat,
                       int n, int c, int hw, float eps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * NormalizedDistanceLoss.forward -- reference losses.py:122-146
 * stats: n x 4 fp64 [sum m*d, sum m, sum m|P-Pw|_1, sum m(d+|dw|)]
 * ------------------------------------------------------------------------------------------- */
int endo_norm_dist_fwd(const float* depth, const float* warped, const float* intersect,
                       const float* K, float* loss, double* stats,
                       int n, int h, int w, float eps, void* stream);
int endo_norm_dist_bwd(const float* grad_loss, const float* depth, const float* warped,
                       const float* intersect, const float* K, const double* stats,
                       float* grad_depth, float* grad_warped,
                       int n, int h, int w, float eps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ScaleInvariantLoss.forward -- reference losses.py:22-32
 * stats: n x 3 fp64 [sum r^2, sum r, sum b]
 * ------------------------------------------------------------------------------------------- */
int endo_scale_inv_fwd(const float* pred, const float* goal, const float* boundary,
                       float* loss, double* stats, int n, int hw, float eps, void* stream);
int endo_scale_inv_bwd(const float* grad_loss, const float* pred, const float* goal,
                       const float* boundary, const double* stats,
                       float* grad_pred, float* grad_goal, int n, int hw, float eps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The loss head of a training iteration in one call -- reference train.py:279-315 (depth scaling, flow from depth, boundary
 * masking, sparse-flow loss, depth warping both ways, depth-consistency loss, weighted sum) AND its backward down to
 * d loss / d prediction, composed from the entry points above (same kernels, same arithmetic; the modules remain for callers
 * that want the pieces).  pred_*: n x 1 x H x W network outputs; the other inputs are the batch tensors of train.py:245-270
 * (sparse flows n x 2 x H x W; t n x 3, R / K n x 9).  losses: FOUR fp32 on the device = total, depth-consistency, sparse-flow
 * (weights applied: w * 0.5 * (term_1 + term_2)) and the guard flag of train.py:317 (1.0 when the total is NaN / Inf, else 0.0).
 * grad_pred_*: n x 1 x H x W, written.  workspace: 16-byte aligned, endo_loss_head_workspace_floats(n, h, w) floats.  The caller
 * feeds grad_pred_* to endo_net_bwd without waiting for the host and hands &losses[3] (or its all-reduced sum) to
 * endo_sgd_clip_step as `skip_flag`: the reference's guarded branch also runs backward() and then a step() that changes nothing
 * (train.py:318-321).
 * ------------------------------------------------------------------------------------------- */
int64_t endo_loss_head_workspace_floats(int n, int h, int w);
int endo_loss_head(const float* pred_1, const float* pred_2, const float* boundaries,
                   const float* sparse_depths_1, const float* sparse_depths_2,
                   const float* sparse_depth_masks_1, const float* sparse_depth_masks_2,
                   const float* sparse_flows_1, const float* sparse_flows_2,
                   const float* sparse_flow_masks_1, const float* sparse_flow_masks_2,
                   const float* t_1_wrt_2, const float* r_1_wrt_2, const float* t_2_wrt_1, const float* r_2_wrt_1,
                   const float* intrinsics, float sfl_weight, float dcl_weight, float eps,
                   float* losses, float* grad_pred_1, float* grad_pred_2, float* workspace,
                   int n, int h, int w, void* stream);

/* Depth warp both ways + depth-consistency loss, forward AND backward, in one call -- reference models.py:454-554 (DepthWarpingLayer,
 * once per direction), losses.py:112-146 (NormalizedDistanceLoss, once per direction), train.py:305-314, and their backward:
 *   loss[0] = dcl_weight * 0.5 * (NDL(depth_1, warp(depth_2 -> 1)) + NDL(depth_2, warp(depth_1 -> 2)))
 *   grad_depth_k = d loss / d depth_k  (through the loss terms, the sampling grids and the sampled images)
 * depth_k: the (scaled) depth maps N x 1 x H x W; poses and intrinsics as endo_loss_head.  This is the chain BASELINE.json's second
 * metric times ("depth-warp fwd+bwd ms / pair").  workspace: endo_warp_consistency_workspace_floats(n, h, w) floats, 16-byte aligned. */
int64_t endo_warp_consistency_workspace_floats(int n, int h, int w);
/* algorithmic HBM bytes of one call (160 B per pixel of a frame pair, both directions, forward and backward) */
int64_t endo_warp_consistency_bytes(int n, int h, int w);
int endo_warp_consistency(const float* depth_1, const float* depth_2, const float* boundaries, const float* t_1_wrt_2,
                          const float* r_1_wrt_2, const float* t_2_wrt_1, const float* r_2_wrt_1, const float* intrinsics,
                          float dcl_weight, float eps, float* loss, float* grad_depth_1, float* grad_depth_2,
                          float* workspace, int n, int h, int w, void* stream);


/* ---------------------------------------------------------------------------------------------
 * train.py glue that is pure elementwise work between the modules
 *   endo_mask_mul: out[n,c,hw] = a[n,c,hw] * mask[n,0,hw]   (train.py:272-273, 293-298)
 * ------------------------------------------------------------------------------------------- */
int endo_mask_mul(const float* a, const float* mask, float* out, int n, int c, int hw, void* stream);

/* ---------------------------------------------------------------------------------------------
 * FCDenseNet57 -- reference models.py:100-194 (layers models.py:19-97)
 *
 * The network is driven through a handle that fixes (N, H, W) and owns only host-side launch
 * tables.  Parameters, gradients, BN buffers, activations and gradient workspaces live in caller
 * memory (torch tensors):
 *   params     fp32, the 210 trainable tensors packed in the reference's .parameters() order
 *   grads      fp32, same packing; endo_net_bwd ACCUMULATES into it (two forwards share one
 *              backward accumulation per step, train.py:276-277, 325)
 *   bn_running fp32, running_mean then running_var of the 49 BN layers in module order
 *   tape       fp32 activation workspace of endo_net_tape_floats() elements, written by fwd and
 *              read by bwd (one tape per forward call that will be differentiated)
 *   gradws     fp32 gradient workspace of endo_net_gradws_floats() elements (scratch for bwd)
 * ------------------------------------------------------------------------------------------- */
typedef struct endo_net endo_net;

int endo_net_create(endo_net** out, int n, int h, int w);
/* Grouped batch: `groups` independent forward / backward passes of n samples each -- the two frames of a training
 * pair, train.py:276-277 -- run inside every kernel launch: x / out / grad_out hold groups * n samples, each
 * group keeps its own BatchNorm batch statistics (so the result equals `groups` separate calls, running statistics
 * updated in group order), parameter gradients sum over the groups.  tape and gradws are then groups *
 * endo_net_group_stride() floats (what endo_net_tape_floats / endo_net_gradws_floats return).  groups <= 4. */
int endo_net_create_grouped(endo_net** out, int n, int h, int w, int groups);
int endo_net_groups(const endo_net* net);
/* Kernel-form / precision options of ONE network handle (the reference's modules are independent objects, train.py:191,
 * 206-211: two models in one process, or a second thread configuring its own model, never change this one's arithmetic).
 * endo_net_create* sets the defaults below; nothing is read from the environment.  endo_net_set_option returns the previous
 * value, ENDO_E_BADARG for a null handle or an unknown option; it takes effect with the next endo_net_fwd / endo_net_bwd on
 * that handle (a forward and the backward that differentiates it must run under the same ENDO_OPT_MFMA_BF16 value).
 * Every setting but ENDO_OPT_MFMA_BF16 computes the same function up to fp32 summation order; tests use
 * ENDO_OPT_WINO_MIN_TILES = 1 to reach the Winograd kernels at small sizes.
 *   ENDO_OPT_WINO_FWD        dense-layer forward at the fine levels: 0 direct convolution, 1 Winograd F(2x2,3x3), 3 / 4 = with 3 / 4 LDS stages,
 *                            5 (default since round 5) = F(4x4,3x3) for the launches whose 64 x 16 blocks fill the chip (level 0 at 256 x 320; csrc/wino4_fwd_kernels.h)
 *                            and F(2x2,3x3) below: +3 % frame-pairs/s; the depth is 5e-6 of its maximum from fp64 instead of 1e-6, against the
 *                            1e-4 of the parity target
 *   ENDO_OPT_WINO_DGRAD      fused base-channel data gradient at the fine levels: 0 direct, 1 Winograd with phase-skewed workers, one block per tile,
 *                            2 Winograd, the round-2 kernel, 3 (default) = 1 as persistent blocks that walk a run of tiles where that form applies
 *                            (csrc/dgrad_wino3p_kernels.h: at most 144 base channels; it also forms the final convolution's weight gradient of the
 *                            last up block's base channels)
 *   ENDO_OPT_DGRAD_VEC       new-channel data-gradient passes: 2 (default) = persistent blocks that walk a run of tiles (csrc/dgrad_newmap_kernels.h),
 *                            1 = one block per tile with 16-byte DMA of the gradient tiles, 0 = the same with dword DMA
 *   ENDO_OPT_WINO_MIN_TILES  tiles per launch from which a Winograd kernel is chosen (default 1024)
 *   ENDO_OPT_MFMA_BF16       NOT the same function: 1 = the dense layers' convolution kernels round their MFMA operands to bf16
 *                            (v_mfma_f32_16x16x16_bf16; fp32 accumulation, fp32 tensors in memory) -- the mixed-precision mode of
 *                            BASELINE configs[2], with its own tolerance; default 0 = fp32 operands.
 *                            Development values 2 * mask (mask bit 0 weight gradients, 1 forward, 2 data gradients) select families
 *   ENDO_OPT_MFMA_X3         the SAME function as fp32 operands, on the bf16 matrix cores: a bit mask (1 weight gradients, 2 forward, 4 data
 *                            gradients) of the dense layers' kernel families that evaluate their fp32 products as three-term bf16 splits
 *                            (v = hi + mid + lo exactly; six exact bf16 products per fp32 product, fp32 accumulation; csrc/common.h
 *                            split_bf16x8) -- fp32 operands and fp32-level accuracy (held to the fp32 bounds by the parity tests) at
 *                            6 / 16 of the fp32 matrix instructions' issue time.  Ignored where ENDO_OPT_MFMA_BF16 selects rounded operands.
 *   ENDO_OPT_WGRAD_OVERLAP   endo_net_bwd runs the weight gradients on a side stream of its own, overlapped with the data-gradient
 *                            chain and joined before it returns: 1 (default); 0 puts them back in line on the
 *                            caller's stream (clean per-kernel timings)
 *   ENDO_OPT_WGRAD_F34       dense-layer weight gradient where the height is a multiple of 16 and the width of 4 (levels 0-4 at 256 x 320): 1 (default) = in the
 *                            Winograd domain, F(3x3, 4x4) -- 36 multiplications per 4 x 4 tile of the output gradient instead of 144,
 *                            fp32 throughout (csrc/wgrad_f34_kernels.h); 0 = the direct kernels.  Ignored where ENDO_OPT_MFMA_BF16 or
 *                            ENDO_OPT_MFMA_X3 select another operand form for the weight gradients.
 *   ENDO_OPT_FINAL_VIRTUAL   1 (default) = the data gradient of the final 1x1 convolution (models.py:167, 186), g(pixel) * w[channel] with
 *                            g = grad_out * sign(pre), is not written to the 192 level-0 gradient planes: g goes to one plane and the last up
 *                            block's backward kernels form the product where they first touch a channel (two passes over 1 GB less per
 *                            step at 16 x 256 x 320); the forward sum of that convolution over the 180 input channels of the network's last dense
 *                            layer is formed by that layer's F(4x4,3x3) launch (final_fwd_kernel reads 12 planes instead of 192), and the first
 *                            convolution's gradient preparation (G = d + P x + Q, bias gradient) is folded into its weight-gradient kernel;
 *                            0 = the separate kernels (final_bwd_data_kernel, the full final_fwd_kernel, prep_dy) as before.  Same function up
 *                            to summation order.
 *   ENDO_OPT_TD_PERSIST      transition-down layers (models.py:56-67) with 96 / 144 channels on whole 32 x 8 tiles (levels 0 / 1 of configs[1]) as
 *                            persistent blocks that keep the 1x1 weights in LDS: bit 0 (1) = the data gradient (csrc/td_dgrad_kernels.h; also the
 *                            128-pixel-run kernel where the pooled rows have no whole code dwords), bit 1 (2) = the training-mode forward
 *                            (csrc/td_fwd_kernels.h).  Default 3; 0 = the per-tile kernels (conv_dma_kernel).  Same function up to summation order. */
#define ENDO_OPT_WINO_FWD 0
#define ENDO_OPT_WINO_DGRAD 1
#define ENDO_OPT_DGRAD_VEC 2
#define ENDO_OPT_WINO_MIN_TILES 3
#define ENDO_OPT_MFMA_BF16 4
#define ENDO_OPT_WGRAD_OVERLAP 5
#define ENDO_OPT_MFMA_X3 6
#define ENDO_OPT_WGRAD_F34 7
#define ENDO_OPT_FINAL_VIRTUAL 8
#define ENDO_OPT_TD_PERSIST 9
#define ENDO_OPT_COUNT 10
int endo_net_set_option(endo_net* net, int option_id, int value);
int endo_net_get_option(const endo_net* net, int option_id);
int64_t endo_net_group_stride(const endo_net* net);
void endo_net_destroy(endo_net* net);
int64_t endo_net_param_floats(void);                 /* 1 374 865 */
int64_t endo_net_bn_floats(void);                    /* 2 * sum of BN widths */
int64_t endo_net_tape_floats(const endo_net* net);
int64_t endo_net_gradws_floats(const endo_net* net);
/* offset (in floats) of the i-th trainable tensor inside params/grads, i in [0, 210) */
int64_t endo_net_param_offset(int index);
/* offset of running_mean / running_var of the i-th BN layer inside bn_running, i in [0, 49) */
int64_t endo_net_bn_offset(int bn_index, int which /*0 mean, 1 var*/);

/* x: n x 3 x H x W (already multiplied by the boundary, train.py:272-273); out: n x 1 x H x W >= 0.
 * training != 0: batch statistics + running-stat update (momentum 0.1, eps 1e-5); 0: running stats.
 * The tape is always required (it also holds the level buffers the forward pass works in). */
int endo_net_fwd(endo_net* net, const float* params, float* bn_running, const float* x, float* out,
                 float* tape, int training, void* stream);
/* grad_out: n x 1 x H x W; x: the forward call's input.  Accumulates parameter gradients into grads.
 * The tape must come from a endo_net_fwd call with the same params, x and training flag. */
int endo_net_bwd(endo_net* net, const float* params, const float* x, const float* tape,
                 const float* grad_out, float* grads, float* gradws, int training, void* stream);
/* layout queries (tests inspect intermediate activations through these):
 * channel count of level buffer `level` (0..5) and its offset (floats) inside the tape */
int endo_net_level_channels(int level);
int64_t endo_net_act_offset(const endo_net* net, int level);
/* more of the tape's layout, for tests that rebuild the activation pattern a forward pass took (which ReLU inputs were
 * positive -- models.py:24, which pixel each 2x2 max-pool kept -- models.py:66, the sign under the final |.| --
 * models.py:186) so that gradients can be compared against an exact evaluation on the SAME pattern:
 *   ENDO_TAPE_PRE     (index ignored)        float offset of the final conv's pre-activation, n x 1 x H x W
 *   ENDO_TAPE_BN_SAVED index = BN layer in module order (0..48): float offset of its (mean, rstd) pairs, 2 per channel
 *   ENDO_TAPE_POOL    index = level 0..4:    BYTE offset of the argmax codes of that level's max-pool,
 *                                            n x C x H/2 x W/2 bytes, code = 2 * (row & 1) + (col & 1)
 * Offsets are inside one group's tape; group g of a grouped handle starts endo_net_group_stride() floats further.
 * Returns -1 for a bad argument. */
#define ENDO_TAPE_PRE 0
#define ENDO_TAPE_BN_SAVED 1
#define ENDO_TAPE_POOL 2
int64_t endo_net_tape_offset(const endo_net* net, int what, int index);

/* ---------------------------------------------------------------------------------------------
 * clip_grad_norm_(params, max_norm) + SGD(momentum) -- reference train.py:327-328, 202
 * grads are scaled in place by grad_scale (1/world after the all-reduce) and then by the clip
 * coefficient min(1, max_norm / (norm + 1e-6)); momentum buf = mu * buf + g; p -= lr * buf.
 * first_step != 0: buf = g.  norm_out: 2 fp64 [sum of squares, pre-clip global L2 norm], written by the call.
 * skip_flag: null, or one fp32 on the device -- when it is non-zero (the non-finite-loss guard of train.py:317-322, endo_loss_head's
 * losses[3], summed over ranks by the gradient all-reduce) parameters and momentum are left untouched; norm_out is still written.
 * ------------------------------------------------------------------------------------------- */
int endo_sgd_clip_step(float* params, float* grads, float* momentum, double* norm_out,
                       int64_t count, float lr, float mu, float max_norm, float grad_scale,
                       int first_step, const float* skip_flag, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sparse SfM scatter -- reference utils.py:460-612 (get_torch_training_data) for a batch of pairs of ONE
 * sequence, on the device.  points [P][4] fp64 (homogeneous), projections [B][2][3][4] fp64,
 * extrinsics [B][2][4][4] fp64, visibility [B][P][2] (> 0.5 = the point is seen in that frame of the pair),
 * clean [P] or NULL (utils.py:496-497), mask [H][W] uint8 (255 = inside the endoscope boundary).
 * Outputs, zero-filled by the call, NCHW with the frame index outermost: depth_masks / depths / flow_masks
 * [2][B][1][H][W], flows [2][B][2][H][W] (u then v, divided by W and H; entries with |flow| > 5 zeroed,
 * utils.py:566-569,603-606).  depths are multiplied by depth_multiplier (1 = the reference function;
 * 1 / global_scale folds in dataset.py:391-392).  Pixel collisions: the highest point index wins (numpy
 * fancy-index assignment).  winner_scratch: 2*B*H*W int32 of workspace.
 * ------------------------------------------------------------------------------------------- */
int endo_sparse_scatter(const double* points, int n_points, const double* projections, const double* extrinsics,
                        const float* visibility, const float* clean, const uint8_t* mask, int batch, int height, int width,
                        float depth_multiplier, int32_t* winner_scratch, float* depth_masks, float* depths, float* flow_masks,
                        float* flows, void* stream);

/* Relative camera motion of a batch of frame pairs -- reference dataset.py:384-399, on the device.
 * pair_extrinsics [B][2][4][4] fp64 (world-to-camera of frame 1 and frame 2); scale = the sequence's estimated global scale.
 * relative = E_1 inv(E_2) in fp64;  r_1_wrt_2 [B][3][3] = fp32(relative[:3,:3]);  t_1_wrt_2 [B][3] = fp32(relative[:3,3] / scale);
 * r_2_wrt_1 = r_1_wrt_2^T;  t_2_wrt_1 = -r_1_wrt_2^T t_1_wrt_2 in fp32 -- the four pose tensors of a training batch, with no
 * host arithmetic and no host-to-device copy per sample. */
int endo_relative_poses(const double* pair_extrinsics, int batch, double scale,
                        float* r_1_wrt_2, float* t_1_wrt_2, float* r_2_wrt_1, float* t_2_wrt_1, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Coloured point cloud of one depth map -- reference utils.py:823-852 (point_cloud_from_depth), the per-frame
 * back-projection of evaluate.py:272,340.  depth, mask [H][W] fp32; color_bgr [H][W][3] uint8 (cv2 order);
 * intrinsics [3][3] fp32.  A pixel is kept when h % downsampling == 0, w % downsampling == 0, mask > 0.5 and, with
 * use_threshold != 0, max(r,g,b) >= max_threshold && min(r,g,b) <= min_threshold.  points receives
 * (x, y, z, r, g, b) = ((w - cx) / fx * z, (h - cy) / fy * z, z, r, g, b) per kept pixel in row-major pixel order
 * (capacity H * W rows of 6 floats); *count_out the number of rows written.  row_offsets: H + 1 int32 of workspace.
 * ------------------------------------------------------------------------------------------- */
int endo_point_cloud(const float* depth, const uint8_t* color_bgr, const float* mask, const float* intrinsics, int height,
                     int width, int downsampling, int use_threshold, float min_threshold, float max_threshold,
                     int32_t* row_offsets, float* points, int32_t* count_out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Colour frames from the sequence folder's .jpg files -- reference utils.py:441-457 (get_pair_color_imgs: cv2.imread,
 * cv2.resize(img, (0, 0), fx = fy = 1 / downsampling), crop [start_h:end_h, start_w:end_w], BGR2RGB), utils.py:72-83
 * (get_test_color_img) and dataset.py:148,446-451 (albumentations Normalize(0.5, 0.5) + img_to_tensor).
 * Baseline / extended-sequential Huffman JPEG, 8 bit, one interleaved scan; grey, 4:4:4, 4:2:2 (h2v1) and 4:2:0 (h2v2);
 * restart intervals.  Anything else (progressive, arithmetic, 12 bit, CMYK) returns ENDO_E_UNSUPPORTED.
 *   endo_jpeg_info            host only.  info[16]: width, height, components, hmax, vmax, MCUs across, MCUs down,
 *                             (blocks across, blocks down) per component, total 8x8 blocks, restart interval, 0.
 *   endo_jpeg_entropy_decode  host only: the coefficient blocks the device stage starts from -- total_blocks x 64 int16,
 *                             component planes one after the other, blocks row-major inside a plane, coefficients in
 *                             natural (row-major, de-zigzagged) order, not dequantised; quant: [3][64] uint16, natural order.
 *   endo_jpeg_workspace_bytes bytes of `staging` (host; pinned for an asynchronous copy) and of `workspace` (device).
 *   endo_jpeg_decode_crop     parses and Huffman-decodes on the calling thread into `staging`, copies to `workspace` on
 *                             `stream` and launches the inverse DCT and the resize/crop kernels there.  out_hwc: device
 *                             uint8 [H][W][3] (rgb_order != 0: R,G,B as rgb_mode "rgb"; 0: B,G,R as cv2.imread) or NULL;
 *                             out_chw: device fp32 [3][H][W] = (v - 127.5) * (1 / 127.5) or NULL.  `staging` must stay
 *                             untouched until `stream` has passed the call.  Pixel values are those of libjpeg's default
 *                             decoder (JDCT_ISLOW, fancy upsampling) followed by cv2's 8-bit INTER_LINEAR arithmetic.
 * ------------------------------------------------------------------------------------------- */
int endo_jpeg_info(const uint8_t* data, int64_t size, int32_t* info);
int endo_jpeg_entropy_decode(const uint8_t* data, int64_t size, int16_t* blocks, int64_t capacity_blocks, uint16_t* quant);
int64_t endo_jpeg_workspace_bytes(const uint8_t* data, int64_t size);
int endo_jpeg_decode_crop(const uint8_t* data, int64_t size, double downsampling, int start_h, int end_h, int start_w,
                          int end_w, int rgb_order, uint8_t* out_hwc, float* out_chw, void* staging, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Contaminated-point filter, the per-pixel half -- reference utils.py:339-404 (get_clean_point_list), dataset.py:96-111.
 * imgs [frames][H][W][3] uint8 in cv2 order (B, G, R) as utils.get_color_imgs returns them (values, before the / 255);
 * points [P][4] fp64; projections [frames][3][4], extrinsics [frames][4][4] fp64; visibility [P][frames] fp32
 * (view_indexes_per_point); mask [H][W] uint8.  For frame f and point p (outputs [frames][P]): valid = visible > 0.5, projects
 * to 0 <= u <= W-1, 0 <= v <= H-1 with camera depth > 0, and mask[round(v)][round(u)] == 255 (round half to even); depth = the
 * camera depth; brightness = max(B, G, R) of cv2.bilateralFilter(img / 255, d, sigma_color, sigma_space) at that pixel (circular
 * window of radius d / 2, BORDER_REFLECT_101, colour distance |db| + |dg| + |dr|), evaluated only there.
 * ------------------------------------------------------------------------------------------- */
int endo_point_brightness(const uint8_t* imgs, int frames, int height, int width, const double* points, int n_points,
                          const double* projections, const double* extrinsics, const float* visibility, const uint8_t* mask,
                          int d, double sigma_color, double sigma_space, int32_t* valid, double* depth, float* brightness,
                          void* stream);

/* cv2.cvtColor(img, COLOR_BGR2HSV_FULL) (blue_index 0) / COLOR_RGB2HSV_FULL (blue_index 2) on 8-bit interleaved pixels -- the reader's
 * HSV input mode, reference utils.py:449-450, 80-81 and dataset.py:434-442 (`--use_hsv_colorspace`).  OpenCV's scalar fixed-point
 * arithmetic (H over [0, 256), tables round((255 << 12) / v), round((256 << 12) / (6 diff))); PARITY UNPINNED against cv2 itself (no
 * converted image in the reference tree).  src [pixels][3] uint8; out_u8 [pixels][3] (H, S, V) and / or out_f32 [3][pixels] =
 * (x / 255 - 0.5) / 0.5 (dataset.py:446-451); either may be null, src == out_u8 is allowed. */
int endo_hsv_full(const uint8_t* src, int64_t pixels, int blue_index, uint8_t* out_u8, float* out_f32, void* stream);

/* live per-kernel-family timing for bench.py's roofline line: HIP events recorded on the launch
 * stream around every entry of the selected families.  family_mask: bit f enables family f
 * (0 = off, -1 = all); calling it also discards previously recorded events.  endo_prof_read
 * synchronises on the recorded events and returns totals (ms, launches, algorithmic flops/bytes) of the TIMED launches.
 * endo_prof_sample(period): time one launch in `period` of each enabled family (default 1 = every launch) -- two events around a
 * launch on the caller's stream serialise it with its neighbours, and a family of 44 launches per step then costs the step 3 %; a
 * period coprime with the family's launches per step visits every launch of the step in rotation.  endo_prof_seen: launches of
 * an enabled family since endo_prof_enable, timed or not. */
#define ENDO_PROF_FAMILIES 16
int endo_prof_enable(int family_mask);
int endo_prof_sample(int period);
int endo_prof_seen(int family, int64_t* launches);
int endo_prof_read(int family, double* total_ms, int64_t* launches, double* total_flops, double* total_bytes);
const char* endo_prof_family_name(int family);

/* ---------------------------------------------------------------------------------------------
 * bf16-STORAGE family (BASELINE configs[2] / [4]): bf16 level buffers in 32-channel blocks
 * ([n][t / blk][h][w][blk] bf16; blk = t is plain channels-last) and the convolution over them on the bf16 matrix cores
 * (v_mfma_f32_16x16x32_bf16, fp32 accumulation), with BatchNorm + ReLU applied once per staged element (reference models.py:19-28:
 * DenseLayer; 70-80: TransitionUp).
 *   endo_bf16_pack_nhwc / unpack_nhwc   fp32 NCHW <-> a channel slice of a bf16 buffer (round to nearest even); blk 0 = t
 *   endo_bf16_conv_weights              W[cout][cin][ks][ks] fp32 -> the kernel's bf16 layout (endo_bf16_conv_weight_elems elements)
 *   endo_bf16_conv                      out[.., oc0 : oc0 + cout] = conv_ks([nearest x2]([relu(x * scale + shift)])) + bias
 *                                       bn: [cin][2] fp32 (scale, shift) or null; out_sums: [cout][2] fp64, accumulated, or null
 * Constraints: ks in {1, 3}; cin, cout, oc0, out_blk multiples of 4; ic0, in_blk multiples of 8; t a multiple of blk.
 * ------------------------------------------------------------------------------------------- */
int endo_bf16_pack_nhwc(const float* x, void* out, int n, int c, int h, int w, int t, int blk, int oc0, void* stream);
int endo_bf16_unpack_nhwc(const void* in, float* x, int n, int c, int h, int w, int t, int blk, int ic0, void* stream);
int64_t endo_bf16_conv_weight_elems(int cout, int cin, int ks);
int endo_bf16_conv_weights(const float* w, int cout, int cin, int ks, void* out, void* stream);
int endo_bf16_conv(const void* in, int in_t, int in_blk, int ic0, int cin, const float* bn, const void* wgt, const float* bias, void* out,
                   int out_t, int out_blk, int oc0, int cout, double* out_sums, int n, int h, int w, int ks, int ups, void* stream);
/* FCDenseNet57 FORWARD over bf16 level buffers (reference models.py:171-187): same parameters / running statistics / input / output
 * tensors as endo_net_fwd (fp32), activations stored as bf16 in 32-channel blocks, BatchNorm statistics and the output in fp32.  The input is
 * rounded to bf16 on the way in.  training != 0: batch statistics + running-statistics update; 0: running statistics (the
 * evaluate.py path).  tape: endo_net16_tape_bytes() bytes of device memory, 256-byte aligned.  H and W multiples of 32.
 * endo_net16_bwd: its backward pass.  tape: the forward call's tape, untouched since; grad_out: fp32 [n][1][H][W]; grads: the flat
 * fp32 parameter-gradient buffer (offsets of endo_net_param_offset), ACCUMULATED into; ws: endo_net16_bwd_workspace_bytes() bytes,
 * 256-byte aligned; training as in the forward call (0: BatchNorm as a fixed affine map).  Gradients between layers are stored as
 * bf16; BatchNorm sums (fp64), parameter gradients and the deferred BatchNorm terms (fp32) are not. */
typedef struct endo_net16 endo_net16;
/* n_per_group x groups samples per call (groups 1 or 2): every group has its own BatchNorm batch statistics and the running statistics are
 * updated with group 0's first -- one call with groups = 2 equals the reference's two network calls of a training step (train.py:276-277) */
int endo_net16_create(endo_net16** out, int n_per_group, int h, int w, int groups);
void endo_net16_destroy(endo_net16* net);
int64_t endo_net16_tape_bytes(const endo_net16* net);
int endo_net16_fwd(endo_net16* net, const float* params, float* bn_running, const float* x, float* out, void* tape, int training,
                   void* stream);
int64_t endo_net16_bwd_workspace_bytes(const endo_net16* net);
/* 1 (default): endo_net16_bwd runs the weight gradients on a side stream of its own, forked from and joined back into the caller's
 * stream by events before it returns; 0: everything in line on the caller's stream */
int endo_net16_set_wgrad_overlap(endo_net16* net, int on);
/* The same family over IEEE HALF storage (BASELINE configs[4]: "mixed fp16 storage / fp32 accum"): identical signatures and buffer
 * layouts (csrc/net16h.hip compiles the bf16 sources with another element type), v_mfma_f32_16x16x32_f16, and a power-of-two gradient
 * scale chosen per backward call from max |grad_out| (per-pixel gradients of a mean loss are far below half's smallest normal):
 * stored gradients carry it, parameter gradients leave without it.  endo_f16_pack_nhwc / unpack_nhwc: the layout helpers for half. */
int endo_net16h_create(endo_net16** out, int n_per_group, int h, int w, int groups);
void endo_net16h_destroy(endo_net16* net);
int64_t endo_net16h_tape_bytes(const endo_net16* net);
int endo_net16h_fwd(endo_net16* net, const float* params, float* bn_running, const float* x, float* out, void* tape, int training,
                    void* stream);
int64_t endo_net16h_bwd_workspace_bytes(const endo_net16* net);
int endo_net16h_set_wgrad_overlap(endo_net16* net, int on);
int64_t endo_net16h_offset(const endo_net16* net, int what, int index);
int endo_net16h_bwd(endo_net16* net, const float* params, const void* tape, const float* grad_out, float* grads, void* ws, int training,
                    void* stream);
int endo_f16_pack_nhwc(const float* x, void* out, int n, int c, int h, int w, int t, int blk, int oc0, void* stream);
int endo_f16_unpack_nhwc(const void* in, float* x, int n, int c, int h, int w, int t, int blk, int ic0, void* stream);
/* byte offsets into the tape (what 0: final pre-activation fp32; 1: (mean, rstd) of BatchNorm `index` in module order; 2: max-pool
 * codes of transition down `index` ([n][h / 2][w / 2][cout] bytes); 3: level buffer `index`) or the backward workspace (4: gradient
 * buffer of level `index`); 5: channels of level buffer `index`; 7: bytes between two sample groups' (mean, rstd) tables.  Level buffers: [n][t / 32][h][w][32] bf16, channels [0, S) the
 * down path, [S, S + 48) the transition-up output, [S + 48, S + 96) the up block's maps (S = 96 + 48 level; bottleneck: 288 + 48). */
int64_t endo_net16_offset(const endo_net16* net, int what, int index);
int endo_net16_bwd(endo_net16* net, const float* params, const void* tape, const float* grad_out, float* grads, void* ws, int training,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ENDO_HIP_H */
