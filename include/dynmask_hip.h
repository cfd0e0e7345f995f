/*
 * include/dynmask_hip.h -- C ABI of the CondInst-style dynamic mask head of UNINEXT on MI355X (gfx950), part of
 * libmsda_hip.so.  SURVEY.md 8(f) rank 2: the step after the decoder on the inference path.
 *
 * Replaces, for inference, the body of DDETRSegmUniDN.dynamic_mask_with_coords
 * (projects/UNINEXT/uninext/models/ddetrs_dn.py:755-844) between "build mask_head_inputs" and "upsample":
 *   - compute_locations + relative coordinates (ddetrs_dn.py:765-784, 1199-1212),
 *   - the repeat/cat that materialises [1, n_inst*(C+2), H, W] (ddetrs_dn.py:786-808; 1.2 GB at 1800 instances),
 *   - parse_dynamic_params (ddetrs_dn.py:1148-1171) and the three grouped 1x1 convolutions of
 *     mask_heads_forward (ddetrs_dn.py:734-752): (C+2) -> 8 -> 8 -> 1 with ReLU between,
 * by one kernel that keeps the C mask-feature channels of a pixel in registers and walks the instances of the
 * image with their 169 parameters staged in LDS; and aligned_bilinear (ddetrs_dn.py:1174-1196) by a second one.
 *
 * All pointers are device pointers except `num_insts` (host), contiguous fp32 unless noted; `stream` is a
 * hipStream_t as void*.  Kernels are only enqueued.  Returns 0, a negative DYNMASK_ERR_*, or a positive hipError_t;
 * the message is available from msda_hip_last_error().
 */
#ifndef DYNMASK_HIP_H_
#define DYNMASK_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DYNMASK_ERR_NULL_POINTER (-1)
#define DYNMASK_ERR_BAD_DIMS (-2)
#define DYNMASK_ERR_UNSUPPORTED (-5)   /* feature channels != 8: use the PyTorch composition */

/*
 * mask_feats   [batch, 8, H, W]            output of the mask-feature branch (hidden_dim / 32 = 8 channels)
 * inst_xy      [n_inst_all, 2]             instance reference points (x, y) in input-image pixels
 * params       [n_inst_all, 169]           controller output: w0 (8 x 10) w1 (8 x 8) w2 (1 x 8) b0 (8) b1 (8) b2 (1);
 *                                          with rel_coord == 0: w0 is 8 x 8 and a row has 153 values
 * num_insts    [batch] (HOST ints)         instances per image, in order; n_inst_all = sum
 * out_logits   [n_inst_all, H, W]          mask logits at the feature stride
 * stride       mask_feat_stride (8): pixel (y, x) sits at (x * stride + stride / 2, y * stride + stride / 2)
 */
int dynmask_hip_forward_f32(const float* mask_feats, const float* inst_xy, const float* params,
                            const int* num_insts, int batch, int channels, int H, int W, int stride,
                            int rel_coord, float* out_logits, void* stream);

/*
 * Kernel behind dynmask_hip_forward_f32: 0 = auto (the faster one measured on MI355X), 1 = packed-FMA VALU kernel,
 * 2 / 3 = MFMA kernel (v_mfma_f32_4x4x1_16b_f32; 2 / 4 pixels per lane).  The environment variable
 * DYNMASK_HIP_VARIANT seeds the choice.  Returns 0 or DYNMASK_ERR_BAD_DIMS.  dynmask_hip_last_kernel names the
 * kernel the last forward call enqueued ("dynmask_fwd_pkfma", "dynmask_fwd_mfma_q2", "dynmask_fwd_mfma_q4").
 */
int dynmask_hip_set_variant(int variant);
const char* dynmask_hip_last_kernel(void);

/*
 * aligned_bilinear (ddetrs_dn.py:1174-1196): in [n, h, w] -> out [n, factor*h, factor*w]; factor >= 1.
 */
int aligned_bilinear_hip_f32(const float* in, int n, int h, int w, int factor, float* out, void* stream);

/*
 * Backward of dynmask_hip_forward_f32 (training: BASELINE configs[4] trains the CondInst head, ddetrs_dn.py:493-560 calls
 * dynamic_mask_with_coords under autograd).  Same inputs as the forward plus
 *   grad_logits  [n_inst_all, H, W]     gradient of the loss with respect to out_logits
 * and the gradients, every element written (no accumulation, no float atomics: results are bitwise repeatable):
 *   grad_feats   [batch, 8, H, W] or NULL       sum over the image's instances (zeros for an image without instances)
 *   grad_params  [n_inst_all, 169|153] or NULL  sum over the pixels, in the layout of `params`
 *   grad_xy      [n_inst_all, 2] or NULL        gradient with respect to inst_xy (zeros with rel_coord == 0); needs grad_params
 * A NULL gradient is not computed: grad_feats == NULL skips the pixel-major kernels, grad_params == NULL (with grad_xy == NULL)
 * the instance-major ones and the workspace (frozen mask features / detached parameters; round 6).
 * workspace: device memory of at least dynmask_hip_backward_workspace_bytes(n_inst_all, H, W) bytes (partial sums of the
 * pixel slices, dynmask_hip_backward_parts of them per instance), contents undefined before and after; borrowed for the call
 * in stream order.  At most DYNMASK_HIP_BWD_MAX_BATCH images per call (DYNMASK_ERR_UNSUPPORTED beyond).
 * The activations are recomputed per (instance, pixel) from the inputs: nothing of the forward has to be kept.
 */
#define DYNMASK_HIP_BWD_MAX_BATCH 64
size_t dynmask_hip_backward_workspace_bytes(int n_inst_all, int H, int W);
int dynmask_hip_backward_parts(int n_inst_all, int H, int W);
int dynmask_hip_backward_f32(const float* mask_feats, const float* inst_xy, const float* params, const int* num_insts,
                             int batch, int channels, int H, int W, int stride, int rel_coord, const float* grad_logits,
                             float* grad_feats, float* grad_params, float* grad_xy, void* workspace, size_t workspace_bytes,
                             void* stream);

/*
 * Backward of aligned_bilinear_hip_f32: grad_out [n, factor*h, factor*w] -> grad_in [n, h, w], every element written; a
 * gather (each input pixel sums the output pixels that read it, in a fixed order): bitwise repeatable.
 */
int aligned_bilinear_hip_backward_f32(const float* grad_out, int n, int h, int w, int factor, float* grad_in, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DYNMASK_HIP_H_ */
