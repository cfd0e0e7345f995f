/*
 * include/matcher_cost_hip.h -- C ABI of the fused cost matrix of UNINEXT's Hungarian matcher on MI355X (gfx950), part of
 * libmsda_hip.so.  SURVEY.md 8(f) rank 4, the remainder of row f4.
 *
 * Replaces the PyTorch composition of HungarianMatcherVL.forward
 * (projects/UNINEXT/uninext/models/deformable_detr/matcher.py:476-498): sigmoid of the token logits, the focal
 * positive / negative terms over ALL tokens, a Python loop over the targets gathering each target's positive tokens
 * with a boolean mask and averaging them (:482-488), torch.cdist(p = 1) of the boxes (:491), the generalised IoU of the
 * xyxy boxes (:494, util/box_ops.py:17-85) and the weighted sum (:498) -- about 40 kernel launches and a [num_pred, T]
 * table for a [num_pred, num_gt] result -- by ONE kernel that evaluates, per (prediction, target) pair, the same
 * float32 operations in the same order (no FMA contraction, IEEE division, expf / logf of the device library).  Measured
 * against ATen on the MI355X (tools/matcher_cost_dbg2.py, tests/test_matcher_gpu.py): exp, the sigmoid, the L1 distance
 * and the GIoU chain are bitwise ATen's; logf of ROCm 7.2's device library and of the one PyTorch was built with differ by
 * one unit in the last place on a third of the arguments, so the class term agrees to 5e-7 absolute, not bitwise.  The
 * assignment indices of all six reference-minted matcher fixtures are unchanged.
 *
 *   p = 1 / (1 + exp(-logit));  neg = (0.75 * (p * p)) * (-log((1 - p) + 1e-8));  pos = (0.25 * ((1 - p) * (1 - p))) * (-log(p + 1e-8))
 *   class = (sum over the target's positive tokens, in index order, of (pos - neg)) * (1 / count)
 *   bbox  = (|d0| + |d2|) + (|d1| + |d3|) of the cxcywh boxes            (the summation tree of ATen's cdist kernel)
 *   giou  = iou - (hull - union) / (hull + 1e-7)                          (box_ops.py:62-85 on the xyxy corners)
 *   cost  = ((w_bbox * bbox) + (w_class * class)) + (w_giou * (-giou))
 *
 * logits     [num_pred, T] fp32 (num_pred = batch x queries, flattened), boxes [num_pred, 4] cxcywh
 * tgt_boxes  [num_gt, 4] cxcywh; tok_off int32 [num_gt + 1], tok_idx int32 [tok_off[num_gt]]: the positive tokens of
 *            every target in ascending order (CSR of the boolean positive map); a target without tokens gets NaN
 *            (the mean of an empty selection, as in the reference)
 * cost       [num_pred, num_gt] fp32, every element written.  Device pointers; the kernel is only enqueued on `stream`.
 * Returns 0, a negative MATCHER_COST_ERR_*, or a positive hipError_t; the message is available from msda_hip_last_error().
 */
#ifndef MATCHER_COST_HIP_H_
#define MATCHER_COST_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MATCHER_COST_ERR_NULL_POINTER (-1)
#define MATCHER_COST_ERR_BAD_DIMS (-2)

int matcher_cost_hip_f32(const float* logits, const float* boxes, const float* tgt_boxes, const int32_t* tok_off,
                         const int32_t* tok_idx, int num_pred, int num_tokens, int num_gt, float w_class, float w_bbox,
                         float w_giou, float* cost, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MATCHER_COST_HIP_H_ */
