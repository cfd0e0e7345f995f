/*
 * include/ota_hip.h -- C ABI of the on-device simOTA (dynamic-k) assignment of UNINEXT's matcher on MI355X (gfx950), part of
 * libmsda_hip.so.  SURVEY.md 8(a) row a10 and 8(f) rank 4 (the half the shipped configs use: MODEL.OTA: True, every decoder
 * layer of every training step).
 *
 * Replaces HungarianMatcherVL.forward_ota's per-image work
 * (projects/UNINEXT/uninext/models/deformable_detr/matcher.py: compute_cost :313-342, get_in_boxes_info :344-385,
 * dynamic_k_matching :387-447): ~60 small PyTorch kernels per image, a Python loop over the targets with a boolean-mask
 * gather each (:332-334, a host sync per target), a `.item()` per target (:399-402), `.sum() > 0` / `.any()` host syncs
 * around the conflict resolution and the repair `while` (:409,:417,:427) -- by two kernels for the whole batch that never
 * return to the host.  The results are INTEGERS (which queries serve which ground truth), so every float that decides one is
 * formed by the reference's float32 operations in the reference's order (no FMA contraction, IEEE division), and every
 * selection uses PyTorch's tie rule (lowest index first).  The two multi-term reductions -- a target's class cost over its
 * positive tokens and the sum of its ten largest IoUs -- are SEQUENTIAL sums here (ascending token order / descending IoU);
 * PyTorch's GPU reduction kernels may associate three or more terms differently, so a cost can differ from the PyTorch
 * composition's by one unit in the last place (exact against oracle/ota_oracle.py, which is pinned to the reference's
 * integers on the fixtures):
 *
 *   ota_cost_hip_f32     per (query, target) pair of every image
 *       class = (sum over the target's positive tokens, ascending, of class_table[q, t]) * (1 / count)         (:329-334)
 *       iou   = inter / ((area_q + area_g) - inter)                                   (torchvision.ops.box_iou, :326)
 *       giou  = iou - (hull - union) / (hull + 1e-7)                                          (util/box_ops.py:62-85)
 *       in_box    = cx > gx0 & cx < gx1 & cy > gy0 & cy < gy1      (the target's xyxy corners, :354-360)
 *       in_centre = the same against (gcx -+ 2.5 / 32, gcy -+ 2.5 / 32)                                     (:367-372)
 *       cost  = (class + 3 * (-giou)) + 100 * !(in_box & in_centre)                                              (:338)
 *     class_table [batch, Q, T] is the focal table pos - neg of matcher.py:327-330, which the caller forms with the
 *     reference's own elementwise PyTorch operations (bitwise the reference's on the same device -- the device library's
 *     logf and the one PyTorch was built with differ in the last place on a third of the arguments, so the table is not
 *     recomputed here).  Also written: iou and one byte per pair: bit 0 = in_box | in_centre (the foreground test of :374),
 *     bit 1 = a box of the pair is degenerate (x1 < x0, y1 < y0 or NaN) -- the reference's generalized_box_iou asserts
 *     against those (util/box_ops.py:76-77) and aborts the training step.
 *
 *   ota_dynamic_k_hip    one 1024-thread workgroup per image, everything of :340 and :387-447 in order
 *       cost[q, :] += 10000 for queries inside no box and no centre square                                       (:340)
 *       k_g = max(int(sum of the 10 largest IoUs of column g, added in descending order), 1)                 (:394-397)
 *       the k_g cheapest queries of column g claim g                                                          (:399-402)
 *       a query claimed more than once keeps the cheapest of its row                                          (:406-411)
 *       while a target has no query: cost[taken rows] += 100000; every such target takes its cheapest query; if any
 *         query now holds two targets, the rows that were multiply claimed BEFORE the loop (the reference never refreshes
 *         that mask, :406 vs :432) are reset to their cheapest target                                          (:417-435)
 *       selected queries ascending with the first target of their row; per target the cheapest query among its own (:441-447)
 *     `cost` is modified in place exactly as the reference modifies it.  A repair loop that does not terminate within
 *     `max_rounds` rounds (the reference would spin forever) sets status 2; an image with a degenerate box (bit 1 of a flag)
 *     gets status 4 on top (its assignment is computed anyway, NaN costs sorting last: the caller decides -- the Python
 *     binding raises the reference's AssertionError behind its one host copy).
 *
 * Batch layout: image b has targets gt_off[b] .. gt_off[b + 1] - 1 of the concatenated target arrays; its [Q, G_b] blocks
 * of cost / iou / flags / matching start at element Q * gt_off[b].  All pointers are device memory; kernels are only
 * enqueued on `stream`; no allocation, no synchronisation.  At most OTA_HIP_MAX_BATCH images per call.
 * Returns 0, a negative OTA_ERR_*, or a positive hipError_t; the message is available from msda_hip_last_error().
 */
#ifndef OTA_HIP_H_
#define OTA_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OTA_ERR_NULL_POINTER (-1)
#define OTA_ERR_BAD_DIMS (-2)
#define OTA_HIP_MAX_BATCH 64

/*
 * class_table [batch, Q, T] fp32; boxes [batch, Q, 4] cxcywh fp32; tgt_boxes [G_total, 4] cxcywh fp32;
 * positive_map [G_total, T] bytes (0 / non-zero); gt_off host array int32 [batch + 1] (gt_off[0] = 0).
 * Outputs: cost, iou fp32 [Q * G_total], flags uint8 [Q * G_total] in the batch layout above, every element written.
 */
int ota_cost_hip_f32(const float* class_table, const float* boxes, const float* tgt_boxes, const uint8_t* positive_map,
                     const int32_t* gt_off, int batch, int num_queries, int num_tokens, float* cost, float* iou,
                     uint8_t* flags, void* stream);

/*
 * cost (modified in place), iou, flags: as written by ota_cost_hip_f32.  matching uint8 [Q * G_total]: workspace, contents
 * undefined before, the final 0 / 1 matching matrix after.
 * Outputs per image b (G_b > 0): sel_query[b * Q .. ] int64, ascending, and sel_gt[b * Q ..] int64 -- the first
 * num_selected[b] entries are valid; matched_query int64 [G_total]: per target its cheapest own query; num_selected int32
 * [batch] (0 for an image without targets); status int32 [batch]: 0 ok, bit 1 (2) repair loop cut off after max_rounds,
 * bit 2 (4) a predicted or target box of the image is degenerate.
 */
int ota_dynamic_k_hip(float* cost, const float* iou, const uint8_t* flags, uint8_t* matching, const int32_t* gt_off,
                      int batch, int num_queries, int max_rounds, int64_t* sel_query, int64_t* sel_gt,
                      int64_t* matched_query, int32_t* num_selected, int32_t* status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OTA_HIP_H_ */
