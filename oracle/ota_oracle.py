"""CPU oracle for the simOTA (dynamic-k) assignment of the matcher.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

A numpy restatement, in float32 and in the reference's order of operations, of
projects/UNINEXT/uninext/models/deformable_detr/matcher.py: compute_cost (:313-342), get_in_boxes_info (:344-385) and
dynamic_k_matching (:387-447) with the box helpers of uninext/util/box_ops.py:17-85 -- written the way the HIP kernels of
include/ota_hip.h are organised (one (query, target) pair at a time for the cost; rows and columns of the matching matrix
for the assignment; ties to the lowest index everywhere), so that the kernels can be held to it array for array, and
pinned itself against the fixtures the reference's own matcher minted (tests/golden/matcher_*.npz, tests/test_matcher_cpu.py).

Inputs are per image.  `class_table` [Q, T] float32 is the focal table pos - neg of matcher.py:327-330 (formed by the
caller with the reference's elementwise operations); everything here is float32 numpy arithmetic, one rounding per
operation, no fused multiply-adds.
"""
import numpy as np

F = np.float32
GIOU_WEIGHT, PRIOR_PENALTY, BG_PENALTY, TAKEN_PENALTY = F(3.0), F(100.0), F(10000.0), F(100000.0)   # matcher.py:338,340,415
CENTRE_HALF = F(2.5 / 32)       # 1 * center_radius / expanded_strides (matcher.py:323,367)
TOP_IOU = 10                    # matcher.py:391


def cost_terms(class_table, boxes, tgt_boxes, positive_map):
    """-> cost [Q, G] (before the background penalty of :340), iou [Q, G], flags [Q, G] uint8: bit 0 = in box | in centre,
    bit 1 = a box of the pair fails generalized_box_iou's assert (util/box_ops.py:76-77: x1 < x0, y1 < y0 or NaN)."""
    table = np.asarray(class_table, F)
    bq, g = np.asarray(boxes, F), np.asarray(tgt_boxes, F)
    pm = np.asarray(positive_map) != 0
    Q, G = bq.shape[0], g.shape[0]
    cls = np.empty((Q, G), F)
    for k in range(G):
        toks = np.nonzero(pm[k])[0]
        s = np.zeros(Q, F)
        for t in toks:                                  # ascending token order, one float32 add each
            s = (s + table[:, t]).astype(F)
        with np.errstate(divide="ignore", invalid="ignore"):
            cls[:, k] = s * (F(1.0) / F(len(toks)))     # PyTorch's mean: the sum times fl(1 / count)
    half = F(0.5)
    bx0, by0, bx1, by1 = bq[:, 0] - half * bq[:, 2], bq[:, 1] - half * bq[:, 3], bq[:, 0] + half * bq[:, 2], bq[:, 1] + half * bq[:, 3]
    gx0, gy0, gx1, gy1 = g[:, 0] - half * g[:, 2], g[:, 1] - half * g[:, 3], g[:, 0] + half * g[:, 2], g[:, 1] + half * g[:, 3]
    area1, area2 = (bx1 - bx0) * (by1 - by0), (gx1 - gx0) * (gy1 - gy0)
    iw = np.maximum(np.minimum(bx1[:, None], gx1) - np.maximum(bx0[:, None], gx0), F(0))
    ih = np.maximum(np.minimum(by1[:, None], gy1) - np.maximum(by0[:, None], gy0), F(0))
    inter = iw * ih
    uni = (area1[:, None] + area2) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = inter / uni
        hw = np.maximum(np.maximum(bx1[:, None], gx1) - np.minimum(bx0[:, None], gx0), F(0))
        hh = np.maximum(np.maximum(by1[:, None], gy1) - np.minimum(by0[:, None], gy0), F(0))
        hull = hw * hh
        giou = iou - (hull - uni) / (hull + F(1e-7))
    cx, cy = bq[:, 0][:, None], bq[:, 1][:, None]
    in_box = (cx > gx0) & (cx < gx1) & (cy > gy0) & (cy < gy1)
    in_ctr = (cx > g[:, 0] - CENTRE_HALF) & (cx < g[:, 0] + CENTRE_HALF) & (cy > g[:, 1] - CENTRE_HALF) & (cy < g[:, 1] + CENTRE_HALF)
    cost = (cls + GIOU_WEIGHT * (-giou)) + PRIOR_PENALTY * np.where(in_box & in_ctr, F(0), F(1))
    assert cost.dtype == F and iou.dtype == F
    with np.errstate(invalid="ignore"):
        bad_q, bad_g = ~(bx1 >= bx0) | ~(by1 >= by0), ~(gx1 >= gx0) | ~(gy1 >= gy0)
    degenerate = bad_q[:, None] | bad_g[None, :]
    return cost, iou, ((in_box | in_ctr).astype(np.uint8) | (degenerate.astype(np.uint8) << 1))


def _first_argmin(v):
    """torch.min / argmin with indices: the first minimum -- and NaN PROPAGATES: the first NaN is the minimum (top-k, in
    contrast, sorts NaN behind everything: handled where it is used)."""
    return int(np.argmin(np.where(np.isnan(v), -np.inf, v)))


def dynamic_k(cost, iou, flags, max_rounds=10000):
    """cost [Q, G] float32 (MODIFIED IN PLACE as the reference modifies it), iou, flags as from cost_terms.
    -> (selected_query int64 ascending, gt_index int64, matched_query int64 [G], matching uint8 [Q, G], status: bit 1 (2) =
    the repair loop was cut off, bit 2 (4) = a degenerate box)."""
    Q, G = cost.shape
    assert cost.dtype == F and G > 0
    fg = (flags & 1).any(1)
    degenerate = 4 if (flags & 2).any() else 0          # the reference asserts and aborts; include/ota_hip.h: status bit 2
    cost[~fg] = cost[~fg] + BG_PENALTY                                                        # :340
    M = np.zeros((Q, G), np.uint8)
    ncand = min(Q, TOP_IOU)
    for g in range(G):
        col = iou[:, g]
        order = np.lexsort((np.arange(Q), -np.where(np.isnan(col), -np.inf, col)))[:ncand]    # descending value, ascending index
        s = F(0)
        for q in order:
            if not np.isnan(col[q]):
                s = F(s + col[q])                                                             # summed in descending order
        k = 1 if np.isnan(col).any() else max(int(s), 1)                                      # :397 (topk ranks NaN largest: NaN sum -> 1)
        c = np.where(np.isnan(cost[:, g]), np.inf, cost[:, g])
        M[np.lexsort((np.arange(Q), c))[:k], g] = 1                                           # :399-402, ties to the lowest index
    claims = M.sum(1)
    stale = claims > 1                                                                        # never refreshed (:406 vs :432)
    for q in np.nonzero(stale)[0]:
        a = _first_argmin(cost[q])
        M[q] = 0
        M[q, a] = 1
    status = 0
    rounds = 0
    while (M.sum(0) == 0).any():                                                              # :417
        if rounds >= max_rounds:
            status = 2
            break
        rounds += 1
        taken = M.sum(1) > 0
        cost[taken] = cost[taken] + TAKEN_PENALTY                                             # :420
        for g in np.nonzero(M.sum(0) == 0)[0]:
            M[_first_argmin(cost[:, g]), g] = 1                                               # :423-424
        if (M.sum(1) > 1).any():                                                              # :426
            for q in np.nonzero(stale)[0]:                                                    # the STALE rows (:427-429)
                a = _first_argmin(cost[q])
                M[q] = 0
                M[q, a] = 1
    sel = np.nonzero(M.sum(1) > 0)[0].astype(np.int64)
    gt = np.array([int(np.argmax(M[q])) for q in sel], np.int64)                              # first maximum (:440)
    matched = np.zeros(G, np.int64)
    for g in range(G):
        with np.errstate(invalid="ignore"):
            c = np.where(M[:, g] != 0, cost[:, g], cost[:, g] + F(np.inf))                    # :439 (NaN + inf stays NaN)
        matched[g] = _first_argmin(c)                                                         # :440
    return sel, gt, matched, M, status | degenerate
