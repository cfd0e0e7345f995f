"""Query <-> ground-truth assignment used by the UNINEXT criterion: the token-level Hungarian matcher
(encoder proposals, `OTA: False` configs) and the simOTA dynamic-k matcher (every decoder layer of the shipped
configs, `MODEL.OTA: True`).

Host-side mirror of the reference's `HungarianMatcherVL`
(projects/UNINEXT/uninext/models/deformable_detr/matcher.py:261-503: `forward` :449-503, `forward_ota` :286-311,
`compute_cost` :313-342, `get_in_boxes_info` :344-385, `dynamic_k_matching` :387-447) and of the box helpers it
uses (uninext/util/box_ops.py:17-85, torchvision.ops.box_iou).  Same constructor, method names, argument meaning
and return structure.  The result is INTEGER index tensors, so parity is bit-exact equality: the floating-point
cost is built from the same torch operations in the same order (a different summation order could flip a near-tie
in the assignment), pinned by reference-minted fixtures (tests/golden/matcher_*.npz, tests/test_matcher_cpu.py).

What differs from the reference: the focal cost table (pos - neg) is built once instead of inside the per-target
loop, boxes are converted once, and the code is device-agnostic (no `.cuda()`).  The linear-sum-assignment: for GPU
inputs (`device_lsap = True`, the default) it runs ON THE DEVICE (include/lsap_hip.h, csrc/lsap.hip: SciPy's
shortest-augmenting-path algorithm with SciPy's tie rules, index-for-index equal to it, tests/test_lsap_gpu.py) and
only the min(Q, G) index pairs come back to the host; for CPU inputs, or with `device_lsap = False`, it is SciPy's
`scipy.optimize.linear_sum_assignment` (any SciPy > 1.5.1 as in the reference's setup.py:185) on the host, exactly
where the reference runs it (matcher.py:499-502).

simOTA (`forward_ota`, the matcher of every decoder layer in the shipped configs): for GPU fp32 inputs (`device_ota =
True`, the default) the whole batch is assigned by two HIP kernels (include/ota_hip.h, csrc/ota.hip) that follow
matcher.py:313-447 operation for operation and tie rule for tie rule: EXACTLY the float32 numpy restatement oracle/ota_oracle.py
(which returns the reference's integers on every reference-minted fixture), and PyTorch's composition to within one unit in the
last place of a cost where PyTorch's own GPU reductions (`mean(-1)` over three or more tokens, `sum(0)` over the ten IoUs)
add in another order than the sequential one used here -- enough to flip a selection only between candidates whose costs are
that close.  The focal table `pos - neg` that feeds them is formed by the reference's own elementwise PyTorch operations.
Degenerate boxes (x1 < x0, y1 < y0, NaN) raise the reference's AssertionError behind the one host copy.  Nothing synchronises with the host until the
selected-query counts are copied back, once per call, to cut the index tensors to their lengths
(tests/test_matcher_gpu.py asserts that with torch.cuda.set_sync_debug_mode).  CPU inputs and `device_ota = False` run
the PyTorch composition below -- the reference's data flow with its per-target loops.

`fused_cost` (Hungarian cost matrix in one kernel; OPT-IN since round 6) evaluates the composition's float32 operations in its
order, but its logf is this ROCm's device library's, which differs from the one PyTorch was built with by one unit in the last
place on a third of the arguments, and PyTorch's `mean(-1)` over five or more tokens adds in another order: the class term agrees
with the composition to 5e-7, not bitwise (tests/test_matcher_gpu.py).  The assignment of every reference-minted fixture is
unchanged, but a near-tie between two queries inside that margin could flip silently -- so the DEFAULT cost matrix is the
reference's own PyTorch composition on the device (bitwise the reference's by construction), and only the
assignment itself (SciPy's algorithm with SciPy's tie rules, include/lsap_hip.h) is this library's.
"""
import torch
from scipy.optimize import linear_sum_assignment
from torch import nn

FOCAL_ALPHA = 0.25          # matcher.py:327,480
FOCAL_GAMMA = 2.0           # matcher.py:328,481
OTA_GIOU_WEIGHT = 3.0       # matcher.py:338
OTA_PRIOR_PENALTY = 100.0   # matcher.py:338
OTA_BG_PENALTY = 10000.0    # matcher.py:340
OTA_CENTER_RADIUS = 2.5     # matcher.py:367
OTA_STRIDE = 32             # matcher.py:323
OTA_TOPK_CANDIDATES = 10    # matcher.py:391


# ---- box helpers (uninext/util/box_ops.py:17-85) ----------------------------------------------------
def box_cxcywh_to_xyxy(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def box_area(boxes):
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def _inter_union(boxes1, boxes2):
    area1, area2 = box_area(boxes1), box_area(boxes2)
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter, area1[:, None] + area2 - inter


def box_iou(boxes1, boxes2):
    """Pairwise IoU [N, M] of xyxy boxes (torchvision.ops.box_iou)."""
    inter, union = _inter_union(boxes1, boxes2)
    return inter / union


def generalized_box_iou(boxes1, boxes2):
    """Pairwise GIoU [N, M] of xyxy boxes (box_ops.py:62-85), same degenerate-box asserts."""
    assert (boxes1[:, 2:] >= boxes1[:, :2]).all()
    assert (boxes2[:, 2:] >= boxes2[:, :2]).all()
    inter, union = _inter_union(boxes1, boxes2)
    iou = inter / union
    lt = torch.min(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.max(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    hull = wh[:, :, 0] * wh[:, :, 1]
    return iou - (hull - union) / (hull + 1e-7)


# ---- cost terms -----------------------------------------------------------------------------------------
def focal_token_cost(prob, positive_map):
    """[num_pred, num_gt] classification cost: focal(pos) - focal(neg) averaged over each target's positive
    tokens (matcher.py:327-334 / 480-486; the mean handles class names that span several tokens)."""
    neg = (1 - FOCAL_ALPHA) * (prob ** FOCAL_GAMMA) * (-(1 - prob + 1e-8).log())
    pos = FOCAL_ALPHA * ((1 - prob) ** FOCAL_GAMMA) * (-(prob + 1e-8).log())
    table = pos - neg
    cost = torch.zeros((prob.size(0), positive_map.size(0)), device=prob.device)
    for k in range(positive_map.size(0)):
        cost[:, k] = table[:, positive_map[k]].mean(-1)
    return cost


class HungarianMatcherVL(nn.Module):
    """1-to-1 (`forward`) and dynamic-k (`forward_ota`) assignment between predictions and targets.

    outputs: {"pred_logits": [bs, Q, T] token logits, "pred_boxes": [bs, Q, 4] cxcywh in [0, 1]}
    targets: list (bs) of {"boxes": [G, 4] cxcywh, "positive_map": [G, T] bool (or index) token map}
    """

    batched_topk = True   # dynamic-k selection without a host sync per ground-truth box (same indices; see the tests)
    device_lsap = True    # GPU inputs: solve the assignment on the device (include/lsap_hip.h) instead of C.cpu() + SciPy
    fused_cost = False    # opt-in (GPU fp32 inputs): the cost matrix in one kernel (include/matcher_cost_hip.h) instead of ~40 launches;
                          # its class term is 1 ulp of logf from the composition's -- the default is the bit-exact composition
    device_ota = True     # GPU fp32 inputs: simOTA of the whole batch in two kernels (include/ota_hip.h), one host sync per call

    def __init__(self, cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1, cost_mask: float = 1):
        super().__init__()
        self.cost_class, self.cost_bbox, self.cost_giou, self.cost_mask = cost_class, cost_bbox, cost_giou, cost_mask
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0 or cost_mask != 0, "all costs cant be 0"

    def cost_matrix(self, logits, boxes, tgt_map, tgt_boxes):
        """[num_pred, num_gt] cost of matcher.py:476-498: the reference's composition (default: bitwise the reference's on the same
        device); with `fused_cost = True` on GPU fp32 inputs one kernel with the composition's float32 operation order (class
        term to 1 ulp of logf, see the module docstring)."""
        if (self.fused_cost and logits.is_cuda and logits.dtype == torch.float32 and boxes.dtype == torch.float32
                and tgt_boxes.dtype == torch.float32 and tgt_boxes.shape[0] > 0 and tgt_map.dtype in (torch.bool, torch.uint8)):
            from . import ext as _ext
            xyxy, gxyxy = box_cxcywh_to_xyxy(boxes), box_cxcywh_to_xyxy(tgt_boxes)      # the reference's degenerate-box asserts
            assert (xyxy[:, 2:] >= xyxy[:, :2]).all() and (gxyxy[:, 2:] >= gxyxy[:, :2]).all()
            return _ext.matcher_cost(logits, boxes, tgt_boxes, tgt_map, self.cost_class, self.cost_bbox, self.cost_giou)
        cost_class = focal_token_cost(logits.sigmoid(), tgt_map)
        cost_bbox = torch.cdist(boxes, tgt_boxes, p=1)
        cost_giou = -generalized_box_iou(box_cxcywh_to_xyxy(boxes), box_cxcywh_to_xyxy(tgt_boxes))
        return self.cost_bbox * cost_bbox + self.cost_class * cost_class + self.cost_giou * cost_giou

    # -- Hungarian (matcher.py:449-503) ------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, outputs, targets):
        """Returns a list (bs) of (index_pred int64, index_target int64), each of length min(Q, G)."""
        bs, num_queries = outputs["pred_logits"].shape[:2]
        boxes = outputs["pred_boxes"].flatten(0, 1)
        tgt_map = torch.cat([t["positive_map"] for t in targets])
        tgt_boxes = torch.cat([t["boxes"] for t in targets])

        cost = self.cost_matrix(outputs["pred_logits"].flatten(0, 1), boxes, tgt_map, tgt_boxes)
        sizes = [len(t["boxes"]) for t in targets]
        if self.device_lsap and cost.is_cuda and cost.dtype == torch.float32:
            # SciPy's algorithm with SciPy's tie rules on the GPU: no copy of the [Q, G] matrix, no host solve; only the
            # min(Q, G) index pairs come back (the reference returns CPU index tensors, matcher.py:503)
            from . import ext as _ext
            blocks = [blk[b] for b, blk in enumerate(cost.view(bs, num_queries, -1).split(sizes, -1))]
            return [(i.cpu(), j.cpu()) for i, j in _ext.lsap_batch(blocks, check=True)]
        cost = cost.view(bs, num_queries, -1).cpu()   # the one device->host copy; LSAP runs on the host

        result = []
        for b, block in enumerate(cost.split(sizes, -1)):
            rows, cols = linear_sum_assignment(block[b])
            result.append((torch.as_tensor(rows, dtype=torch.int64), torch.as_tensor(cols, dtype=torch.int64)))
        return result

    # -- simOTA (matcher.py:286-447) ---------------------------------------------------------------------
    @torch.no_grad()
    def forward_ota(self, outputs, targets, nf=1):
        """Returns (indices, matched_ids): per image ((selected_query int64, gt_index int64), best query per gt)."""
        bs = outputs["pred_logits"].shape[0]
        prob = outputs["pred_logits"].sigmoid()
        boxes = outputs["pred_boxes"]
        if self._ota_on_device(prob, boxes, targets):
            return self._forward_ota_device(prob, boxes, targets, nf)
        indices, matched_ids = [], []
        for b in range(bs):
            cost, ious, gt_boxes = self.compute_cost(b, boxes, prob, targets, nf)
            if gt_boxes.shape[0] > 0:
                pair, best_query = self.dynamic_k_matching(cost, ious, gt_boxes.shape[0])
            else:
                empty = torch.tensor([], dtype=torch.int64, device=prob.device)
                pair, best_query = (empty, empty.clone()), []
            indices.append(pair)
            matched_ids.append(best_query)
        return indices, matched_ids

    def _ota_on_device(self, prob, boxes, targets):
        if not (self.device_ota and prob.is_cuda and prob.dtype == torch.float32 and boxes.dtype == torch.float32
                and prob.dim() == 3 and 0 < prob.shape[0] <= 64 and prob.shape[1] > 0):
            return False
        T = prob.shape[2]
        for t in targets:
            pm, tb = t["positive_map"], t["boxes"]
            if not (pm.is_cuda and pm.dtype in (torch.bool, torch.uint8) and pm.dim() == 2 and pm.shape[1] == T
                    and tb.is_cuda and tb.dtype == torch.float32 and pm.shape[0] <= 4096):
                return False      # (index-valued positive maps, other dtypes: the composition)
        return True

    def ota_device_launch(self, prob, boxes, targets, nf=1):
        """Everything of forward_ota that runs on the device, WITHOUT the host copy: (sel_query [bs, Q], sel_gt [bs, Q],
        matched_query [G_total], num_selected [bs], status [bs], sizes) -- see uninext_amd.ext.ota_assign."""
        from . import ext as _ext
        neg = (1 - FOCAL_ALPHA) * (prob ** FOCAL_GAMMA) * (-(1 - prob + 1e-8).log())        # matcher.py:329-330, all images at once
        pos = FOCAL_ALPHA * ((1 - prob) ** FOCAL_GAMMA) * (-(prob + 1e-8).log())
        table = pos - neg
        sizes = [len(t["positive_map"]) for t in targets]
        gt_boxes = torch.cat([t["boxes"].reshape(n, nf, 4)[:, 0] for t, n in zip(targets, sizes)])    # matcher.py:318
        pm = torch.cat([t["positive_map"] for t in targets])
        return _ext.ota_assign(table, boxes, gt_boxes, pm, sizes) + (sizes,)

    def _forward_ota_device(self, prob, boxes, targets, nf):
        sel_q, sel_g, matched, count, status, sizes = self.ota_device_launch(prob, boxes, targets, nf)
        host = torch.stack((count, status)).cpu()                   # THE host synchronisation of the call
        counts, stats = host[0].tolist(), host[1].tolist()
        if any(s & 4 for s in stats):
            # the reference's generalized_box_iou asserts `(boxes[:, 2:] >= boxes[:, :2]).all()` on both box sets
            # (util/box_ops.py:76-77; NaN fails it) and aborts the step: same exception, raised behind the one host copy
            raise AssertionError("simOTA: degenerate box (x1 < x0, y1 < y0 or NaN) in image(s) %s"
                                 % [b for b, s in enumerate(stats) if s & 4])
        if any(s & 2 for s in stats):
            raise RuntimeError("simOTA: the repair loop of matcher.py:417-435 did not terminate (the reference would spin)")
        indices, matched_ids, off = [], [], 0
        for b, n in enumerate(sizes):
            if n > 0:
                indices.append((sel_q[b, :counts[b]], sel_g[b, :counts[b]]))
                matched_ids.append(matched[off:off + n])
            else:
                empty = torch.tensor([], dtype=torch.int64, device=prob.device)
                indices.append((empty, empty.clone()))
                matched_ids.append([])
            off += n
        return indices, matched_ids

    def compute_cost(self, batch_idx, out_bbox, out_prob, targets, nf):
        boxes, prob = out_bbox[batch_idx], out_prob[batch_idx]
        tgt_map = targets[batch_idx]["positive_map"]
        num_gt = len(tgt_map)
        gt_boxes = targets[batch_idx]["boxes"].reshape(num_gt, nf, 4)[:, 0]
        fg_mask, in_box_and_center = self.get_in_boxes_info(boxes, gt_boxes, expanded_strides=OTA_STRIDE)
        boxes_xyxy, gt_xyxy = box_cxcywh_to_xyxy(boxes), box_cxcywh_to_xyxy(gt_boxes)
        ious = box_iou(boxes_xyxy, gt_xyxy)
        cost_class = focal_token_cost(prob, tgt_map)
        cost_giou = -generalized_box_iou(boxes_xyxy, gt_xyxy)
        cost = cost_class + OTA_GIOU_WEIGHT * cost_giou + OTA_PRIOR_PENALTY * (~in_box_and_center)
        cost[~fg_mask] = cost[~fg_mask] + OTA_BG_PENALTY
        return cost, ious, gt_boxes

    def get_in_boxes_info(self, boxes, target_gts, expanded_strides):
        """(query centre inside ANY gt box or ANY gt centre square) [Q], (inside box AND centre square) [Q, G]."""
        gt_xyxy = box_cxcywh_to_xyxy(target_gts)
        cx, cy = boxes[:, 0].unsqueeze(1), boxes[:, 1].unsqueeze(1)

        def inside(x0, y0, x1, y1):
            hits = (cx > x0.unsqueeze(0)).long() + (cx < x1.unsqueeze(0)).long() \
                + (cy > y0.unsqueeze(0)).long() + (cy < y1.unsqueeze(0)).long()
            return hits == 4

        in_boxes = inside(gt_xyxy[:, 0], gt_xyxy[:, 1], gt_xyxy[:, 2], gt_xyxy[:, 3])
        r = 1 * OTA_CENTER_RADIUS / expanded_strides
        in_centers = inside(target_gts[:, 0] - r, target_gts[:, 1] - r, target_gts[:, 0] + r, target_gts[:, 1] + r)
        candidate = (in_boxes.sum(1) > 0) | (in_centers.sum(1) > 0)
        return candidate, in_boxes & in_centers

    def dynamic_k_matching(self, cost, pair_wise_ious, num_gt):
        """k_g = clamp(int(sum of the 10 best IoUs of gt g), 1) cheapest queries per gt; a query claimed by
        several gts keeps its cheapest one; gts left without a query get their cheapest still-free query.
        `cost` is modified in place exactly as in the reference (matcher.py:415,437)."""
        matching = torch.zeros_like(cost)
        n_query = len(pair_wise_ious)
        topk_ious, _ = torch.topk(pair_wise_ious, min(n_query, OTA_TOPK_CANDIDATES), dim=0)
        dynamic_ks = torch.clamp(topk_ious.sum(0).int(), min=1)
        if self.batched_topk:
            # the reference loops over the gts with one `.item()` (host sync) each (matcher.py:399-402); k_g <= 10, so one
            # top-10 over all columns and a rank mask select the same entries without leaving the device
            kmax = min(n_query, OTA_TOPK_CANDIDATES)
            _, pos = torch.topk(cost, k=kmax, dim=0, largest=False)                      # [kmax, G], ascending cost
            keep = torch.arange(kmax, device=cost.device)[:, None] < dynamic_ks[None, :]
            cols = torch.arange(num_gt, device=cost.device)[None, :].expand_as(pos)
            matching[pos[keep], cols[keep]] = 1.0
        else:
            for g in range(num_gt):
                _, pos = torch.topk(cost[:, g], k=dynamic_ks[g].item(), largest=False)
                matching[:, g][pos] = 1.0

        claims = matching.sum(1)            # NOTE: the reference never refreshes this inside the repair loop below
        if (claims > 1).sum() > 0:
            _, cheapest = torch.min(cost[claims > 1], dim=1)
            matching[claims > 1] *= 0
            matching[claims > 1, cheapest] = 1

        while (matching.sum(0) == 0).any():
            taken = matching.sum(1) > 0
            cost[taken] += 100000.0
            for g in torch.nonzero(matching.sum(0) == 0, as_tuple=False).squeeze(1):
                matching[:, g][torch.argmin(cost[:, g])] = 1.0
            if (matching.sum(1) > 1).sum() > 0:
                _, cheapest = torch.min(cost[claims > 1], dim=1)
                matching[claims > 1] *= 0
                matching[claims > 1, cheapest] = 1

        assert not (matching.sum(0) == 0).any()
        selected = matching.sum(1) > 0
        gt_indices = matching[selected].max(1)[1]
        assert selected.sum() == len(gt_indices)
        cost[matching == 0] = cost[matching == 0] + float("inf")
        best_query_per_gt = torch.min(cost, dim=0)[1]
        selected_query = torch.arange(len(matching)).to(gt_indices)[selected]
        return (selected_query, gt_indices), best_query_per_gt
