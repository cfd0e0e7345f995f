#!/usr/bin/env python
"""Which term of matcher_cost_hip_f32 differs from the PyTorch composition (GPU box)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_matcher_cpu import _case  # noqa: E402
from uninext_amd import ext  # noqa: E402
from uninext_amd.matcher import focal_token_cost, generalized_box_iou, box_cxcywh_to_xyxy  # noqa: E402

def ulps(a, b):
    return (a.view(torch.int32).to(torch.int64) - b.view(torch.int32).to(torch.int64)).abs()

for name in ("matcher_q900_t1", "matcher_q900_t256", "matcher_encoder_q22223"):
    g, bs, outputs, targets = _case(name, device="cuda:0")
    logits = outputs["pred_logits"].flatten(0, 1); boxes = outputs["pred_boxes"].flatten(0, 1)
    tm = torch.cat([t["positive_map"] for t in targets]); tb = torch.cat([t["boxes"] for t in targets])
    cls = focal_token_cost(logits.sigmoid(), tm)
    l1 = torch.cdist(boxes, tb, p=1)
    gi = -generalized_box_iou(box_cxcywh_to_xyxy(boxes), box_cxcywh_to_xyxy(tb))
    for nm, w, ref in (("class", (1, 0, 0), cls), ("bbox", (0, 1, 0), l1), ("giou", (0, 0, 1), gi)):
        f = ext.matcher_cost(logits, boxes, tb, tm, *w)
        r = (0.0 * l1 + 0.0 * cls) + 0.0 * gi      # the zero terms of the kernel's sum, as it adds them
        r = {"class": (0.0 * l1 + 1.0 * cls) + 0.0 * gi, "bbox": (1.0 * l1 + 0.0 * cls) + 0.0 * gi, "giou": (0.0 * l1 + 0.0 * cls) + 1.0 * gi}[nm]
        d = ulps(f, r)
        i = int(d.argmax()); q, k = divmod(i, f.shape[1])
        print("%-24s %-5s max ulp %6d  max abs %.3e  at (%d,%d): fused %.9g ref %.9g" % (name, nm, int(d.max()), float((f - r).abs().max()), q, k, float(f[q, k]), float(r[q, k])))
    # elementary functions: sigmoid / log through a one-token target
    x = logits[:, :1].contiguous()
    p = x.sigmoid()
    print("   prob range %.3g..%.3g" % (float(p.min()), float(p.max())))

# timing: the PyTorch composition against the one kernel
from uninext_amd.matcher import HungarianMatcherVL  # noqa: E402
for name in ("matcher_q900_t256", "matcher_encoder_q22223", "matcher_many_gt"):
    g, bs, outputs, targets = _case(name, device="cuda:0")
    logits = outputs["pred_logits"].flatten(0, 1); boxes = outputs["pred_boxes"].flatten(0, 1)
    tm = torch.cat([t["positive_map"] for t in targets]); tb = torch.cat([t["boxes"] for t in targets])
    m = HungarianMatcherVL(cost_class=2, cost_bbox=5, cost_giou=2)
    res = {}
    for fused in (False, True):
        m.fused_cost = fused
        for _ in range(3):
            m.cost_matrix(logits, boxes, tm, tb)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            m.cost_matrix(logits, boxes, tm, tb)
        b.record(); torch.cuda.synchronize()
        res[fused] = a.elapsed_time(b) / 20 * 1e3
    print("%-24s [%d x %d], %d tokens: composition %.0f us, fused (incl. the CSR of the positive map and the degenerate-box asserts) %.0f us" % (
        name, logits.shape[0], tb.shape[0], logits.shape[1], res[False], res[True]))
