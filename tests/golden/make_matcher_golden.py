"""Mint known-answer index assignments with the REFERENCE matcher (build container only).

    python tests/golden/make_matcher_golden.py

Loads projects/UNINEXT/uninext/models/deformable_detr/matcher.py and uninext/util/box_ops.py from the
reference checkout under a synthetic package skeleton (the real package __init__ pulls detectron2 and
friends; matcher.py uses relative imports, matcher.py:17-18) with a 2-function stand-in for the missing
torchvision (`box_area`, `box_iou`; only used at matcher.py:86,326 and box_ops.py:14), then runs
`HungarianMatcherVL(cost_class=2, cost_bbox=5, cost_giou=2)` (the shipped weights, uninext/config.py:152-154)
`.forward` (matcher.py:449-503) and `.forward_ota` (matcher.py:286-447) on seeded CPU inputs.  No reference test
pins these results, so these fixtures ARE the pin: inputs and the integer outputs are stored in
tests/golden/matcher_*.npz.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("UNINEXT_REFERENCE", "/root/reference")
UX = os.path.join(REF, "projects/UNINEXT/uninext")
HERE = os.path.dirname(os.path.abspath(__file__))


def _box_area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def _box_iou(b1, b2):  # torchvision.ops.box_iou
    a1, a2 = _box_area(b1), _box_area(b2)
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (a1[:, None] + a2 - inter)


def load_reference_matcher():
    tv = types.ModuleType("torchvision"); tv.__path__ = []
    tvo = types.ModuleType("torchvision.ops"); tvo.__path__ = []
    tvb = types.ModuleType("torchvision.ops.boxes")
    tvb.box_area = _box_area
    tvo.box_iou = _box_iou
    tvo.boxes = tvb
    tv.ops = tvo
    sys.modules.update({"torchvision": tv, "torchvision.ops": tvo, "torchvision.ops.boxes": tvb})
    for name in ("refpkg", "refpkg.util", "refpkg.models", "refpkg.models.deformable_detr"):
        m = types.ModuleType(name); m.__path__ = []
        sys.modules[name] = m

    def load(modname, path):
        spec = importlib.util.spec_from_file_location(modname, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod
    load("refpkg.util.box_ops", os.path.join(UX, "util/box_ops.py"))
    return load("refpkg.models.deformable_detr.matcher", os.path.join(UX, "models/deformable_detr/matcher.py"))


def make_case(seed, bs, Q, T, gts, duplicate=False, one_token=False, clustered=False):
    """pred_logits [bs,Q,T], pred_boxes [bs,Q,4] cxcywh in (0,1); targets: boxes [G,4], positive_map [G,T] bool."""
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(bs, Q, T, generator=g) * 2.0 - 2.0
    cxcy = torch.rand(bs, Q, 2, generator=g)
    wh = 0.02 + 0.4 * torch.rand(bs, Q, 2, generator=g) ** 2
    boxes = torch.cat([cxcy, wh], -1)
    targets = []
    for b in range(bs):
        G = gts[b]
        tc = 0.1 + 0.8 * torch.rand(G, 2, generator=g)
        tw = 0.03 + 0.3 * torch.rand(G, 2, generator=g)
        if clustered:   # many overlapping targets (some identical) fighting for few queries: the repair loop of matcher.py:417-435 runs
            tc = 0.45 + 0.1 * torch.rand(G, 2, generator=g)
            tw = 0.2 + 0.05 * torch.rand(G, 2, generator=g)
            tc[1::3] = tc[0::3][:len(tc[1::3])]
            tw[1::3] = tw[0::3][:len(tw[1::3])]
        tb = torch.cat([tc, tw], -1)
        pm = torch.zeros(G, T, dtype=torch.bool)
        for k in range(G):
            if one_token or T == 1:
                pm[k, int(torch.randint(0, T, (1,), generator=g))] = True
            else:
                start = int(torch.randint(1, T - 4, (1,), generator=g))
                ntok = int(torch.randint(1, 4, (1,), generator=g))   # class names of 1..3 tokens
                pm[k, start:start + ntok] = True
        if duplicate and G >= 2:   # identical ground truths and identical predictions: exact cost ties
            tb[1] = tb[0]
            pm[1] = pm[0]
            boxes[b, 5] = boxes[b, 4]
            logits[b, 5] = logits[b, 4]
        # some predictions sit right on ground-truth boxes so that the in-box/centre prior and IoU top-k matter
        for k in range(min(G, 6)):
            boxes[b, 10 + k] = tb[k] + 0.01 * torch.randn(4, generator=g)
            boxes[b, 10 + k, 2:].clamp_(min=0.01)
        targets.append({"boxes": tb, "positive_map": pm})
    return logits, boxes, targets


CASES = {
    "matcher_q900_t256": dict(seed=1, bs=2, Q=900, T=256, gts=[7, 19]),
    "matcher_q300_t256_ties": dict(seed=2, bs=2, Q=300, T=256, gts=[5, 2], duplicate=True),
    "matcher_q900_t1": dict(seed=3, bs=2, Q=900, T=1, gts=[3, 1]),
    "matcher_empty_image": dict(seed=4, bs=3, Q=100, T=256, gts=[4, 0, 1]),
    "matcher_many_gt": dict(seed=5, bs=1, Q=900, T=256, gts=[60]),
    "matcher_encoder_q22223": dict(seed=6, bs=1, Q=22223, T=16, gts=[11], one_token=True),
    "matcher_ota_conflicts": dict(seed=7, bs=2, Q=120, T=64, gts=[30, 45], clustered=True),
}


def main():
    ref = load_reference_matcher()
    matcher = ref.HungarianMatcherVL(cost_class=2, cost_bbox=5, cost_giou=2)
    for name, kw in CASES.items():
        logits, boxes, targets = make_case(**kw)
        outputs = {"pred_logits": logits, "pred_boxes": boxes}
        hung = matcher.forward(outputs, [dict(t) for t in targets])
        save = {"pred_logits": logits.numpy(), "pred_boxes": boxes.numpy(), "bs": np.int64(kw["bs"])}
        for b, t in enumerate(targets):
            save[f"tgt_boxes_{b}"] = t["boxes"].numpy()
            save[f"tgt_posmap_{b}"] = t["positive_map"].numpy()
            save[f"hung_i_{b}"] = hung[b][0].numpy()
            save[f"hung_j_{b}"] = hung[b][1].numpy()
        if kw["Q"] <= 1000:   # forward_ota is only ever called on decoder queries (ddetrs_dn.py:231)
            ota, matched = matcher.forward_ota(outputs, [dict(t) for t in targets])
            for b in range(kw["bs"]):
                save[f"ota_q_{b}"] = ota[b][0].numpy()
                save[f"ota_g_{b}"] = ota[b][1].numpy()
                save[f"ota_matched_{b}"] = (matched[b].numpy() if torch.is_tensor(matched[b])
                                            else np.asarray(matched[b], dtype=np.int64))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **save)
        print(name, [(len(h[0])) for h in hung])


if __name__ == "__main__":
    main()
