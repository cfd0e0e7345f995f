#!/usr/bin/env python
"""simOTA assignment (HungarianMatcherVL.forward_ota) at the training shapes of BASELINE configs[4] (GPU box only):
the two HIP kernels of include/ota_hip.h beside the PyTorch composition (the reference's data flow, per-target loops and
host syncs included), wall-clock per call incl. the one host copy, and the kernels alone by HIP events.

    python tools/ota_bench.py [--bs 2] [--queries 900] [--tokens 256] [--gts 7,19]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uninext_amd.matcher import HungarianMatcherVL  # noqa: E402


def make(bs, Q, T, gts, seed, dev):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(bs, Q, T, generator=g) * 2.0 - 2.0
    boxes = torch.cat([torch.rand(bs, Q, 2, generator=g), 0.02 + 0.4 * torch.rand(bs, Q, 2, generator=g) ** 2], -1)
    targets = []
    for b in range(bs):
        G = gts[b % len(gts)]
        tb = torch.cat([0.1 + 0.8 * torch.rand(G, 2, generator=g), 0.03 + 0.3 * torch.rand(G, 2, generator=g)], -1)
        pm = torch.zeros(G, T, dtype=torch.bool)
        for k in range(G):
            s = int(torch.randint(1, max(T - 4, 2), (1,), generator=g))
            pm[k, s:s + int(torch.randint(1, 4, (1,), generator=g))] = True
        for k in range(min(G, 6)):
            boxes[b, 10 + k] = tb[k] + 0.01 * torch.randn(4, generator=g)
            boxes[b, 10 + k, 2:].clamp_(min=0.01)
        targets.append({"boxes": tb.to(dev), "positive_map": pm.to(dev)})
    return {"pred_logits": logits.to(dev), "pred_boxes": boxes.to(dev)}, targets


def wall(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=2)
    ap.add_argument("--queries", type=int, default=900)
    ap.add_argument("--tokens", type=int, default=256)
    ap.add_argument("--gts", default="7,19")
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for gts in ([int(x) for x in a.gts.split(",")], [1, 3], [40, 60]):
        outputs, targets = make(a.bs, a.queries, a.tokens, gts, 1, dev)
        m = HungarianMatcherVL(cost_class=2, cost_bbox=5, cost_giou=2)
        m.device_ota = True
        dev_idx, _ = m.forward_ota(outputs, targets)
        t_dev = wall(lambda: m.forward_ota(outputs, targets), a.reps)
        prob = outputs["pred_logits"].sigmoid()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        m.ota_device_launch(prob, outputs["pred_boxes"], targets)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.reps):
            m.ota_device_launch(prob, outputs["pred_boxes"], targets)
        e1.record()
        torch.cuda.synchronize()
        t_k = e0.elapsed_time(e1) / a.reps * 1e3
        m.device_ota = False
        cmp_idx, _ = m.forward_ota(outputs, targets)
        t_cmp = wall(lambda: m.forward_ota(outputs, targets), max(a.reps // 3, 3))
        m.batched_topk = False
        t_ref = wall(lambda: m.forward_ota(outputs, targets), max(a.reps // 3, 3))
        same = all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(dev_idx, cmp_idx))
        print("bs %d Q %d T %d targets/image %s: device path %.0f us per call (table + two kernels + allocations on the stream %.0f us), "
              "PyTorch composition %.0f us (batched top-k) / %.0f us (the reference's per-target loop); same indices: %s"
              % (a.bs, a.queries, a.tokens, gts, t_dev, t_k, t_cmp, t_ref, same))


if __name__ == "__main__":
    main()
