#!/usr/bin/env python
"""Static mask head (MaskHeadSmallConv.forward, ddetrs_dn.py:991-1025) at the R50 800x1333 shapes, bs 2, exact fp32 both ways:
every 3x3 convolution through PyTorch-ROCm / MIOpen (the module's default) against every one through this library's own exact
MFMA convolution (conv3x3_hip_packed_exact_f32, cached packed weights).  Module forward and the five layers one by one, rounds
interleaved so that both routes see the same clocks.  GPU box only.

    [MSDA_HIP_LIB=...] python tools/maskhead_exact_ab.py [--reps 20] [--rounds 5]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uninext_amd import ext, mask_head  # noqa: E402
from uninext_amd._cache import packed_weight  # noqa: E402

LAYERS = [("lay3", 256, 256, 25, 42), ("lay4", 256, 256, 50, 84), ("jia_dcn", 256, 256, 100, 167),
          ("lay1", 256, 64, 100, 167), ("lay2", 64, 8, 100, 167)]


def own_conv3x3_relu(x, conv, exact=True):
    pe = packed_weight(conv, lambda w: ext.conv3x3_pack_weight(w, exact=True), slot="_msda_packed_exact")
    return ext.conv3x3_packed_forward(x.contiguous(), pe, conv.weight.shape[0], conv.bias, relu=True, exact=True)


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    print("library:", os.environ.get("MSDA_HIP_LIB", "uninext_amd/lib/libmsda_hip.so"))
    with torch.no_grad():
        for name, cin, cout, H, W in LAYERS:
            x = torch.randn(2, cin, H, W, device=dev)
            conv = torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
            to, tm = [], []
            for _ in range(args.rounds):
                to.append(timeit(lambda: own_conv3x3_relu(x, conv), args.reps))
                tm.append(timeit(lambda: torch.relu_(conv(x)), args.reps))
            ref = torch.relu(F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1))
            eo = float((own_conv3x3_relu(x, conv).double() - ref).abs().max()) / float(ref.abs().max())
            em = float((torch.relu_(conv(x)).double() - ref).abs().max()) / float(ref.abs().max())
            print("%-8s own exact %7.1f us (min %7.1f) err %.1e | MIOpen %7.1f us (min %7.1f) err %.1e"
                  % (name, sorted(to)[len(to) // 2], min(to), eo, sorted(tm)[len(tm) // 2], min(tm), em))
        head = mask_head.MaskHeadSmallConv(256, None, 256).to(dev).eval()
        xs = [torch.randn(2, 256, h, w, device=dev) for h, w in ((100, 167), (50, 84), (25, 42))]
        head.exact_fp32 = True
        to, tm = [], []
        for _ in range(args.rounds):
            head.own_exact_conv = True                 # the default since round 6: conv3x3_hip_packed_exact_f32 on every layer
            to.append(timeit(lambda: head(xs, None), args.reps))
            y_own = head(xs, None)
            head.own_exact_conv = False                # MIOpen
            tm.append(timeit(lambda: head(xs, None), args.reps))
            y_lib = head(xs, None)
        d = float((y_own - y_lib).abs().max()) / float(y_lib.abs().max())
        print("MaskHeadSmallConv.forward: own exact %7.1f us (min %7.1f) | MIOpen route %7.1f us (min %7.1f) | max difference %.1e of the output scale"
              % (sorted(to)[len(to) // 2], min(to), sorted(tm)[len(tm) // 2], min(tm), d))


if __name__ == "__main__":
    main()
