#!/usr/bin/env python
"""Dynamic mask head timing on the GPU box (inference, bs 2, 100x167 mask features, 900 instances per image as in
UNINEXT's coco_inference, ddetrs_dn.py:469-488): HIP kernels vs the PyTorch composition and the reference-style
(materialising) algorithm.  fp32 VALU roofline: 157.3 TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uninext_amd import ext, mask_head  # noqa: E402


def timeit(fn, reps=10):
    for _ in range(2):
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
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    N, H, W = 2, 100, 167
    num_insts = [900, 900]
    n_all = sum(num_insts)
    feats = torch.randn(N, 8, H, W, generator=g).to(dev)
    ref = (torch.rand(1, n_all, 2, generator=g) * torch.tensor([W * 8.0, H * 8.0])).to(dev)
    params = (torch.randn(1, n_all, 169, generator=g) * 0.3).to(dev)
    xy, pr = ref.reshape(-1, 2).contiguous(), params.flatten(0, 1).contiguous()
    flop = 2.0 * 152 * n_all * H * W
    with torch.no_grad():
        from uninext_amd import _lib
        lib = _lib.load()
        base = None
        for v in (1, 2, 3):      # A/B of the kernels behind the same entry point (include/dynmask_hip.h)
            lib.dynmask_hip_set_variant(v)
            o = ext.dynmask_forward(feats, xy, pr, num_insts, 8, True)
            name = lib.dynmask_hip_last_kernel().decode()
            us_v = timeit(lambda: ext.dynmask_forward(feats, xy, pr, num_insts, 8, True))
            base = o if base is None else base
            print("  variant %d %-22s %9.1f us  %6.1f TFLOP/s (%.1f%% of 157.3 TF)  max |diff| vs variant 1 %.3g"
                  % (v, name, us_v, flop / us_v / 1e6, flop / us_v / 1e6 / 1.573, float((o - base).abs().max())))
        lib.dynmask_hip_set_variant(0)
        if "--ab-only" in sys.argv:
            return
        us = timeit(lambda: ext.dynmask_forward(feats, xy, pr, num_insts, 8, True))
        print("dynmask_hip_forward_f32      %9.1f us  %6.1f TFLOP/s (%.1f%% of the 157.3 TF fp32 VALU peak), %.0f GB/s written"
              % (us, flop / us / 1e6, flop / us / 1e6 / 1.573, n_all * H * W * 4 / us / 1e3))
        logits = ext.dynmask_forward(feats, xy, pr, num_insts, 8, True).reshape(-1, 1, H, W)
        us2 = timeit(lambda: ext.aligned_bilinear_forward(logits, 2))
        print("aligned_bilinear_hip_f32 x2  %9.1f us  %6.0f GB/s (read + write)" % (us2, n_all * H * W * 4 * 5 / us2 / 1e3))
        us3 = timeit(lambda: mask_head._dynamic_convs_torch(feats, xy, pr, num_insts, 8, True), reps=3)
        print("torch composition (convs)    %9.1f us" % us3)
        us4 = timeit(lambda: mask_head._aligned_bilinear_torch(logits, 2), reps=3)
        print("torch aligned_bilinear       %9.1f us" % us4)
        try:
            from oracle.dynmask_torch import dynamic_mask_oracle
            us5 = timeit(lambda: dynamic_mask_oracle(feats, ref, params, num_insts, 8), reps=2)
            print("reference algorithm (repeat/cat + grouped conv2d + upsample) %9.1f us" % us5)
        except Exception as e:  # e.g. out of memory
            print("reference algorithm failed:", type(e).__name__)


if __name__ == "__main__":
    main()
