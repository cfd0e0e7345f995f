#!/usr/bin/env python
"""Static mask head (MaskHeadSmallConv, ddetrs_dn.py:923-1031) at the R50 800x1333 shapes, bs 2 (GPU box only):
include/conv3x3_hip.h (fp32 MFMA implicit GEMM, bias + ReLU fused) vs the PyTorch-ROCm convolutions.

    python tools/maskhead_bench.py [--reps 20]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uninext_amd import ext  # noqa: E402
from uninext_amd.mask_head import MaskHeadSmallConv  # noqa: E402
MaskHeadSmallConv.exact_fp32 = False   # opt-in since round 4: the module timing below is the split-bf16 path

PEAK_TF = 157.3
LAYERS = [("lay3", 256, 256, 25, 42), ("lay4", 256, 256, 50, 84), ("jia_dcn", 256, 256, 100, 167),
          ("lay1", 256, 64, 100, 167), ("lay2", 64, 8, 100, 167)]


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
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = 2
    tot_h = tot_s = tot_t = tot_f = 0.0
    with torch.no_grad():
        for name, cin, cout, H, W in LAYERS:
            x = torch.randn(B, cin, H, W, device=dev)
            conv = torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
            flop = 2.0 * B * H * W * cout * cin * 9
            t_hip = timeit(lambda: ext.conv3x3_forward(x, conv.weight, conv.bias, relu=True), args.reps)
            packed = ext.conv3x3_pack_weight(conv.weight.detach())
            t_split = timeit(lambda: ext.conv3x3_packed_forward(x, packed, cout, conv.bias, relu=True), args.reps)
            t_torch = timeit(lambda: torch.relu_(conv(x)), args.reps)
            if cin % 16 == 0:      # round 4: the exact fp32 convolution in the halo structure
                pe = ext.conv3x3_pack_weight(conv.weight.detach(), exact=True)
                t_ex = timeit(lambda: ext.conv3x3_packed_forward(x, pe, cout, conv.bias, relu=True, exact=True), args.reps)
                ref64_ = torch.relu(torch.nn.functional.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1))
                e_ex = float((ext.conv3x3_packed_forward(x, pe, cout, conv.bias, relu=True, exact=True).double() - ref64_).abs().max()) / float(ref64_.abs().max())
                print("%-8s exact fp32 HALO kernel (conv3x3_hip_packed_exact_f32) %7.1f us %5.1f TFLOP/s (%4.1f %% of 157.3) err %.0e"
                      % (name, t_ex, flop / t_ex * 1e-6, 100 * flop / t_ex * 1e-6 / PEAK_TF, e_ex))
                tot_e = globals().setdefault("_tot_e", [0.0]); tot_e[0] += t_ex
            ref = torch.relu(conv(x.double().cpu()).float() if False else conv(x))
            ref64 = torch.relu(torch.nn.functional.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1))
            sc = float(ref64.abs().max())
            err0 = float((ext.conv3x3_forward(x, conv.weight, conv.bias, relu=True).double() - ref64).abs().max()) / sc
            err1 = float((ext.conv3x3_packed_forward(x, packed, cout, conv.bias, relu=True).double() - ref64).abs().max()) / sc
            errt = float((ref.double() - ref64).abs().max()) / sc
            tf = flop / t_hip * 1e-6
            tot_h += t_hip; tot_s += t_split; tot_t += t_torch; tot_f += flop
            print("%-8s %3d->%3d @ %3dx%3d %6.2f GFLOP | exact fp32 MFMA %7.1f us %5.1f TFLOP/s (%4.1f %% of 157.3) err %.0e | split-bf16 %7.1f us %6.1f TFLOP/s err %.0e | torch conv+relu %7.1f us err %.0e"
                  % (name, cin, cout, H, W, flop * 1e-9, t_hip, tf, 100 * tf / PEAK_TF, err0, t_split, flop / t_split * 1e-6, err1,
                     t_torch, errt))
        print("five convolutions: exact %.1f us (%.1f TFLOP/s), split-bf16 %.1f us (%.1f TFLOP/s), torch %.1f us; exact halo kernel %.1f us"
              % (tot_h, tot_f / tot_h * 1e-6, tot_s, tot_f / tot_s * 1e-6, tot_t, globals().get("_tot_e", [0.0])[0]))
        head = MaskHeadSmallConv(256, None, 256).to(dev).eval()
        xs = [torch.randn(B, 256, h, w, device=dev) for h, w in ((100, 167), (50, 84), (25, 42))]
        MaskHeadSmallConv.exact_fp32 = True
        t_lib_mod = timeit(lambda: head(xs, None), args.reps)
        print("MaskHeadSmallConv.forward (bs 2), exact fp32 (the default: MIOpen route) %.1f us" % t_lib_mod)
        MaskHeadSmallConv.exact_fp32 = False
        t_mod = timeit(lambda: head(xs, None), args.reps)
        F = torch.nn.functional

        def ref():
            f = F.relu(head.lay3(xs[-1]))
            f = F.relu(head.lay4(xs[-2] + F.interpolate(f, size=xs[-2].shape[-2:], mode="nearest")))
            f = F.relu(head.jia_dcn(xs[-3] + F.interpolate(f, size=xs[-3].shape[-2:], mode="nearest")))
            return F.relu(head.lay2(F.relu(head.lay1(f))))
        t_ref = timeit(ref, args.reps)
        print("MaskHeadSmallConv.forward (bs 2): this repo %.1f us, PyTorch composition %.1f us" % (t_mod, t_ref))


if __name__ == "__main__":
    main()
