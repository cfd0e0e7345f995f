#!/usr/bin/env python
"""Patch-embedding convolutions at the backbone shapes of BASELINE.json configs 3-4 (GPU box only):
include/patch_embed_hip.h (fp32 MFMA implicit GEMM) vs the PyTorch-ROCm convolution the reference runs.

    python tools/patch_embed_bench.py [--reps 30]

Prints per shape: launch time (HIP events), TFLOP/s, fraction of the 157.3 TFLOP/s dense fp32 matrix peak
(MI355X_MICROARCH.md) = MFMA utilisation, and the PyTorch time for the same output.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uninext_amd import ext  # noqa: E402

PEAK_TF = 157.3

SHAPES = [
    # name, B, C, H, W, E, k, channels_last
    ("ViT-Huge patch_embed 16x16 (bs 2, 800x1333)", 2, 3, 800, 1333, 1280, 16, True),
    ("ViT-Huge patch_embed 16x16 (bs 2, 1024x1024)", 2, 3, 1024, 1024, 1280, 16, True),
    ("ConvNeXt-L stem 4x4 (bs 2, 800x1333)", 2, 3, 800, 1333, 192, 4, False),
    ("ConvNeXt-L downsample 1 2x2 192->384", 2, 192, 200, 333, 384, 2, False),
    ("ConvNeXt-L downsample 2 2x2 384->768", 2, 384, 100, 166, 768, 2, False),
    ("ConvNeXt-L downsample 3 2x2 768->1536", 2, 768, 50, 83, 1536, 2, False),
]


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
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.backends.cudnn.allow_tf32 = False
    for name, B, C, H, W, E, k, cl in SHAPES:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, C, H, W, generator=g).to(dev)
        w = (torch.randn(E, C, k, k, generator=g) / (C * k * k) ** 0.5).to(dev)
        b = torch.randn(E, generator=g).to(dev)
        flop = 2.0 * B * (H // k) * (W // k) * E * C * k * k
        with torch.no_grad():
            out = ext.patch_embed_forward(x, w, b, channels_last=cl)
            ref = torch.nn.functional.conv2d(x, w, b, stride=k)
            ref = ref.permute(0, 2, 3, 1) if cl else ref
            err = float((out - ref).abs().max()) / float(ref.abs().max())
            t_hip = timeit(lambda: ext.patch_embed_forward(x, w, b, channels_last=cl), args.reps)
            t_split, err_s = float("nan"), float("nan")
            if ext.patch_embed_packed_supported(w):
                packed = ext.patch_embed_pack_weight(w)
                o2 = ext.patch_embed_packed_forward(x, packed, E, k, b, cl)
                err_s = float((o2 - ref).abs().max()) / float(ref.abs().max())
                t_split = timeit(lambda: ext.patch_embed_packed_forward(x, packed, E, k, b, cl), args.reps)
            t_conv = timeit(lambda: torch.nn.functional.conv2d(x, w, b, stride=k), args.reps)
            t_conv_cl = timeit(lambda: torch.nn.functional.conv2d(x, w, b, stride=k).permute(0, 2, 3, 1).contiguous(),
                               args.reps) if cl else float("nan")
        tf = flop / t_hip * 1e-6
        print("%-48s M=%6d N=%4d K=%4d  exact %7.1f us %6.1f TFLOP/s = %4.1f %% of fp32 MFMA peak (err %.0e) | split-bf16 %7.1f us "
              "%6.1f TFLOP/s (err %.0e) | torch conv %7.1f us%s"
              % (name, B * (H // k) * (W // k), E, C * k * k, t_hip, tf, 100 * tf / PEAK_TF, err, t_split,
                 flop / t_split * 1e-6, err_s, t_conv, (" (+permute copy %7.1f us)" % t_conv_cl) if cl else ""))


if __name__ == "__main__":
    main()
