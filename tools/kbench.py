#!/usr/bin/env python
"""Kernel-level A/B timing of the MSDeformAttn variants at the R50 shapes (GPU box only).

    python tools/kbench.py [--reps 30] [--variants-fwd 1,2] [--variants-bwd 1,2] [--kinds encoder,decoder]

Prints one line per (direction, kind, flavour, variant): mean launch time over `reps` back-to-back launches
(HIP events on the launch stream), algorithmic GB/s and the fraction of the 8 TB/s HBM roofline.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uninext_amd import _lib, ext, workloads  # noqa: E402


def timeit(fn, reps, warm_ms=60.0):
    # warm up by TIME, not by count: the first launches of a process (and the first after an idle gap) run at a lower
    # clock -- with three warm-up launches the first variant of a list came out 8-12 % slower than the same kernel measured
    # after another one (round 3: an A/B "gain" of 10 % turned out to be the order of the runs)
    import time
    t0 = time.perf_counter()
    n = 0
    while n < 3 or (time.perf_counter() - t0) * 1e3 < warm_ms:
        fn()
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--variants-fwd", default="")
    ap.add_argument("--variants-bwd", default="")
    ap.add_argument("--kinds", default="encoder,decoder")
    ap.add_argument("--flavours", default="model,uniform,wide",
                    help="model (init-time offsets, sigma 1 px), uniform (ops/test.py:34), wide (model-like, sigma 6 px)")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--no-bwd", action="store_true")
    ap.add_argument("--sigma", type=float, default=1.0)
    ap.add_argument("--far", type=float, default=0.05)
    ap.add_argument("--rotate", type=int, default=1, help="cycle through this many distinct input sets (cold caches)")
    ap.add_argument("--workloads", default="", help="comma list of named workloads (uninext_amd.workloads.WORKLOADS, or 'all') "
                    "instead of --kinds at the R50 inference shapes")
    args = ap.parse_args()
    _lib.load()
    # default: every forward kernel THIS library carries (the default build names the experiments "exp:..." and refuses them)
    vf = [int(v) for v in args.variants_fwd.split(",") if v] or [
        k for k, n in enumerate(_lib.variants("forward")) if k >= 1 and not n.startswith("exp:")]
    vb = [int(v) for v in args.variants_bwd.split(",") if v] or [1, 2, 3]
    names = list(workloads.WORKLOADS) if args.workloads == "all" else [w for w in args.workloads.split(",") if w]
    for kind in (names or args.kinds.split(",")):
        for flavour in args.flavours.split(","):
            fl, sigma = ("model", 6.0) if flavour == "wide" else (flavour, args.sigma)
            if names:
                xs = [workloads.make_workload(kind, fl, seed=1 + r, offset_sigma=sigma, far_fraction=args.far)
                      for r in range(args.rotate)]
            else:
                xs = [workloads.make_inputs(kind, fl, batch=args.batch, seed=1 + r, offset_sigma=sigma,
                                            far_fraction=args.far) for r in range(args.rotate)]
            x = xs[0]
            N, S = x["value"].shape[:2]
            Lq = x["loc"].shape[1]
            a = (x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"])
            sets = [(y["value"], y["shapes"], y["lsi"], y["loc"], y["attn"]) for y in xs]
            fb = workloads.algorithmic_bytes_forward(N, S, Lq)
            counter = [0]

            def fwd_rot():
                counter[0] += 1
                return ext.ms_deform_attn_forward(*sets[counter[0] % len(sets)], 64)
            for v in vf:
                _lib.set_variant("forward", v)
                us = timeit(fwd_rot, args.reps)
                print("fwd %-22s %-8s %-22s %9.1f us  %8.1f GB/s  %5.1f%% of 8TB/s" % (
                    kind, flavour, _lib.last_kernel("forward") + "#%d" % v, us, fb / us / 1e3, fb / us / 1e3 / 80), flush=True)
            _lib.set_variant("forward", 0)
            if args.no_bwd:
                continue
            go = torch.randn(N, Lq, x["value"].shape[2] * x["value"].shape[3], device="cuda")
            bb = workloads.algorithmic_bytes_backward(N, S, Lq)
            for v in vb:
                _lib.set_variant("backward", v)
                us = timeit(lambda: ext.ms_deform_attn_backward(*a, go, 64), max(3, args.reps // 3))
                print("bwd %-22s %-8s %-22s %9.1f us  %8.1f GB/s  %5.1f%% of 8TB/s (incl. grad_value memset)" % (
                    kind, flavour, _lib.last_kernel("backward") + "#%d" % v, us, bb / us / 1e3, bb / us / 1e3 / 80), flush=True)
            _lib.set_variant("backward", 0)


if __name__ == "__main__":
    main()
