#!/usr/bin/env python
"""Where the per-site kernel choices cross (GPU box): for model-like sampling patterns of growing spread (offset sigma in pixels of the
level) the far fraction the window forward kernel reports, and the launch time of every candidate kernel, forward and backward.

    python tools/crossover_sweep.py [--sigmas 1,2,3,4,5,6,8] [--reps 12]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from uninext_amd import _lib, ext, workloads  # noqa: E402
from kbench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sigmas", default="1,2,3,4,5,6,8")
    ap.add_argument("--reps", type=int, default=12)
    a = ap.parse_args()
    _lib.load()
    print("sigma   far     fwd: win     lg3   |  bwd: win    tiled   regions   (us per launch, bwd incl. the grad_value memset)")
    for k, sg in enumerate(float(s) for s in a.sigmas.split(",")):
        xs = [workloads.make_inputs("encoder", batch=2, seed=1 + r, flavour="model", offset_sigma=sg, device="cuda") for r in range(3)]
        x = xs[0]
        with ext.call_site(20 + k):
            for _ in range(3):
                ext.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
        torch.cuda.synchronize()
        _, far = _lib.forward_locality()
        cnt = [0]

        def fwd():
            cnt[0] += 1
            y = xs[cnt[0] % 3]
            return ext.ms_deform_attn_forward(y["value"], y["shapes"], y["lsi"], y["loc"], y["attn"], 64)
        t = {}
        for name in ("msda_fwd_win", "msda_fwd_lg3"):
            _lib.set_variant("forward", name)
            t[name] = timeit(fwd, a.reps * 2)
        _lib.set_variant("forward", "auto")
        go = torch.randn(2, x["loc"].shape[1], 256, device="cuda")
        for name in ("msda_bwd_win", "msda_bwd_tiled", "msda_bwd_regions"):
            _lib.set_variant("backward", name)
            t[name] = timeit(lambda: ext.ms_deform_attn_backward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], go, 64), a.reps)
        _lib.set_variant("backward", "auto")
        print("%5.1f  %.3f   %8.1f %8.1f  |  %8.1f %8.1f %8.1f" % (sg, far, t["msda_fwd_win"], t["msda_fwd_lg3"], t["msda_bwd_win"],
                                                                  t["msda_bwd_tiled"], t["msda_bwd_regions"]), flush=True)


if __name__ == "__main__":
    main()
