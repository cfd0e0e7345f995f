#!/usr/bin/env python
"""msda_bwd_win against msda_bwd_tiled / msda_bwd_generic (whose parity with the oracle the test suite establishes): full-size
flavours, odd pyramids, odd head counts; prints the max differences per output.  GPU box only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uninext_amd import _lib, ext, workloads  # noqa: E402

ODD = [((100, 168), (50, 84), (25, 42), (13, 21)), ((50, 84), (25, 42), (13, 21), (7, 11)), ((33, 47), (17, 24), (9, 12), (5, 6)),
       ((40, 40), (80, 80), (3, 3), (1, 1)), ((3, 400), (2, 200), (1, 100), (1, 50)), ((64, 80), (32, 40), (16, 20), (17, 17)),
       ((31, 37), (31, 37), (31, 37), (31, 37))]


def run(x, go, variant):
    _lib.set_variant("backward", variant)
    try:
        return ext.ms_deform_attn_backward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], go, 64)
    finally:
        _lib.set_variant("backward", "auto")


def main():
    kernel = sys.argv[1] if len(sys.argv) > 1 else "msda_bwd_win"
    _lib.load()
    cases = [(fl, workloads.R50_LEVELS_INFER, 8, 2) for fl in ("model", "uniform", "wide")]
    cases += [(fl, lv, 8, 2) for lv in ODD for fl in ("model", "wide")]
    cases += [("model", ((25, 42), (13, 21), (7, 11), (4, 6)), h, b) for h, b in ((8, 5), (3, 3), (16, 1), (1, 2), (5, 2))]
    bad = 0
    for fl, lv, heads, batch in cases:
        kw = dict(flavour="model", offset_sigma=6.0) if fl == "wide" else dict(flavour=fl)
        x = workloads.make_inputs("encoder", batch=batch, levels=lv, heads=heads, seed=21 + len(lv[0]) + heads, **kw)
        S = x["value"].shape[1]
        if S < 1024:
            continue
        x["loc"][0, 3, 0, 0, 0, 0] = float("nan")
        x["loc"][0, 5, heads - 1, 3, 3, 1] = float("inf")
        go = torch.randn(batch, S, heads * 32, generator=torch.Generator().manual_seed(9)).cuda()
        rv, rl, ra = run(x, go, "msda_bwd_generic")
        gv, gl, ga = run(x, go, kernel)
        kern = _lib.last_kernel("backward")
        e_v = float((gv - rv).abs().max()); e_a = float((ga - ra).abs().max())
        dl = (gl - rl).abs()
        e_l = [float(dl[:, :, :, l].max()) / (1e-4 * max(lv[l])) for l in range(4)]
        fin = bool(torch.isfinite(gv).all() and torch.isfinite(gl).all() and torch.isfinite(ga).all())
        ok = kern == kernel and e_v < 1e-4 and e_a < 5e-4 and max(e_l) < 1.0 and fin
        print("%-8s M=%-2d N=%d %-48s %-13s grad_value %.2e (max %.1f) grad_attn %.2e grad_loc/bound %s %s%s" % (
            fl, heads, batch, str(lv), kern, e_v, float(rv.abs().max()), e_a, ["%.2f" % e for e in e_l], "" if fin else "NON-FINITE ",
            "" if ok else "  <-- MISMATCH"), flush=True)
        if not ok:
            bad += 1
            bv = (gv - rv).abs().view(batch, S, heads, 32).amax(-1)
            idx = torch.nonzero(bv > 1e-4)
            print("   grad_value: %d bad (image, pixel, head) of %d; first %s" % (len(idx), bv.numel(), idx[:8].tolist()))
            ba = (ga - ra).abs()
            idx = torch.nonzero(ba > 5e-4)
            print("   grad_attn: %d bad of %d; first %s" % (len(idx), ba.numel(), idx[:6].tolist()))
            for l in range(4):
                idx = torch.nonzero(dl[:, :, :, l] > 1e-4 * max(lv[l]))
                if len(idx):
                    print("   grad_loc level %d: %d bad; first %s" % (l, len(idx), idx[:6].tolist()))
    print("MISMATCHES: %d of %d" % (bad, len(cases)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
