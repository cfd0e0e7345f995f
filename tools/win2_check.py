#!/usr/bin/env python
"""msda_fwd_win2 against msda_fwd_lg3 (whose parity with the oracle the test suite establishes) on the full-size flavours,
the odd pyramids and odd head counts; prints max |diff| and, on a mismatch, where the worst queries are.  GPU box only."""
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


def run(x, variant):
    _lib.set_variant("forward", variant)
    try:
        return ext.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
    finally:
        _lib.set_variant("forward", "auto")


def main():
    target = sys.argv[1] if len(sys.argv) > 1 else "msda_fwd_win2"
    _lib.load()
    bad = 0
    cases = []
    for fl in ("model", "uniform", "wide"):
        cases.append((fl, workloads.R50_LEVELS_INFER, 8, 2))
    for lv in ODD:
        for fl in ("model", "uniform", "wide"):
            cases.append((fl, lv, 8, 2))
    for heads, batch in ((8, 5), (3, 3), (16, 1), (1, 2), (5, 2), (7, 2)):
        cases.append(("model", ((25, 42), (13, 21), (7, 11), (4, 6)), heads, batch))
        cases.append(("wide", ODD[2], heads, batch))
    for fl, lv, heads, batch in cases:
        kw = dict(flavour="model", offset_sigma=6.0) if fl == "wide" else dict(flavour=fl)
        x = workloads.make_inputs("encoder", batch=batch, levels=lv, heads=heads, seed=17 + len(lv[0]) + heads, **kw)
        if x["loc"].shape[1] < 1024:
            continue
        x["loc"][0, 3, 0, 0, 0, 0] = float("nan")
        x["loc"][0, 5, heads - 1, 3, 3, 1] = float("inf")
        x["loc"][batch - 1, 17, min(2, heads - 1), 1, 2, 0] = -1e30
        ref = run(x, "msda_fwd_lg3")
        out = run(x, target)
        k = _lib.last_kernel("forward")
        again = run(x, target)
        d = (out - ref).abs()
        d = torch.where(torch.isfinite(d), d, torch.full_like(d, 1e9))
        err = float(d.max())
        ok = err < 2e-5 and bool(torch.equal(out, again)) and k == target
        print("%-8s M=%-2d N=%d %-48s %-14s max|diff| %.2e %s%s" % (fl, heads, batch, str(lv), k, err, "repeatable" if torch.equal(out, again) else "NOT-repeatable",
                                                              "" if ok else "   <-- MISMATCH"), flush=True)
        if not ok:
            bad += 1
            per_q = d.view(batch, -1, heads, 32).amax(-1)                 # [N, Lq, M]
            idx = torch.nonzero(per_q > 2e-5)
            print("   %d bad (image, query, head) pairs of %d; first: %s" % (len(idx), per_q.numel(), idx[:12].tolist()))
            starts = [0]
            for h, w in lv:
                starts.append(starts[-1] + h * w)
            lvl_of = np.searchsorted(np.array(starts[1:]), idx[:, 1].cpu().numpy(), side="right")
            print("   bad pairs per level:", np.bincount(lvl_of, minlength=4).tolist(), " per head:", np.bincount(idx[:, 2].cpu().numpy(), minlength=heads).tolist())
            for (bb, q, mm) in idx[:4].tolist():
                l = int(np.searchsorted(np.array(starts[1:]), q, side="right"))
                qq = q - starts[l]
                print("      image %d level %d (y %d, x %d) head %d: got %s want %s" % (bb, l, qq // lv[l][1], qq % lv[l][1], mm, out[bb, q, mm * 32:mm * 32 + 4].tolist(), ref[bb, q, mm * 32:mm * 32 + 4].tolist()))
    print("MISMATCHES: %d of %d cases" % (bad, len(cases)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
