#!/usr/bin/env python
"""Development check of one forward variant against the C oracle on EVERY query (GPU box):
    python tools/win_check.py [--variant msda_fwd_win] [--quick]
Full R50 size in the three location flavours + odd pyramids; prints the max error and, on failure, where
the wrong (query, head) pairs are (level, position, count) to localise kernel bugs."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import msda_oracle  # noqa: E402
from uninext_amd import _lib, ext, workloads  # noqa: E402

PYRAMIDS = [
    workloads.R50_LEVELS_INFER,
    workloads.R50_LEVELS_TRAIN,
    ((50, 84), (25, 42), (13, 21), (7, 11)),
    ((33, 47), (17, 24), (9, 12), (5, 6)),
    ((40, 40), (80, 80), (3, 3), (1, 1)),            # a finer level after the first one
    ((3, 400), (2, 200), (1, 100), (1, 50)),         # thin image
    ((64, 80), (32, 40), (16, 20), (17, 17)),
    ((31, 37), (31, 37), (31, 37), (31, 37)),        # four levels of equal resolution
]


def check(levels, flavour, variant, batch, seed):
    kw = dict(flavour="model", offset_sigma=6.0) if flavour == "wide" else dict(flavour=flavour)
    x = workloads.make_inputs("encoder", batch=batch, levels=levels, seed=seed, device="cuda", **kw)
    if seed % 2:   # a few poisoned locations
        x["loc"][0, 3, 0, 0, 0, 0] = float("nan")
        x["loc"][0, 5, 7, 3, 3, 1] = float("inf")
        x["loc"][batch - 1, 17, 2, 1, 2, 0] = -1e30
    _lib.set_variant("forward", variant)
    try:
        out = ext.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
        again = ext.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
    finally:
        _lib.set_variant("forward", 0)
    torch.cuda.synchronize()
    kern = _lib.last_kernel("forward")
    ref = msda_oracle.forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"])
    o = out.cpu().numpy().astype(np.float64)
    err = np.abs(o - ref).reshape(batch, -1, 8, 32)
    finite = bool(np.isfinite(o).all())
    mx = float(np.nanmax(err)) if err.size else 0.0
    ok = finite and mx < 1e-4 and torch.equal(out, again)
    print("%-5s %-8s %-44s kernel %-18s max err %.2e  finite %s  deterministic %s" % (
        "ok" if ok else "FAIL", flavour, str(levels), kern, mx, finite, torch.equal(out, again)), flush=True)
    if not ok:
        bad = np.argwhere(~(err.max(-1) < 1e-4))
        print("   wrong pairs: %d of %d" % (len(bad), err.shape[0] * err.shape[1] * 8))
        starts = np.cumsum([0] + [h * w for h, w in levels])
        for l, (h, w) in enumerate(levels):
            sel = bad[(bad[:, 1] >= starts[l]) & (bad[:, 1] < starts[l + 1])]
            if len(sel):
                q = sel[:, 1] - starts[l]
                print("   level %d: %d wrong; rows %d..%d cols %d..%d heads %s; first %s" % (
                    l, len(sel), (q // w).min(), (q // w).max(), (q % w).min(), (q % w).max(),
                    sorted(set(sel[:, 2].tolist())), [(int(a), int(b // w), int(b % w), int(c)) for a, b, c in
                                                       zip(sel[:6, 0], q[:6], sel[:6, 2])]))
        ch = err.max(axis=(0, 1, 2))
        print("   max err per channel:", np.array2string(ch, precision=1, max_line_width=200))
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="msda_fwd_win")
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    _lib.load()
    ok = True
    seed = 10
    for levels in (PYRAMIDS[:1] if args.quick else PYRAMIDS):
        for flavour in ("model", "uniform", "wide"):
            seed += 1
            ok &= check(levels, flavour, args.variant, 2, seed)
    print("ALL OK" if ok else "FAILURES")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
