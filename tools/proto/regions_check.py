#!/usr/bin/env python
"""Development check of msda_bwd_regions against the C oracle (GPU box):
    python tools/proto/regions_check.py [--quick] [--time]
Every element of grad_value / grad_sampling_loc / grad_attn_weight on small and odd pyramids and at the full R50 size in the
three location flavours, two calls in a row (the workspace must come back clean), grad_loc / grad_attn bitwise against
msda_bwd_tiled (the same query-side pass); on a mismatch, where the wrong pixels are.  MSDA_BWD_REGIONS_HIST=8 runs the file
kernels without their LDS table."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import msda_oracle  # noqa: E402
from uninext_amd import _lib, ext, workloads  # noqa: E402

PYRAMIDS = [
    ((25, 42), (13, 21), (7, 11), (4, 6)),
    ((33, 47), (17, 24), (9, 12), (5, 6)),
    ((40, 40), (80, 80), (3, 3), (1, 1)),            # a finer level after the first one
    ((3, 400), (2, 200), (1, 100), (1, 50)),         # thin image
    ((64, 80), (32, 40), (16, 20), (17, 17)),
    ((31, 37), (31, 37), (31, 37), (31, 37)),        # four levels of equal resolution
    workloads.R50_LEVELS_INFER,
]


def bwd(x, go, variant):
    _lib.set_variant("backward", variant)
    try:
        out = ext.ms_deform_attn_backward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], go, 64)
    finally:
        _lib.set_variant("backward", 0)
    torch.cuda.synchronize()
    return out, _lib.last_kernel("backward")


def check(levels, flavour, batch, heads, seed):
    kw = dict(flavour="model", offset_sigma=6.0) if flavour == "wide" else dict(flavour=flavour)
    x = workloads.make_inputs("encoder", batch=batch, levels=levels, heads=heads, seed=seed, device="cuda", **kw)
    if seed % 2:   # a few poisoned locations
        x["loc"][0, 3, 0, 0, 0, 0] = float("nan")
        x["loc"][0, 5, heads - 1, 3, 3, 1] = float("inf")
        x["loc"][batch - 1, 17, 1, 1, 2, 0] = -1e30
    S = sum(h * w for h, w in levels)
    go = torch.randn(batch, S, heads * 32, generator=torch.Generator().manual_seed(seed + 100)).cuda()
    (gv, gl, ga), kern = bwd(x, go, "msda_bwd_regions")
    (gv2, gl2, ga2), _ = bwd(x, go, "msda_bwd_regions")
    (tv, tl, ta), tk = bwd(x, go, "msda_bwd_tiled")
    ogv, ogl, oga = msda_oracle.backward(go.double(), x["value"].double(), x["shapes"], x["lsi"], x["loc"].double(), x["attn"].double())
    e_gv = np.abs(gv.cpu().numpy().astype(np.float64) - ogv)
    e_t = float(np.abs(tv.cpu().numpy().astype(np.float64) - ogv).max())
    again = torch.equal(gv, gv2) and torch.equal(gl, gl2) and torch.equal(ga, ga2)
    same_q = torch.equal(gl, tl) and torch.equal(ga, ta)
    mx = float(np.nanmax(e_gv))
    ok = kern == "msda_bwd_regions" and mx < 1e-4 and again and same_q and bool(torch.isfinite(gv).all())
    print("%-5s %-8s %-44s N=%d M=%d kernel %-17s grad_value max err %.2e (tiled %.2e)  repeatable %s  grad_loc/attn == tiled's %s" % (
        "ok" if ok else "FAIL", flavour, str(levels), batch, heads, kern, mx, e_t, again, same_q), flush=True)
    if not (mx < 1e-4):
        bad = np.argwhere(~(e_gv.max(-1) < 1e-4))     # (b, pixel, head)
        print("   wrong (pixel, head) rows: %d of %d" % (len(bad), e_gv.shape[0] * e_gv.shape[1] * e_gv.shape[2]))
        starts = np.cumsum([0] + [h * w for h, w in levels])
        for l, (h, w) in enumerate(levels):
            sel = bad[(bad[:, 1] >= starts[l]) & (bad[:, 1] < starts[l + 1])]
            if len(sel):
                p = sel[:, 1] - starts[l]
                print("   level %d: %d wrong; rows %d..%d cols %d..%d heads %s images %s; first %s" % (
                    l, len(sel), (p // w).min(), (p // w).max(), (p % w).min(), (p % w).max(), sorted(set(sel[:, 2].tolist())),
                    sorted(set(sel[:, 0].tolist())), [(int(a), int(q // w), int(q % w), int(c)) for a, q, c in zip(sel[:6, 0], p[:6], sel[:6, 2])]))
        b0, p0, m0 = bad[0]
        print("   first wrong row: got %s\n                    ref %s" % (np.array2string(gv[b0, p0, m0 * 32:m0 * 32 + 6].cpu().numpy(), precision=5),
                                                                       np.array2string(ogv[b0, p0, m0, :6], precision=5)))
    return ok


def timing():
    for flavour in ("model", "wide", "uniform"):
        x = workloads.make_workload("r50_infer_encoder", flavour=flavour, device="cuda")
        go = torch.randn(2, 22223, 256, generator=torch.Generator().manual_seed(5)).cuda()
        for variant in ("msda_bwd_tiled", "msda_bwd_regions"):
            _lib.set_variant("backward", variant)
            try:
                for _ in range(6):
                    ext.ms_deform_attn_backward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], go, 64)
                torch.cuda.synchronize()
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                ev[0].record()
                for _ in range(12):
                    ext.ms_deform_attn_backward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], go, 64)
                ev[1].record()
                torch.cuda.synchronize()
            finally:
                _lib.set_variant("backward", 0)
            print("time  %-8s %-18s %8.1f us per call (incl. output allocation + memset)" % (flavour, _lib.last_kernel("backward"), ev[0].elapsed_time(ev[1]) / 12 * 1e3), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--time", action="store_true")
    args = ap.parse_args()
    _lib.load()
    ok = True
    seed = 30
    for levels in (PYRAMIDS[:2] if args.quick else PYRAMIDS):
        full = levels == workloads.R50_LEVELS_INFER
        for flavour in ("model", "uniform", "wide"):
            seed += 1
            ok &= check(levels, flavour, 2, 8 if full else 3, seed)
    print("ALL OK" if ok else "SOME FAILED")
    if args.time:
        timing()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
