#!/usr/bin/env python
"""numpy restatement of the destination-side ("owner computes") grad_value pass planned in DESIGN.md section 7 -- a
development aid for the next round, NOT product code and not on any product path.

    python tools/proto/owner_computes_ref.py            # self-check against the C oracle on small seeded cases

Two steps, as the kernels would do them:
  file_samples()  every in-range sample is filed once under each destination region that one of its in-image bilinear corners
                  falls into (1, 2 or 4 regions): records (image, query, head, level, point, region)
  region_pass()   a region adds, for each of its records, ONLY the corners that lie inside the region -- every corner of
                  every sample is applied exactly once, by the one region that owns its pixel, so regions can be STORED
                  without atomics.
The arithmetic per corner is the reference's (ops/src/cuda/ms_deform_im2col_cuda.cuh:87-159: w1..w4 from (lh, lw),
grad_value += w * attn * grad_out).  Device-side bins and records of a future kernel can be compared with file_samples()."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import msda_oracle  # noqa: E402  (development check only)
from uninext_amd import workloads  # noqa: E402


def _corners(loc, H, W):
    """Per sample of one level: in-range flag, top-left corner, the four (dy, dx, weight-without-attn, valid) corners."""
    px = loc[..., 0] * W - 0.5
    py = loc[..., 1] * H - 0.5
    inr = (py > -1) & (px > -1) & (py < H) & (px < W)            # cuh:285 (and :38-46 for the per-corner validity)
    y0 = np.floor(py).astype(np.int64)
    x0 = np.floor(px).astype(np.int64)
    lh, lw = py - y0, px - x0
    hh, hw = 1.0 - lh, 1.0 - lw
    out = []
    for dy, dx, w in ((0, 0, hh * hw), (0, 1, hh * lw), (1, 0, lh * hw), (1, 1, lh * lw)):
        cy, cx = y0 + dy, x0 + dx
        ok = inr & (cy >= 0) & (cy <= H - 1) & (cx >= 0) & (cx <= W - 1)
        out.append((cy, cx, w, ok))
    return inr, out


def file_samples(loc, levels, sizes):
    """-> list per level of dict(b, q, m, p, ry, rx): one row per (sample, touched region)."""
    N, Lq, M, L, P, _ = loc.shape
    idx = np.indices((N, Lq, M, P))
    filed = []
    for l, (H, W) in enumerate(levels):
        rh, rw = sizes[l]
        inr, corners = _corners(loc[:, :, :, l], H, W)
        keys = []
        rows = {k: [] for k in ("b", "q", "m", "p", "ry", "rx")}
        for cy, cx, _, ok in corners:
            ry, rx = np.where(ok, cy // rh, -1), np.where(ok, cx // rw, -1)
            fresh = ok.copy()
            for (py_, px_, pok) in keys:                            # a region already filed by an earlier corner of the sample
                fresh &= ~(pok & (py_ == ry) & (px_ == rx))
            keys.append((ry, rx, ok))
            sel = np.nonzero(fresh)
            rows["b"].append(idx[0][sel]); rows["q"].append(idx[1][sel]); rows["m"].append(idx[2][sel]); rows["p"].append(idx[3][sel])
            rows["ry"].append(ry[sel]); rows["rx"].append(rx[sel])
        filed.append({k: np.concatenate(v) for k, v in rows.items()})
    return filed


def region_pass(filed, grad_out, loc, attn, levels, sizes, S, lsi):
    """grad_value [N, S, M, D] in float64 from the records: each record applies the corners inside ITS region only."""
    N, Lq, M, L, P, _ = loc.shape
    D = grad_out.shape[-1] // M
    go = grad_out.reshape(N, Lq, M, D).astype(np.float64)
    gv = np.zeros((N, S, M, D))
    applied = 0
    for l, (H, W) in enumerate(levels):
        rh, rw = sizes[l]
        r = filed[l]
        b, q, m, p = r["b"], r["q"], r["m"], r["p"]
        _, corners = _corners(loc[b, q, m, l, p].astype(np.float64), H, W)
        a = attn[b, q, m, l, p].astype(np.float64)
        for cy, cx, w, ok in corners:
            mine = ok & (cy // rh == r["ry"]) & (cx // rw == r["rx"])
            sel = np.nonzero(mine)[0]
            pix = lsi[l] + cy[sel] * W + cx[sel]
            np.add.at(gv, (b[sel], pix, m[sel]), (w[sel] * a[sel])[:, None] * go[b[sel], q[sel], m[sel]])
            applied += sel.size
    return gv, applied


def check(levels, sizes, flavour, seed):
    x = workloads.make_inputs("encoder", flavour, batch=2, levels=levels, heads=3, seed=seed, device="cpu")
    loc, attn = x["loc"].numpy(), x["attn"].numpy()
    S = sum(h * w for h, w in levels)
    lsi = [int(v) for v in x["lsi"].tolist()]
    go = torch.randn(2, S, 3 * 32, generator=torch.Generator().manual_seed(seed + 1))
    filed = file_samples(loc.astype(np.float64), levels, sizes)
    gv, applied = region_pass(filed, go.numpy(), loc, attn, levels, sizes, S, lsi)
    ogv, _, _ = msda_oracle.backward(go.double(), x["value"].double(), x["shapes"], x["lsi"], x["loc"].double(), x["attn"].double())
    nrec = sum(f["b"].size for f in filed)
    inr = sum(int(_corners(loc[:, :, :, l].astype(np.float64), H, W)[0].sum()) for l, (H, W) in enumerate(levels))
    err = float(np.abs(gv - ogv).max())
    print("%-8s %-44s regions %-22s %7d in-range samples -> %7d records (x %.3f), %8d corner adds, max |grad_value - oracle| %.2e"
          % (flavour, str(levels), ",".join("%dx%d" % s for s in sizes), inr, nrec, nrec / max(inr, 1), applied, err))
    return err < 1e-12


def main():
    ok = True
    cases = [(((25, 42), (13, 21), (7, 11), (4, 6)), ((16, 16), (8, 16), (4, 8), (2, 4))),
             (((25, 42), (13, 21), (7, 11), (4, 6)), ((16, 16),) * 4),
             (((33, 47), (17, 24), (9, 12), (5, 6)), ((8, 8), (8, 8), (3, 5), (5, 6))),
             (((3, 40), (2, 20), (1, 10), (1, 5)), ((2, 16), (1, 16), (1, 4), (1, 2)))]
    for levels, sizes in cases:
        for flavour in ("model", "uniform"):
            ok &= check(levels, sizes, flavour, seed=11)
    print("OK" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
