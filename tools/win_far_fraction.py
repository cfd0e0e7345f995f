#!/usr/bin/env python
"""CPU model of msda_fwd_win's window placement: fraction of in-range samples that are 'far' (a corner outside the
tile's window) per level, for the bench inputs.  Mirrors the kernel: tile = 8 x 16 level-0 pixels, queries of all
levels by centre, window origin = round(mean top-left corner of the in-range samples of the first 128 queries) -
(W - 2) / 2, clamped to [-1, size + 1 - W]."""
import argparse
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uninext_amd import workloads

ap = argparse.ArgumentParser()
ap.add_argument("--flavour", default="model")
ap.add_argument("--wh", default="12,10,10,10")   # rounds 2-4: 14,10,8,7
ap.add_argument("--ww", default="20,14,12,10")   # rounds 2-4: 22,14,10,8
ap.add_argument("--th", type=int, default=8)
ap.add_argument("--tw", type=int, default=16)
args = ap.parse_args()
WH = [int(v) for v in args.wh.split(",")]; WW = [int(v) for v in args.ww.split(",")]
TH, TW = args.th, args.tw
kw = dict(flavour="model", offset_sigma=6.0) if args.flavour == "wide" else dict(flavour=args.flavour)
x = workloads.make_inputs("encoder", batch=1, seed=100, device="cpu", **kw)
levels = [tuple(r) for r in x["shapes"].tolist()]
loc = x["loc"][0].numpy()    # [Lq, M, L, P, 2]
H0, W0 = levels[0]
starts = np.cumsum([0] + [h * w for h, w in levels])
# tile id of every query
tile = np.zeros(loc.shape[0], dtype=np.int64)
order = np.zeros(loc.shape[0], dtype=np.int64)   # rank inside the tile (level-major, raster)
TX = (W0 + TW - 1) // TW
for l, (h, w) in enumerate(levels):
    q = np.arange(h * w); y, xx = q // w, q % w
    tx = np.minimum(((2 * xx + 1) * W0) // (2 * TW * w), TX - 1)
    ty = np.minimum(((2 * y + 1) * H0) // (2 * TH * h), (H0 + TH - 1) // TH - 1)
    tile[starts[l]:starts[l + 1]] = ty * TX + tx
ntiles = tile.max() + 1
far_tot = np.zeros(4); inr_tot = np.zeros(4)
first128 = np.zeros(loc.shape[0], dtype=bool)
for t in range(ntiles):
    idx = np.nonzero(tile == t)[0]          # level-major raster order == kernel order
    first128[idx[:128]] = True
for l, (h, w) in enumerate(levels):
    px = loc[:, :, l, :, 0] * w - 0.5; py = loc[:, :, l, :, 1] * h - 0.5     # [Lq, M, P]
    inr = (py > -1) & (px > -1) & (py < h) & (px < w)
    x0 = np.floor(px); y0 = np.floor(py)
    for m in range(loc.shape[1]):
        sx = np.bincount(tile, weights=(x0[:, m] * inr[:, m] * first128[:, None]).sum(1), minlength=ntiles)
        sy = np.bincount(tile, weights=(y0[:, m] * inr[:, m] * first128[:, None]).sum(1), minlength=ntiles)
        sn = np.bincount(tile, weights=(inr[:, m] * first128[:, None]).sum(1), minlength=ntiles)
        sn1 = np.maximum(sn, 1)
        ox = np.floor(sx / sn1 + 0.5) - (WW[l] - 2) // 2; oy = np.floor(sy / sn1 + 0.5) - (WH[l] - 2) // 2
        ox = np.maximum(-1, np.minimum(ox, w + 1 - WW[l])); oy = np.maximum(-1, np.minimum(oy, h + 1 - WH[l]))
        cx = x0[:, m] - ox[tile][:, None]; ry = y0[:, m] - oy[tile][:, None]
        near = inr[:, m] & (cx >= 0) & (cx <= WW[l] - 2) & (ry >= 0) & (ry <= WH[l] - 2)
        far_tot[l] += (inr[:, m] & ~near).sum(); inr_tot[l] += inr[:, m].sum()
print("flavour %s  windows %s x %s  tile %dx%d (%d tiles)" % (args.flavour, WH, WW, TH, TW, ntiles))
for l in range(4):
    print("  level %d: far %.2f %% of in-range samples" % (l, 100 * far_tot[l] / max(inr_tot[l], 1)))
print("  all: far %.2f %%  (in range %.1f %% of all samples)" % (100 * far_tot.sum() / inr_tot.sum(), 100 * inr_tot.sum() / (loc.size / 2)))
