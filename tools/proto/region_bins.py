#!/usr/bin/env python
"""Sizing study for the planned owner-computes backward: how the samples of an encoder call fall into
destination regions.  CPU only (torch), R50 inference shapes, N = 2.

    python tools/proto/region_bins.py [h0xw0,h1xw1,h2xw2,h3xw3]      (region size per level; default 16x16 on every level)

Per location flavour: records per (image, head, level, region) when every sample is filed under each region that one of its
four corners touches -- total (the workspace), duplication over the sample count, and the spread over the regions (the
load balance of a region pass with one workgroup per region)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from uninext_amd import workloads  # noqa: E402


def study(flavour, sizes):
    x = workloads.make_workload("r50_infer_encoder", flavour=flavour, device="cpu")
    loc = x["loc"]                                    # [N, Lq, M, L, P, 2]
    N, Lq, M, L, P, _ = loc.shape
    levels = [tuple(int(v) for v in hw) for hw in x["shapes"].tolist()]
    tot_rec = 0
    tot_in = 0
    per_region = []
    rows = []
    for l, (H, W) in enumerate(levels):
        rh, rw = sizes[l]
        px = loc[:, :, :, l, :, 0] * W - 0.5
        py = loc[:, :, :, l, :, 1] * H - 0.5
        inr = (px > -1) & (py > -1) & (px < W) & (py < H)
        x0 = torch.floor(px).long()
        y0 = torch.floor(py).long()
        ry, rx = -(-H // rh), -(-W // rw)
        # regions touched by the valid corners of a sample: columns {x0, x0 + 1} and rows {y0, y0 + 1} clipped to the image
        cx0 = (x0.clamp(0, W - 1) // rw)
        cx1 = ((x0 + 1).clamp(0, W - 1) // rw)
        cy0 = (y0.clamp(0, H - 1) // rh)
        cy1 = ((y0 + 1).clamp(0, H - 1) // rh)
        nm = torch.arange(N).view(N, 1, 1, 1) * M + torch.arange(M).view(1, 1, M, 1)
        counts = torch.zeros(N * M * ry * rx, dtype=torch.long)
        seen = []
        for cy in (cy0, cy1):
            for cx in (cx0, cx1):
                key = (nm * ry + cy) * rx + cx
                dup = torch.zeros_like(inr)
                for k in seen:
                    dup |= (k == key)
                sel = inr & ~dup
                counts += torch.bincount(key[sel].reshape(-1), minlength=counts.numel())
                seen.append(key)
        n_in = int(inr.sum())
        n_rec = int(counts.sum())
        tot_rec += n_rec
        tot_in += n_in
        per_region.append(counts)
        c = counts.float()
        rows.append("    level %d (%3d x %3d, %3d regions per head): %8d in-range samples, %8d records (x %.3f); per region mean %7.0f  p99 %7.0f  max %7d"
                    % (l, H, W, ry * rx, n_in, n_rec, n_rec / max(n_in, 1), c.mean(), c.quantile(0.99), int(c.max())))
    allc = torch.cat(per_region).float()
    print("%-8s regions %s: %d records for %d in-range samples (x %.3f) = %.1f MB at 16 B; %d regions, mean %.0f, p99 %.0f, max %d records"
          % (flavour, ",".join("%dx%d" % s_ for s_ in sizes), tot_rec, tot_in, tot_rec / tot_in, tot_rec * 16 / 1e6, allc.numel(), allc.mean(), allc.quantile(0.99), int(allc.max())))
    for r in rows:
        print(r)
    # a region pass with one workgroup per region and 256 CUs: the busiest CU when regions are dealt largest first
    loads = torch.sort(allc, descending=True)[0]
    cu = torch.zeros(256)
    for v in loads.tolist():
        i = int(torch.argmin(cu))
        cu[i] += v
    print("    largest-first over 256 CUs: busiest CU %.0f records, mean %.0f (x %.3f)" % (float(cu.max()), float(cu.mean()), float(cu.max() / cu.mean())))


def main():
    spec = sys.argv[1] if len(sys.argv) > 1 else "16x16,16x16,16x16,16x16"
    sizes = [tuple(int(v) for v in t.split("x")) for t in spec.split(",")]
    for flavour in ("model", "wide", "uniform"):
        study(flavour, sizes)


if __name__ == "__main__":
    main()
