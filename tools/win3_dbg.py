#!/usr/bin/env python
"""Where msda_fwd_win3 differs from msda_fwd_lg3 on the R50 model flavour: histograms by tile row / column / pipeline iteration."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uninext_amd import _lib, ext, workloads  # noqa: E402
_lib.load()
target = sys.argv[1] if len(sys.argv) > 1 else "msda_fwd_win3"
x = workloads.make_inputs("encoder", "model", batch=2, seed=3)
def run(v):
    _lib.set_variant("forward", v)
    try:
        return ext.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
    finally:
        _lib.set_variant("forward", "auto")
ref = run("msda_fwd_lg3")
for rep in range(3):
    out = run(target)
    d = (out - ref).abs().view(2, -1, 8, 32)
    per = d.amax(-1)
    idx = torch.nonzero(per > 2e-5).cpu().numpy()
    print("run %d: %d bad pairs, max %.3g" % (rep, len(idx), float(d.max())))
    if not len(idx):
        continue
    q = idx[:, 1]
    l0 = q < 16700
    y, xx = q[l0] // 167, q[l0] % 167
    print("  level-0 bad: %d; by y%%8 %s; by x%%16 %s" % (l0.sum(), np.bincount(y % 8, minlength=8).tolist(), np.bincount(xx % 16, minlength=16).tolist()))
    tile = (y // 8) * 11 + xx // 16 + idx[l0, 0] * 143
    it = tile // 32
    print("  by pipeline iteration (item // K): %s" % np.bincount(it, minlength=9).tolist())
    print("  by workgroup (item %% K): %s" % np.bincount(tile % 32, minlength=32).tolist())
    ch = torch.nonzero(d[idx[0, 0], idx[0, 1], idx[0, 2]] > 2e-5).flatten().tolist()
    print("  first bad pair %s: bad channels %s" % (idx[0].tolist(), ch))
    nbad_ch = (d > 2e-5).sum(-1)[per > 2e-5]
    print("  bad channels per bad pair: %s" % np.bincount(nbad_ch.cpu().numpy(), minlength=33).tolist())
    for (bb, q, mm) in idx[:6].tolist():
        g = out[bb, q, mm * 32:mm * 32 + 32].cpu().numpy(); w = ref[bb, q, mm * 32:mm * 32 + 32].cpu().numpy()
        print("  pair", (bb, q, mm), "got-want on channels 16..31:", np.round(g[16:] - w[16:], 4).tolist())
    rest = idx[~l0]
    if len(rest):
        qq = rest[:, 1] - 16700
        l1 = qq < 4200
        print("  level-1 bad: x%%8 %s y%%4 %s" % (np.bincount((qq[l1] % 84) % 8, minlength=8).tolist(), np.bincount((qq[l1] // 84) % 4, minlength=4).tolist()))
