#!/usr/bin/env python
"""Timeline of msda_fwd_win4's second item per workgroup from in-kernel s_memrealtime stamps (library built with -DMSDA_WIN4_PROF)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uninext_amd import _lib, ext, workloads  # noqa: E402
NAMES = {(0, 1): "scalar loads, geometry", (1, 2): "barrier #1", (2, 4): "query, loads, coordinates, placement sums", (4, 5): "barrier #2",
         (5, 6): "origins", (6, 8): "classify + window DMA issue", (8, 9): "far steps", (9, 10): "own DMA landed", (10, 11): "barrier #3",
         (11, 13): "LDS pass", (13, 15): "stores acknowledged", (0, 15): "ITEM"}
fl = sys.argv[1] if len(sys.argv) > 1 else "model"
lib = _lib.load()
kw = dict(flavour="model", offset_sigma=6.0) if fl == "wide" else dict(flavour=fl)
x = workloads.make_inputs("encoder", batch=2, seed=3, **kw)
_lib.set_variant("forward", "msda_fwd_win4")
for _ in range(3):
    ext.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
torch.cuda.synchronize()
nb, W = 512, 6
buf = np.zeros((nb, W, 16), dtype=np.uint64)
assert lib.msda_debug_read_prof4(buf.ctypes.data_as(ctypes.c_void_p), nb) == 0
t = buf.astype(np.int64)
t = t[t[:, 0, 15] > 0]
us = (t - t[:, :, 0].min()) * 1e-2
print("flavour %s: %d workgroups" % (fl, len(t)))
for grp, sel in (("level-0 waves", slice(0, 4)), ("waves of levels 1..3", slice(4, 6))):
    print(" " + grp)
    for (a, b), n in NAMES.items():
        dd = (us[:, sel, b] - us[:, sel, a]).reshape(-1)
        print("   %-44s median %6.2f  mean %6.2f  p90 %6.2f us" % (n, np.median(dd), dd.mean(), np.percentile(dd, 90)))
st = (t[:, 0, 0] - t[:, :, 0].min()) * 1e-2
print(" start of the second item after the earliest one: percentiles 10/50/90/100: %s us" % np.round(np.percentile(st, [10, 50, 90, 100]), 1).tolist())
