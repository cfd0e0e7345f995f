#!/usr/bin/env python
"""Phase timeline of msda_bwd_win2 from in-kernel timestamps (second item of every workgroup).  GPU box only; library built
with -DMSDA_BWIN2_PROF from the experiments build (uninext_amd/csrc/experiments/msda_bwd_win2.hip; profiles/r04_backward_two_phase.txt)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uninext_amd import _lib, ext, workloads  # noqa: E402
NAMES = {(0, 1): "prefetched loads arrive, maxima, placement sums", (1, 2): "barrier A", (2, 3): "origins, DMA issue, flush table",
         (3, 4): "own DMA landed", (4, 5): "barrier B", (5, 6): "gather: near samples of round 0", (6, 7): "far samples, stores, later rounds",
         (7, 8): "scatter-side fetch issue + barrier C", (8, 9): "zero the windows, scale", (9, 10): "barrier D", (10, 11): "scatter phase",
         (11, 12): "next item: decode + issue loads", (12, 13): "barrier E", (13, 14): "flush", (0, 14): "ITEM"}
fl = sys.argv[1] if len(sys.argv) > 1 else "model"
lib = _lib.load()
kw = dict(flavour="model", offset_sigma=6.0) if fl == "wide" else dict(flavour=fl)
x = workloads.make_inputs("encoder", batch=2, seed=3, **kw)
S = x["value"].shape[1]
go = torch.randn(2, S, 256, device="cuda")
_lib.set_variant("backward", "msda_bwd_win2")
for _ in range(3):
    ext.ms_deform_attn_backward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], go, 64)
torch.cuda.synchronize()
nb, W = 512, 8
buf = np.zeros((nb, W, 16), dtype=np.uint64)
assert lib.msda_debug_read_prof_bwin2(buf.ctypes.data_as(ctypes.c_void_p), nb) == 0
t = buf.astype(np.int64)
t = t[t[:, 0, 14] > 0]
us = (t - t[:, :, 0].min()) * 1e-2
print("flavour %s: %d workgroups, second item of each" % (fl, len(t)))
for grp, sel in (("waves 0..2 (they also take the queries of levels 1..3)", slice(0, 3)), ("waves 3..7", slice(3, 8))):
    print(" " + grp)
    for (a, b), n in NAMES.items():
        dd = (us[:, sel, b] - us[:, sel, a]).reshape(-1)
        print("   %-50s median %6.2f  mean %6.2f  p90 %6.2f us" % (n, np.median(dd), dd.mean(), np.percentile(dd, 90)))
