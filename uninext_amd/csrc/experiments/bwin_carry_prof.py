#!/usr/bin/env python
"""Phase timeline of msda_bwd_win (round 5: deferred, carried flush) from in-kernel timestamps (second item of every workgroup).  GPU box only; library built
with -DMSDA_BWIN_PROF (tools/abl_build.sh bwprof msda_bwd_win -DMSDA_BWIN_PROF)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uninext_amd import _lib, ext, workloads  # noqa: E402
NAMES = {(1, 4): "prefetched loads arrive, maxima, placement sums",
         (4, 5): "barrier #2", (5, 2): "scale, origins, carry, DMA issue", (2, 6): "transition: move / rescale accumulators (LDS)",
         (6, 7): "classify", (7, 8): "own DMA landed", (8, 9): "barrier #3", (9, 15): "transition: float atomics of what left",
         (15, 10): "pass (gather, gradients, scatter)", (10, 11): "far samples, stores of grad_loc / grad_attn", (11, 14): "next item: decode + issue loads", (14, 12): "barrier #4", (0, 13): "ITEM"}
fl = sys.argv[1] if len(sys.argv) > 1 else "model"
lib = _lib.load()
kw = dict(flavour="model", offset_sigma=6.0) if fl == "wide" else dict(flavour=fl)
x = workloads.make_inputs("encoder", batch=2, seed=3, **kw)
S = x["value"].shape[1]
go = torch.randn(2, S, 256, device="cuda")
_lib.set_variant("backward", "msda_bwd_win")
for _ in range(3):
    ext.ms_deform_attn_backward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], go, 64)
torch.cuda.synchronize()
nb, W = 256, 11
buf = np.zeros((nb, W, 16), dtype=np.uint64)
assert lib.msda_debug_read_prof_bwin(buf.ctypes.data_as(ctypes.c_void_p), nb) == 0
t = buf.astype(np.int64)
t = t[t[:, 0, 13] > 0]
us = (t - t[:, :, 0].min()) * 1e-2
print("flavour %s: %d workgroups" % (fl, len(t)))
for grp, sel in (("level-0 waves", slice(0, 8)), ("waves of levels 1..3", slice(8, 11))):
    print(" " + grp)
    for (a, b), n in NAMES.items():
        dd = (us[:, sel, b] - us[:, sel, a]).reshape(-1)
        print("   %-40s median %6.2f  mean %6.2f  p90 %6.2f us" % (n, np.median(dd), dd.mean(), np.percentile(dd, 90)))
