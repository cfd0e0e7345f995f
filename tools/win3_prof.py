#!/usr/bin/env python
"""Timeline of msda_fwd_win3's steady-state iteration (third item of every workgroup) from in-kernel s_memrealtime stamps.
Needs a library built with -DMSDA_WIN3_PROF (tools/abl_build.sh w3prof msda_fwd_win3 "-DMSDA_WIN3_PROF"; MSDA_HIP_LIB=...)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uninext_amd import _lib, ext, workloads  # noqa: E402
NAMES = {(0, 1): "barrier B", (1, 2): "next item: query decode, loads issued", (2, 3): "classify + gather levels 0-1",
         (3, 4): "next item: coordinates, placement sums", (4, 5): "barrier A", (5, 6): "next item: origins, window DMA issued",
         (6, 7): "gather levels 2-3", (7, 8): "far samples", (8, 9): "wait DMA + stores", (0, 9): "ITERATION"}
fl = sys.argv[1] if len(sys.argv) > 1 else "model"
lib = _lib.load()
kw = dict(flavour="model", offset_sigma=6.0) if fl == "wide" else dict(flavour=fl)
x = workloads.make_inputs("encoder", batch=2, seed=3, **kw)
_lib.set_variant("forward", "msda_fwd_win3")
for _ in range(3):
    ext.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
torch.cuda.synchronize()
nb, W = 256, 12
buf = np.zeros((nb, W, 16), dtype=np.uint64)
assert lib.msda_debug_read_prof3(buf.ctypes.data_as(ctypes.c_void_p), nb) == 0
t = buf.astype(np.int64)
t = t[t[:, 0, 9] > 0]
us = (t - t[:, :, 0].min()) * 1e-2
print("flavour %s: %d workgroups" % (fl, len(t)))
for grp, sel in (("level-0 waves", slice(0, 8)), ("waves of levels 1..3", slice(8, 12))):
    print(" " + grp)
    for (a, b), n in NAMES.items():
        dd = (us[:, sel, b] - us[:, sel, a]).reshape(-1)
        print("   %-44s median %6.2f  mean %6.2f  p90 %6.2f us" % (n, np.median(dd), dd.mean(), np.percentile(dd, 90)))
