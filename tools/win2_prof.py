#!/usr/bin/env python
"""Phase timeline of msda_fwd_win2 from in-kernel timestamps of every wave (profiling build: `make -C uninext_amd/csrc prof`,
run with MSDA_HIP_LIB=uninext_amd/lib/libmsda_hip_prof.so).  GPU box only."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSDA_HIP_LIB", os.path.join(ROOT, "uninext_amd", "lib", "libmsda_hip_prof.so"))
from uninext_amd import _lib, ext, workloads  # noqa: E402

NAMES = {(0, 1): "scalar loads, geometry, load issue", (1, 2): "barrier #1", (2, 4): "locations arrive, coords, placement sums",
         (4, 5): "barrier #2", (5, 6): "origins", (6, 7): "classification", (7, 8): "far issue + window DMA issue", (8, 9): "far consume (far loads arrive)",
         (9, 10): "own DMA landed", (10, 11): "barrier #3", (11, 12): "further far steps", (12, 13): "LDS pass", (13, 15): "stores acknowledged", (14, 0): "kernel entry -> item start (prologue)", (0, 13): "TOTAL"}
WAVES = 11


def main():
    flavour = sys.argv[1] if len(sys.argv) > 1 else "model"
    lib = _lib.load()
    kw = dict(flavour="model", offset_sigma=6.0) if flavour == "wide" else dict(flavour=flavour)
    xs = [workloads.make_inputs("encoder", batch=2, seed=1 + r, **kw) for r in range(4)]
    _lib.set_variant("forward", "msda_fwd_win2")
    for r in range(8):
        x = xs[r % 4]
        ext.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
    torch.cuda.synchronize()
    nb = 2784
    buf = np.zeros((nb, WAVES, 16), dtype=np.uint64)
    rc = lib.msda_debug_read_prof2(buf.ctypes.data_as(ctypes.c_void_p), nb)
    assert rc == 0, rc
    t = buf.astype(np.int64)
    real = t[:, 0, 13] > 0
    t = t[real]
    t0 = t[:, :, 0].min()
    us = (t - t0) * 1e-2                    # 100 MHz
    print("flavour %s: %d workgroups with an item; launch span %.1f us" % (flavour, real.sum(), us[:, :, 13].max()))
    for grp, sel in (("level-0 waves (0..7)", slice(0, 8)), ("waves of levels 1..3 (8..10)", slice(8, 11))):
        print(" " + grp)
        for (a, b), n in NAMES.items():
            dd = (us[:, sel, b] - us[:, sel, a]).reshape(-1)
            print("   %-44s median %6.2f  mean %6.2f  p10 %6.2f  p90 %6.2f us" % (n, np.median(dd), dd.mean(), np.percentile(dd, 10), np.percentile(dd, 90)))
    wg_start, wg_end = us[:, :, 14].min(1), us[:, :, 15].max(1)
    print(" workgroup lifetime (first wave enters the kernel .. last store acknowledged): median %.2f mean %.2f us" % (np.median(wg_end - wg_start), (wg_end - wg_start).mean()))
    ev = np.concatenate([np.stack([wg_start, np.ones(len(us))], 1), np.stack([wg_end, -np.ones(len(us))], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    conc = np.cumsum(ev[:, 1])
    dur = np.diff(ev[:, 0], append=ev[-1, 0])
    print(" mean resident workgroups over the span: %.1f (512 fit)" % ((conc * dur).sum() / max(wg_end.max(), 1e-9)))


if __name__ == "__main__":
    main()
