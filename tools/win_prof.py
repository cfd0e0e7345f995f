#!/usr/bin/env python
"""Phase timeline of msda_fwd_win from in-kernel timestamps (profiling build: `make -C uninext_amd/csrc prof`, run with
MSDA_HIP_LIB=uninext_amd/lib/libmsda_hip_prof.so).  GPU box only."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSDA_HIP_LIB", os.path.join(ROOT, "uninext_amd", "lib", "libmsda_hip_prof.so"))
from uninext_amd import _lib, ext, workloads  # noqa: E402

NAMES = ["geometry (meta, barrier)", "loc/attn load + coords", "placement sums + barrier", "origin + window DMA issue",
         "prepare samples", "far pass", "DMA wait + barrier", "LDS pass round 0", "store + later rounds"]


def main():
    flavour = sys.argv[1] if len(sys.argv) > 1 else "model"
    lib = _lib.load()
    kw = dict(flavour="model", offset_sigma=6.0) if flavour == "wide" else dict(flavour=flavour)
    xs = [workloads.make_inputs("encoder", batch=2, seed=1 + r, **kw) for r in range(4)]
    _lib.set_variant("forward", "msda_fwd_win")
    for r in range(8):
        x = xs[r % 4]
        ext.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
    torch.cuda.synchronize()
    S = xs[0]["value"].shape[1]
    nb = 512                                # persistent grid: 2 workgroups per CU
    buf = np.zeros((nb, 16), dtype=np.uint64)
    rc = lib.msda_debug_read_prof(buf.ctypes.data_as(ctypes.c_void_p), nb)
    assert rc == 0, rc
    t = buf[:, :10].astype(np.int64)
    real = t[:, 9] > 0                     # workgroups that had a tile
    t = t[real]
    t0 = t[:, 0].min()
    us = (t - t0) * 1e-2                    # 100 MHz
    print("flavour %s: first work item of each of %d persistent workgroups; its span %.1f us" % (flavour, real.sum(), us[:, 9].max()))
    for i, n in enumerate(NAMES):
        dd = us[:, i + 1] - us[:, i]
        print("  %-32s median %6.2f  mean %6.2f  p10 %6.2f  p90 %6.2f us" % (n, np.median(dd), dd.mean(),
                                                                          np.percentile(dd, 10), np.percentile(dd, 90)))
    tot = us[:, 9] - us[:, 0]
    print("  workgroup total                  median %6.2f  mean %6.2f" % (np.median(tot), tot.mean()))
    starts = np.sort(us[:, 0])
    print("  starts: 10%% by %.1f us, 50%% by %.1f, 90%% by %.1f, last %.1f" % tuple(np.percentile(starts, [10, 50, 90, 100])))
    # concurrency over time
    ev = np.concatenate([np.stack([us[:, 0], np.ones(len(us))], 1), np.stack([us[:, 9], -np.ones(len(us))], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    conc = np.cumsum(ev[:, 1])
    dur = np.diff(ev[:, 0], append=ev[-1, 0])
    print("  mean resident workgroups over the span: %.1f (512 fit)" % ((conc * dur).sum() / max(us[:, 9].max(), 1e-9)))


if __name__ == "__main__":
    main()
