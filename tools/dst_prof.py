#!/usr/bin/env python
"""Slot timeline of msda_bwd_dst from in-kernel timestamps.  Build (the instrumentation is a patch, not in the product source):
    (cd uninext_amd/csrc && patch -p0 -i experiments/msda_bwd_dst_prof.patch) && bash tools/abl_build.sh dstprof msda_bwd_dst -DMSDA_DST_PROF
    (cd uninext_amd/csrc && patch -p0 -R -i experiments/msda_bwd_dst_prof.patch)
GPU box only."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSDA_HIP_LIB", os.path.join(ROOT, "uninext_amd", "lib", "abl", "libmsda_dstprof.so"))
from uninext_amd import _lib, ext, workloads  # noqa: E402


def main():
    lib = _lib.load()
    xs = [workloads.make_inputs("decoder", "model", batch=2, levels=workloads.R50_LEVELS_TRAIN, num_query=1100, seed=1 + r) for r in range(3)]
    go = torch.randn(2, 1100, 256, device="cuda")
    _lib.set_variant("backward", "msda_bwd_dst")
    for r in range(6):
        x = xs[r % 3]
        ext.ms_deform_attn_backward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], go, 64)
    torch.cuda.synchronize()
    nb = 512
    buf = np.zeros((nb, 8, 6), dtype=np.uint64)
    rc = lib.msda_debug_read_dst_prof(buf.ctypes.data_as(ctypes.c_void_p), nb)
    assert rc == 0, rc
    t = buf.astype(np.int64)
    used = t[:, :, 4] > 0
    t0 = t[:, 0, 0][t[:, 0, 0] > 0].min()
    us = (t[:, :, :5] - t0) * 1e-2
    lvl = t[:, :, 5] >> 16
    nrec = t[:, :, 5] & 0xffff
    print("workgroups with a slot: %d; slots per workgroup: %s" % (used[:, 0].sum(), np.bincount(used.sum(1))))
    print("kernel span (first start to last end): %.1f us" % us[:, :, 4][used].max())
    for l in (3, 2, 1, 0):
        sel = used & (lvl == l)
        if not sel.any():
            continue
        d = us[sel]
        print("level %d: %4d slots (of sampled), wave-0 records median %3d max %3d | scan %5.2f  process-tail %5.2f  barrier %5.2f  flush %5.2f  total %5.2f us (medians; p90 total %5.2f)" % (
            l, sel.sum(), np.median(nrec[sel]), nrec[sel].max(), np.median(d[:, 1] - d[:, 0]), np.median(d[:, 2] - d[:, 1]), np.median(d[:, 3] - d[:, 2]),
            np.median(d[:, 4] - d[:, 3]), np.median(d[:, 4] - d[:, 0]), np.percentile(d[:, 4] - d[:, 0], 90)))
    ends = us[:, :, 4].max(1)
    print("workgroup end times: p10 %.1f  p50 %.1f  p90 %.1f  max %.1f" % tuple(np.percentile(ends[used[:, 0]], [10, 50, 90, 100])))
    starts = us[:, 0, 0][used[:, 0]]
    print("workgroup start times: p50 %.1f  p90 %.1f  max %.1f" % tuple(np.percentile(starts, [50, 90, 100])))


if __name__ == "__main__":
    main()
