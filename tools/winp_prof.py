#!/usr/bin/env python
"""Phase timeline of msda_fwd_winp's iteration 4 from in-kernel timestamps of every wave (experiments library with
experiments/msda_fwd_winp.hip rebuilt -DWINP_PROF: tools/exp_build.sh p_prof msda_fwd_winp -DWINP_PROF; GPU box only)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSDA_HIP_LIB", os.path.join(ROOT, "uninext_amd", "lib", "abl", "libmsda_p_prof.so"))
from uninext_amd import _lib, ext, workloads  # noqa: E402

CONS = {(0, 1): "previous stores issued -> vmcnt(0)", (1, 2): "barrier", (2, 5): "records, gather (+ next item's location loads)",
        (5, 6): "quad sums", (6, 8): "far samples + stores issued"}


def main():
    flavour = sys.argv[1] if len(sys.argv) > 1 else "model"
    lib = _lib.load()
    kw = dict(flavour="model", offset_sigma=6.0) if flavour == "wide" else dict(flavour=flavour)
    xs = [workloads.make_inputs("encoder", batch=2, seed=1 + r, **kw) for r in range(4)]
    _lib.set_variant("forward", "msda_fwd_winp")
    for r in range(8):
        x = xs[r % 4]
        ext.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
    torch.cuda.synchronize()
    nb = 256
    buf = np.zeros((nb, 12, 16), dtype=np.uint64)
    lib.msda_debug_read_prof_winp.argtypes, lib.msda_debug_read_prof_winp.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int
    assert lib.msda_debug_read_prof_winp(buf.ctypes.data_as(ctypes.c_void_p), nb) == 0
    t = buf.astype(np.int64) * 1e-2                                # us (100 MHz)
    ok = t[:, 0, 0] > 0
    t = t[ok]
    print("flavour %s: iteration 4 of %d workgroups" % (flavour, ok.sum()))
    for name, waves in (("level-0 waves 0..7", slice(0, 8)), ("rest waves 8..10", slice(8, 11))):
        w = t[:, waves, :]
        print(" %s" % name)
        for (a, b) in CONS:
            dd = (w[:, :, b] - w[:, :, a]).reshape(-1)
            dd = dd[(w[:, :, b].reshape(-1) > 0) & (w[:, :, a].reshape(-1) > 0)]
            if len(dd):
                print("   %-48s median %6.2f  mean %6.2f  p10 %6.2f  p90 %6.2f us" % (CONS[(a, b)], np.median(dd), dd.mean(),
                                                                                      np.percentile(dd, 10), np.percentile(dd, 90)))
        tot = (w[:, :, 8] - w[:, :, 0]).reshape(-1)
        print("   iteration (stamp 0 -> stores issued)               median %6.2f  mean %6.2f" % (np.median(tot), tot.mean()))
    p = t[:, 11, :]
    print(" producer (wave 11)")
    for a, b, name in ((0, 1, "barrier"), (1, 2, "window DMA of item n + 1 issued (its share)"), (2, 3, "finish(n + 2): sums, origins"),
                       (3, 4, "produce(n + 3): geometry, subsample loads"), (4, 5, "window DMA landed")):
        dd = p[:, b] - p[:, a]
        print("   %-48s median %6.2f  mean %6.2f  p10 %6.2f  p90 %6.2f us" % (name, np.median(dd), dd.mean(), np.percentile(dd, 10), np.percentile(dd, 90)))
    # barrier-to-barrier: consumers' stamp 2 of iteration 4 is all we have; the iteration length = launch time / iterations
    skew = t[:, :11, 1].max(axis=1) - t[:, :11, 1].min(axis=1)
    print(" arrival skew of the consumers at the barrier: median %.2f us; producer arrives %.2f us (median) before the last consumer" % (
        np.median(skew), np.median(t[:, :11, 1].max(axis=1) - p[:, 0])))


if __name__ == "__main__":
    main()
