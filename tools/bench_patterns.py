#!/usr/bin/env python
"""Why does the encoder launch take ~79 us inside bench.py's step and ~72 us in tools/kbench.py?  Same kernel, same inputs:
time the launch pattern piece by piece (HIP events over 40 repetitions after a 100 ms warm-up)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from uninext_amd import _lib, ext as MSDA, workloads  # noqa: E402

_lib.load()
enc = [workloads.make_inputs("encoder", batch=2, seed=10 + i) for i in range(6)]
dec = [workloads.make_inputs("decoder", batch=2, seed=60 + i) for i in range(6)]

def fwd(x, site=None):
    if site is None:
        return MSDA.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
    return bench.call(x, site)

def timeit(fn, n_enc, reps=40):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) < 0.1:
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps

def show(name, fn, n_enc, n_dec=0, dec_us=14.0):
    us = timeit(fn, n_enc)
    print("%-78s %8.1f us per repetition -> %6.1f us per encoder launch%s" % (name, us, (us - n_dec * dec_us) / n_enc, " (decoder launches at %.1f us taken off)" % dec_us if n_dec else ""), flush=True)

_lib.set_variant("forward", "msda_fwd_win")
show("pinned msda_fwd_win, 6 input sets, no call sites", lambda: [fwd(x) for x in enc], 6)
show("pinned msda_fwd_win, ONE input set 6 times", lambda: [fwd(enc[0]) for _ in range(6)], 6)
_lib.set_variant("forward", "auto")
for _ in range(4):
    for i, x in enumerate(enc):
        fwd(x, 1 + i)
show("variant 0, six call sites (as bench.py), 6 input sets", lambda: [fwd(x, 1 + i) for i, x in enumerate(enc)], 6)
dus = timeit(lambda: [fwd(x) for x in dec], 6) / 6
print("decoder launch alone: %.1f us" % dus)
show("bench.py's step: 6 encoder (six sites) + 6 decoder launches", lambda: bench.run_step(enc, dec), 6, 6, dus)
_lib.set_variant("forward", "msda_fwd_win")
show("pinned msda_fwd_win: 6 encoder + 6 decoder launches", lambda: ([fwd(x) for x in enc], [fwd(x) for x in dec]), 6, 6, dus)
show("pinned: encoder and decoder launches alternating", lambda: [(fwd(e), fwd(d)) for e, d in zip(enc, dec)], 6, 6, dus)
_lib.set_variant("forward", "auto")
