#!/usr/bin/env python
"""Far fraction the window forward kernel reports for itself on the three location flavours (GPU box; A/B builds via
MSDA_HIP_LIB): python tools/far_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uninext_amd import _lib, ext, workloads  # noqa: E402

_lib.load()
for site, flavour in enumerate(("model", "wide", "uniform")):
    kw = dict(flavour="model", offset_sigma=6.0) if flavour == "wide" else dict(flavour=flavour)
    x = workloads.make_inputs("encoder", batch=2, seed=3, device="cuda", **kw)
    with ext.call_site(10 + site):
        for _ in range(3):
            ext.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
    torch.cuda.synchronize()
    n, far = _lib.forward_locality()
    print("%-8s kernel %-14s reports %d far fraction %.4f" % (flavour, _lib.last_kernel("forward"), n, far))
