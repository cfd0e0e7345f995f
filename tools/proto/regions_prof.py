#!/usr/bin/env python
"""Calls msda_bwd_regions a few times per location flavour at the R50 shapes (for rocprofv3 --kernel-trace --stats:
tools/rocprof_summary.py trace prints the per-kernel averages).  argv[1]: flavour (default: all three)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from uninext_amd import _lib, ext, workloads  # noqa: E402

_lib.load()
for flavour in ([sys.argv[1]] if len(sys.argv) > 1 else ["model", "wide", "uniform"]):
    x = workloads.make_workload("r50_infer_encoder", flavour=flavour, device="cuda")
    go = torch.randn(2, 22223, 256, generator=torch.Generator().manual_seed(5)).cuda()
    _lib.set_variant("backward", "msda_bwd_regions")
    for _ in range(8):
        ext.ms_deform_attn_backward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], go, 64)
    torch.cuda.synchronize()
    _lib.set_variant("backward", 0)
