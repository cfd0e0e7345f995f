"""msda_fwd_res (uninext_amd/csrc/experiments/, forward variant 13 once experiments/msda_fwd_res_wiring.patch is applied) against the C oracle and
msda_fwd_lg3 on eight shapes (GPU box): full R50 call in three flavours, everything resident, level 3 too big, level 2 partly resident,
arbitrary queries, 5 heads; NaN / inf / negative locations planted."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import MultiScaleDeformableAttention as MSDA
from uninext_amd import _lib, workloads
from oracle import msda_oracle
dev = torch.device("cuda:0")
R50 = ((100, 167), (50, 84), (25, 42), (13, 21))
cases = [
    ("r50 model", dict(kind="encoder", flavour="model", batch=2, levels=R50)),
    ("r50 uniform", dict(kind="encoder", flavour="uniform", batch=2, levels=R50)),
    ("r50 wide", dict(kind="encoder", flavour="model", batch=2, levels=R50, offset_sigma=6.0)),
    ("small all resident", dict(kind="encoder", flavour="uniform", batch=2, levels=((64, 80), (32, 40), (16, 20), (8, 10)))),
    ("level 3 too big", dict(kind="encoder", flavour="uniform", batch=1, levels=((60, 60), (50, 50), (40, 40), (36, 36)))),
    ("level 2 29 of 30 rows", dict(kind="encoder", flavour="uniform", batch=3, levels=((80, 100), (40, 50), (30, 40), (10, 10)))),
    ("decoder queries", dict(kind="decoder", flavour="model", batch=2, levels=R50, num_query=5000)),
    ("5 heads", dict(kind="encoder", flavour="wide_or_uniform", batch=1, levels=((64, 80), (32, 40), (16, 20), (8, 10)), heads=5)),
]
bad = 0
for name, kw in cases:
    if kw["flavour"] == "wide_or_uniform":
        kw["flavour"] = "uniform"
    x = workloads.make_inputs(seed=5, device=dev, **kw)
    loc = x["loc"]
    loc[0, 7, 0, 3, 2, 0] = float("nan"); loc[0, 9, 1, 2, 1, 1] = float("inf"); loc[0, 11, 0, 2, 0, 0] = -3.0
    outs = {}
    for v in ("msda_fwd_res", "msda_fwd_lg3"):
        _lib.set_variant("forward", v)
        try:
            outs[v] = MSDA.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], loc, x["attn"], 64)
            kn = _lib.last_kernel("forward")
        finally:
            _lib.set_variant("forward", "auto")
        if v == "msda_fwd_res":
            took = kn
    torch.cuda.synchronize()
    d_lg3 = float((outs["msda_fwd_res"] - outs["msda_fwd_lg3"]).abs().max())
    Lq = loc.shape[1]
    idx = torch.cat([torch.arange(0, min(700, Lq)), torch.arange(max(0, Lq // 2 - 300), Lq // 2 + 300), torch.arange(Lq - 700, Lq)]).unique().to(dev)
    ref = msda_oracle.forward(x["value"], x["shapes"], x["lsi"], loc[:, idx].contiguous(), x["attn"][:, idx].contiguous())
    d_or = float(np.abs(outs["msda_fwd_res"][:, idx].cpu().numpy() - ref).max())
    fin = bool(torch.isfinite(outs["msda_fwd_res"]).all())
    ok = took == "msda_fwd_res" and d_lg3 < 2e-5 and d_or < 1e-4 and fin
    bad += not ok
    print(f"{name:26s} kernel {took:14s} vs lg3 {d_lg3:.2e}  vs oracle {d_or:.2e} finite {fin}  {'OK' if ok else 'FAIL'}", flush=True)
print("FAILED" if bad else "ALL OK")
