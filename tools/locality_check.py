#!/usr/bin/env python
"""Automatic forward-kernel choice on the GPU box (include/msda_hip.h, msda_hip_forward_locality): per location
flavour the launch time of the window kernel, of the gather kernel and of variant 0, the far fraction the window
kernel reports, and the kernels variant 0 runs while the inputs change flavour."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from uninext_amd import _lib, workloads
_lib.load()
def t(fn, reps=12):
    evs=[]
    for _ in range(reps):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a,b))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a,b in evs)/reps*1e3
for fl in ("model","wide","uniform","model"):
    xs=[workloads.make_inputs("encoder", batch=2, seed=70+i, **bench.flavour_kwargs(fl)) for i in range(3)]
    k=[0]
    def one():
        k[0]+=1; bench.call(xs[k[0]%3])
    res=[]
    for var in ("msda_fwd_win","msda_fwd_lg3","auto"):
        _lib.set_variant("forward", var)
        one(); one(); torch.cuda.synchronize()
        us=t(one)
        res.append("%s %.1f us (%s)" % (var, us, _lib.last_kernel("forward")))
        if var == "msda_fwd_win":
            res.append("locality %s" % (_lib.forward_locality(),))
    print(fl, " | ".join(res), flush=True)
# transitions under auto: how many calls until the kernel follows the inputs
_lib.set_variant("forward","auto")
seq=[]
for fl in ("uniform","model","uniform"):
    xs=[workloads.make_inputs("encoder", batch=2, seed=90+i, **bench.flavour_kwargs(fl)) for i in range(2)]
    names=[]
    for i in range(70):
        bench.call(xs[i%2]); torch.cuda.synchronize()
        names.append(_lib.last_kernel("forward")[9:])
    seq.append(fl+": "+" ".join(names[:6])+" ... "+" ".join(names[60:70]))
print("\n".join(seq))
