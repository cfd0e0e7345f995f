import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uninext_amd import _lib, ext, workloads
_lib.load()
x = workloads.make_inputs("encoder", "model", batch=2, seed=3)
_lib.set_variant("forward", "msda_fwd_win2")
out = ext.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
torch.cuda.synchronize()
print("ran", _lib.last_kernel("forward"), float(out.abs().max()))
