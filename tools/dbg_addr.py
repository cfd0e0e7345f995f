import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uninext_amd import _lib, ext, workloads
_lib.load()
x = workloads.make_inputs("encoder", "model", batch=2, seed=3)
_lib.set_variant("forward", "msda_fwd_win2")
out = ext.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
torch.cuda.synchronize()
o = out.view(torch.int32).view(-1, 4).cpu()
M, T = 8, 704
def show(kk, m, tids):
    for t in tids:
        r = o[(kk * M + m) * T + t].tolist()
        print("kk %3d m %d tid %3d: live %d qidx %6d pair %8d pair_img %d hi %d" % (kk, m, t, (r[0] >> 31) & 1, r[0] & 0x7fffffff, r[1], r[2], r[3]))
show(0, 0, [0, 4, 60, 64, 448, 512, 516, 572, 576, 640, 700])
show(5, 3, [0, 4, 64, 512, 516, 640])
show(150, 7, [0, 4, 64, 512, 516, 640])
show(285, 1, [0, 4, 64, 512, 516, 640])
