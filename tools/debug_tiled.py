import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uninext_amd import _lib, ext, workloads
_lib.load()
levels = ((25, 42), (13, 21), (7, 11), (4, 6))
x = workloads.make_inputs("encoder", "model", batch=1, levels=levels, seed=33)
a = (x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"])
def run(v, args):
    _lib.set_variant("forward", v); o = ext.ms_deform_attn_forward(*args, 64); _lib.set_variant("forward", 0); return o
ref = run(2, a); out = run(3, a)
err = (out - ref).abs().view(1, -1, 8, 8, 4)   # [N, q, head, chunk, 4]
print("max err", float(err.max()), "frac bad elems", float((err > 1e-4).float().mean()))
print("bad frac per head", (err > 1e-4).float().mean((0, 1, 3, 4)).tolist())
print("bad frac per chunk", (err > 1e-4).float().mean((0, 1, 2, 4)).tolist())
S = ref.shape[1]
perq = (err > 1e-4).float().mean((0, 2, 3, 4))
lsi = x["lsi"].tolist() + [S]
for l in range(4):
    seg = perq[lsi[l]:lsi[l+1]]
    print("level", l, "bad query frac", float((seg > 0).float().mean()), "first bad", (seg > 0).nonzero()[:10].flatten().tolist())
# one-hot attention on sample s: isolates per-sample errors
for s in [0, 1, 4, 5, 15]:
    at = torch.zeros_like(x["attn"]).view(1, S, 8, 16); at[..., s] = 1.0
    at = at.view_as(x["attn"]).contiguous()
    aa = (x["value"], x["shapes"], x["lsi"], x["loc"], at)
    e = (run(3, aa) - run(2, aa)).abs()
    print("one-hot sample", s, "max err", float(e.max()), "bad frac", float((e > 1e-4).float().mean()))
# value = ones
ones = torch.ones_like(x["value"])
ao = (ones, x["shapes"], x["lsi"], x["loc"], x["attn"])
e = (run(3, ao) - run(2, ao)).abs()
print("ones: max err", float(e.max()), "bad frac", float((e > 1e-4).float().mean()))
# value varying only by pixel index
pix = torch.arange(S, device="cuda", dtype=torch.float32).view(1, S, 1, 1).expand(1, S, 8, 32).contiguous()
ap = (pix, x["shapes"], x["lsi"], x["loc"], x["attn"])
o3, o2 = run(3, ap), run(2, ap)
e = (o3 - o2).abs()
print("pixel-index value: max err", float(e.max()), "bad frac", float((e > 1e-2).float().mean()))
q = int((e.view(S, -1).max(1)[0] > 1e-2).nonzero()[0]) if (e > 1e-2).any() else 0
print("q", q, "tiled", o3[0, q, :8].tolist(), "ref", o2[0, q, :8].tolist())
# value varying only by channel
ch = torch.arange(256, device="cuda", dtype=torch.float32).view(1, 1, 8, 32).expand(1, S, 8, 32).contiguous()
ac = (ch, x["shapes"], x["lsi"], x["loc"], x["attn"])
o3, o2 = run(3, ac), run(2, ac)
print("channel-index value: max err", float((o3 - o2).abs().max()))
print("tiled", o3[0, 100, :40].tolist()); print("ref  ", o2[0, 100, :40].tolist())
