import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.environ["MSDA_HIP_LIB"])
p, i = ctypes.c_void_p, ctypes.c_int
lib.matcher_cost_debug_unary_f32.argtypes = [p, i, i, p, p]
g = torch.Generator().manual_seed(0)
x = (torch.randn(1 << 20, generator=g) * 4).cuda()
pos = torch.rand(1 << 20, generator=g).cuda() + 1e-8
def run(inp, op):
    y = torch.empty_like(inp)
    assert lib.matcher_cost_debug_unary_f32(inp.data_ptr(), inp.numel(), op, y.data_ptr(), None) == 0
    torch.cuda.synchronize(); return y
def cmp(name, a, b):
    d = (a.view(torch.int32).to(torch.int64) - b.view(torch.int32).to(torch.int64)).abs()
    print("%-34s differing %7d of %d, max ulp %d" % (name, int((d > 0).sum()), d.numel(), int(d.max())))
cmp("expf(-x) vs torch.exp(-x)", run(x, 0), torch.exp(-x))
cmp("__expf(-x) vs torch.exp(-x)", run(x, 3), torch.exp(-x))
cmp("logf vs torch.log", run(pos, 1), torch.log(pos))
cmp("__logf vs torch.log", run(pos, 4), torch.log(pos))
cmp("1/(1+expf(-x)) vs torch.sigmoid", run(x, 2), torch.sigmoid(x))
cmp("1/(1+__expf(-x)) vs torch.sigmoid", run(x, 5), torch.sigmoid(x))
cmp("rcp(1+expf(-x)) vs torch.sigmoid", run(x, 6), torch.sigmoid(x))
cmp("torch 1/(1+exp(-x)) vs torch.sigmoid", 1.0 / (1.0 + torch.exp(-x)), torch.sigmoid(x))
