"""Mint golden input/output vectors by running the REFERENCE's own Python oracle.

Run in the build container only (needs /root/reference; the GPU box has no copy):

    python tests/golden/make_golden.py

It imports projects/UNINEXT/uninext/models/deformable_detr/ops/functions/ms_deform_attn_func.py
from the reference checkout (the file imports the compiled CUDA extension at top level,
func.py:18, so an empty stub module of that name is planted in sys.modules first), runs
`ms_deform_attn_core_pytorch` (func.py:43-63) in float64 on seeded inputs, differentiates
through it with autograd for the backward vectors, and writes one .npz per case next to
this script.  The fixtures pin oracle/msda_oracle.c (tests/test_oracle_golden.py) and the
HIP path (tests/test_msda_gpu.py).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("UNINEXT_REFERENCE", "/root/reference")
FUNC = os.path.join(REF, "projects/UNINEXT/uninext/models/deformable_detr/ops/functions/ms_deform_attn_func.py")
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_func():
    sys.modules.setdefault("MultiScaleDeformableAttention", types.ModuleType("MultiScaleDeformableAttention"))
    spec = importlib.util.spec_from_file_location("_ref_ms_deform_attn_func", FUNC)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def lsi_of(shapes):
    hw = shapes.prod(1)
    return torch.cat((shapes.new_zeros((1,)), hw.cumsum(0)[:-1]))


def case_testpy(seed):
    # ops/test.py:21-37 fixture and distributions
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    torch.manual_seed(seed)
    value = torch.rand(N, S, M, D) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    attn = torch.rand(N, Lq, M, L, P) + 1e-5
    attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    return value, shapes, loc, attn


def case_random(seed, N, M, D, Lq, shapes, P, lo=-0.15, hi=1.15, scale=1.0):
    shapes = torch.as_tensor(shapes, dtype=torch.long)
    L = shapes.shape[0]
    S = int(shapes.prod(1).sum())
    g = torch.Generator().manual_seed(seed)
    value = torch.randn(N, S, M, D, generator=g) * scale
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g) * (hi - lo) + lo
    attn = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P)
    return value, shapes, loc, attn


def case_border(seed):
    # locations that land exactly on pixel centres, cell edges and the -1 / H cut-offs
    shapes = torch.as_tensor([(4, 5), (2, 3)], dtype=torch.long)
    N, M, D, P = 1, 2, 4, 4
    L = 2
    S = int(shapes.prod(1).sum())
    g = torch.Generator().manual_seed(seed)
    value = torch.randn(N, S, M, D, generator=g)
    specials = []
    for (H, W) in shapes.tolist():
        xs = [0.0, 1.0, 0.5 / W, (W - 0.5) / W, -0.5 / W, (W + 0.5) / W, 1.5 / W, -0.49 / W, (W + 0.49) / W, 0.5]
        ys = [0.0, 1.0, 0.5 / H, (H - 0.5) / H, -0.5 / H, (H + 0.5) / H, 1.5 / H, -0.49 / H, (H + 0.49) / H, 0.5]
        specials.append((xs, ys))
    Lq = 25
    loc = torch.empty(N, Lq, M, L, P, 2)
    for q in range(Lq):
        for m in range(M):
            for l in range(L):
                xs, ys = specials[l]
                for p in range(P):
                    k = (q * 7 + m * 3 + p) % len(xs)
                    j = (q * 5 + m + 2 * p + l) % len(ys)
                    loc[0, q, m, l, p, 0] = xs[k]
                    loc[0, q, m, l, p, 1] = ys[j]
    attn = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P)
    return value, shapes, loc, attn


CASES = {
    # name: (builder, kwargs)
    "testpy_seed3": (case_testpy, dict(seed=3)),
    "d32_l4_p4": (case_random, dict(seed=11, N=2, M=8, D=32, Lq=37, shapes=[(10, 17), (5, 9), (3, 5), (2, 3)], P=4)),
    "d32_inrange_small_values": (case_random, dict(seed=12, N=1, M=8, D=32, Lq=64, shapes=[(12, 9), (6, 5), (3, 3), (2, 2)], P=4, lo=0.0, hi=1.0, scale=0.01)),
    "border": (case_border, dict(seed=13)),
    "d30_odd": (case_random, dict(seed=14, N=1, M=3, D=30, Lq=11, shapes=[(7, 5), (4, 3), (2, 2)], P=2)),
    "d71_odd": (case_random, dict(seed=15, N=2, M=1, D=71, Lq=5, shapes=[(5, 6), (3, 3)], P=3)),
    "d64_m4": (case_random, dict(seed=16, N=1, M=4, D=64, Lq=19, shapes=[(8, 8), (4, 4), (2, 2), (1, 1)], P=4)),
    "single_level_point": (case_random, dict(seed=17, N=3, M=2, D=8, Lq=1, shapes=[(3, 4)], P=1)),
    "d16_m16_p8": (case_random, dict(seed=18, N=1, M=16, D=16, Lq=23, shapes=[(9, 11), (5, 6)], P=8)),
    "far_outside": (case_random, dict(seed=19, N=1, M=8, D=32, Lq=16, shapes=[(6, 7), (3, 4), (2, 2), (1, 1)], P=4, lo=-3.0, hi=4.0)),
}


def mint_encoder_shaped(ref):
    """An encoder-shaped call (num_query == spatial_size >= 1024, 32 channels, 4 levels x 4 points) with the model-like
    locations of the bench (uninext_amd/workloads.py): the shape the window forward kernel and the tiled backward kernel
    take.  Inputs are float32 values (run through the reference in float64), everything is stored as float32 to keep
    the file small; the tests hold the float64 oracle to 1e-6 and the kernels to 1e-4 against it."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from uninext_amd import workloads
    levels = ((24, 33), (12, 17), (6, 9), (3, 5))            # S = 1065
    x = workloads.make_inputs("encoder", "model", batch=1, levels=levels, heads=2, seed=31, device="cpu")
    value, loc, attn = x["value"].double(), x["loc"].double(), x["attn"].double()
    value.requires_grad_(True)
    loc.requires_grad_(True)
    attn.requires_grad_(True)
    out = ref.ms_deform_attn_core_pytorch(value, x["shapes"], loc, attn)
    grad_out = torch.randn(out.shape, generator=torch.Generator().manual_seed(1031)).double()   # float32 values
    gv, gl, ga = torch.autograd.grad(out, (value, loc, attn), grad_out)
    f32 = lambda t: t.detach().float().numpy()
    np.savez_compressed(os.path.join(HERE, "encshape_s1065_m2.npz"), value=f32(value), shapes=x["shapes"].numpy(),
                        lsi=x["lsi"].numpy(), loc=f32(loc), attn=f32(attn), out=f32(out), grad_out=f32(grad_out),
                        grad_value=f32(gv), grad_loc=f32(gl), grad_attn=f32(ga))
    print(f"encshape_s1065_m2: out{tuple(out.shape)} |out|max={out.abs().max():.3e}")


def main():
    ref = load_reference_func()
    mint_encoder_shaped(ref)
    for name, (builder, kw) in CASES.items():
        value, shapes, loc, attn = builder(**kw)
        value, loc, attn = value.double(), loc.double(), attn.double()
        value.requires_grad_(True)
        loc.requires_grad_(True)
        attn.requires_grad_(True)
        out = ref.ms_deform_attn_core_pytorch(value, shapes, loc, attn)
        g = torch.Generator().manual_seed(1000 + kw["seed"])
        grad_out = torch.randn(out.shape, generator=g, dtype=torch.float64)
        gv, gl, ga = torch.autograd.grad(out, (value, loc, attn), grad_out)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            value=value.detach().numpy(), shapes=shapes.numpy(), lsi=lsi_of(shapes).numpy(),
            loc=loc.detach().numpy(), attn=attn.detach().numpy(), out=out.detach().numpy(),
            grad_out=grad_out.numpy(), grad_value=gv.numpy(), grad_loc=gl.numpy(), grad_attn=ga.numpy())
        print(f"{name}: out{tuple(out.shape)} |out|max={out.abs().max():.3e}")


if __name__ == "__main__":
    main()
