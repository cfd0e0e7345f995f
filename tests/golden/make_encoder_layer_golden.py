"""Mint golden vectors for the deformable encoder layer with the REFERENCE's own code (build container only).

    python tests/golden/make_encoder_layer_golden.py

The reference module files are executed as they are, in fp64 on the CPU:
  * ops/modules/ms_deform_attn.py is loaded under a synthetic package whose `functions.MSDeformAttnFunction.apply`
    is the reference's own CPU path `ms_deform_attn_core_pytorch` (ops/functions/ms_deform_attn_func.py:43-63; the
    compiled CUDA extension it would otherwise call cannot exist here);
  * class DeformableTransformerEncoderLayer and _get_activation_fn are cut out of
    models/deformable_detr/deformable_transformer_dino.py (:330-370, :543-552) with `ast` (the file imports the whole
    model zoo) and bound to that MSDeformAttn.
Only inputs, parameters and outputs are stored (tests/golden/enclayer_*.npz).
"""
import ast
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = os.environ.get("UNINEXT_REFERENCE", "/root/reference")
OPS = os.path.join(REF, "projects/UNINEXT/uninext/models/deformable_detr/ops")
DINO = os.path.join(REF, "projects/UNINEXT/uninext/models/deformable_detr/deformable_transformer_dino.py")
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_layer():
    sys.modules.setdefault("MultiScaleDeformableAttention", types.ModuleType("MultiScaleDeformableAttention"))
    spec = importlib.util.spec_from_file_location("_ref_func", os.path.join(OPS, "functions/ms_deform_attn_func.py"))
    func = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(func)

    class CpuFunction:   # stands in for the autograd Function: same call signature, the reference's CPU arithmetic
        @staticmethod
        def apply(value, shapes, level_start, loc, attn, im2col_step):
            return func.ms_deform_attn_core_pytorch(value, shapes, loc, attn)

    pkg = types.ModuleType("refops"); pkg.__path__ = []
    fpk = types.ModuleType("refops.functions"); fpk.MSDeformAttnFunction = CpuFunction
    mpk = types.ModuleType("refops.modules"); mpk.__path__ = []
    sys.modules.update({"refops": pkg, "refops.functions": fpk, "refops.modules": mpk})
    spec = importlib.util.spec_from_file_location("refops.modules.ms_deform_attn", os.path.join(OPS, "modules/ms_deform_attn.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["refops.modules.ms_deform_attn"] = mod
    spec.loader.exec_module(mod)

    tree = ast.parse(open(DINO).read())
    body = [n for n in tree.body if (isinstance(n, ast.ClassDef) and n.name == "DeformableTransformerEncoderLayer")
            or (isinstance(n, ast.FunctionDef) and n.name == "_get_activation_fn")]
    assert len(body) == 2
    ns = {"torch": torch, "nn": nn, "F": F, "MSDeformAttn": mod.MSDeformAttn}
    exec(compile(ast.Module(body=body, type_ignores=[]), DINO, "exec"), ns)
    return ns["DeformableTransformerEncoderLayer"]


def main():
    cls = load_reference_layer()
    torch.manual_seed(5)
    levels = [(9, 12), (5, 6), (3, 3), (2, 2)]
    S = sum(h * w for h, w in levels)
    shapes = torch.as_tensor(levels, dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    for name, masked in (("enclayer_plain", False), ("enclayer_masked", True)):
        layer = cls(d_model=64, d_ffn=128, dropout=0.1, activation="relu", n_levels=4, n_heads=2, n_points=4).double().eval()
        with torch.no_grad():   # move off the zero-weight initialisation so that every parameter matters
            layer.self_attn.sampling_offsets.weight.normal_(0, 0.05)
            layer.self_attn.attention_weights.weight.normal_(0, 0.2)
            for p in layer.parameters():
                if p.dim() == 1:
                    p.add_(torch.randn_like(p) * 0.1)
        N = 2
        src = torch.randn(N, S, 64, dtype=torch.float64)
        pos = torch.randn(N, S, 64, dtype=torch.float64) * 0.5
        ref = torch.rand(N, S, 4, 2, dtype=torch.float64)
        mask = None
        if masked:
            mask = torch.zeros(N, S, dtype=torch.bool)
            mask[1, -20:] = True
        with torch.no_grad():
            out = layer(src, pos, ref, shapes, lsi, mask)
        arrays = {"src": src.numpy(), "pos": pos.numpy(), "ref": ref.numpy(), "shapes": shapes.numpy(), "lsi": lsi.numpy(),
                  "out": out.numpy()}
        if masked:
            arrays["mask"] = mask.numpy()
        arrays.update({"p:" + k: v.detach().numpy() for k, v in layer.state_dict().items()})
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
        print(name, tuple(src.shape), "->", tuple(out.shape))


STACK_LEVELS = [(24, 33), (12, 17), (6, 9), (3, 5)]       # S = 1065 >= 1024: the encoder-sized kernels of the HIP path
STACK_LAYERS = 6


def stack_inputs(dtype=torch.float64):
    """Seeded inputs and parameters of the 6-layer stack (shared with tests/test_encoder_layer_gpu.py: only the
    reference's output is stored, the 4.7 M parameters are re-drawn from the same CPU generator on both sides and
    pinned by a digest)."""
    g = torch.Generator().manual_seed(2024)
    S = sum(h * w for h, w in STACK_LEVELS)
    src = torch.randn(1, S, 256, generator=g, dtype=torch.float64)
    pos = torch.randn(1, S, 256, generator=g, dtype=torch.float64) * 0.5
    ref = torch.rand(1, S, 4, 2, generator=g, dtype=torch.float64)
    params = []
    for _ in range(STACK_LAYERS):
        p = {}
        def lin(name, o, i, wstd):
            p[name + ".weight"] = torch.randn(o, i, generator=g, dtype=torch.float64) * wstd
            p[name + ".bias"] = torch.randn(o, generator=g, dtype=torch.float64) * 0.1
        lin("self_attn.sampling_offsets", 256, 256, 0.02)
        lin("self_attn.attention_weights", 128, 256, 0.1)
        lin("self_attn.value_proj", 256, 256, 1.0 / 16)
        lin("self_attn.output_proj", 256, 256, 1.0 / 16)
        lin("linear1", 1024, 256, 1.0 / 16)
        lin("linear2", 256, 1024, 1.0 / 32)
        for n in ("norm1", "norm2"):
            p[n + ".weight"] = 1.0 + 0.1 * torch.randn(256, generator=g, dtype=torch.float64)
            p[n + ".bias"] = 0.1 * torch.randn(256, generator=g, dtype=torch.float64)
        # the MSDeformAttn bias pattern of the offsets (one direction per head x point index) on top of the noise
        theta = torch.arange(8, dtype=torch.float64) * (2.0 * torch.pi / 8)
        grid = torch.stack([theta.cos(), theta.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(8, 1, 1, 2).repeat(1, 4, 4, 1)
        grid = grid * torch.arange(1, 5, dtype=torch.float64).view(1, 1, 4, 1)
        p["self_attn.sampling_offsets.bias"] = p["self_attn.sampling_offsets.bias"] + grid.reshape(-1)
        params.append({k: v.to(dtype) for k, v in p.items()})
    digest = float(sum(float(v.double().abs().sum()) for p in params for v in p.values()))
    shapes = torch.as_tensor(STACK_LEVELS, dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    return src.to(dtype), pos.to(dtype), ref.to(dtype), shapes, lsi, params, digest


def stack():
    """Six DISTINCT reference encoder layers back to back (what DeformableTransformerEncoder.forward does,
    deformable_transformer_dino.py:303-327, without the text-fusion branch), d_model 256 / 8 heads / d_ffn 1024, in
    fp64: the fixture that bounds how the split-bf16 projections of the HIP inference path compound over the stack."""
    cls = load_reference_layer()
    src, pos, ref, shapes, lsi, params, digest = stack_inputs()
    out = src
    with torch.no_grad():
        for p in params:
            layer = cls(d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4).double().eval()
            layer.load_state_dict(p)
            out = layer(out, pos, ref, shapes, lsi, None)
    np.savez_compressed(os.path.join(HERE, "encstack_6layers.npz"), out=out.numpy().astype(np.float32),
                        digest=np.float64(digest))
    print("encstack_6layers", tuple(out.shape), "scale", float(out.abs().max()), "digest", digest)


if __name__ == "__main__":
    main()
    stack()
