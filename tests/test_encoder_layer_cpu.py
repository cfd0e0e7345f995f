"""CPU tests of the encoder-layer mirror and of the C ABI surface it adds (include/layernorm_hip.h, the extended
Linear entry point): reference-minted fixtures in fp64, parameter names, argument handling.  No GPU work."""
import os
import re

import numpy as np
import pytest
import torch

from golden_util import enclayer_names, load_golden, max_abs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _layer_from(g):
    from uninext_amd.modules import DeformableTransformerEncoderLayer
    params = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("p:")}
    d_model, d_ffn = params["linear1.weight"].shape[1], params["linear1.weight"].shape[0]
    layer = DeformableTransformerEncoderLayer(d_model=d_model, d_ffn=d_ffn, dropout=0.1, activation="relu", n_levels=4,
                                              n_heads=d_model // 32, n_points=4)
    assert sorted(layer.state_dict()) == sorted(params)               # the reference's parameter names
    return layer, params


@pytest.mark.parametrize("name", enclayer_names())
def test_layer_mirrors_reference_on_cpu(name):
    """On the CPU the operator itself has no implementation (by design); swap in the grid_sample port for it and check
    everything around it against the reference layer's fp64 output."""
    from oracle.msda_gridsample import msda_gridsample
    from uninext_amd.modules import ms_deform_attn as mod
    g = load_golden(name)
    layer, params = _layer_from(g)
    layer = layer.double().eval()
    layer.load_state_dict(params)
    levels = [tuple(int(v) for v in hw) for hw in g["shapes"]]

    class CpuFunction:
        @staticmethod
        def apply(value, shapes, level_start, loc, attn, im2col_step):
            return msda_gridsample(value, levels, loc, attn)

    saved = mod.MSDeformAttnFunction
    mod.MSDeformAttnFunction = CpuFunction
    try:
        mask = torch.from_numpy(g["mask"]) if "mask" in g else None
        with torch.no_grad():
            out = layer(torch.from_numpy(g["src"]), torch.from_numpy(g["pos"]), torch.from_numpy(g["ref"]),
                        torch.from_numpy(g["shapes"]), torch.from_numpy(g["lsi"]), mask)
    finally:
        mod.MSDeformAttnFunction = saved
    assert max_abs(out.numpy(), g["out"]) < 1e-10


def test_header_symbols_are_exported():
    from uninext_amd import _lib
    text = open(os.path.join(ROOT, "include", "layernorm_hip.h")).read()
    assert set(re.findall(r"\b(add_layernorm_hip_\w+)\s*\(", text)) == set(_lib.LAYERNORM_EXPORTS)
    lib = _lib.load()
    for sym in _lib.LAYERNORM_EXPORTS:
        assert hasattr(lib, sym)


def test_argument_errors_need_no_gpu():
    from uninext_amd import _lib
    lib = _lib.load()
    one = 16
    assert lib.add_layernorm_hip_f32(one, None, None, None, 1e-5, 10, 6, one, None) == -5        # 6 % 4 != 0
    assert lib.add_layernorm_hip_f32(one, None, None, None, 1e-5, 10, 8192, one, None) == -5
    assert lib.add_layernorm_hip_f32(one, None, None, None, 1e-5, -1, 256, one, None) == -2
    assert lib.add_layernorm_hip_f32(None, None, None, None, 1e-5, 10, 256, one, None) == -1
    assert lib.add_layernorm_hip_f32(None, None, None, None, 1e-5, 0, 256, None, None) == 0
    assert lib.linear_hip_packed_ex_f32(one, None, one, None, None, 10, 64, 8, 2, one, None) == -2   # unknown activation
    assert "activation" in _lib.last_error()


def test_unknown_activation_is_rejected():
    from uninext_amd.modules import DeformableTransformerEncoderLayer
    with pytest.raises(RuntimeError, match="activation"):
        DeformableTransformerEncoderLayer(activation="swish")
