"""CPU tests of include/linear_hip.h: exported symbols, argument handling, sizes (no GPU work)."""
import os
import re

import numpy as np
import torch


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    from uninext_amd import _lib
    text = open(os.path.join(ROOT, "include", "linear_hip.h")).read()
    declared = set(re.findall(r"\b(linear_hip_\w+)\s*\(", text))
    assert declared == set(_lib.LINEAR_EXPORTS)
    lib = _lib.load()
    for sym in _lib.LINEAR_EXPORTS:
        assert hasattr(lib, sym)


def test_sizes_and_argument_errors_need_no_gpu():
    from uninext_amd import _lib
    lib = _lib.load()
    assert lib.linear_hip_packed_weight_bytes(256, 256) == 16 * 2 * 256 * 16 * 2
    assert lib.linear_hip_packed_weight_bytes(130, 64) == 4 * 2 * 256 * 16 * 2           # columns padded to 256
    assert lib.linear_hip_packed_weight_bytes(256, 100) == 0
    one = 16
    assert lib.linear_hip_packed_f32(one, one, None, None, 10, 100, 8, one, None) == -5
    assert "multiple of 64" in _lib.last_error()
    assert lib.linear_hip_packed_f32(one, one, None, None, -1, 64, 8, one, None) == -2
    assert lib.linear_hip_packed_f32(None, one, None, None, 10, 64, 8, one, None) == -1
    assert lib.linear_hip_packed_f32(None, None, None, None, 0, 64, 8, None, None) == 0   # no rows
    assert lib.linear_hip_pack_weight_f32(None, 8, 64, one, None) == -1


def test_oracle_matches_torch_linear():
    from oracle import linear_oracle
    rng = np.random.default_rng(0)
    x, w, b = rng.standard_normal((3, 7, 64)), rng.standard_normal((10, 64)), rng.standard_normal(10)
    mask = rng.random((3, 7)) < 0.3
    ref = torch.nn.functional.linear(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b))
    ref = ref.masked_fill(torch.from_numpy(mask)[..., None], 0.0)
    assert np.abs(linear_oracle.forward(x, w, b, mask) - ref.numpy()).max() < 1e-12


def test_module_projection_route_on_cpu():
    """Off the GPU the module runs nn.Linear + masked_fill exactly as the reference."""
    from uninext_amd.modules import MSDeformAttn
    layer = MSDeformAttn(256, 4, 8, 4)
    x = torch.randn(2, 5, 256)
    mask = torch.zeros(2, 5, dtype=torch.bool)
    mask[1, 3] = True
    y = layer._project(layer.value_proj, x, mask)
    assert torch.equal(y, layer.value_proj(x).masked_fill(mask[..., None], 0.0))


def test_ffn_argument_errors_and_support_predicate_need_no_gpu():
    from uninext_amd import _lib, ext
    lib = _lib.load()
    one = 16
    f = lambda rows, d_model, d_ffn, x=one, p1=one, p2=one, out=one: lib.linear_hip_packed_ffn_f32(
        x, p1, None, p2, None, None, None, None, 1e-5, 1, rows, d_model, d_ffn, out, None)
    assert f(10, 128, 1024) == -5 and "d_model must be 256" in _lib.last_error()
    assert f(10, 256, 192) == -5
    assert f(-1, 256, 1024) == -2
    assert f(10, 256, 1024, x=None) == -1
    assert f(10, 256, 1024, p2=None) == -1
    assert f(0, 256, 1024, x=None, p1=None, p2=None, out=None) == 0     # no rows: nothing is dereferenced
    x = torch.zeros(4, 256)
    w1, w2 = torch.zeros(1024, 256), torch.zeros(256, 1024)
    assert not ext.ffn_packed_supported(x, w1, w2, (256,))               # CPU tensors never take the kernel
    from uninext_amd.modules.encoder_layer import DeformableTransformerEncoderLayer
    layer = DeformableTransformerEncoderLayer().eval()
    with torch.no_grad():   # off the GPU the FFN block is the reference's composition
        src = torch.randn(1, 5, 256)
        want = layer.norm2(src + layer.linear2(torch.relu(layer.linear1(src))))
        assert layer.self_attn._ffn_norm(layer.linear1, layer.linear2, src, layer.norm2) is None
        assert torch.equal(layer.forward_ffn(src), want)
