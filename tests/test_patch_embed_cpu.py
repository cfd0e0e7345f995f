"""CPU tests of the patch-embedding path: the numpy oracle against the reference-minted fixtures, the C ABI surface,
and the host-side layer (no GPU: only argument handling and the PyTorch route are exercised)."""
import os
import re

import numpy as np
import pytest
import torch

from golden_util import load_golden, max_abs, patch_names

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", patch_names())
def test_oracle_matches_reference_fixture(name):
    from oracle import patch_embed_oracle
    g = load_golden(name)
    out = patch_embed_oracle.forward(g["x"], g["weight"], g["bias"], bool(int(g["channels_last"])))
    assert out.shape == g["out"].shape
    assert max_abs(out, g["out"]) < 1e-11     # fp64 vs fp64: summation order only


def test_fixture_set_is_complete():
    assert set(patch_names()) == {"patch_vit_small", "patch_vit_tiles", "patch_convnext_stem", "patch_convnext_down"}


def test_header_symbols_are_exported():
    from uninext_amd import _lib
    text = open(os.path.join(ROOT, "include", "patch_embed_hip.h")).read()
    declared = set(re.findall(r"\b(patch_embed_hip_\w+)\s*\(", text))
    assert declared == set(_lib.PATCH_EMBED_EXPORTS)
    lib = _lib.load()                           # loading needs no GPU
    for sym in _lib.PATCH_EMBED_EXPORTS:
        assert hasattr(lib, sym)


def test_argument_errors_need_no_gpu():
    """Dimension / support checks come before anything touches the device."""
    from uninext_amd import _lib
    lib = _lib.load()
    one = 16   # dummy non-null pointer value; rejected calls never dereference
    assert lib.patch_embed_hip_f32(one, one, None, 1, 3, 32, 32, 8, 3, 1, one, None) == -5       # patch 3
    assert "patch size" in _lib.last_error()
    assert lib.patch_embed_hip_f32(one, one, None, 1, 1, 32, 32, 8, 2, 1, one, None) == -5       # K = 4
    assert lib.patch_embed_hip_f32(one, one, None, 1, 3, 0, 32, 8, 16, 1, one, None) == -2
    assert lib.patch_embed_hip_f32(None, one, None, 1, 3, 32, 32, 8, 16, 1, one, None) == -1
    assert lib.patch_embed_hip_f32(one, one, None, 1, 3, 8, 32, 8, 16, 1, one, None) == 0        # no patch fits: no-op
    assert lib.patch_embed_hip_f32(None, None, None, 0, 3, 32, 32, 8, 16, 1, None, None) == 0    # empty batch


def test_layer_mirrors_reference_on_cpu():
    """PatchEmbed keeps the reference's parameter names and, off the GPU, its PyTorch arithmetic."""
    from uninext_amd.backbone import PatchEmbed, patch_conv2d
    g = load_golden("patch_vit_small")
    pe = PatchEmbed(kernel_size=(16, 16), stride=(16, 16), padding=(0, 0), in_chans=3, embed_dim=g["weight"].shape[0])
    assert sorted(pe.state_dict()) == ["proj.bias", "proj.weight"]
    pe = pe.double()
    pe.load_state_dict({"proj.weight": torch.from_numpy(g["weight"]), "proj.bias": torch.from_numpy(g["bias"])})
    out = pe(torch.from_numpy(g["x"]))
    assert out.shape == g["out"].shape and max_abs(out.detach().numpy(), g["out"]) < 1e-11
    g = load_golden("patch_convnext_stem")
    conv = torch.nn.Conv2d(3, g["weight"].shape[0], kernel_size=4, stride=4).double()
    conv.load_state_dict({"weight": torch.from_numpy(g["weight"]), "bias": torch.from_numpy(g["bias"])})
    out = patch_conv2d(torch.from_numpy(g["x"]), conv)
    assert max_abs(out.detach().numpy(), g["out"]) < 1e-11


def test_packed_weight_sizes_need_no_gpu():
    from uninext_amd import _lib
    lib = _lib.load()
    assert lib.patch_embed_hip_packed_weight_bytes(1280, 3, 16) == (768 // 16) * 2 * 1280 * 16 * 2
    assert lib.patch_embed_hip_packed_weight_bytes(192, 3, 4) == 3 * 2 * 256 * 16 * 2        # E padded to 256
    assert lib.patch_embed_hip_packed_weight_bytes(64, 5, 4) == 0                             # K = 80
    assert lib.patch_embed_hip_packed_weight_bytes(64, 3, 3) == 0
    one = 16
    assert lib.patch_embed_hip_packed_f32(one, one, None, 1, 5, 32, 32, 8, 4, 1, one, None) == -5
    assert lib.patch_embed_hip_pack_weight_f32(None, 8, 3, 16, one, None) == -1


def test_supported_predicate():
    from uninext_amd import ext
    x = torch.zeros(1, 3, 32, 32)
    w = torch.zeros(8, 3, 16, 16)
    assert not ext.patch_embed_supported(x, w, (16, 16), (0, 0))          # CPU tensor
