"""CPU tests of the static mask head path: numpy oracle and the host-side MaskHeadSmallConv mirror against fixtures
minted by the reference class, plus the C ABI surface of include/conv3x3_hip.h (no GPU work)."""
import os
import re

import numpy as np
import pytest
import torch

from golden_util import load_golden, maskhead_names, max_abs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _params(g):
    return {k[2:]: v for k, v in g.items() if k.startswith("p:")}


def test_oracle_matches_reference_fixture():
    from oracle import conv3x3_oracle
    g = load_golden("maskhead_nofpn")
    out = conv3x3_oracle.mask_head_small_conv([g["x0"], g["x1"], g["x2"]], _params(g))
    assert out.shape == g["out"].shape and max_abs(out, g["out"]) < 1e-10


def test_oracle_conv_matches_torch():
    from oracle import conv3x3_oracle
    rng = np.random.default_rng(0)
    x, w, b = rng.standard_normal((2, 5, 7, 9)), rng.standard_normal((4, 5, 3, 3)), rng.standard_normal(4)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), padding=1)
    assert max_abs(conv3x3_oracle.conv3x3(x, w, b), ref.numpy()) < 1e-12
    assert max_abs(conv3x3_oracle.conv3x3(x, w, b, relu=True), ref.clamp(min=0).numpy()) < 1e-12


@pytest.mark.parametrize("name", maskhead_names())
def test_module_mirrors_reference_on_cpu(name):
    from uninext_amd.mask_head import MaskHeadSmallConv
    g = load_golden(name)
    params = _params(g)
    fpn_dims = [g["fpn%d" % i].shape[1] for i in range(3)] if "fpn0" in g else None
    head = MaskHeadSmallConv(g["x0"].shape[1], fpn_dims, g["x0"].shape[1]).double()
    assert sorted(head.state_dict()) == sorted(params)            # the reference's parameter names
    head.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    x = [torch.from_numpy(g["x%d" % i]) for i in range(3)]
    fpns = [torch.from_numpy(g["fpn%d" % i]) for i in range(3)] if fpn_dims else None
    out = head(x, fpns)
    assert out.shape == g["out"].shape and max_abs(out.detach().numpy(), g["out"]) < 1e-10


@pytest.mark.filterwarnings("ignore")
def test_initialisation_follows_reference():
    from uninext_amd.mask_head import MaskHeadSmallConv
    head = MaskHeadSmallConv(256, None, 256)
    assert all(float(m.bias.abs().max()) == 0.0 for m in head.modules() if isinstance(m, torch.nn.Conv2d))
    assert head.lay2.weight.shape == (8, 64, 3, 3) and head.jia_dcn.weight.shape == (256, 256, 3, 3)
    with pytest.raises(NotImplementedError):
        MaskHeadSmallConv(256, None, 256, use_raft=True)


def test_header_symbols_are_exported():
    from uninext_amd import _lib
    text = open(os.path.join(ROOT, "include", "conv3x3_hip.h")).read()
    declared = set(re.findall(r"\b((?:conv3x3|upsample_add)_hip_\w+)\s*\(", text))
    assert declared == set(_lib.CONV3X3_EXPORTS)
    lib = _lib.load()
    for sym in _lib.CONV3X3_EXPORTS:
        assert hasattr(lib, sym)


def test_argument_errors_need_no_gpu():
    from uninext_amd import _lib
    lib = _lib.load()
    one = 16
    assert lib.conv3x3_hip_f32(one, one, None, 1, 8, 4, 4, 8, 1, 0, one, None) == -5        # 72 % 16 != 0
    assert "multiple of 16" in _lib.last_error()
    assert lib.conv3x3_hip_f32(one, one, None, 1, 16, 0, 4, 8, 1, 0, one, None) == -2
    assert lib.conv3x3_hip_f32(None, one, None, 1, 16, 4, 4, 8, 1, 0, one, None) == -1
    assert lib.conv3x3_hip_f32(None, None, None, 0, 16, 4, 4, 8, 1, 1, None, None) == 0     # empty batch
    assert lib.conv3x3_hip_f32(one, one, None, 1, 16, 4, 4, 8, 1, 7, one, None) == -2        # unknown precision
