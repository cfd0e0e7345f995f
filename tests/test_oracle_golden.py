"""CPU: pin oracle/msda_oracle.c (and the grid_sample port) to the reference-minted fixtures.

Fixtures in tests/golden/*.npz come from the reference's own ms_deform_attn_core_pytorch
(ops/functions/ms_deform_attn_func.py:43-63) in float64 + autograd (tests/golden/make_golden.py).
Mirrors ops/test.py:31-60 (forward fp64 / fp32 checks) and :63-78 (gradient checks).
"""
import numpy as np
import pytest
import torch

from golden_util import golden_names, load_golden, max_abs, scaled_err
from oracle import msda_gridsample, msda_oracle

NAMES = golden_names()


def test_fixtures_present():
    assert "testpy_seed3" in NAMES and len(NAMES) >= 8


@pytest.mark.parametrize("name", NAMES)
def test_oracle_forward_f64(name):
    g = load_golden(name)
    out = msda_oracle.forward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"])
    assert out.dtype == np.float64 and out.shape == g["out"].shape
    assert max_abs(out, g["out"]) < 1e-12


@pytest.mark.parametrize("name", NAMES)
def test_oracle_forward_f32(name):
    g = load_golden(name)
    out = msda_oracle.forward(g["value"].astype(np.float32), g["shapes"], g["lsi"],
                              g["loc"].astype(np.float32), g["attn"].astype(np.float32))
    assert out.dtype == np.float32
    # north_star tolerance: 1e-4 abs in fp32 (the reference's own fp32 check is rtol 1e-2 / atol 1e-3, ops/test.py:56)
    assert max_abs(out, g["out"]) < 1e-4


@pytest.mark.parametrize("name", NAMES)
def test_oracle_backward_f64(name):
    g = load_golden(name)
    gv, gl, ga = msda_oracle.backward(g["grad_out"], g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"])
    assert max_abs(gv, g["grad_value"]) < 1e-11
    assert max_abs(ga, g["grad_attn"]) < 1e-11
    # grid_sample's location gradient and the CUDA formula agree except exactly on a cell edge,
    # where the one-sided derivative is a convention; the border fixture is built on such edges.
    if name != "border":
        assert max_abs(gl, g["grad_loc"]) < 1e-9


@pytest.mark.parametrize("name", NAMES)
def test_oracle_backward_f32(name):
    g = load_golden(name)
    f = np.float32
    gv, gl, ga = msda_oracle.backward(g["grad_out"].astype(f), g["value"].astype(f), g["shapes"], g["lsi"],
                                      g["loc"].astype(f), g["attn"].astype(f))
    assert scaled_err(gv, g["grad_value"]) < 1e-4
    assert scaled_err(ga, g["grad_attn"]) < 1e-4
    if name != "border":
        assert scaled_err(gl, g["grad_loc"]) < 1e-3  # fp32 loc*W-0.5 rounding moves lw/lh by ~1e-6*W


@pytest.mark.parametrize("name", NAMES)
def test_gridsample_port_matches_golden(name):
    g = load_golden(name)
    t = lambda k: torch.from_numpy(g[k])
    out = msda_gridsample.msda_gridsample(t("value"), g["shapes"].tolist(), t("loc"), t("attn"))
    assert max_abs(out.numpy(), g["out"]) < 1e-12


def test_oracle_testpy_distribution_many_seeds():
    """ops/test.py:31-44 repeated over seeds: C oracle vs grid_sample port, fp64."""
    shapes = np.array([(6, 4), (3, 2)], dtype=np.int64)
    lsi = np.array([0, 24], dtype=np.int64)
    for seed in range(20):
        torch.manual_seed(seed)
        value = (torch.rand(1, 30, 2, 2) * 0.01).double()
        loc = torch.rand(1, 2, 2, 2, 2, 2).double()
        attn = (torch.rand(1, 2, 2, 2, 2) + 1e-5).double()
        attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
        ref = msda_gridsample.msda_gridsample(value, shapes.tolist(), loc, attn).numpy()
        out = msda_oracle.forward(value, shapes, lsi, loc, attn)
        assert np.allclose(out, ref)  # default allclose, as ops/test.py:40


def test_oracle_gradcheck_against_autograd_random():
    """Backward oracle vs autograd through the grid_sample port at a non-fixture shape."""
    g = torch.Generator().manual_seed(5)
    shapes = [(5, 7), (3, 4), (2, 2)]
    S = sum(h * w for h, w in shapes)
    N, M, D, Lq, L, P = 2, 4, 12, 9, 3, 3
    value = torch.randn(N, S, M, D, generator=g, dtype=torch.float64, requires_grad=True)
    loc = (torch.rand(N, Lq, M, L, P, 2, generator=g, dtype=torch.float64) * 1.2 - 0.1).requires_grad_(True)
    attn = torch.rand(N, Lq, M, L, P, generator=g, dtype=torch.float64, requires_grad=True)
    out = msda_gridsample.msda_gridsample(value, shapes, loc, attn)
    go = torch.randn(out.shape, generator=g, dtype=torch.float64)
    gv, gl, ga = torch.autograd.grad(out, (value, loc, attn), go)
    sh = np.array(shapes, dtype=np.int64)
    lsi = np.concatenate(([0], np.cumsum(sh.prod(1))[:-1])).astype(np.int64)
    ogv, ogl, oga = msda_oracle.backward(go, value, sh, lsi, loc, attn)
    assert max_abs(ogv, gv.numpy()) < 1e-11
    assert max_abs(ogl, gl.numpy()) < 1e-9
    assert max_abs(oga, ga.numpy()) < 1e-11


def test_oracle_on_the_encoder_shaped_reference_fixture():
    """tests/golden/encshape_s1065_m2.npz (make_golden.py: mint_encoder_shaped): an encoder-shaped call with model-like
    locations, run through the reference in float64 and stored as float32.  The float64 oracle on the stored float32
    inputs reproduces it to float32 round-off of the stored values."""
    from golden_util import load_golden
    from oracle import msda_oracle
    import torch
    g = load_golden("encshape_s1065_m2")
    t = lambda k: torch.from_numpy(g[k]).double()
    shapes, lsi = torch.from_numpy(g["shapes"]), torch.from_numpy(g["lsi"])
    out = msda_oracle.forward(t("value"), shapes, lsi, t("loc"), t("attn"))
    assert float(np.abs(out - g["out"]).max()) < 1e-6
    gv, gl, ga = msda_oracle.backward(t("grad_out"), t("value"), shapes, lsi, t("loc"), t("attn"))
    assert float(np.abs(gv - g["grad_value"]).max()) < 1e-5 * max(1.0, float(np.abs(g["grad_value"]).max()))
    assert float(np.abs(ga - g["grad_attn"]).max()) < 1e-5 * max(1.0, float(np.abs(g["grad_attn"]).max()))
    # grad_loc: the oracle follows the CUDA formula, the fixture grid_sample's autograd -- they agree away from cell
    # edges (module docstring); model-like locations are generic, so a handful of edge cases at most
    d = np.abs(gl - g["grad_loc"])
    assert float(np.quantile(d, 0.999)) < 1e-4 * max(1.0, float(np.abs(g["grad_loc"]).max()))
