"""CPU: the dynamic mask head against fixtures produced by the reference's own code
(tests/golden/make_dynmask_golden.py): the test-side oracle (reference algorithm restated) and the product's
differentiable PyTorch composition (used for training; a different, non-materialising formulation)."""
import numpy as np
import pytest
import torch

from golden_util import dynmask_bwd_names, dynmask_names, load_golden
from oracle.dynmask_torch import dynamic_mask_oracle, upsample_aligned
from uninext_amd import mask_head

NAMES = dynmask_names()


def _case(name, device="cpu"):
    g = load_golden(name)
    t = lambda k: torch.from_numpy(g[k]).to(device)
    return g, t("mask_feats"), t("reference_points"), t("mask_head_params"), g["num_insts"].tolist(), bool(g["rel_coord"]), int(g["mask_out_stride"])


def test_fixtures_present():
    assert len(NAMES) >= 4


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference(name):
    g, feats, ref, params, num_insts, rel, mos = _case(name)
    out = dynamic_mask_oracle(feats, ref, params, num_insts, 8, rel_coord=rel, mask_out_stride=mos)
    assert out.shape == g["out"].shape
    assert float(np.abs(out.numpy() - g["out"]).max()) < 1e-5 * max(1.0, float(np.abs(g["out"]).max()))


@pytest.mark.parametrize("name", NAMES)
def test_product_torch_path_matches_reference(name):
    g, feats, ref, params, num_insts, rel, mos = _case(name)
    out = mask_head.dynamic_mask_with_coords(feats, ref, params, num_insts, 8, rel_coord=rel, mask_out_stride=mos)
    assert out.shape == g["out"].shape
    assert float(np.abs(out.numpy() - g["out"]).max()) < 1e-4 * max(1.0, float(np.abs(g["out"]).max()))


def test_product_torch_path_is_differentiable():
    g, feats, ref, params, num_insts, rel, mos = _case("dynmask_rel_up2")
    params = params.clone().requires_grad_(True)
    feats = feats.clone().requires_grad_(True)
    out = mask_head.dynamic_mask_with_coords(feats, ref, params, num_insts, 8, rel_coord=rel, mask_out_stride=mos)
    out.square().mean().backward()
    assert torch.isfinite(params.grad).all() and torch.isfinite(feats.grad).all() and params.grad.abs().sum() > 0


@pytest.mark.parametrize("factor", [1, 2, 3, 4])
def test_aligned_bilinear_matches_reference(factor):
    g = load_golden("dynmask_aligned_bilinear")
    x = torch.from_numpy(g["x"])
    assert np.allclose(upsample_aligned(x, factor).numpy(), g[f"f{factor}"], atol=1e-6)
    assert np.allclose(mask_head.aligned_bilinear(x, factor).numpy(), g[f"f{factor}"], atol=1e-6)


def test_no_instances_returns_empty():
    out = mask_head.dynamic_mask_with_coords(torch.zeros(1, 8, 4, 5), torch.zeros(1, 0, 2), torch.zeros(1, 0, 169), [0], 8)
    assert out.shape == (1, 0, 4, 5)


# ---- gradients (round 5: the head under autograd, include/dynmask_hip.h backward entry points) ---------------------------------
def _grads(out_fn, g, device="cpu", dtype=torch.float32):
    t = lambda k: torch.from_numpy(g[k]).to(device=device, dtype=dtype)
    feats, ref, params = (t(k).requires_grad_(True) for k in ("mask_feats", "reference_points", "mask_head_params"))
    out = out_fn(feats, ref, params, g["num_insts"].tolist(), 8, rel_coord=bool(g["rel_coord"]), mask_out_stride=int(g["mask_out_stride"]))
    gf, gr, gp = torch.autograd.grad((out * t("upstream")).sum(), (feats, ref, params), allow_unused=True)
    return out, gf, (gr if gr is not None else torch.zeros_like(ref)), gp


@pytest.mark.parametrize("name", dynmask_bwd_names())
def test_composition_gradients_match_the_reference_under_autograd(name):
    """The product's differentiable composition (the CPU / fallback route, and the GPU kernels' second checker) against the
    gradients the REFERENCE's code produced under autograd in float64 (tests/golden/make_dynmask_bwd_golden.py)."""
    g = load_golden(name)
    out, gf, gr, gp = _grads(mask_head.dynamic_mask_with_coords, g, dtype=torch.float64)
    assert float(np.abs(out.detach().numpy() - g["out"]).max()) < 1e-9 * max(1.0, float(np.abs(g["out"]).max()))
    for got, key in ((gf, "grad_mask_feats"), (gr, "grad_reference_points"), (gp, "grad_mask_head_params")):
        want = g[key]
        assert got.shape == want.shape
        assert float(np.abs(got.numpy() - want).max()) < 1e-9 * max(1.0, float(np.abs(want).max())), key
    # float32, as trained: within float32 accumulation error of the float64 truth
    _, gf, gr, gp = _grads(mask_head.dynamic_mask_with_coords, g)
    for got, key in ((gf, "grad_mask_feats"), (gr, "grad_reference_points"), (gp, "grad_mask_head_params")):
        assert float(np.abs(got.numpy() - g[key]).max()) < 2e-5 * max(1.0, float(np.abs(g[key]).max())), key


def test_composition_passes_float64_gradcheck():
    g = torch.Generator().manual_seed(21)
    feats = torch.randn(2, 8, 3, 4, generator=g, dtype=torch.float64, requires_grad=True)
    # (the relative coordinates pass through `.float()` as in the reference, ddetrs_dn.py:783: a finite difference in the
    # reference points is below float32 resolution -- their analytic gradient is held to the reference's by the fixtures above)
    ref = torch.rand(1, 3, 2, generator=g, dtype=torch.float64) * 30
    params = (torch.randn(1, 3, 169, generator=g, dtype=torch.float64) * 0.3).requires_grad_(True)
    fn = lambda f, p: mask_head.dynamic_mask_with_coords(f, ref, p, [2, 1], 8, rel_coord=True, mask_out_stride=4)
    assert torch.autograd.gradcheck(fn, (feats, params), eps=1e-6, atol=1e-5, rtol=1e-4, nondet_tol=0.0)


@pytest.mark.parametrize("factor", [2, 3, 4])
def test_aligned_bilinear_gradient_matches_reference(factor):
    g = load_golden("dynmask_bwd_aligned_bilinear")
    x = torch.from_numpy(g["x"]).double().requires_grad_(True)
    (gx,) = torch.autograd.grad((mask_head.aligned_bilinear(x, factor) * torch.from_numpy(g[f"up{factor}"]).double()).sum(), (x,))
    assert np.allclose(gx.numpy(), g[f"g{factor}"], atol=1e-12)
