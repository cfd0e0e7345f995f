"""CPU: the dynamic mask head against fixtures produced by the reference's own code
(tests/golden/make_dynmask_golden.py): the test-side oracle (reference algorithm restated) and the product's
differentiable PyTorch composition (used for training; a different, non-materialising formulation)."""
import numpy as np
import pytest
import torch

from golden_util import dynmask_names, load_golden
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
