"""GPU: the HIP dynamic mask head (include/dynmask_hip.h) against the reference-minted fixtures and, at the
R50 COCO size, against the test-side oracle."""
import numpy as np
import pytest
import torch

from golden_util import dynmask_names, load_golden
from test_dynmask_cpu import _case
from uninext_amd import ext, mask_head

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name", dynmask_names())
def test_hip_path_matches_reference(name):
    g, feats, ref, params, num_insts, rel, mos = _case(name, DEV)
    with torch.no_grad():
        out = mask_head.dynamic_mask_with_coords(feats, ref, params, num_insts, 8, rel_coord=rel, mask_out_stride=mos)
    assert out.shape == g["out"].shape
    assert float(np.abs(out.cpu().numpy() - g["out"]).max()) < 1e-4 * max(1.0, float(np.abs(g["out"]).max()))


@pytest.mark.parametrize("variant", [2, 3])
@pytest.mark.parametrize("rel", [True, False])
def test_mfma_variants_equal_the_packed_fma_kernel(variant, rel):
    """The MFMA form (include/dynmask_hip.h: dynmask_hip_set_variant) is the same fmaf chain: bitwise equal, with and
    without relative coordinates, on an image whose pixel count is not a multiple of the chunk (tail lanes)."""
    from uninext_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    H, W = 37, 53
    num_insts = [5, 0, 9]
    n_all = sum(num_insts)
    feats = torch.randn(3, 8, H, W, generator=g).to(DEV)
    xy = (torch.rand(n_all, 2, generator=g) * torch.tensor([W * 8.0, H * 8.0])).to(DEV)
    params = (torch.randn(n_all, 169 if rel else 153, generator=g) * 0.3).to(DEV)
    try:
        assert lib.dynmask_hip_set_variant(1) == 0
        base = ext.dynmask_forward(feats, xy, params, num_insts, 8, rel)
        assert lib.dynmask_hip_last_kernel().decode() == "dynmask_fwd_pkfma"
        assert lib.dynmask_hip_set_variant(variant) == 0
        out = ext.dynmask_forward(feats, xy, params, num_insts, 8, rel)
        assert lib.dynmask_hip_last_kernel().decode() == "dynmask_fwd_mfma_q%d" % (2 if variant == 2 else 4)
        assert torch.equal(out, base)
        assert lib.dynmask_hip_set_variant(99) != 0
    finally:
        lib.dynmask_hip_set_variant(0)


@pytest.mark.parametrize("factor", [1, 2, 3, 4])
def test_hip_aligned_bilinear(factor):
    g = load_golden("dynmask_aligned_bilinear")
    x = torch.from_numpy(g["x"]).to(DEV)
    out = ext.aligned_bilinear_forward(x, factor) if factor > 1 else mask_head.aligned_bilinear(x, factor)
    assert np.allclose(out.cpu().numpy(), g[f"f{factor}"], atol=1e-5)


def test_hip_path_at_r50_size_vs_oracle_and_torch_path():
    """bs 2, 100x167 mask features, 300 + 157 instances: HIP vs the materialising oracle (sub-sampled instances) and
    vs the product's differentiable composition (all instances)."""
    from oracle.dynmask_torch import dynamic_mask_oracle
    g = torch.Generator().manual_seed(3)
    N, H, W = 2, 100, 167
    num_insts = [300, 157]
    n_all = sum(num_insts)
    feats = torch.randn(N, 8, H, W, generator=g).to(DEV)
    ref = (torch.rand(1, n_all, 2, generator=g) * torch.tensor([W * 8.0, H * 8.0])).to(DEV)
    params = (torch.randn(1, n_all, 169, generator=g) * 0.3).to(DEV)
    with torch.no_grad():
        hip = mask_head.dynamic_mask_with_coords(feats, ref, params, num_insts, 8)
        assert hip.shape == (1, n_all, 2 * H, 2 * W)
        tor = mask_head._aligned_bilinear_torch(
            mask_head._dynamic_convs_torch(feats, ref.reshape(-1, 2), params.flatten(0, 1), num_insts, 8, True)
            .reshape(-1, 1, H, W), 2).reshape(1, n_all, 2 * H, 2 * W)
        scale = float(tor.abs().max())
        assert float((hip - tor).abs().max()) < 1e-4 * scale
        sub = [0, 1, 299, 300, 456]
        counts = [3, 2]
        ora = dynamic_mask_oracle(feats, ref[:, sub], params[:, sub], counts, 8)
        assert float((hip[:, sub] - ora).abs().max()) < 1e-4 * scale


def test_unsupported_geometry_falls_back_to_torch():
    feats = torch.randn(1, 4, 6, 7, device=DEV)           # 4 feature channels: no HIP kernel
    ref = torch.rand(1, 3, 2, device=DEV) * 40
    params = torch.randn(1, 3, (4 + 2) * 8 + 64 + 8 + 8 + 8 + 1, device=DEV)
    with torch.no_grad():
        out = mask_head.dynamic_mask_with_coords(feats, ref, params, [3], 8)
    assert out.shape == (1, 3, 12, 14) and torch.isfinite(out).all()
    with pytest.raises(RuntimeError, match="8 mask-feature channels|inconsistent"):
        ext.dynmask_forward(feats, ref.reshape(-1, 2), params.flatten(0, 1), [3], 8)
