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


# ---- the head under autograd (round 5; include/dynmask_hip.h: dynmask_hip_backward_f32, aligned_bilinear_hip_backward_f32) -----
def _hip_grads(g, up_key="upstream"):
    t = lambda k: torch.from_numpy(g[k]).float().to(DEV)
    feats, ref, params = (t(k).requires_grad_(True) for k in ("mask_feats", "reference_points", "mask_head_params"))
    out = mask_head.dynamic_mask_with_coords(feats, ref, params, g["num_insts"].tolist(), 8, rel_coord=bool(g["rel_coord"]),
                                             mask_out_stride=int(g["mask_out_stride"]))
    gf, gr, gp = torch.autograd.grad((out * t(up_key)).sum(), (feats, ref, params), allow_unused=True)
    return out, gf, (gr if gr is not None else torch.zeros_like(ref)), gp


@pytest.mark.parametrize("name", __import__("golden_util").dynmask_bwd_names())
def test_hip_gradients_match_the_reference_under_autograd(name):
    """DynMaskFunction + AlignedBilinearFunction against the gradients the reference's own code produced under autograd in
    float64: within float32 accumulation error, relative to the largest gradient of each tensor; bitwise repeatable."""
    from golden_util import load_golden
    g = load_golden(name)
    out, gf, gr, gp = _hip_grads(g)
    assert out.grad_fn is not None and type(out.grad_fn).__name__ != "CatBackward0"
    assert float(np.abs(out.detach().cpu().numpy() - g["out"]).max()) < 1e-5 * max(1.0, float(np.abs(g["out"]).max()))
    for got, key in ((gf, "grad_mask_feats"), (gr, "grad_reference_points"), (gp, "grad_mask_head_params")):
        want = g[key]
        assert tuple(got.shape) == want.shape
        err = float(np.abs(got.cpu().numpy() - want).max())
        print("%s %s: max |err| %.2e of max |grad| %.2e" % (name, key, err, float(np.abs(want).max())))
        assert err < 2e-5 * max(1.0, float(np.abs(want).max())), key
    _, gf2, gr2, gp2 = _hip_grads(g)
    assert torch.equal(gf, gf2) and torch.equal(gr, gr2) and torch.equal(gp, gp2)        # no float atomics anywhere


@pytest.mark.parametrize("factor", [2, 3, 4])
def test_hip_aligned_bilinear_gradient(factor):
    from golden_util import load_golden
    g = load_golden("dynmask_bwd_aligned_bilinear")
    x = torch.from_numpy(g["x"]).to(DEV).requires_grad_(True)
    y = mask_head.aligned_bilinear(x, factor)
    assert type(y.grad_fn).__name__.startswith("AlignedBilinearFunction")
    (gx,) = torch.autograd.grad((y * torch.from_numpy(g[f"up{factor}"]).to(DEV)).sum(), (x,))
    assert np.allclose(gx.cpu().numpy(), g[f"g{factor}"], atol=2e-6)


def test_hip_gradients_at_the_training_shape_vs_the_composition():
    """BASELINE configs[4]: bs 2 padded to 800 x 1344 (mask features 100 x 168), 7 + 19 matched instances, x2 up-sampling: the
    HIP Functions against the PyTorch composition's autograd (same float32 data flow), with time and peak memory of both."""
    import time
    g = torch.Generator().manual_seed(31)
    N, H, W, counts = 2, 100, 168, [7, 19]
    n_all = sum(counts)
    feats = torch.randn(N, 8, H, W, generator=g).to(DEV)
    ref = (torch.rand(1, n_all, 2, generator=g) * torch.tensor([W * 8.0, H * 8.0])).to(DEV)
    params = (torch.randn(1, n_all, 169, generator=g) * 0.3).to(DEV)
    up = torch.randn(1, n_all, 2 * H, 2 * W, generator=g).to(DEV)

    def run(hip):
        f, r, p = (t.clone().requires_grad_(True) for t in (feats, ref, params))
        if hip:
            out = mask_head.dynamic_mask_with_coords(f, r, p, counts, 8)
        else:
            logits = mask_head._dynamic_convs_torch(f, r.reshape(-1, 2), p.flatten(0, 1), counts, 8, True).reshape(-1, 1, H, W)
            out = mask_head._aligned_bilinear_torch(logits, 2).reshape(1, n_all, 2 * H, 2 * W)
        return (out,) + torch.autograd.grad((out * up).sum(), (f, r, p))

    res = {}
    for hip in (True, False):
        run(hip)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        t0 = time.perf_counter()
        for _ in range(10):
            o = run(hip)
        torch.cuda.synchronize()
        res[hip] = (o, (time.perf_counter() - t0) / 10 * 1e3, (torch.cuda.max_memory_allocated() - base) / 2 ** 20)
    print("dynamic mask head forward + backward, bs 2, 100 x 168, %d instances: HIP %.3f ms / %.1f MiB peak, PyTorch composition "
          "%.3f ms / %.1f MiB peak" % (n_all, res[True][1], res[True][2], res[False][1], res[False][2]))
    for a, b, name in zip(res[True][0], res[False][0], ("out", "grad_feats", "grad_ref", "grad_params")):
        scale = max(1.0, float(b.abs().max()))
        assert float((a - b).abs().max()) < 1e-4 * scale, name
    assert res[True][2] < res[False][2]


def test_hip_gradients_with_an_image_without_instances_and_detached_inputs():
    g = torch.Generator().manual_seed(32)
    feats = torch.randn(3, 8, 9, 14, generator=g).to(DEV).requires_grad_(True)
    ref = (torch.rand(1, 5, 2, generator=g) * 60).to(DEV)                          # detached, as ddetrs_dn.py:391 hands them over
    params = (torch.randn(1, 5, 169, generator=g) * 0.3).to(DEV).requires_grad_(True)
    out = mask_head.dynamic_mask_with_coords(feats, ref, params, [2, 0, 3], 8)
    out.square().sum().backward()
    assert torch.isfinite(feats.grad).all() and float(feats.grad[1].abs().max()) == 0.0 and float(feats.grad[0].abs().max()) > 0
    assert torch.isfinite(params.grad).all() and ref.grad is None


def test_hip_gradients_of_one_input_alone_equal_those_of_the_full_backward():
    """Round 6: frozen mask features or detached parameters -- the backward launches only the kernel family the wanted
    gradient needs (NULL grad_feats / grad_params in the C ABI), and what it returns is bitwise what the full backward returns."""
    g = torch.Generator().manual_seed(33)
    feats0 = torch.randn(2, 8, 12, 20, generator=g).to(DEV)
    ref0 = (torch.rand(1, 6, 2, generator=g) * 90).to(DEV)
    params0 = (torch.randn(1, 6, 169, generator=g) * 0.3).to(DEV)

    def run(need):
        t = [x.clone().requires_grad_(n) for x, n in zip((feats0, ref0, params0), need)]
        out = mask_head.dynamic_mask_with_coords(t[0], t[1], t[2], [4, 2], 8)
        out.square().sum().backward()
        return [x.grad for x in t]

    full = run((True, True, True))
    for need in ((True, False, False), (False, False, True), (False, True, False), (True, False, True)):
        got = run(need)
        for a, b, n in zip(got, full, need):
            assert (a is None) == (not n)
            if n:
                assert torch.equal(a, b), need
