"""MI355X: out of the box the caller-side modules compute in the reference's fp32; split-bf16 is opt-in.

Both arithmetic routes of MSDeformAttn's projections, the static mask head's 3x3 convolutions and the patch embedding are
held against float64 on operands at TRAINED-CHECKPOINT scales -- heavy-tailed weights with |w| up to ~5 and activations up
to ~1e2, not the unit-scale random-init data of the other test files: the default route to fp32 accumulation error, the
opt-in route to 1e-4 of the output scale (north_star's bound)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def heavy(shape, sigma, peak, frac, gen):
    """N(0, sigma) with a fraction `frac` of outliers of magnitude up to `peak` (both signs)."""
    t = torch.randn(shape, generator=gen) * sigma
    out = torch.rand(shape, generator=gen) < frac
    mag = peak * (0.3 + 0.7 * torch.rand(shape, generator=gen)) * torch.sign(torch.randn(shape, generator=gen))
    return torch.where(out, mag, t)


def test_defaults_are_the_exact_paths(monkeypatch):
    import importlib
    import os
    assert os.environ.get("UNINEXT_AMD_SPLIT_BF16", "0") != "1", "run this test without the opt-in in the environment"
    from uninext_amd.backbone import PatchEmbed
    from uninext_amd.mask_head import MaskHeadSmallConv
    from uninext_amd.modules import MSDeformAttn
    assert MSDeformAttn.fast_linear is False and MaskHeadSmallConv.exact_fp32 is True and PatchEmbed.exact_fp32 is True


@pytest.mark.parametrize("fast", [False, True])
def test_msdeformattn_layer_at_trained_scales(fast, dev):
    from uninext_amd import _lib, workloads
    from uninext_amd.modules import MSDeformAttn
    gen = torch.Generator().manual_seed(5)
    levels = ((40, 53), (20, 27), (10, 14), (5, 7))          # S = 2835: the encoder-sized kernels
    S = sum(h * w for h, w in levels)
    layer = MSDeformAttn(256, 4, 8, 4).eval()
    with torch.no_grad():
        for lin, sig in ((layer.value_proj, 0.08), (layer.output_proj, 0.08), (layer.attention_weights, 0.05)):
            lin.weight.copy_(heavy(lin.weight.shape, sig, 5.0, 0.002, gen))
            lin.bias.copy_(heavy(lin.bias.shape, 0.3, 3.0, 0.02, gen))
        layer.sampling_offsets.weight.copy_(torch.randn(layer.sampling_offsets.weight.shape, generator=gen) * 0.01)
    src = heavy((2, S, 256), 1.0, 100.0, 0.001, gen)
    ref_pts = workloads.encoder_reference_points(levels, "cpu")[None, :, None, :].expand(2, S, 4, 2).contiguous()
    sh, lsi = workloads.level_tensors(levels, "cpu")
    # float64 yardstick: the same module in double on the CPU (host variants of the operator)
    want = layer.double()(src.double(), ref_pts.double(), src.double(), sh, lsi, None).detach()
    layer = layer.float().to(dev)
    old = MSDeformAttn.fast_linear
    MSDeformAttn.fast_linear = fast
    try:
        with torch.no_grad():
            got = layer(src.to(dev), ref_pts.to(dev), src.to(dev), sh.to(dev), lsi.to(dev), None)
    finally:
        MSDeformAttn.fast_linear = old
    assert ("_msda_packed" in layer.value_proj.__dict__) == fast          # the default route never packs bf16 weights
    scale = float(want.abs().max())
    err = float((got.double().cpu() - want).abs().max())
    print("MSDeformAttn %s: max |err| %.3e = %.2e of the output scale %.1f" % ("split-bf16" if fast else "fp32 (default)", err, err / scale, scale))
    # the sampling LOCATIONS depend on the projected offsets: a perturbed offset moves a bilinear sample, so the bound is
    # on the output scale for both routes; fp32 GEMM accumulation sits two orders below the split's
    assert err < (1e-4 if fast else 2e-5) * scale, (err, scale)


@pytest.mark.parametrize("exact", [True, False])
def test_mask_head_at_trained_scales(exact, dev):
    from uninext_amd.mask_head import MaskHeadSmallConv
    gen = torch.Generator().manual_seed(6)
    head = MaskHeadSmallConv(256, None, 256).eval()
    with torch.no_grad():
        for m in head.modules():
            if isinstance(m, torch.nn.Conv2d):
                fan = m.weight.shape[1] * 9
                m.weight.copy_(heavy(m.weight.shape, 1.5 / fan ** 0.5, 5.0 / fan ** 0.5 * 8, 0.002, gen))
                m.bias.copy_(heavy(m.bias.shape, 0.2, 2.0, 0.02, gen))
    x = [heavy((1, 256, h, w), 1.0, 100.0, 0.0005, gen) for h, w in ((50, 84), (25, 42), (13, 21))]
    with torch.no_grad():
        want = head.double()([t.double() for t in x], None)
    head = head.float().to(dev)
    old = MaskHeadSmallConv.exact_fp32
    # True: the default -- exact fp32 (since round 6 through this library's own MFMA convolution); False: split-bf16
    MaskHeadSmallConv.exact_fp32 = bool(exact)
    try:
        with torch.no_grad():
            got = head([t.to(dev) for t in x], None)
    finally:
        MaskHeadSmallConv.exact_fp32 = old
    scale = float(want.abs().max())
    err = float((got.double().cpu() - want).abs().max())
    print("MaskHeadSmallConv %s: max |err| %.3e = %.2e of the output scale %.1f" % (
        {True: "fp32 (default: own exact MFMA convolution)", False: "split-bf16"}[exact], err, err / scale, scale))
    assert err < (1e-5 if exact else 1e-4) * scale, (err, scale)


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("patch,cin,dim", [(16, 3, 1280), (4, 3, 192)])
def test_patch_embed_at_trained_scales(patch, cin, dim, exact, dev):
    from uninext_amd.backbone import PatchEmbed
    gen = torch.Generator().manual_seed(7)
    pe = PatchEmbed(kernel_size=(patch, patch), stride=(patch, patch), in_chans=cin, embed_dim=dim).eval()
    with torch.no_grad():
        fan = cin * patch * patch
        pe.proj.weight.copy_(heavy(pe.proj.weight.shape, 1.0 / fan ** 0.5, 5.0, 0.001, gen))
        pe.proj.bias.copy_(heavy(pe.proj.bias.shape, 0.2, 2.0, 0.02, gen))
    x = heavy((2, cin, 8 * patch, 12 * patch), 1.0, 100.0, 0.001, gen)      # (normalised images reach a few units; 1e2 is the bound asked for)
    with torch.no_grad():
        want = pe.double()(x.double())
    pe = pe.float().to(dev)
    old = PatchEmbed.exact_fp32
    PatchEmbed.exact_fp32 = exact
    try:
        with torch.no_grad():
            got = pe(x.to(dev))
    finally:
        PatchEmbed.exact_fp32 = old
    scale = float(want.abs().max())
    err = float((got.double().cpu() - want).abs().max())
    print("PatchEmbed %dx%d %s: max |err| %.3e = %.2e of the output scale %.1f" % (patch, patch, "fp32 (default)" if exact else "split-bf16", err, err / scale, scale))
    assert err < (1e-5 if exact else 1e-4) * scale, (err, scale)
