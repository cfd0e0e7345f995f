"""MI355X parity tests of patch_embed_hip_f32 (include/patch_embed_hip.h): reference-minted fixtures, the numpy
oracle on seeded inputs (tile tails, odd image sizes, every patch size), and size-independent properties at the
ViT-Huge / ConvNeXt-Large shapes of BASELINE.json configs 3-4.  Tolerance: 1e-4 of the output scale (north_star);
the kernel is an exact-fp32 fmaf chain, so the observed error is ~1e-6."""
import numpy as np
import pytest
import torch

from golden_util import load_golden, max_abs, patch_names

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("split_bf16_paths")]   # this file is about the opt-in fast paths


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _run(x, w, b, channels_last, dev):
    from uninext_amd import ext
    t = lambda a: None if a is None else torch.as_tensor(np.asarray(a), dtype=torch.float32, device=dev).contiguous()
    return ext.patch_embed_forward(t(x), t(w), t(b), channels_last=channels_last).cpu().numpy()


@pytest.mark.parametrize("name", patch_names())
def test_fixture(name, dev):
    g = load_golden(name)
    cl = bool(int(g["channels_last"]))
    out = _run(g["x"], g["weight"], g["bias"], cl, dev)
    assert out.shape == g["out"].shape
    assert max_abs(out, g["out"]) < 1e-4 * max(1.0, float(np.abs(g["out"]).max()))
    out2 = _run(g["x"], g["weight"], g["bias"], not cl, dev)            # the other layout: same numbers, transposed
    ref2 = g["out"].transpose(0, 3, 1, 2) if cl else g["out"].transpose(0, 2, 3, 1)
    assert max_abs(out2, ref2) < 1e-4 * max(1.0, float(np.abs(g["out"]).max()))


@pytest.mark.parametrize("k,C,E,B,H,W", [
    (16, 3, 1280, 1, 80, 112),     # ViT-Huge width, 35 patches: one partial row tile
    (16, 3, 130, 3, 161, 207),     # odd image sizes (rows only 4-byte aligned), 3 images, partial column tile
    (8, 4, 96, 2, 40, 72),
    (4, 3, 192, 2, 100, 135),      # ConvNeXt-Large stem
    (4, 5, 64, 1, 64, 64),         # K = 80
    (2, 192, 384, 1, 50, 68),      # ConvNeXt-Large downsample 1
    (2, 8, 16, 2, 9, 11),          # K = 32, tiny
])
@pytest.mark.parametrize("channels_last", [True, False])
def test_packed_vs_oracle(k, C, E, B, H, W, channels_last, dev):
    """Split-bf16 path from packed weights (where the geometry has one): 1e-4 of the output scale."""
    from oracle import patch_embed_oracle
    from uninext_amd import ext
    rng = np.random.default_rng(k * 1000 + E + 1)
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((E, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32)
    b = rng.standard_normal(E).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    if (C * k * k) % 48 != 0:
        assert not ext.patch_embed_packed_supported(t(w))
        with pytest.raises(RuntimeError, match="multiple of 48"):
            ext.patch_embed_pack_weight(t(w))
        return
    packed = ext.patch_embed_pack_weight(t(w))
    ref = patch_embed_oracle.forward(x, w, b, channels_last)
    out = ext.patch_embed_packed_forward(t(x), packed, E, k, t(b), channels_last).cpu().numpy()
    assert out.shape == ref.shape and max_abs(out, ref) < 1e-4 * max(1.0, float(np.abs(ref).max()))
    out = ext.patch_embed_packed_forward(t(x), packed, E, k, None, channels_last).cpu().numpy()
    assert max_abs(out, patch_embed_oracle.forward(x, w, None, channels_last)) < 1e-4 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("k,C,E,B,H,W", [
    (16, 3, 1280, 1, 80, 112),
    (16, 3, 130, 3, 161, 207),
    (8, 3, 96, 2, 40, 72),         # K = 192
    (4, 3, 192, 2, 100, 135),      # K = 48: one step
    (4, 5, 64, 1, 64, 64),         # K = 80: no packed path
    (2, 192, 384, 1, 50, 68),
    (2, 12, 16, 2, 9, 11),         # K = 48, tiny
])
@pytest.mark.parametrize("channels_last", [True, False])
def test_vs_oracle(k, C, E, B, H, W, channels_last, dev):
    from oracle import patch_embed_oracle
    rng = np.random.default_rng(k * 1000 + E)
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((E, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32)
    b = rng.standard_normal(E).astype(np.float32)
    ref = patch_embed_oracle.forward(x, w, b, channels_last)
    out = _run(x, w, b, channels_last, dev)
    assert out.shape == ref.shape
    assert max_abs(out, ref) < 1e-4 * max(1.0, float(np.abs(ref).max()))
    out_nb = _run(x, w, None, channels_last, dev)                        # bias == NULL
    assert max_abs(out_nb, patch_embed_oracle.forward(x, w, None, channels_last)) < 1e-4 * max(1.0, float(np.abs(ref).max()))


def test_full_size_vit_huge(dev):
    """bs 2 x 800 x 1333 -> 2 x 50 x 83 patches x 1280: oracle on a patch subset, torch conv on everything,
    linearity, layout agreement."""
    from oracle import patch_embed_oracle
    from uninext_amd import ext
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 800, 1333, generator=g).to(dev)
    w = (torch.randn(1280, 3, 16, 16, generator=g) / 27.7).to(dev)
    b = torch.randn(1280, generator=g).to(dev)
    out = ext.patch_embed_forward(x, w, b, channels_last=True)
    assert out.shape == (2, 50, 83, 1280)
    # (1) oracle on the image corners (the patches there exercise the first / last row tiles)
    crop = x[:, :, :32, :48].cpu().numpy()
    ref = patch_embed_oracle.forward(crop, w.cpu().numpy(), b.cpu().numpy(), True)
    assert max_abs(out[:, :2, :3].cpu().numpy(), ref) < 1e-4 * float(np.abs(ref).max())
    crop = x[:, :, 768:800, 1280:1328].cpu().numpy()
    ref = patch_embed_oracle.forward(crop, w.cpu().numpy(), b.cpu().numpy(), True)
    assert max_abs(out[:, 48:50, 80:83].cpu().numpy(), ref) < 1e-4 * float(np.abs(ref).max())
    # (2) PyTorch-ROCm convolution (the arithmetic the reference runs on a GPU)
    tc = torch.nn.functional.conv2d(x, w, b, stride=16).permute(0, 2, 3, 1)
    assert float((out - tc).abs().max()) < 1e-4 * float(tc.abs().max())
    # (3) the NCHW variant holds the same numbers
    nchw = ext.patch_embed_forward(x, w, b, channels_last=False)
    assert torch.equal(nchw.permute(0, 2, 3, 1), out)
    # (4) linearity in x (bias-free)
    y = torch.randn_like(x)
    f = lambda t: ext.patch_embed_forward(t, w, None, channels_last=True)
    assert float((f(2 * x + y) - (2 * f(x) + f(y))).abs().max()) < 1e-4 * float(out.abs().max())


def test_layer_and_stream(dev):
    """PatchEmbed / patch_conv2d route inference through the HIP kernel, also on a side stream and in a HIP graph."""
    from uninext_amd.backbone import PatchEmbed, patch_conv2d
    torch.manual_seed(3)
    pe = PatchEmbed(in_chans=3, embed_dim=96).to(dev)
    x = torch.randn(2, 3, 64, 96, device=dev)
    with torch.no_grad():
        got = pe(x)                                   # split-bf16 from cached packed weights
        want = pe.proj(x).permute(0, 2, 3, 1)
        pe.exact_fp32 = True
        got_exact = pe(x)
        pe.exact_fp32 = False
        assert float((got_exact - want).abs().max()) < 1e-5
        pe.proj.weight.mul_(2.0)                      # in-place update: the packed copy must be rebuilt
        assert float((pe(x) - pe.proj(x).permute(0, 2, 3, 1)).abs().max()) < 2e-4
        pe.proj.weight.mul_(0.5)
    assert got.is_contiguous() and float((got - want).abs().max()) < 1e-4
    loss = pe(x).sum()              # autograd recording: PyTorch route, has a backward
    loss.backward()
    assert pe.proj.weight.grad is not None
    conv = torch.nn.Conv2d(16, 32, kernel_size=2, stride=2).to(dev)
    xc = torch.randn(1, 16, 10, 12, device=dev)
    s = torch.cuda.Stream()
    with torch.no_grad(), torch.cuda.stream(s):
        a = patch_conv2d(xc, conv)
    s.synchronize()
    with torch.no_grad():
        assert float((a - conv(xc)).abs().max()) < 1e-4
        graph = torch.cuda.CUDAGraph()
        static_out = None
        with torch.cuda.graph(graph):
            static_out = patch_conv2d(xc, conv)
        xc.copy_(torch.randn_like(xc))
        graph.replay()
        torch.cuda.synchronize()
        assert float((static_out - conv(xc)).abs().max()) < 1e-4


def test_errors(dev):
    from uninext_amd import ext
    x = torch.randn(1, 3, 32, 32, device=dev)
    with pytest.raises(RuntimeError, match="patch size"):
        ext.patch_embed_forward(x, torch.randn(8, 3, 3, 3, device=dev))
    with pytest.raises(RuntimeError, match="multiple of 16"):
        ext.patch_embed_forward(torch.randn(1, 1, 32, 32, device=dev), torch.randn(8, 1, 2, 2, device=dev))
    with pytest.raises(RuntimeError, match="contiguous"):
        ext.patch_embed_forward(x.permute(0, 1, 3, 2), torch.randn(8, 3, 16, 16, device=dev))
    with pytest.raises(RuntimeError, match="float32"):
        ext.patch_embed_forward(x.double(), torch.randn(8, 3, 16, 16, device=dev).double())
    out = ext.patch_embed_forward(torch.randn(0, 3, 32, 32, device=dev), torch.randn(8, 3, 16, 16, device=dev))
    assert out.shape == (0, 2, 2, 8)
