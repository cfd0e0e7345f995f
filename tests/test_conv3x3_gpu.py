"""MI355X parity tests of conv3x3_hip_f32 (include/conv3x3_hip.h) and of the MaskHeadSmallConv mirror that uses it:
numpy oracle on seeded inputs (tile tails, borders, both tile sizes), fixtures minted by the reference class, and the
real R50 shapes (jia_dcn 256 -> 256 at 100 x 167, lay2 64 -> 8) against the PyTorch-ROCm convolution plus
size-independent properties.  Tolerance 1e-4 of the output scale (north_star); observed ~1e-6."""
import numpy as np
import pytest
import torch

from golden_util import load_golden, maskhead_names, max_abs

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("split_bf16_paths")]   # this file is about the opt-in fast paths


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("B,C,H,W,E", [
    (1, 16, 5, 7, 8),          # tiny: one partial 64 x 64 tile, every pixel on a border
    (2, 16, 33, 47, 70),       # odd sizes, partial channel tile
    (1, 64, 100, 167, 8),      # lay2 of the R50 model (cout 8 -> 64 x 64 tiles)
    (2, 16, 100, 167, 130),    # 128 x 128 tiles (522 of them), partial channel tile, two images
    (3, 48, 9, 300, 130),      # wide rows, partial 128-channel tile
    (2, 16, 6, 5, 8),          # round 6, the exact kernel's 16-byte halo loads: widths of every residue mod 4 -- a row's last load
    (1, 32, 3, 2, 8),          # reaches past the row, in the tensor's last row past the tensor (read as zeros, masked) --,
    (2, 16, 17, 18, 16),       # an image narrower than one load, exactly one halo wide, a single pixel
    (1, 16, 1, 1, 8),
    (1, 16, 4, 21, 8),
])
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("precision", [0, 1, 2, 3])     # 2: the packed-weight fast path of precision 1; 3: the exact fp32 halo kernel
def test_vs_oracle(B, C, H, W, E, relu, precision, dev):
    from oracle import conv3x3_oracle
    from uninext_amd import ext
    rng = np.random.default_rng(C * 100 + E)
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((E, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
    b = rng.standard_normal(E).astype(np.float32)
    ref = conv3x3_oracle.conv3x3(x, w, b, relu)
    t = lambda a: torch.from_numpy(a).to(dev)
    if precision == 2:
        packed = ext.conv3x3_pack_weight(t(w))
        run = lambda bias: ext.conv3x3_packed_forward(t(x), packed, E, bias, relu=relu).cpu().numpy()
    elif precision == 3:
        packed = ext.conv3x3_pack_weight(t(w), exact=True)
        run = lambda bias: ext.conv3x3_packed_forward(t(x), packed, E, bias, relu=relu, exact=True).cpu().numpy()
    else:
        run = lambda bias: ext.conv3x3_forward(t(x), t(w), bias, relu=relu, precision=precision).cpu().numpy()
    out = run(t(b))
    assert max_abs(out, ref) < 1e-4 * max(1.0, float(np.abs(ref).max()))
    if precision in (0, 3):
        assert max_abs(out, ref) < 5e-6 * max(1.0, float(np.abs(ref).max()))     # exact-fp32 products
    out = run(None)
    assert max_abs(out, conv3x3_oracle.conv3x3(x, w, None, relu)) < 1e-4 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("B,C,H,W,E", [
    (2, 16, 300, 470, 64),     # >= 2048 units of 8 x 16 x 64: the exact kernel's second unit (two sub-tiles per wave)
    (2, 16, 200, 330, 130),    # ... and its first (8 x 16 x 128, partial channel tile)
])
def test_exact_kernel_larger_units(B, C, H, W, E, dev):
    """conv3x3_hip_packed_exact_f32 picks 8 x 16 pixel units when a launch has >= 2048 of them (feature maps beyond the R50 head's):
    the same kernel body with two sub-tiles per wave and, since round 6, the 16-byte halo loads -- against PyTorch's float64
    convolution on the device (the numpy oracle takes minutes at these sizes)."""
    from uninext_amd import ext
    g = torch.Generator().manual_seed(C + E)
    x = torch.randn(B, C, H, W, generator=g).to(dev)
    w = (torch.randn(E, C, 3, 3, generator=g) / (9 * C) ** 0.5).to(dev)
    b = torch.randn(E, generator=g).to(dev)
    packed = ext.conv3x3_pack_weight(w, exact=True)
    got = ext.conv3x3_packed_forward(x, packed, E, b, relu=True, exact=True)
    want = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1))
    assert float((got.double() - want).abs().max()) < 5e-6 * max(1.0, float(want.abs().max()))
    assert torch.equal(got, ext.conv3x3_packed_forward(x, packed, E, b, relu=True, exact=True))


@pytest.mark.parametrize("name", maskhead_names())
def test_module_vs_reference_fixture(name, dev):
    from uninext_amd.mask_head import MaskHeadSmallConv
    g = load_golden(name)
    params = {k[2:]: torch.from_numpy(v).float() for k, v in g.items() if k.startswith("p:")}
    fpn_dims = [g["fpn%d" % i].shape[1] for i in range(3)] if "fpn0" in g else None
    head = MaskHeadSmallConv(g["x0"].shape[1], fpn_dims, g["x0"].shape[1])
    head.load_state_dict(params)
    head = head.to(dev).eval()
    x = [torch.from_numpy(g["x%d" % i]).float().to(dev) for i in range(3)]
    fpns = [torch.from_numpy(g["fpn%d" % i]).float().to(dev) for i in range(3)] if fpn_dims else None
    with torch.no_grad():
        out = head(x, fpns)
    assert max_abs(out.cpu().numpy(), g["out"]) < 1e-4 * max(1.0, float(np.abs(g["out"]).max()))


def test_full_size_r50_head(dev):
    """The five convolutions at the R50 800 x 1333 shapes, bs 2, through the module: PyTorch-ROCm convolutions on the
    same weights as the cross-check, oracle on a crop of the largest one, linearity of the bias-free convolution."""
    from oracle import conv3x3_oracle
    from uninext_amd import ext
    from uninext_amd.mask_head import MaskHeadSmallConv
    torch.manual_seed(2)
    head = MaskHeadSmallConv(256, None, 256).to(dev).eval()
    for m in head.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.uniform_(m.bias, -0.1, 0.1)
    x = [torch.randn(2, 256, h, w, device=dev) for h, w in ((100, 167), (50, 84), (25, 42))]
    with torch.no_grad():
        head.exact_fp32 = False
        out = head(x, None)                                          # HIP route: split-bf16 from packed weights
        head.exact_fp32 = True
        assert head.own_exact_conv is True                           # the default since round 6
        out_own = head(x, None)                                      # the default: exact fp32 through conv3x3_hip_packed_exact_f32
        assert torch.equal(head(x, None), out_own)                   # one fixed summation order: bitwise repeatable
        head.own_exact_conv = False
        out_exact = head(x, None)                                    # fp32 through MIOpen
        head.own_exact_conv = True
        F = torch.nn.functional
        f = F.relu(head.lay3(x[-1]))
        f = F.relu(head.lay4(x[-2] + F.interpolate(f, size=x[-2].shape[-2:], mode="nearest")))
        f = F.relu(head.jia_dcn(x[-3] + F.interpolate(f, size=x[-3].shape[-2:], mode="nearest")))
        want = F.relu(head.lay2(F.relu(head.lay1(f))))
    assert out.shape == (2, 8, 100, 167)
    assert float((out - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))
    assert float((out_exact - want).abs().max()) < 1e-5 * max(1.0, float(want.abs().max()))
    assert float((out_own - want).abs().max()) < 1e-5 * max(1.0, float(want.abs().max()))
    # oracle on the top-left 12 x 14 crop of jia_dcn's output (needs a 13 x 15 input crop; zero padding on two sides)
    with torch.no_grad():
        xin = torch.randn(2, 256, 100, 167, device=dev)
        got = ext.conv3x3_forward(xin, head.jia_dcn.weight, head.jia_dcn.bias, relu=True)
    ref = conv3x3_oracle.conv3x3(xin[:, :, :13, :15].cpu().numpy(), head.jia_dcn.weight.detach().cpu().numpy(),
                                 head.jia_dcn.bias.detach().cpu().numpy(), relu=True)[:, :, :12, :14]
    assert max_abs(got[:, :, :12, :14].cpu().numpy(), ref) < 1e-4 * max(1.0, float(np.abs(ref).max()))
    with torch.no_grad():
        y = torch.randn_like(xin)
        c = lambda t: ext.conv3x3_forward(t, head.jia_dcn.weight, None, relu=False)
        assert float((c(2 * xin + y) - (2 * c(xin) + c(y))).abs().max()) < 1e-4 * float(c(xin).abs().max())


def test_stream_graph_and_autograd_route(dev):
    from uninext_amd import ext
    from uninext_amd.mask_head import conv3x3_relu
    conv = torch.nn.Conv2d(16, 24, 3, padding=1).to(dev)
    x = torch.randn(2, 16, 11, 13, device=dev)
    with torch.no_grad():
        want = torch.relu(conv(x))
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            a = ext.conv3x3_forward(x, conv.weight, conv.bias, relu=True)   # this library's exact-fp32 kernel, on a side stream
        s.synchronize()
        assert float((a - want).abs().max()) < 1e-4
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out = conv3x3_relu(x, conv, False)     # the split-bf16 kernel from packed weights
        x.copy_(torch.randn_like(x))
        graph.replay()
        torch.cuda.synchronize()
        assert float((static_out - torch.relu(conv(x))).abs().max()) < 1e-4
    y = conv3x3_relu(x.requires_grad_(True), conv)      # autograd recording: PyTorch route
    y.sum().backward()
    assert x.grad is not None


def test_split_precision_error_level(dev):
    """precision 1 on badly scaled data: operands spanning 8 orders of magnitude still land within 1e-4 of the output
    scale (the split keeps 16 mantissa bits per operand whatever the exponent)."""
    from oracle import conv3x3_oracle
    from uninext_amd import ext
    rng = np.random.default_rng(9)
    x = (rng.standard_normal((1, 32, 20, 24)) * 10.0 ** rng.uniform(-4, 4, (1, 32, 1, 1))).astype(np.float32)
    w = (rng.standard_normal((16, 32, 3, 3)) * 10.0 ** rng.uniform(-2, 2, (16, 1, 1, 1))).astype(np.float32)
    ref = conv3x3_oracle.conv3x3(x, w, None, False)
    out = ext.conv3x3_forward(torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), None, precision=1).cpu().numpy()
    scale = np.abs(ref).max(axis=(0, 2, 3), keepdims=True)                # per output channel
    assert float((np.abs(out - ref) / scale).max()) < 1e-4


@pytest.mark.parametrize("H,W,h,w", [(50, 84, 25, 42), (100, 167, 50, 84), (13, 18, 7, 9), (7, 9, 4, 5), (5, 5, 5, 5), (9, 31, 2, 3)])
def test_upsample_add_equals_torch(H, W, h, w, dev):
    from uninext_amd import ext
    g = torch.Generator().manual_seed(H * W)
    skip, low = torch.randn(2, 6, H, W, generator=g).to(dev), torch.randn(2, 6, h, w, generator=g).to(dev)
    want = skip + torch.nn.functional.interpolate(low, size=(H, W), mode="nearest")
    assert torch.equal(ext.upsample_add(skip, low), want)             # same source pixel, one fp32 add: bitwise
    with pytest.raises(RuntimeError, match="expected float32"):
        ext.upsample_add(skip, low[:, :3].contiguous())


def test_errors(dev):
    from uninext_amd import ext
    x = torch.randn(1, 8, 4, 4, device=dev)
    with pytest.raises(RuntimeError, match="multiple of 16"):
        ext.conv3x3_forward(x, torch.randn(4, 8, 3, 3, device=dev))
    with pytest.raises(RuntimeError, match="contiguous"):
        ext.conv3x3_forward(torch.randn(1, 16, 4, 4, device=dev).permute(0, 1, 3, 2), torch.randn(4, 16, 3, 3, device=dev))
    out = ext.conv3x3_forward(torch.randn(0, 16, 4, 4, device=dev), torch.randn(4, 16, 3, 3, device=dev))
    assert out.shape == (0, 4, 4, 4)
