"""MI355X parity tests of linear_hip_packed_f32 (include/linear_hip.h) and of MSDeformAttn's inference projections:
numpy oracle on seeded inputs (row / column tails, masks, every supported K), the encoder shape against hipBLASLt,
badly scaled operands, and the whole layer with and without the fast projections.  Tolerance 1e-4 of the output
scale; observed ~2e-5."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("rows,k,n", [(1, 64, 1), (130, 64, 70), (300, 256, 256), (1000, 256, 384), (77, 128, 128),
                                        (5000, 256, 128), (260, 512, 40)])
@pytest.mark.parametrize("masked", [False, True])
def test_vs_oracle(rows, k, n, masked, dev):
    from oracle import linear_oracle
    from uninext_amd import ext
    rng = np.random.default_rng(rows + k + n)
    x = rng.standard_normal((rows, k)).astype(np.float32)
    w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    mask = (rng.random(rows) < 0.25) if masked else None
    t = lambda a: torch.from_numpy(a).to(dev)
    packed = ext.linear_pack_weight(t(w))
    ref = linear_oracle.forward(x, w, b, mask)
    out = ext.linear_packed_forward(t(x), packed, n, t(b), t(mask) if masked else None).cpu().numpy()
    assert out.shape == ref.shape
    assert float(np.abs(out - ref).max()) < 1e-4 * max(1.0, float(np.abs(ref).max()))
    if masked:
        assert np.all(out[mask] == 0.0)
    out = ext.linear_packed_forward(t(x), packed, n, None, None).cpu().numpy()
    assert float(np.abs(out - linear_oracle.forward(x, w)).max()) < 1e-4 * max(1.0, float(np.abs(ref).max()))


def test_encoder_shape_vs_hipblaslt_and_scaling(dev):
    from uninext_amd import ext
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 22223, 256, generator=g).to(dev)
    lin = torch.nn.Linear(256, 256).to(dev)
    with torch.no_grad():
        want = lin(x)
        packed = ext.linear_pack_weight(lin.weight)
        got = ext.linear_packed_forward(x, packed, 256, lin.bias)
        assert got.shape == want.shape
        assert float((got - want).abs().max()) < 1e-4 * float(want.abs().max())
        # per-feature scales spanning 8 orders of magnitude: the split keeps 16 mantissa bits whatever the exponent
        sx = 10.0 ** torch.linspace(-4, 4, 256, device=dev)
        want = torch.nn.functional.linear((x * sx).double(), lin.weight.double(), lin.bias.double())
        got = ext.linear_packed_forward((x * sx).contiguous(), packed, 256, lin.bias)
        assert float((got.double() - want).abs().max()) < 1e-4 * float(want.abs().max())


@pytest.mark.parametrize("ref_dim", [2, 4])
def test_layer_with_fast_projections(ref_dim, dev):
    """The whole MSDeformAttn layer at inference: packed projections vs the fp32 library GEMMs, and an in-place
    weight update must invalidate the packed copy."""
    from uninext_amd.modules import MSDeformAttn
    from uninext_amd.workloads import level_tensors
    torch.manual_seed(21)
    levels = ((20, 27), (10, 14), (5, 7), (3, 4))
    S = sum(h * w for h, w in levels)
    N, Lq = 2, S if ref_dim == 2 else 300
    layer = MSDeformAttn(256, 4, 8, 4).to(dev).eval()
    with torch.no_grad():
        layer.sampling_offsets.weight.normal_(0, 0.02)
        layer.attention_weights.weight.normal_(0, 0.1)
        for lin in (layer.value_proj, layer.output_proj, layer.sampling_offsets, layer.attention_weights):
            lin.bias.add_(torch.randn_like(lin.bias) * 0.1)
    query, src = torch.randn(N, Lq, 256, device=dev), torch.randn(N, S, 256, device=dev)
    ref = torch.rand(N, Lq, 4, ref_dim, device=dev)
    if ref_dim == 4:
        ref[..., 2:] = ref[..., 2:] * 0.3 + 0.05
    mask = torch.zeros(N, S, dtype=torch.bool, device=dev)
    mask[1, -40:] = True
    sh, lsi = level_tensors(levels, dev)
    with torch.no_grad():
        fast = layer(query, ref, src, sh, lsi, mask)
        assert "_msda_packed" in layer.value_proj.__dict__
        MSDeformAttn.fast_linear = False
        try:
            exact = layer(query, ref, src, sh, lsi, mask)
        finally:
            MSDeformAttn.fast_linear = True
        scale = float(exact.abs().max())
        assert float((fast - exact).abs().max()) < 1e-4 * max(1.0, scale)
        layer.output_proj.weight.mul_(1.5)
        fast2 = layer(query, ref, src, sh, lsi, mask)
        MSDeformAttn.fast_linear = False
        try:
            exact2 = layer(query, ref, src, sh, lsi, mask)
        finally:
            MSDeformAttn.fast_linear = True
        assert float((fast2 - exact2).abs().max()) < 1e-4 * max(1.0, float(exact2.abs().max()))
        assert float((fast2 - fast).abs().max()) > 1e-3          # the update took effect
    q2 = query.clone().requires_grad_(True)                     # autograd recording: library GEMMs, has a backward
    layer(q2, ref, src, sh, lsi, mask).sum().backward()
    assert q2.grad is not None


def test_errors(dev):
    from uninext_amd import ext
    with pytest.raises(RuntimeError, match="multiple of 64"):
        ext.linear_pack_weight(torch.randn(8, 100, device=dev))
    packed = ext.linear_pack_weight(torch.randn(8, 64, device=dev))
    with pytest.raises(RuntimeError, match="does not belong"):
        ext.linear_packed_forward(torch.randn(4, 128, device=dev), packed, 8)
    with pytest.raises(RuntimeError, match="contiguous"):
        ext.linear_packed_forward(torch.randn(64, 4, device=dev).t(), packed, 8)
    assert ext.linear_packed_forward(torch.randn(0, 64, device=dev), packed, 8).shape == (0, 8)
