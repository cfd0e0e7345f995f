"""MI355X parity tests of linear_hip_packed_f32 (include/linear_hip.h) and of MSDeformAttn's inference projections:
numpy oracle on seeded inputs (row / column tails, masks, every supported K), the encoder shape against hipBLASLt,
badly scaled operands, and the whole layer with and without the fast projections.  Tolerance 1e-4 of the output
scale; observed ~2e-5."""
import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("split_bf16_paths")]   # this file is about the opt-in fast paths


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
    from uninext_amd import _lib
    with torch.no_grad():
        fast = layer(query, ref, src, sh, lsi, mask)
        assert "_msda_packed" in layer.value_proj.__dict__
        assert _lib.last_kernel("forward") == ("msda_fwd_lg3_fused" if Lq >= 1024 else "msda_fwd_fused")
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


@pytest.mark.parametrize("ref_dim", [2, 4])
def test_fused_lg3_and_head_major_vs_oracle(ref_dim, dev):
    """Encoder-sized fused forward (msda_fwd_lg3 with the prologue folded in) on both value layouts against the C
    oracle fed with the PyTorch prologue, and the head-major Linear epilogue against a transpose."""
    import torch.nn.functional as F
    from oracle import msda_oracle
    from uninext_amd import _lib, ext
    from uninext_amd.workloads import level_tensors
    torch.manual_seed(31)
    levels = ((40, 50), (20, 25), (10, 13), (5, 7))
    S = sum(h * w for h, w in levels)
    N, M, L, P, Lq = 2, 8, 4, 4, 1500
    sh, lsi = level_tensors(levels, dev)
    value = torch.randn(N, S, M, 32, device=dev)
    ref = torch.rand(N, Lq, L, ref_dim, device=dev)
    if ref_dim == 4:
        ref[..., 2:] = ref[..., 2:] * 0.3 + 0.05
    offsets = torch.randn(N, Lq, M * L * P * 2, device=dev) * 3
    logits = torch.randn(N, Lq, M * L * P, device=dev)
    offsets[1, 5, 7] = float("nan")
    out = ext.ms_deform_attn_forward_fused(value, sh, lsi, ref, offsets, logits, P)
    assert _lib.last_kernel("forward") == "msda_fwd_lg3_fused"
    value_hm = value.permute(0, 2, 1, 3).contiguous()
    out_hm = ext.ms_deform_attn_forward_fused(value_hm, sh, lsi, ref, offsets, logits, P, value_head_major=True)
    assert _lib.last_kernel("forward") == "msda_fwd_lg3_fused"
    assert torch.equal(torch.nan_to_num(out), torch.nan_to_num(out_hm))            # same arithmetic, other addresses
    off = offsets.view(N, Lq, M, L, P, 2)
    w = F.softmax(logits.view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
    if ref_dim == 2:
        wh = torch.stack([sh[..., 1], sh[..., 0]], -1)
        loc = ref[:, :, None, :, None, :] + off / wh[None, None, None, :, None, :]
    else:
        loc = ref[:, :, None, :, None, :2] + off / P * ref[:, :, None, :, None, 2:] * 0.5
    idx = torch.cat([torch.arange(0, 300), torch.arange(Lq - 300, Lq)]).to(dev)
    want = msda_oracle.forward(value, sh, lsi, loc[:, idx].contiguous(), w[:, idx].contiguous())
    got = out[:, idx].cpu().numpy()
    ok = np.isfinite(want)                                   # the NaN offset poisons one (query, head) in the oracle
    assert np.abs(got[ok] - want[ok]).max() < 1e-4
    assert torch.isfinite(out).all()                         # the kernel drops the out-of-range sample instead
    with pytest.raises(RuntimeError, match="head-major"):    # decoder-sized call: no head-major kernel
        ext.ms_deform_attn_forward_fused(value_hm, sh, lsi, ref[:, :100].contiguous(), offsets[:, :100].contiguous(),
                                         logits[:, :100].contiguous(), P, value_head_major=True)
    # head-major Linear epilogue == transpose of the row-major result
    lin = torch.nn.Linear(256, 256).to(dev)
    x = torch.randn(N, S, 256, device=dev)
    mask = torch.rand(N, S, device=dev) < 0.1
    with torch.no_grad():
        packed = ext.linear_pack_weight(lin.weight)
        rm = ext.linear_packed_forward(x, packed, 256, lin.bias, mask)
        hm = ext.linear_packed_forward(x, packed, 256, lin.bias, mask, head_major_rows=S)
    assert hm.shape == (N, 8, S, 32) and torch.equal(hm, rm.view(N, S, 8, 32).permute(0, 2, 1, 3))


@pytest.mark.parametrize("rows,k", [(1, 64), (63, 256), (64, 256), (1000, 256), (333, 1024)])
@pytest.mark.parametrize("with_res", [True, False])
def test_linear_layernorm_epilogue(rows, k, with_res, dev):
    from uninext_amd import ext
    g = torch.Generator().manual_seed(rows + k)
    x = torch.randn(rows, k, generator=g).to(dev)
    w = (torch.randn(256, k, generator=g) / k ** 0.5).to(dev)
    b, gm, bt = (torch.randn(256, generator=g).to(dev) for _ in range(3))
    res = (torch.randn(rows, 256, generator=g) * 2).to(dev) if with_res else None
    packed = ext.linear_pack_weight(w)
    got = ext.linear_packed_ln(x, packed, b, res, gm, bt, 1e-5)
    y = torch.nn.functional.linear(x.double(), w.double(), b.double()) + (res.double() if with_res else 0)
    want = torch.nn.functional.layer_norm(y, (256,), gm.double(), bt.double(), 1e-5)
    assert float((got.double() - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))
    plain = ext.linear_packed_ln(x, packed, None, res, None, None, 1e-5)
    y = torch.nn.functional.linear(x.double(), w.double()) + (res.double() if with_res else 0)
    assert float((plain.double() - torch.nn.functional.layer_norm(y, (256,), None, None, 1e-5)).abs().max()) < 1e-4
    with pytest.raises(RuntimeError, match="packed copy"):
        ext.linear_packed_ln(x, ext.linear_pack_weight(torch.randn(128, k, device=dev)), None, None, None, None, 1e-5)


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


@pytest.mark.parametrize("rows,d_ffn", [(1, 128), (63, 256), (64, 1024), (333, 1024), (1000, 384)])
@pytest.mark.parametrize("layer_norm", [True, False])
def test_fused_ffn_equals_two_kernel_path_and_oracle(rows, d_ffn, layer_norm, dev):
    """linear_hip_packed_ffn_f32: bitwise the result of linear_packed_forward(relu) -> linear_packed_ln / forward, and
    within 1e-4 of the float64 composition (deformable_transformer_dino.py:354-357)."""
    from uninext_amd import ext
    g = torch.Generator().manual_seed(rows * 7 + d_ffn)
    x = torch.randn(rows, 256, generator=g).to(dev)
    l1, l2 = torch.nn.Linear(256, d_ffn).to(dev), torch.nn.Linear(d_ffn, 256).to(dev)
    ln = torch.nn.LayerNorm(256).to(dev)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.uniform_(-0.5, 0.5)
        p1, p2 = ext.linear_pack_weight(l1.weight), ext.linear_pack_weight(l2.weight)
        hidden = ext.linear_packed_forward(x, p1, d_ffn, l1.bias, relu=True)
        if layer_norm:
            two = ext.linear_packed_ln(hidden, p2, l2.bias, x, ln.weight, ln.bias, ln.eps)
        else:
            two = ext.linear_packed_forward(hidden, p2, 256, l2.bias) + x
        one = ext.ffn_packed(x, p1, l1.bias, p2, l2.bias, d_ffn, x, ln.weight, ln.bias, ln.eps, layer_norm=layer_norm)
        assert one.shape == two.shape
        if layer_norm and d_ffn % 256 != 0:
            assert torch.equal(one, two)      # the 4-wave kernel repeats the two-kernel arithmetic exactly
        else:   # 8 waves reduce the LayerNorm statistics in another order; without LayerNorm the residual is added
            # inside the kernel: a few ulp either way
            assert float((one - two).abs().max()) <= 4e-6 * float(two.abs().max())
        xd = x.double()
        want = xd + torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(xd, l1.weight.double(), l1.bias.double())),
                                               l2.weight.double(), l2.bias.double())
        if layer_norm:
            want = torch.nn.functional.layer_norm(want, (256,), ln.weight.double(), ln.bias.double(), ln.eps)
        assert float((one.double() - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))
        # no biases, no residual
        one = ext.ffn_packed(x, p1, None, p2, None, d_ffn, None, None, None, layer_norm=False)
        want = torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(xd, l1.weight.double())), l2.weight.double())
        assert float((one.double() - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))


def test_fused_ffn_rejects_unsupported(dev):
    from uninext_amd import ext
    x = torch.randn(8, 256, device=dev)
    w1, w2 = torch.randn(192, 256, device=dev), torch.randn(256, 192, device=dev)
    assert not ext.ffn_packed_supported(x, w1, w2, (256,))
    p1, p2 = ext.linear_pack_weight(w1), ext.linear_pack_weight(w2)
    with pytest.raises(RuntimeError):
        ext.ffn_packed(x, p1, None, p2, None, 192)
    with pytest.raises(RuntimeError):
        ext.ffn_packed(x, p1, None, p2, None, 256)   # packed sizes do not match d_ffn


@pytest.mark.parametrize("rows,with_add", [(44446, True), (130, False), (64, True)])
def test_two_output_linear_equals_the_two_separate_calls(rows, with_add, dev):
    """linear_hip_packed_split_f32: sampling_offsets and attention_weights of MSDeformAttn (256 -> 256 and 256 -> 128) as
    one product with two outputs.  Same packed arithmetic per element: bitwise the two separate calls."""
    from uninext_amd import ext
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, 256, generator=g).to(dev)
    xa = torch.randn(rows, 256, generator=g).to(dev) * 0.3 if with_add else None
    wa, wb = (torch.randn(256, 256, generator=g) / 16).to(dev), (torch.randn(128, 256, generator=g) / 16).to(dev)
    ba, bb = torch.randn(256, generator=g).to(dev), torch.randn(128, generator=g).to(dev)
    pa, pb = ext.linear_pack_weight(wa), ext.linear_pack_weight(wb)
    pab = ext.linear_pack_weight(torch.cat([wa, wb], 0).contiguous())
    oa = ext.linear_packed_forward(x, pa, 256, ba, x_add=xa)
    ob = ext.linear_packed_forward(x, pb, 128, bb, x_add=xa)
    sa, sb = ext.linear_packed_split_forward(x, pab, 256, 384, torch.cat([ba, bb]).contiguous(), xa)
    assert sa.shape == oa.shape and sb.shape == ob.shape
    assert torch.equal(sa, oa) and torch.equal(sb, ob)
    with pytest.raises(RuntimeError):
        ext.linear_packed_split_forward(x, pab, 100, 384, None, xa)          # split column not a multiple of 128
