"""MI355X parity tests of the encoder layer around the operator: add + LayerNorm kernel, the extended Linear entry
point (input addend, ReLU), and the whole DeformableTransformerEncoderLayer at inference against reference-minted
fixtures and against its own autograd (PyTorch-ops) route.  Tolerance 1e-4 of the output scale."""
import numpy as np
import pytest
import torch

from golden_util import enclayer_names, load_golden, max_abs

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("split_bf16_paths")]   # this file is about the opt-in fast paths


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("rows,d", [(1, 4), (7, 256), (1000, 256), (333, 64), (50, 1024), (9, 4096), (130, 260)])
@pytest.mark.parametrize("with_res", [True, False])
def test_add_layernorm_vs_torch_fp64(rows, d, with_res, dev):
    from uninext_amd import ext
    g = torch.Generator().manual_seed(rows + d)
    x = (torch.randn(rows, d, generator=g) * 3 + 1.5).to(dev)
    r = torch.randn(rows, d, generator=g).to(dev) if with_res else None
    w, b = torch.randn(d, generator=g).to(dev), torch.randn(d, generator=g).to(dev)
    got = ext.add_layernorm(x, r, w, b, 1e-5)
    s = x.double() + (r.double() if with_res else 0)
    want = torch.nn.functional.layer_norm(s, (d,), w.double(), b.double(), 1e-5)
    assert float((got.double() - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))
    plain = ext.add_layernorm(x, r, None, None, 1e-5)          # no affine parameters
    want = torch.nn.functional.layer_norm(s, (d,), None, None, 1e-5)
    assert float((plain.double() - want).abs().max()) < 1e-4


def test_add_layernorm_constant_rows_and_errors(dev):
    from uninext_amd import ext
    x = torch.full((3, 256), 7.25, device=dev)
    out = ext.add_layernorm(x, None, None, None, 1e-5)
    assert torch.isfinite(out).all() and float(out.abs().max()) < 1e-3      # zero variance: eps keeps it finite
    with pytest.raises(RuntimeError, match="multiple of 4"):
        ext.add_layernorm(torch.randn(2, 6, device=dev), None, None, None, 1e-5)
    with pytest.raises(RuntimeError, match="shape of x"):
        ext.add_layernorm(x, torch.randn(3, 128, device=dev), None, None, 1e-5)
    assert ext.add_layernorm(torch.randn(0, 256, device=dev), None, None, None, 1e-5).shape == (0, 256)


@pytest.mark.parametrize("rows,k,n", [(500, 256, 1024), (77, 1024, 256), (130, 64, 70)])
def test_linear_addend_and_relu(rows, k, n, dev):
    from oracle import linear_oracle
    from uninext_amd import ext
    rng = np.random.default_rng(rows + n)
    x, xa = rng.standard_normal((rows, k)).astype(np.float32), rng.standard_normal((rows, k)).astype(np.float32)
    w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    packed = ext.linear_pack_weight(t(w))
    ref = linear_oracle.forward(x.astype(np.float64) + xa, w, b)
    scale = max(1.0, float(np.abs(ref).max()))
    got = ext.linear_packed_forward(t(x), packed, n, t(b), x_add=t(xa)).cpu().numpy()
    assert float(np.abs(got - ref).max()) < 1e-4 * scale
    got = ext.linear_packed_forward(t(x), packed, n, t(b), x_add=t(xa), relu=True).cpu().numpy()
    assert float(np.abs(got - np.maximum(ref, 0)).max()) < 1e-4 * scale and got.min() >= 0.0
    got = ext.linear_packed_forward(t(x), packed, n, t(b), relu=True).cpu().numpy()
    assert float(np.abs(got - np.maximum(linear_oracle.forward(x, w, b), 0)).max()) < 1e-4 * scale


@pytest.mark.parametrize("name", enclayer_names())
def test_layer_vs_reference_fixture(name, dev):
    from uninext_amd.modules import DeformableTransformerEncoderLayer
    g = load_golden(name)
    params = {k[2:]: torch.from_numpy(v).float() for k, v in g.items() if k.startswith("p:")}
    d_model, d_ffn = params["linear1.weight"].shape[1], params["linear1.weight"].shape[0]
    layer = DeformableTransformerEncoderLayer(d_model=d_model, d_ffn=d_ffn, n_heads=d_model // 32)
    layer.load_state_dict(params)
    layer = layer.to(dev).eval()
    f = lambda k: torch.from_numpy(g[k]).to(dev)
    src, pos, ref = f("src").float(), f("pos").float(), f("ref").float()
    mask = f("mask") if "mask" in g else None
    with torch.no_grad():
        out = layer(src, pos, ref, f("shapes"), f("lsi"), mask)                    # fused inference route
    scale = max(1.0, float(np.abs(g["out"]).max()))
    assert max_abs(out.cpu().numpy(), g["out"]) < 1e-4 * scale
    src_g = src.clone().requires_grad_(True)                                       # autograd route: PyTorch ops + operator
    out_g = layer(src_g, pos, ref, f("shapes"), f("lsi"), mask)
    assert max_abs(out_g.detach().cpu().numpy(), g["out"]) < 1e-4 * scale
    out_g.sum().backward()
    assert src_g.grad is not None and torch.isfinite(src_g.grad).all()


def test_layer_full_size_routes_agree(dev):
    """R50 encoder shapes, bs 2: inference route (packed projections, head-major value, fused add + LayerNorm) vs the
    same layer forced onto PyTorch's GEMMs and LayerNorm."""
    from uninext_amd import _lib, workloads
    from uninext_amd.modules import DeformableTransformerEncoderLayer, MSDeformAttn
    torch.manual_seed(8)
    levels = workloads.R50_LEVELS_INFER
    S = sum(h * w for h, w in levels)
    layer = DeformableTransformerEncoderLayer().to(dev).eval()
    with torch.no_grad():
        layer.self_attn.sampling_offsets.weight.normal_(0, 0.02)
        layer.self_attn.attention_weights.weight.normal_(0, 0.1)
    src, pos = torch.randn(2, S, 256, device=dev), torch.randn(2, S, 256, device=dev) * 0.3
    ref = workloads.encoder_reference_points(levels, dev)[None, :, None, :].expand(2, S, 4, 2).contiguous()
    sh, lsi = workloads.level_tensors(levels, dev)
    with torch.no_grad():
        fast = layer(src, pos, ref, sh, lsi, None)
        assert _lib.last_kernel("forward") in ("msda_fwd_lg3_fused", "msda_fwd_win_fused")
        MSDeformAttn.fast_linear = False
        try:
            q = src + pos
            src2 = layer.self_attn(q, ref, src, sh, lsi, None)
            mid = layer.norm1(src + src2)
            want = layer.norm2(mid + layer.linear2(torch.relu(layer.linear1(mid))))
        finally:
            MSDeformAttn.fast_linear = True
    assert float((fast - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))


def test_layer_is_hip_graph_capturable(dev):
    """Inference through the layer only enqueues kernels (no sync, no host-side data dependence): capture it once,
    replay it on new inputs, compare with eager execution."""
    from uninext_amd import workloads
    from uninext_amd.modules import DeformableTransformerEncoderLayer
    torch.manual_seed(13)
    levels = ((32, 40), (16, 20), (8, 10), (4, 5))
    S = sum(h * w for h, w in levels)
    layer = DeformableTransformerEncoderLayer().to(dev).eval()
    with torch.no_grad():
        layer.self_attn.sampling_offsets.weight.normal_(0, 0.02)
        layer.self_attn.attention_weights.weight.normal_(0, 0.1)
    src, pos = torch.randn(2, S, 256, device=dev), torch.randn(2, S, 256, device=dev) * 0.3
    ref = workloads.encoder_reference_points(levels, dev)[None, :, None, :].expand(2, S, 4, 2).contiguous()
    sh, lsi = workloads.level_tensors(levels, dev)
    mask = torch.zeros(2, S, dtype=torch.bool, device=dev)
    mask[0, :50] = True
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            layer(src, pos, ref, sh, lsi, mask)                      # warm-up: packs the weights, checks the shapes
        torch.cuda.current_stream().wait_stream(side)
        static_src = src.clone()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out = layer(static_src, pos, ref, sh, lsi, mask)
        new_src = torch.randn_like(src)
        static_src.copy_(new_src)
        graph.replay()
        torch.cuda.synchronize()
        # a capture always takes the gather kernel (include/msda_hip.h: events cannot be waited for inside one), an eager
        # call may take the window kernel (another fp32 summation order): ask for determinism to compare bit for bit
        torch.use_deterministic_algorithms(True)
        try:
            eager = layer(new_src, pos, ref, sh, lsi, mask)
        finally:
            torch.use_deterministic_algorithms(False)
        same = torch.equal(static_out, eager)
        torch.cuda.synchronize()
        del graph                                   # release the captured graph while the runtime is fully alive
    assert same


def test_six_layer_stack_vs_reference_fixture(dev):
    """Six distinct encoder layers back to back (d_model 256, 8 heads, d_ffn 1024, S = 1065) against the reference's
    own layers run in fp64 (tests/golden/make_encoder_layer_golden.py::stack).  Bounds how the per-layer error of the
    default inference path (split-bf16 projections, ~2e-5 of the output scale per GEMM) compounds over the stack:
    1e-4 of the output scale, the parity bound of BASELINE.json's north_star, must hold for the STACK, not per layer.
    The exact-fp32 route (fast_linear = False) is held to the same bound with a wide margin."""
    import importlib.util
    import os
    from uninext_amd.modules import DeformableTransformerEncoderLayer, MSDeformAttn
    spec = importlib.util.spec_from_file_location(
        "make_encoder_layer_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_encoder_layer_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    g = load_golden("encstack_6layers")
    src, pos, ref, shapes, lsi, params, digest = gen.stack_inputs(torch.float32)
    assert abs(digest - float(g["digest"])) < 1e-6 * float(g["digest"]), "parameter generator drifted from the fixture"
    layers = []
    for p in params:
        layer = DeformableTransformerEncoderLayer(d_model=256, d_ffn=1024, n_heads=8)
        layer.load_state_dict(p)
        layers.append(layer.to(dev).eval())
    src, pos, ref, shapes, lsi = (t.to(dev) for t in (src, pos, ref, shapes, lsi))
    want = g["out"].astype(np.float64)
    scale = float(np.abs(want).max())
    errs = {}
    for fast in (True, False):
        old = MSDeformAttn.fast_linear
        MSDeformAttn.fast_linear = fast
        try:
            with torch.no_grad():
                x = src
                for layer in layers:
                    x = layer(x, pos, ref, shapes, lsi, None)
        finally:
            MSDeformAttn.fast_linear = old
        errs[fast] = max_abs(x.cpu().numpy(), want) / scale
    print("six-layer stack, error / output scale: split-bf16 %.2e, exact fp32 %.2e" % (errs[True], errs[False]))
    assert errs[False] < 2e-5, errs
    assert errs[True] < 1e-4, errs
