"""GPU parity tests of the HIP MSDeformAttn path (run on the MI355X box: pytest -m gpu).

Everything goes through the C ABI (uninext_amd.ext -> libmsda_hip.so).  The CPU oracle
(oracle/msda_oracle.c, pinned to reference-minted fixtures by tests/test_oracle_golden.py) and the
golden fixtures themselves are the checkers.  Tolerances: fp64 ~1e-12; fp32 forward 1e-4 abs
(BASELINE.json north_star; the reference's own fp32 check is rtol 1e-2 / atol 1e-3, ops/test.py:56);
fp32 gradients 1e-4 relative to the gradient scale (atomics => summation order differs run to run).
"""
import numpy as np
import pytest
import torch

from golden_util import carried, golden_names, grad_loc_err, load_golden, max_abs, scaled_err

pytestmark = pytest.mark.gpu

NAMES = golden_names()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def api():
    import MultiScaleDeformableAttention as MSDA  # the name the reference imports (func.py:18)
    from uninext_amd import _lib
    _lib.load()  # fails loudly if libmsda_hip.so is not built
    return MSDA, _lib


def _to(g, dev, dtype):
    f = lambda k: torch.from_numpy(g[k]).to(device=dev, dtype=dtype).contiguous()
    i = lambda k: torch.from_numpy(g[k]).to(device=dev)
    return f("value"), i("shapes"), i("lsi"), f("loc"), f("attn"), f("grad_out")


def _np(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------------------------------------
# golden fixtures (reference-generated)

@pytest.mark.parametrize("name", NAMES)
def test_golden_forward_f64(name, dev, api):
    MSDA, lib = api
    g = load_golden(name)
    v, sh, lsi, loc, attn, _ = _to(g, dev, torch.float64)
    out = MSDA.ms_deform_attn_forward(v, sh, lsi, loc, attn, 64)
    assert lib.last_kernel("forward") == "msda_fwd_generic"
    assert out.shape == g["out"].shape
    assert max_abs(_np(out), g["out"]) < 1e-12


@pytest.mark.parametrize("variant", ["auto", "msda_fwd_generic", "msda_fwd_lanegroup"])
@pytest.mark.parametrize("name", NAMES)
def test_golden_forward_f32(name, variant, dev, api):
    MSDA, lib = api
    g = load_golden(name)
    v, sh, lsi, loc, attn, _ = _to(g, dev, torch.float32)
    lib.set_variant("forward", variant)
    try:
        out = MSDA.ms_deform_attn_forward(v, sh, lsi, loc, attn, 64)
    finally:
        lib.set_variant("forward", "auto")
    if variant == "msda_fwd_generic" or v.shape[3] % 4 != 0:
        assert lib.last_kernel("forward") == "msda_fwd_generic"
    elif v.shape[3] == 32:  # the UNINEXT head size must take the lane-group kernel
        assert lib.last_kernel("forward") == "msda_fwd_lanegroup"
    assert max_abs(_np(out), g["out"]) < 1e-4


@pytest.mark.parametrize("name", NAMES)
def test_golden_backward_f64(name, dev, api):
    MSDA, lib = api
    g = load_golden(name)
    v, sh, lsi, loc, attn, go = _to(g, dev, torch.float64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, sh, lsi, loc, attn, go, 64)
    assert lib.last_kernel("backward") == "msda_bwd_generic"
    assert max_abs(_np(gv), g["grad_value"]) < 1e-11
    assert max_abs(_np(ga), g["grad_attn"]) < 1e-11
    if name != "border":  # one-sided derivative convention exactly on cell edges (see test_oracle_golden.py)
        assert max_abs(_np(gl), g["grad_loc"]) < 1e-9


@pytest.mark.parametrize("variant", ["auto", "msda_bwd_generic", "msda_bwd_lanegroup"])
@pytest.mark.parametrize("name", NAMES)
def test_golden_backward_f32(name, variant, dev, api):
    MSDA, lib = api
    g = load_golden(name)
    v, sh, lsi, loc, attn, go = _to(g, dev, torch.float32)
    lib.set_variant("backward", variant)
    try:
        gv, gl, ga = MSDA.ms_deform_attn_backward(v, sh, lsi, loc, attn, go, 64)
    finally:
        lib.set_variant("backward", "auto")
    assert scaled_err(_np(gv), g["grad_value"]) < 1e-4
    assert scaled_err(_np(ga), g["grad_attn"]) < 1e-4
    if name != "border":
        assert grad_loc_err(_np(gl).reshape(g["grad_loc"].shape), g["grad_loc"], g["shapes"]) < 1.0


def test_border_fixture_backward_matches_c_oracle(dev, api):
    """On cell edges the HIP kernels must follow the CUDA formula (as the C oracle does), fp64 exact."""
    from oracle import msda_oracle
    MSDA, _ = api
    g = load_golden("border")
    v, sh, lsi, loc, attn, go = _to(g, dev, torch.float64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, sh, lsi, loc, attn, go, 64)
    ogv, ogl, oga = msda_oracle.backward(g["grad_out"], g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"])
    assert max_abs(_np(gv), ogv) < 1e-12 and max_abs(_np(gl), ogl) < 1e-11 and max_abs(_np(ga), oga) < 1e-12


# ------------------------------------------------------------------------------------------------
# seeded workloads vs the C oracle

@pytest.mark.parametrize("flavour", ["model", "uniform"])
@pytest.mark.parametrize("kind", ["encoder", "decoder"])
def test_quarter_scale_r50_vs_oracle(kind, flavour, dev, api):
    """R50 pyramid at 1/4 linear scale (S = 1394), M=8 D=32 L=4 P=4, N=2: forward + backward vs oracle."""
    from oracle import msda_oracle
    from uninext_amd import workloads
    MSDA, lib = api
    levels = ((25, 42), (13, 21), (7, 11), (4, 6))
    x = workloads.make_inputs(kind, flavour, batch=2, levels=levels, num_query=None if kind == "encoder" else 300,
                              seed=21, device=dev)
    out = MSDA.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
    # encoder shape: the automatic choice follows the reported sample locality (include/msda_hip.h)
    assert lib.last_kernel("forward") in (("msda_fwd_lg3", "msda_fwd_win") if x["loc"].shape[1] >= 1024 else ("msda_fwd_lanegroup",))
    ref = msda_oracle.forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"])
    assert max_abs(_np(out), ref) < 1e-4
    go = torch.randn(out.shape, generator=torch.Generator().manual_seed(5)).to(dev)
    ogv, ogl, oga = msda_oracle.backward(go, x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"])
    for variant in ("msda_bwd_generic", "msda_bwd_lanegroup", "msda_bwd_tiled", "msda_bwd_win"):
        if variant in ("msda_bwd_tiled", "msda_bwd_win") and kind != "encoder":
            continue   # the tiled / window backward need Lq == S; other calls fall back to the generic kernel
        lib.set_variant("backward", variant)
        try:
            gv, gl, ga = MSDA.ms_deform_attn_backward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], go, 64)
        finally:
            lib.set_variant("backward", "auto")
        assert lib.last_kernel("backward") == variant
        assert scaled_err(_np(gv), ogv) < 1e-4
        assert scaled_err(_np(ga), oga) < 1e-4
        assert grad_loc_err(_np(gl), ogl, _np(x["shapes"])) < 1.0


@pytest.mark.parametrize("D,M,L,P", [(4, 3, 2, 3), (8, 5, 3, 2), (16, 2, 1, 5), (64, 2, 4, 4), (128, 1, 2, 2),
                                     (256, 1, 2, 1), (32, 8, 4, 8), (32, 8, 5, 4)])
def test_lanegroup_shapes_vs_oracle(D, M, L, P, dev, api):
    """Every lane-group width G = D/4 in {1..64} and runtime L*P paths."""
    from oracle import msda_oracle
    MSDA, lib = api
    g = torch.Generator().manual_seed(D * 131 + M)
    levels = [(9, 7), (5, 4), (3, 3), (2, 2), (1, 2)][:L]
    S = sum(h * w for h, w in levels)
    N, Lq = 2, 53
    value = torch.randn(N, S, M, D, generator=g).to(dev)
    loc = (torch.rand(N, Lq, M, L, P, 2, generator=g) * 1.3 - 0.15).to(dev)
    attn = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P).to(dev)
    from uninext_amd.workloads import level_tensors
    sh, lsi = level_tensors(levels, dev)
    out = MSDA.ms_deform_attn_forward(value, sh, lsi, loc, attn, 64)
    assert lib.last_kernel("forward") == "msda_fwd_lanegroup"
    assert max_abs(_np(out), msda_oracle.forward(value, sh, lsi, loc, attn)) < 1e-4
    go = torch.randn(out.shape, generator=g).to(dev)
    lib.set_variant("backward", "msda_bwd_lanegroup")
    try:
        gv, gl, ga = MSDA.ms_deform_attn_backward(value, sh, lsi, loc, attn, go, 64)
    finally:
        lib.set_variant("backward", "auto")
    assert lib.last_kernel("backward") == "msda_bwd_lanegroup"
    ogv, ogl, oga = msda_oracle.backward(go, value, sh, lsi, loc, attn)
    assert scaled_err(_np(gv), ogv) < 1e-4 and scaled_err(_np(ga), oga) < 1e-4 and grad_loc_err(_np(gl), ogl, _np(sh)) < 1.0


TILED_PYRAMIDS = [
    ((25, 42), (13, 21), (7, 11), (4, 6)),      # R50 pyramid / 4
    ((8, 8),),                                  # a single tile
    ((9, 7), (5, 4)),
    ((16, 16), (16, 16)),                       # two levels of equal resolution
    ((3, 50), (2, 25), (1, 13)),                # thin images
    ((10, 10), (20, 20)),                       # a finer level after the first one
    ((40, 40), (80, 80), (3, 3)),               # windows exceed the LDS budget: a level is served from L2
    ((17, 23), (9, 12), (5, 6), (3, 3)),
]


TILED_VARIANTS = carried("forward", "msda_fwd_tiled", "msda_fwd_tiled_l0", "msda_fwd_tiled_l0big")   # experiments build only


@pytest.mark.parametrize("variant", TILED_VARIANTS)
@pytest.mark.parametrize("flavour", ["model", "uniform", "wide"])
@pytest.mark.parametrize("levels", TILED_PYRAMIDS)
def test_tiled_forward_vs_oracle(levels, flavour, variant, dev, api):
    """The LDS-tiled encoder kernel (Lq == S) on odd pyramids; far / window-missing samples take its
    global-memory path, so 'uniform' and 'wide' stress that path and 'model' the LDS path."""
    from oracle import msda_oracle
    from uninext_amd import workloads
    MSDA, lib = api
    kw = dict(offset_sigma=6.0) if flavour == "wide" else {}
    x = workloads.make_inputs("encoder", "model" if flavour == "wide" else flavour, batch=2, levels=levels,
                              seed=33, device=dev, **kw)
    lib.set_variant("forward", variant)
    try:
        out = MSDA.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
    finally:
        lib.set_variant("forward", "auto")
    assert lib.last_kernel("forward") == "msda_fwd_tiled"
    ref = msda_oracle.forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"])
    assert max_abs(_np(out), ref) < 1e-4


@pytest.mark.parametrize("variant", TILED_VARIANTS)
@pytest.mark.parametrize("M,L,P", [(1, 1, 4), (3, 2, 4), (16, 3, 4), (8, 4, 2), (5, 1, 16), (2, 3, 5)])
def test_tiled_forward_head_point_counts(M, L, P, variant, dev, api):
    from oracle import msda_oracle
    from uninext_amd import workloads
    MSDA, lib = api
    levels = ((21, 18), (11, 9), (6, 5), (3, 3))[:L]
    x = workloads.make_inputs("encoder", "model", batch=1, levels=levels, heads=M, points=P, seed=9, device=dev)
    x["loc"][0, 3, 0, 0, 0, 0] = float("nan")
    x["loc"][0, 5, M - 1, L - 1, P - 1, 1] = float("inf")
    lib.set_variant("forward", variant)
    try:
        out = MSDA.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
    finally:
        lib.set_variant("forward", "auto")
    assert lib.last_kernel("forward") == ("msda_fwd_tiled" if P == 4 else "msda_fwd_lanegroup")  # P != 4 falls back
    ref = msda_oracle.forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"])
    assert torch.isfinite(out).all() and max_abs(_np(out), ref) < 1e-4


@pytest.mark.parametrize("kind,flavour,levels", [
    ("encoder", "model", ((40, 50), (20, 25), (10, 13), (5, 7))),          # S = 2665 < 4096: falls back
    ("encoder", "uniform", ((64, 80), (32, 40), (16, 20), (8, 10))),       # last level 80 px: resident
    ("encoder", "model", ((64, 80), (32, 40), (16, 20), (17, 17))),        # last level 289 px: too big, nothing resident
    ("decoder", "model", ((64, 80), (32, 40), (16, 20), (8, 10))),         # Lq = 5000 arbitrary queries
])
@pytest.mark.parametrize("variant", carried("forward", "msda_fwd_lgcl", "msda_fwd_lg3", "msda_fwd_lgp"))
def test_lgcl_forward_vs_oracle(kind, flavour, levels, variant, dev, api):
    """Lane-group kernel with the last pyramid level resident in LDS (any query set, any sampling pattern)."""
    from oracle import msda_oracle
    from uninext_amd import workloads
    MSDA, lib = api
    x = workloads.make_inputs(kind, flavour, batch=2, levels=levels, num_query=None if kind == "encoder" else 5000,
                              seed=41, device=dev)
    x["loc"][1, 7, 3, 3, 2, 0] = float("nan")
    lib.set_variant("forward", variant)
    try:
        out = MSDA.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
    finally:
        lib.set_variant("forward", "auto")
    S = x["value"].shape[1]
    min_q = 4096 if variant == "msda_fwd_lgcl" else 1024
    assert lib.last_kernel("forward") == (variant if x["loc"].shape[1] >= min_q else "msda_fwd_lanegroup"), S
    idx = torch.cat([torch.arange(0, 600), torch.arange(x["loc"].shape[1] - 600, x["loc"].shape[1])]).to(dev)
    ref = msda_oracle.forward(x["value"], x["shapes"], x["lsi"], x["loc"][:, idx].contiguous(),
                              x["attn"][:, idx].contiguous())
    assert torch.isfinite(out).all() and max_abs(_np(out[:, idx]), ref) < 1e-4


# ------------------------------------------------------------------------------------------------
# full BASELINE sizes: oracle on a query subset + size-independent properties

@pytest.mark.parametrize("flavour", ["model", "uniform"])
def test_full_size_encoder_forward(flavour, dev, api):
    from oracle import msda_oracle
    from uninext_amd import workloads
    MSDA, lib = api
    x = workloads.make_inputs("encoder", flavour, batch=2, seed=3, device=dev)
    out = MSDA.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
    assert out.shape == (2, 22223, 256)
    auto_kernel = lib.last_kernel("forward")
    assert auto_kernel in ("msda_fwd_lg3", "msda_fwd_win")
    for other in carried("forward", "msda_fwd_lanegroup", "msda_fwd_tiled_l0", "msda_fwd_lgcl", "msda_fwd_lg3", "msda_fwd_lgp", "msda_fwd_win"):   # every fast kernel
        lib.set_variant("forward", other)
        try:
            out_o = MSDA.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
        finally:
            lib.set_variant("forward", "auto")
        assert lib.last_kernel("forward") == other.replace("_l0", "")
        assert float((out - out_o).abs().max()) < 2e-5  # different HIP kernels: summation order only
    # (1) the oracle on a subset of queries (outputs of different queries are independent)
    idx = torch.cat([torch.arange(0, 300), torch.arange(16600, 16800), torch.arange(22000, 22223),
                     torch.randint(0, 22223, (500,), generator=torch.Generator().manual_seed(1))])
    ref = msda_oracle.forward(x["value"], x["shapes"], x["lsi"], x["loc"][:, idx.to(dev)].contiguous(),
                              x["attn"][:, idx.to(dev)].contiguous())
    assert max_abs(_np(out[:, idx.to(dev)]), ref) < 1e-4
    # (2) linearity in value:  f(2 v + w) = 2 f(v) + f(w)
    w = torch.randn_like(x["value"])
    f_w = MSDA.ms_deform_attn_forward(w, x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
    f_mix = MSDA.ms_deform_attn_forward(2 * x["value"] + w, x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
    assert float((f_mix - (2 * out + f_w)).abs().max()) < 1e-4
    # (3) value == 1 everywhere: every output channel of a (query, head) equals the total weight of its valid
    #     corners, so all 32 channels agree and lie in [0, 1]
    ones = MSDA.ms_deform_attn_forward(torch.ones_like(x["value"]), x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
    ones = ones.view(2, 22223, 8, 32)
    assert float((ones - ones[..., :1]).abs().max()) < 1e-6
    assert float(ones.min()) >= 0.0 and float(ones.max()) <= 1.0 + 1e-5
    # (4) determinism: no forward kernel has atomics -> bitwise repeatable per kernel (the automatic choice may move
    #     between the window and the gather kernel from one call to the next: summation order, checked above)
    for pinned in ("msda_fwd_lg3", "msda_fwd_win"):
        lib.set_variant("forward", pinned)
        try:
            a = MSDA.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
            b = MSDA.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
        finally:
            lib.set_variant("forward", "auto")
        assert torch.equal(a, b)


def test_full_size_decoder_forward(dev, api):
    from oracle import msda_oracle
    from uninext_amd import workloads
    MSDA, _ = api
    x = workloads.make_inputs("decoder", "model", batch=2, seed=4, device=dev)
    out = MSDA.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
    ref = msda_oracle.forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"])
    assert max_abs(_np(out), ref) < 1e-4


@pytest.mark.parametrize("flavour", ["model", "uniform", "wide"])
@pytest.mark.parametrize("levels", TILED_PYRAMIDS)
def test_tiled_backward_vs_oracle(levels, flavour, dev, api):
    """LDS-privatised backward (Lq == S) on odd pyramids: 'model' exercises the LDS accumulators, 'uniform' and
    'wide' the far path (direct full-line global atomics)."""
    from oracle import msda_oracle
    from uninext_amd import workloads
    MSDA, lib = api
    kw = dict(offset_sigma=6.0) if flavour == "wide" else {}
    x = workloads.make_inputs("encoder", "model" if flavour == "wide" else flavour, batch=2, levels=levels,
                              seed=35, device=dev, **kw)
    S = x["value"].shape[1]
    go = torch.randn(2, S, 256, generator=torch.Generator().manual_seed(36)).to(dev)
    lib.set_variant("backward", "msda_bwd_tiled")
    try:
        gv, gl, ga = MSDA.ms_deform_attn_backward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], go, 64)
    finally:
        lib.set_variant("backward", "auto")
    assert lib.last_kernel("backward") == "msda_bwd_tiled"
    ogv, ogl, oga = msda_oracle.backward(go, x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"])
    assert scaled_err(_np(gv), ogv) < 1e-4
    assert scaled_err(_np(ga), oga) < 1e-4
    assert grad_loc_err(_np(gl), ogl, _np(x["shapes"])) < 1.0


@pytest.mark.parametrize("variant", ["auto", "msda_bwd_lanegroup", "msda_bwd_generic"])
@pytest.mark.parametrize("levels", ["infer", "train"])
def test_full_size_encoder_backward(levels, variant, dev, api):
    from oracle import msda_oracle
    from uninext_amd import workloads
    MSDA, lib = api
    lv = workloads.R50_LEVELS_INFER if levels == "infer" else workloads.R50_LEVELS_TRAIN
    x = workloads.make_inputs("encoder", "model", batch=2, levels=lv, seed=6, device=dev)
    S = x["value"].shape[1]
    out = MSDA.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
    go = torch.randn(out.shape, generator=torch.Generator().manual_seed(8)).to(dev)
    lib.set_variant("backward", variant)
    try:
        gv, gl, ga = MSDA.ms_deform_attn_backward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], go, 64)
    finally:
        lib.set_variant("backward", "auto")
    assert lib.last_kernel("backward") in (("msda_bwd_tiled", "msda_bwd_win", "msda_bwd_regions") if variant == "auto" else (variant,))
    # per-query gradients: oracle on a query subset
    idx = torch.cat([torch.arange(0, 200), torch.arange(S - 200, S),
                     torch.randint(0, S, (400,), generator=torch.Generator().manual_seed(2))]).to(dev)
    _, ogl, oga = msda_oracle.backward(go[:, idx].contiguous(), x["value"], x["shapes"], x["lsi"],
                                       x["loc"][:, idx].contiguous(), x["attn"][:, idx].contiguous())
    assert scaled_err(_np(ga[:, idx]), oga) < 1e-4
    assert grad_loc_err(_np(gl[:, idx]), ogl, _np(x["shapes"])) < 1.0
    # out is linear in value and in attn:  <grad_value, value> = <grad_out, out> = <grad_attn, attn>
    dot = float((go.double() * out.double()).sum())
    scale = float((go.double() * out.double()).abs().sum())
    assert abs(float((gv.double() * x["value"].double()).sum()) - dot) < 1e-5 * scale
    assert abs(float((ga.double() * x["attn"].double()).sum()) - dot) < 1e-5 * scale
    # grad_value of a query subset alone (small enough for the oracle) -- checks the scatter addresses
    sub = torch.arange(1000, 1400).to(dev)
    gv_sub, _, _ = MSDA.ms_deform_attn_backward(x["value"], x["shapes"], x["lsi"], x["loc"][:, sub].contiguous(),
                                                x["attn"][:, sub].contiguous(), go[:, sub].contiguous(), 64)
    ogv, _, _ = msda_oracle.backward(go[:, sub].contiguous(), x["value"], x["shapes"], x["lsi"],
                                     x["loc"][:, sub].contiguous(), x["attn"][:, sub].contiguous())
    assert scaled_err(_np(gv_sub), ogv) < 1e-4


# ------------------------------------------------------------------------------------------------
# the reference's own self-test procedure (ops/test.py) on top of the Function

def _testpy_inputs(dev, channels, dtype):
    N, M, Lq, L, P = 1, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long, device=dev)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    value = (torch.rand(N, S, M, channels, device=dev) * 0.01).to(dtype)
    loc = torch.rand(N, Lq, M, L, P, 2, device=dev).to(dtype)
    attn = torch.rand(N, Lq, M, L, P, device=dev) + 1e-5
    attn = (attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)).to(dtype)
    return value, shapes, lsi, loc, attn


def test_reference_selftest_forward(dev, api):
    """ops/test.py:31-60 with the same tolerances."""
    from oracle.msda_gridsample import msda_gridsample
    from uninext_amd.functions import MSDeformAttnFunction
    torch.manual_seed(3)
    v, sh, lsi, loc, attn = _testpy_inputs(dev, 2, torch.float64)
    out = MSDeformAttnFunction.apply(v, sh, lsi, loc, attn, 2)
    ref = msda_gridsample(v.cpu(), sh.tolist(), loc.cpu(), attn.cpu())
    assert torch.allclose(out.cpu(), ref)
    v, sh, lsi, loc, attn = _testpy_inputs(dev, 2, torch.float32)
    out = MSDeformAttnFunction.apply(v, sh, lsi, loc, attn, 2)
    ref = msda_gridsample(v.cpu(), sh.tolist(), loc.cpu(), attn.cpu())
    assert torch.allclose(out.cpu(), ref, rtol=1e-2, atol=1e-3)
    assert float((out.cpu() - ref).abs().max()) < 1e-4


@pytest.mark.parametrize("channels", [30, 32, 64, 71, 1025, 2048, 3096])
def test_reference_selftest_gradcheck(channels, dev, api):
    """ops/test.py:63-78,85: fp64 gradcheck at the channel counts that exercised each CUDA backward variant."""
    from torch.autograd import gradcheck
    from uninext_amd.functions import MSDeformAttnFunction
    torch.manual_seed(3)
    v, sh, lsi, loc, attn = _testpy_inputs(dev, channels, torch.float64)
    v.requires_grad_(True)
    loc.requires_grad_(True)
    attn.requires_grad_(True)
    # the full numerical Jacobian at D >= 1025 is ~10^5 forward launches; use torch's fast mode there
    assert gradcheck(MSDeformAttnFunction.apply, (v, sh, lsi, loc, attn, 2), fast_mode=channels > 100,
                     nondet_tol=1e-12)  # fp64 atomics: summation order varies run to run (as in the reference)


def test_autocast_casts_to_fp32(dev, api):
    """custom_fwd(cast_inputs=float32), ops/functions/ms_deform_attn_func.py:23."""
    from uninext_amd.functions import MSDeformAttnFunction
    torch.manual_seed(0)
    v, sh, lsi, loc, attn = _testpy_inputs(dev, 32, torch.float16)
    with torch.autocast("cuda", dtype=torch.float16):
        out = MSDeformAttnFunction.apply(v, sh, lsi, loc, attn, 64)
    assert out.dtype == torch.float32


# ------------------------------------------------------------------------------------------------
# boundary behaviour

def test_errors_and_empty(dev, api):
    MSDA, lib = api
    g = load_golden("d32_l4_p4")
    v, sh, lsi, loc, attn, go = _to(g, dev, torch.float32)
    # all-CPU calls are served by the host-pointer variants of the C ABI (msda_host_*; the reference raises
    # "Not implemented on the CPU", restored by MSDA_HIP_STRICT_DEVICE=1 -- tests/test_host_logic_cpu.py); GPU tensors
    # never take that route, and mixed devices are an error as in the reference
    cpu_out = MSDA.ms_deform_attn_forward(v.cpu(), sh.cpu(), lsi.cpu(), loc.cpu(), attn.cpu(), 64)
    assert not cpu_out.is_cuda and max_abs(_np(cpu_out), g["out"]) < 1e-4
    with pytest.raises(RuntimeError, match="value is on the CPU"):
        MSDA.ms_deform_attn_forward(v.cpu(), sh, lsi.cpu(), loc.cpu(), attn.cpu(), 64)
    with pytest.raises(RuntimeError, match="contiguous"):
        MSDA.ms_deform_attn_forward(v.transpose(1, 2).contiguous().transpose(1, 2), sh, lsi, loc, attn, 64)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        MSDA.ms_deform_attn_forward(v, sh.cpu(), lsi, loc, attn, 64)
    with pytest.raises(RuntimeError, match="must divide"):
        MSDA.ms_deform_attn_forward(torch.cat([v, v[:1]]), sh, lsi, torch.cat([loc, loc[:1]]),
                                    torch.cat([attn, attn[:1]]), 2)  # batch 3, im2col_step 2 (cu:50-52)
    with pytest.raises(RuntimeError, match="dtype"):
        MSDA.ms_deform_attn_forward(v.half(), sh, lsi, loc.half(), attn.half(), 64)
    # empty query set / empty batch: no launch, empty result
    out = MSDA.ms_deform_attn_forward(v, sh, lsi, loc[:, :0].contiguous(), attn[:, :0].contiguous(), 64)
    assert out.shape == (2, 0, 256)
    out = MSDA.ms_deform_attn_forward(v[:0], sh, lsi, loc[:0], attn[:0], 64)
    assert out.shape == (0, 37, 256)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, sh, lsi, loc[:, :0].contiguous(), attn[:, :0].contiguous(),
                                              go[:, :0].contiguous(), 64)
    assert gv.shape == v.shape and float(gv.abs().sum()) == 0.0 and gl.numel() == 0 and ga.numel() == 0


def test_nonfinite_and_huge_locations(dev, api):
    """NaN / inf / 1e30 sampling locations are 'outside' (the reference's range test is false for them)."""
    from oracle import msda_oracle
    MSDA, _ = api
    g = load_golden("d32_l4_p4")
    v, sh, lsi, loc, attn, go = _to(g, dev, torch.float32)
    loc = loc.clone()
    loc[0, 0, 0, 0, 0, 0] = float("nan")
    loc[0, 1, 1, 1, 1, 1] = float("inf")
    loc[1, 2, 2, 2, 2, 0] = -1e30
    loc[1, 3, 3, 3, 3, 1] = 3e9
    out = MSDA.ms_deform_attn_forward(v, sh, lsi, loc, attn, 64)
    ref = msda_oracle.forward(v, sh, lsi, loc, attn)
    assert torch.isfinite(out).all() and max_abs(_np(out), ref) < 1e-4
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, sh, lsi, loc, attn, go, 64)
    ogv, ogl, oga = msda_oracle.backward(go, v, sh, lsi, loc, attn)
    assert torch.isfinite(gv).all() and torch.isfinite(gl).all() and torch.isfinite(ga).all()
    assert scaled_err(_np(gv), ogv) < 1e-4 and scaled_err(_np(ga), oga) < 1e-4 and grad_loc_err(_np(gl), ogl, _np(sh)) < 1.0


def test_runs_on_current_stream_and_in_graph(dev, api):
    """Kernels are enqueued on PyTorch's current stream (ms_deform_attn_cuda.cu:65) and are capturable."""
    MSDA, _ = api
    g = load_golden("d32_l4_p4")
    v, sh, lsi, loc, attn, _ = _to(g, dev, torch.float32)
    base = MSDA.ms_deform_attn_forward(v, sh, lsi, loc, attn, 64)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        on_side = MSDA.ms_deform_attn_forward(v, sh, lsi, loc, attn, 64)
    s.synchronize()
    assert torch.equal(base, on_side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = MSDA.ms_deform_attn_forward(v, sh, lsi, loc, attn, 64)
    captured.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(base, captured)


# ------------------------------------------------------------------------------------------------
# the module on top (ops/modules/ms_deform_attn.py)

@pytest.mark.parametrize("ref_dim", [2, 4])
def test_module_forward_backward_vs_gridsample_port(ref_dim, dev, api):
    import torch.nn.functional as F
    from oracle.msda_gridsample import msda_gridsample
    from uninext_amd.modules import MSDeformAttn
    from uninext_amd.workloads import level_tensors
    torch.manual_seed(11)
    levels = ((10, 13), (5, 7), (3, 4), (2, 2))
    S = sum(h * w for h, w in levels)
    N, Lq = 2, 31
    attn = MSDeformAttn(256, 4, 8, 4).to(dev)
    with torch.no_grad():  # move off the zero-weight init so every projection matters
        attn.sampling_offsets.weight.normal_(0, 0.02)
        attn.attention_weights.weight.normal_(0, 0.1)
    query = torch.randn(N, Lq, 256, device=dev, requires_grad=True)
    src = torch.randn(N, S, 256, device=dev, requires_grad=True)
    ref_pts = torch.rand(N, Lq, 4, ref_dim, device=dev)
    if ref_dim == 4:
        ref_pts[..., 2:] = ref_pts[..., 2:] * 0.3 + 0.05
    mask = torch.zeros(N, S, dtype=torch.bool, device=dev)
    mask[1, -7:] = True
    sh, lsi = level_tensors(levels, dev)
    out = attn(query, ref_pts, src, sh, lsi, mask)
    go = torch.randn_like(out)
    params = list(attn.parameters())
    grads = torch.autograd.grad(out, [query, src] + params, go)

    # the same computation with the sampling core swapped for the grid_sample port
    value = attn.value_proj(src).masked_fill(mask[..., None], 0.0).view(N, S, 8, 32)
    off = attn.sampling_offsets(query).view(N, Lq, 8, 4, 4, 2)
    w = F.softmax(attn.attention_weights(query).view(N, Lq, 8, 16), -1).view(N, Lq, 8, 4, 4)
    if ref_dim == 2:
        wh = torch.stack([sh[..., 1], sh[..., 0]], -1)
        loc = ref_pts[:, :, None, :, None, :] + off / wh[None, None, None, :, None, :]
    else:
        loc = ref_pts[:, :, None, :, None, :2] + off / 4 * ref_pts[:, :, None, :, None, 2:] * 0.5
    ref_out = attn.output_proj(msda_gridsample(value, levels, loc, w))
    ref_grads = torch.autograd.grad(ref_out, [query, src] + params, go)
    assert float((out - ref_out).abs().max()) < 1e-4
    for a, b in zip(grads, ref_grads):
        assert float((a - b).abs().max()) < 2e-4 * max(1.0, float(b.abs().max()))


# ------------------------------------------------------------------------------------------------
# fused prologue: softmax + sampling locations inside the kernel

@pytest.mark.parametrize("ref_dim", [2, 4])
def test_fused_forward_vs_oracle_and_unfused(ref_dim, dev, api):
    import torch.nn.functional as F
    from oracle import msda_oracle
    from uninext_amd import ext, workloads
    MSDA, lib = api
    levels = ((25, 42), (13, 21), (7, 11), (4, 6))
    S = sum(h * w for h, w in levels)
    N, M, L, P = 2, 8, 4, 4
    Lq = S if ref_dim == 2 else 300
    g = torch.Generator().manual_seed(50 + ref_dim)
    value = torch.randn(N, S, M, 32, generator=g).to(dev)
    offsets = (torch.randn(N, Lq, M * L * P * 2, generator=g) * 2.5).to(dev)
    logits = (torch.randn(N, Lq, M * L * P, generator=g) * 3.0).to(dev)
    logits[0, 1, :16] = 80.0   # equal large logits: softmax must not overflow
    if ref_dim == 2:
        ref = workloads.encoder_reference_points(levels, dev)[None, :, None, :].expand(N, S, L, 2).contiguous()
    else:
        ref = torch.rand(N, Lq, L, 4, generator=g).to(dev)
        ref[..., 2:] = ref[..., 2:] * 0.4 + 0.02
    sh, lsi = workloads.level_tensors(levels, dev)
    assert ext.fused_forward_supported(value, ref, L, P)
    out = ext.ms_deform_attn_forward_fused(value, sh, lsi, ref, offsets, logits, P)
    # encoder-shaped calls take the window or the gather kernel (by the reported locality), small ones the lane-group one
    assert lib.last_kernel("forward") in (("msda_fwd_lg3_fused", "msda_fwd_win_fused") if Lq >= 1024 else ("msda_fwd_fused",))
    # the reference's prologue (ops/modules/ms_deform_attn.py:99-112) in torch, then the C oracle
    off = offsets.view(N, Lq, M, L, P, 2)
    attn = F.softmax(logits.view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
    if ref_dim == 2:
        wh = torch.stack([sh[..., 1], sh[..., 0]], -1)
        loc = ref[:, :, None, :, None, :] + off / wh[None, None, None, :, None, :]
    else:
        loc = ref[:, :, None, :, None, :2] + off / P * ref[:, :, None, :, None, 2:] * 0.5
    oracle_out = msda_oracle.forward(value, sh, lsi, loc.contiguous(), attn.contiguous())
    assert max_abs(_np(out), oracle_out) < 1e-4
    unfused = MSDA.ms_deform_attn_forward(value, sh, lsi, loc.contiguous(), attn.contiguous(), 64)
    assert float((out - unfused).abs().max()) < 2e-5


def test_fused_forward_rejects_unsupported_geometry(dev, api):
    from uninext_amd import ext, workloads
    levels = ((6, 5), (3, 3))
    sh, lsi = workloads.level_tensors(levels, dev)
    value = torch.randn(1, 39, 2, 16, device=dev)            # 16 channels per head: no fused kernel
    ref = torch.rand(1, 5, 2, 2, device=dev)
    assert not ext.fused_forward_supported(value, ref, 2, 8)
    with pytest.raises(RuntimeError, match="fused forward needs"):
        ext.ms_deform_attn_forward_fused(value, sh, lsi, ref, torch.zeros(1, 5, 2 * 2 * 8 * 2, device=dev),
                                         torch.zeros(1, 5, 2 * 2 * 8, device=dev), 8)


@pytest.mark.parametrize("ref_dim", [2, 4])
def test_module_inference_uses_fused_kernel_and_matches_autograd_path(ref_dim, dev, api):
    from uninext_amd.modules import MSDeformAttn
    from uninext_amd.workloads import level_tensors
    _, lib = api
    torch.manual_seed(12)
    levels = ((10, 13), (5, 7), (3, 4), (2, 2))
    S = sum(h * w for h, w in levels)
    N, Lq = 2, 31
    layer = MSDeformAttn(256, 4, 8, 4).to(dev).eval()
    with torch.no_grad():
        layer.sampling_offsets.weight.normal_(0, 0.02)
        layer.attention_weights.weight.normal_(0, 0.1)
    query, src = torch.randn(N, Lq, 256, device=dev), torch.randn(N, S, 256, device=dev)
    ref = torch.rand(N, Lq, 4, ref_dim, device=dev)
    mask = torch.zeros(N, S, dtype=torch.bool, device=dev)
    mask[0, :9] = True
    sh, lsi = level_tensors(levels, dev)
    with torch.no_grad():
        fused = layer(query, ref, src, sh, lsi, mask)
        assert lib.last_kernel("forward") == "msda_fwd_fused"          # 31 queries: the small-call fused kernel
        MSDeformAttn.fuse_prologue = False
        try:
            plain = layer(query, ref, src, sh, lsi, mask)
        finally:
            MSDeformAttn.fuse_prologue = True
        assert lib.last_kernel("forward") != "msda_fwd_fused"
    assert float((fused - plain).abs().max()) < 2e-5
    # with gradients required the module takes a differentiable path: the fused Function (same forward kernel) by default,
    # the reference's data flow (PyTorch prologue + MSDeformAttnFunction) with fuse_training_prologue off
    q2 = query.clone().requires_grad_(True)
    out = layer(q2, ref, src, sh, lsi, mask)
    assert lib.last_kernel("forward") == "msda_fwd_fused"
    out.sum().backward()
    assert q2.grad is not None and torch.isfinite(q2.grad).all()
    MSDeformAttn.fuse_training_prologue = False
    try:
        q3 = query.clone().requires_grad_(True)
        out3 = layer(q3, ref, src, sh, lsi, mask)
        assert lib.last_kernel("forward") != "msda_fwd_fused"
        out3.sum().backward()
    finally:
        MSDeformAttn.fuse_training_prologue = True
    assert float((out - out3).abs().max()) < 2e-5 and float((q2.grad - q3.grad).abs().max()) < 1e-4 * max(1.0, float(q3.grad.abs().max()))


def test_module_under_inference_mode(dev, api, split_bf16_paths):
    """torch.inference_mode(): tensors carry no version counter (`_version` raises).  The module -- shapes tensor built
    inside the forward, parameters loaded under inference_mode -- must run as it does under no_grad."""
    from uninext_amd.modules import MSDeformAttn
    from uninext_amd.workloads import encoder_reference_points
    torch.manual_seed(2)
    levels = ((40, 53), (20, 27), (10, 14), (5, 7))          # S = 2835 >= 1024: the encoder-sized fused path
    S = sum(h * w for h, w in levels)
    src = torch.randn(2, S, 256, device=dev)
    ref = encoder_reference_points(levels, dev)[None, :, None, :].expand(2, S, 4, 2).contiguous()
    with torch.inference_mode():
        m = MSDeformAttn(256, 4, 8, 4).to(dev)               # inference parameters
        sh = torch.as_tensor(levels, dtype=torch.long, device=dev)
        lsi = torch.cat((sh.new_zeros((1,)), sh.prod(1).cumsum(0)[:-1]))
        a = m(src, ref, src, sh, lsi, None)
        b = m(src, ref, src, sh, lsi, None)                  # cached packed weights, cached shape check
    m2 = MSDeformAttn(256, 4, 8, 4).to(dev)
    m2.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})
    sh2 = torch.as_tensor(levels, dtype=torch.long, device=dev)
    with torch.no_grad():
        c = m2(src, ref, src, sh2, torch.cat((sh2.new_zeros((1,)), sh2.prod(1).cumsum(0)[:-1])), None)
    assert torch.equal(a, b)
    assert float((a - c).abs().max()) < 1e-6


def test_bare_operator_under_inference_mode_without_a_call_site(dev):
    """INTEGRATION option A -- the operator called with no call_site() block (the reference's unmodified module
    does that) on an encoder-shaped fp32 call whose shapes tensor was built under torch.inference_mode().  The derived-site
    logic must not touch `_version` of an inference tensor; results equal those under no_grad."""
    from uninext_amd import ext, workloads
    levels = ((40, 53), (20, 27), (10, 14), (5, 7))          # S = 2835 >= 1024: takes a call context
    x = workloads.make_inputs("encoder", batch=1, levels=levels, seed=5, device=dev)
    with torch.no_grad():
        want = ext.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
    with torch.inference_mode():
        sh = torch.as_tensor(levels, dtype=torch.long, device=dev)
        lsi = torch.cat((sh.new_zeros((1,)), sh.prod(1).cumsum(0)[:-1]))
        outs = [ext.ms_deform_attn_forward(x["value"], sh, lsi, x["loc"], x["attn"], 64) for _ in range(4)]
    assert ext.last_call_site() >= ext.AUTO_SITE_BASE
    for o in outs:
        assert float((o - want).abs().max()) < 1e-5


# ------------------------------------------------------------------------------------------------
# training-side prologue (include/msda_hip.h: msda_hip_prologue_f32 / msda_hip_prologue_backward_f32, MSDeformAttnFusedFunction)

def _torch_prologue(shapes, ref, offsets, logits, M, P):
    """ops/modules/ms_deform_attn.py:99-112."""
    N, Lq = offsets.shape[:2]
    L = shapes.shape[0]
    off = offsets.view(N, Lq, M, L, P, 2)
    w = torch.nn.functional.softmax(logits.view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
    if ref.shape[-1] == 2:
        wh = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
        loc = ref[:, :, None, :, None, :] + off / wh[None, None, None, :, None, :]
    else:
        loc = ref[:, :, None, :, None, :2] + off / P * ref[:, :, None, :, None, 2:] * 0.5
    return loc, w


@pytest.mark.parametrize("ref_dim", [2, 4])
@pytest.mark.parametrize("M,L,P", [(8, 4, 4), (3, 2, 5), (1, 1, 2)])
def test_prologue_kernels_vs_the_pytorch_composition(ref_dim, M, L, P, dev):
    from uninext_amd import ext
    g = torch.Generator().manual_seed(7 + M)
    N, Lq = 2, 333
    levels = ((30, 41), (15, 21), (8, 11), (4, 6))[:L]
    shapes = torch.as_tensor(levels, dtype=torch.int64, device=dev)
    ref = torch.rand(N, Lq, L, ref_dim, generator=g).to(dev)
    offsets = (torch.randn(N, Lq, M * L * P * 2, generator=g) * 3).to(dev).requires_grad_(True)
    logits = (torch.randn(N, Lq, M * L * P, generator=g) * 2).to(dev).requires_grad_(True)
    ref.requires_grad_(True)
    loc_t, w_t = _torch_prologue(shapes, ref, offsets, logits, M, P)
    loc, w = ext.msda_prologue(shapes, ref.detach(), offsets.detach(), logits.detach(), M, P)
    assert loc.shape == loc_t.shape and w.shape == w_t.shape
    assert float((loc - loc_t).abs().max()) < 1e-6 and float((w - w_t).abs().max()) < 1e-6
    g_loc = torch.randn(loc.shape, generator=g).to(dev)
    g_w = torch.randn(w.shape, generator=g).to(dev)
    want_off, want_logits, want_ref = torch.autograd.grad([loc_t, w_t], [offsets, logits, ref], [g_loc, g_w])
    g_off, g_logits, g_ref = ext.msda_prologue_backward(shapes, ref.detach(), offsets.detach(), w, g_loc, g_w, need_grad_reference=True)
    assert g_off.shape == want_off.shape and g_logits.shape == want_logits.shape and g_ref.shape == want_ref.shape
    assert float((g_off - want_off).abs().max()) < 1e-5 * max(1.0, float(want_off.abs().max()))
    assert float((g_logits - want_logits).abs().max()) < 1e-5 * max(1.0, float(want_logits.abs().max()))
    assert float((g_ref - want_ref).abs().max()) < 1e-4 * max(1.0, float(want_ref.abs().max()))      # a sum of M * P float32 terms
    assert ext.msda_prologue_backward(shapes, ref.detach(), offsets.detach(), w, g_loc, g_w)[2] is None


@pytest.mark.parametrize("ref_dim", [2, 4])
@pytest.mark.parametrize("encoder", [True, False])
def test_fused_training_function_vs_the_reference_data_flow(ref_dim, encoder, dev, api):
    """MSDeformAttnFusedFunction (fused forward from the raw tensors; backward = prologue kernel -> operator backward ->
    prologue backward) against the reference's data flow (PyTorch prologue + MSDeformAttnFunction): outputs and ALL gradients
    (value, reference points, offsets, logits), on an encoder-sized call (window / gather kernels) and a decoder-sized one."""
    from uninext_amd.functions import MSDeformAttnFunction, MSDeformAttnFusedFunction
    from uninext_amd.workloads import level_tensors
    _, lib = api
    g = torch.Generator().manual_seed(21)
    levels = ((40, 53), (20, 27), (10, 14), (5, 7))
    S = sum(h * w for h, w in levels)
    N, M, L, P = 2, 8, 4, 4
    Lq = S if encoder else 300
    sh, lsi = level_tensors(levels, dev)
    value = torch.randn(N, S, M, 32, generator=g).to(dev).requires_grad_(True)
    ref = torch.rand(N, Lq, L, ref_dim, generator=g)
    if ref_dim == 4:
        ref[..., 2:] = ref[..., 2:] * 0.3 + 0.05
    ref = ref.to(dev).requires_grad_(True)
    offsets = (torch.randn(N, Lq, M * L * P * 2, generator=g) * 2).to(dev).requires_grad_(True)
    logits = torch.randn(N, Lq, M * L * P, generator=g).to(dev).requires_grad_(True)
    go = torch.randn(N, Lq, M * 32, generator=g).to(dev)
    out_f = MSDeformAttnFusedFunction.apply(value, sh, lsi, ref, offsets, logits, P)
    kernel = lib.last_kernel("forward")
    assert "fused" in kernel, kernel
    grads_f = torch.autograd.grad(out_f, [value, ref, offsets, logits], go)
    loc, w = _torch_prologue(sh, ref, offsets, logits, M, P)
    out_r = MSDeformAttnFunction.apply(value, sh, lsi, loc, w, 64)
    grads_r = torch.autograd.grad(out_r, [value, ref, offsets, logits], go)
    assert float((out_f - out_r).abs().max()) < 1e-4
    for name, a, b in zip(("value", "reference_points", "offsets", "logits"), grads_f, grads_r):
        scale = max(1.0, float(b.abs().max()))
        err = float((a - b).abs().max())
        print("%-16s max |fused - reference flow| %.2e (scale %.1f)" % (name, err, scale))
        assert err < 2e-4 * scale, (name, err, scale)
