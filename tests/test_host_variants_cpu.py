"""Host-pointer variants of the C ABI (include/msda_hip.h: msda_host_*, uninext_amd/csrc/msda_host.cpp) -- SURVEY.md
8(b)(i), BASELINE configs[0].  They run on the CPU, so they are checked here, without a GPU: against the
reference-minted golden fixtures, against the C oracle on seeded R50-shaped workloads, through the autograd Function
and through the MSDeformAttn module (vs the reference's grid_sample composition, restated in oracle/)."""
import numpy as np
import pytest
import torch

from golden_util import golden_names, load_golden, max_abs, scaled_err

NAMES = golden_names()


def _t(g, dtype):
    f = lambda k: torch.from_numpy(g[k]).to(dtype).contiguous()
    i = lambda k: torch.from_numpy(g[k])
    return f("value"), i("shapes"), i("lsi"), f("loc"), f("attn"), f("grad_out")


@pytest.mark.parametrize("name", NAMES)
def test_golden_f64(name):
    import MultiScaleDeformableAttention as MSDA
    g = load_golden(name)
    v, sh, lsi, loc, attn, go = _t(g, torch.float64)
    out = MSDA.ms_deform_attn_forward(v, sh, lsi, loc, attn, 64)
    assert out.shape == g["out"].shape and max_abs(out.numpy(), g["out"]) < 1e-12
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, sh, lsi, loc, attn, go, 64)
    assert max_abs(gv.numpy(), g["grad_value"]) < 1e-11 and max_abs(ga.numpy(), g["grad_attn"]) < 1e-11
    if name != "border":   # one-sided derivative convention exactly on cell edges (tests/test_oracle_golden.py)
        assert max_abs(gl.numpy(), g["grad_loc"]) < 1e-9


@pytest.mark.parametrize("name", NAMES)
def test_golden_f32(name):
    import MultiScaleDeformableAttention as MSDA
    g = load_golden(name)
    v, sh, lsi, loc, attn, go = _t(g, torch.float32)
    out = MSDA.ms_deform_attn_forward(v, sh, lsi, loc, attn, 64)
    assert max_abs(out.numpy(), g["out"]) < 1e-4
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, sh, lsi, loc, attn, go, 64)
    assert scaled_err(gv.numpy(), g["grad_value"]) < 1e-4 and scaled_err(ga.numpy(), g["grad_attn"]) < 1e-4
    if name != "border":
        wh = float(np.max(g["shapes"]))
        assert max_abs(gl.numpy(), g["grad_loc"]) < 1e-4 * wh * max(1.0, float(np.abs(g["grad_loc"]).max()) / wh)


def test_border_fixture_follows_the_cuda_formula():
    from oracle import msda_oracle
    import MultiScaleDeformableAttention as MSDA
    g = load_golden("border")
    v, sh, lsi, loc, attn, go = _t(g, torch.float64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, sh, lsi, loc, attn, go, 64)
    ogv, ogl, oga = msda_oracle.backward(g["grad_out"], g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"])
    assert max_abs(gv.numpy(), ogv) < 1e-12 and max_abs(gl.numpy(), ogl) < 1e-11 and max_abs(ga.numpy(), oga) < 1e-12


@pytest.mark.parametrize("kind,flavour", [("encoder", "model"), ("encoder", "uniform"), ("decoder", "model")])
def test_seeded_workload_vs_c_oracle(kind, flavour):
    from oracle import msda_oracle
    from uninext_amd import workloads
    import MultiScaleDeformableAttention as MSDA
    levels = ((25, 42), (13, 21), (7, 11), (4, 6))
    x = workloads.make_inputs(kind, flavour, batch=2, levels=levels, num_query=None if kind == "encoder" else 300,
                              seed=3, device="cpu")
    out = MSDA.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
    ref = msda_oracle.forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"])
    assert max_abs(out.numpy(), ref) < 1e-4
    go = torch.randn(out.shape, generator=torch.Generator().manual_seed(4))
    gv, gl, ga = MSDA.ms_deform_attn_backward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], go, 64)
    ogv, ogl, oga = msda_oracle.backward(go, x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"])
    assert max_abs(gv.numpy(), ogv) < 1e-4 and max_abs(ga.numpy(), oga) < 1e-4
    assert max_abs(gl.numpy(), ogl) < 1e-4 * 42


def test_thread_count_does_not_change_the_result(monkeypatch):
    from uninext_amd import ext, workloads
    x = workloads.make_inputs("encoder", "model", batch=2, levels=((9, 11), (5, 6)), seed=8, device="cpu")
    go = torch.randn(2, x["loc"].shape[1], 256, generator=torch.Generator().manual_seed(1))
    res = []
    for threads in (1, 3, 0):
        monkeypatch.setattr(ext, "HOST_THREADS", threads)
        out = ext.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
        res.append([out] + ext.ms_deform_attn_backward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], go, 64))
    for r in res[1:]:
        for a, b in zip(res[0], r):
            assert torch.equal(a, b)    # one thread per (image, head) slice of grad_value: bitwise deterministic


def test_default_thread_count_follows_the_work_not_the_number_of_slices(monkeypatch):
    """the backward's units are (image, head) slices of num_query rows each; 16 slices must not mean one thread."""
    import os
    from uninext_amd import _lib, ext, workloads
    lib = _lib.load()
    hw = os.cpu_count() or 1
    if hw < 2:
        pytest.skip("one hardware thread")
    x = workloads.make_inputs("encoder", "model", batch=2, levels=((25, 42), (13, 21), (7, 11), (4, 6)), seed=8, device="cpu")
    Lq = x["loc"].shape[1]
    go = torch.randn(2, Lq, 256, generator=torch.Generator().manual_seed(1))
    monkeypatch.setattr(ext, "HOST_THREADS", 0)
    ext.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
    assert lib.msda_host_last_num_threads() == min(hw, max(1, 2 * Lq // 256))
    ext.ms_deform_attn_backward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], go, 64)
    # 2 images x 8 heads = 16 slices of 1400 rows: min(hardware threads, 16 slices, 16 * 1400 / 256 rows of work)
    assert lib.msda_host_last_num_threads() == min(hw, 16)
    # a tiny call still gets one thread
    y = workloads.make_inputs("decoder", "model", batch=1, levels=((9, 11), (5, 6)), num_query=20, seed=8, device="cpu")
    ext.ms_deform_attn_backward(y["value"], y["shapes"], y["lsi"], y["loc"], y["attn"], torch.randn(1, 20, 256), 64)
    assert lib.msda_host_last_num_threads() == 1


def test_autograd_function_and_gradcheck_on_cpu():
    """ops/test.py:60-76 (check_gradient_numerical) on the host variants."""
    from uninext_amd.functions import MSDeformAttnFunction
    torch.manual_seed(3)
    N, M, D, Lq, L, P = 1, 2, 4, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    value = (torch.rand(N, S, M, D, dtype=torch.float64) * 0.01).requires_grad_(True)
    loc = torch.rand(N, Lq, M, L, P, 2, dtype=torch.float64).requires_grad_(True)
    attn = torch.rand(N, Lq, M, L, P, dtype=torch.float64) + 1e-5
    attn = (attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)).requires_grad_(True)
    assert torch.autograd.gradcheck(MSDeformAttnFunction.apply, (value, shapes, lsi, loc, attn, 2))


def test_module_on_cpu_matches_the_grid_sample_composition():
    """BASELINE configs[0]: MSDeformAttn on CPU tensors (the reference module cannot run there at all)."""
    from oracle.msda_gridsample import msda_gridsample
    from uninext_amd import workloads
    from uninext_amd.modules import MSDeformAttn
    torch.manual_seed(0)
    levels = ((10, 13), (5, 7), (3, 4), (2, 2))
    S = sum(h * w for h, w in levels)
    shapes, lsi = workloads.level_tensors(levels, "cpu")
    m = MSDeformAttn(256, 4, 8, 4)
    with torch.no_grad():
        m.sampling_offsets.weight.normal_(0, 0.02)
        m.attention_weights.weight.normal_(0, 0.05)
    src = torch.randn(2, S, 256)
    ref = workloads.encoder_reference_points(levels, "cpu")[None, :, None, :].expand(2, S, 4, 2).contiguous()
    mask = torch.zeros(2, S, dtype=torch.bool)
    mask[1, -5:] = True
    q = src.clone().requires_grad_(True)
    out = m(q, ref, src, shapes, lsi, mask)
    # the reference's own data flow with its CPU sampling function
    value = m.value_proj(src).masked_fill(mask[..., None], 0.0).view(2, S, 8, 32)
    off = m.sampling_offsets(q).view(2, S, 8, 4, 4, 2)
    w = torch.softmax(m.attention_weights(q).view(2, S, 8, 16), -1).view(2, S, 8, 4, 4)
    wh = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
    loc = ref[:, :, None, :, None, :] + off / wh[None, None, None, :, None, :]
    want = m.output_proj(msda_gridsample(value, [tuple(r) for r in shapes.tolist()], loc, w))
    assert max_abs(out.detach().numpy(), want.detach().numpy()) < 1e-4
    g1, = torch.autograd.grad(out.sum(), q, retain_graph=True)
    g2, = torch.autograd.grad(want.sum(), q)
    assert scaled_err(g1.numpy(), g2.numpy()) < 1e-4


def test_host_variants_reject_levels_outside_the_value_tensor():
    """the host-pointer entry points index plain memory with spatial_shapes / level_start_index; geometry that
    does not fit spatial_size must be refused (MSDA_ERR_BAD_DIMS), not dereferenced."""
    import torch
    from uninext_amd import ext
    value = torch.randn(1, 20, 2, 4)
    shapes = torch.tensor([[4, 4], [2, 2]], dtype=torch.int64)
    loc = torch.rand(1, 3, 2, 2, 2, 2)
    attn = torch.softmax(torch.randn(1, 3, 2, 4), -1).view(1, 3, 2, 2, 2)
    ok = ext.ms_deform_attn_forward(value, shapes, torch.tensor([0, 16], dtype=torch.int64), loc, attn, 64)
    assert ok.shape == (1, 3, 8)
    for bad_lsi in ([0, 17], [-1, 16], [0, 1 << 40], [0, (1 << 63) - 1], [0, (1 << 63) - 3]):   # (start + H * W would wrap)
        with pytest.raises(RuntimeError):
            ext.ms_deform_attn_forward(value, shapes, torch.tensor(bad_lsi, dtype=torch.int64), loc, attn, 64)
        with pytest.raises(RuntimeError):
            ext.ms_deform_attn_backward(value, shapes, torch.tensor(bad_lsi, dtype=torch.int64), loc, attn, torch.randn(1, 3, 8), 64)
    with pytest.raises(RuntimeError):
        ext.ms_deform_attn_forward(value, torch.tensor([[4, 4], [0, 2]], dtype=torch.int64), torch.tensor([0, 16], dtype=torch.int64), loc, attn, 64)
