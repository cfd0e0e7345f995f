"""The reference's own Python layers run UNMODIFIED on top of this repository's drop-in module.

Skipped where /root/reference is absent (the GPU box).  The reference files are loaded from where they lie
(nothing is copied): ops/functions/ms_deform_attn_func.py does `import MultiScaleDeformableAttention as MSDA`
(:18), which -- with this repository on sys.path -- resolves to ./MultiScaleDeformableAttention.py, i.e. to
uninext_amd.ext and the C ABI.  CPU tensors are served by the host-pointer variants (msda_host_*), so the reference's
`MSDeformAttnFunction` and `MSDeformAttn` can be executed here, end to end, against the reference's own
`ms_deform_attn_core_pytorch` -- the procedure of ops/test.py:31-57 without `.cuda()`."""
import importlib.util
import os
import sys
import types
import warnings

import pytest
import torch

OPS = "/root/reference/projects/UNINEXT/uninext/models/deformable_detr/ops"
pytestmark = pytest.mark.skipif(not os.path.isdir(OPS), reason="reference checkout not present")


@pytest.fixture(scope="module")
def ref():
    """Synthetic package `refops` with the reference's functions/ and modules/ loaded from /root/reference."""
    for k in [k for k in sys.modules if k == "refops" or k.startswith("refops.")]:
        del sys.modules[k]
    pkg = types.ModuleType("refops"); pkg.__path__ = [OPS]
    sys.modules["refops"] = pkg
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # torch.cuda.amp.custom_fwd is deprecated in torch 2.10 (still works)
        for sub in ("functions", "modules"):
            spec = importlib.util.spec_from_file_location("refops." + sub, os.path.join(OPS, sub, "__init__.py"),
                                                          submodule_search_locations=[os.path.join(OPS, sub)])
            mod = importlib.util.module_from_spec(spec)
            sys.modules["refops." + sub] = mod
            spec.loader.exec_module(mod)
    yield sys.modules["refops.functions.ms_deform_attn_func"], sys.modules["refops.modules.ms_deform_attn"]
    for k in [k for k in sys.modules if k == "refops" or k.startswith("refops.")]:
        del sys.modules[k]


def test_reference_function_binds_to_this_library(ref):
    func, _ = ref
    from uninext_amd import ext
    assert func.MSDA.__name__ == "MultiScaleDeformableAttention"
    assert func.MSDA.ms_deform_attn_forward is ext.ms_deform_attn_forward
    assert func.MSDA.ms_deform_attn_backward is ext.ms_deform_attn_backward
    assert func.__file__.startswith("/root/reference/")


def _testpy_inputs(dtype):
    """ops/test.py:21-36."""
    N, M, D = 1, 2, 2
    Lq, L, P = 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = sum((H * W).item() for H, W in shapes)
    torch.manual_seed(3)
    value = torch.rand(N, S, M, D) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    attn = torch.rand(N, Lq, M, L, P) + 1e-5
    attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    return value.to(dtype), shapes, lsi, loc.to(dtype), attn.to(dtype)


def test_reference_function_forward_equals_its_pytorch_path(ref):
    """check_forward_equal_with_pytorch_double / _float (ops/test.py:31-57) through the reference's own Function."""
    func, _ = ref
    for dtype, tol in ((torch.float64, 1e-12), (torch.float32, 1e-6)):
        value, shapes, lsi, loc, attn = _testpy_inputs(dtype)
        want = func.ms_deform_attn_core_pytorch(value, shapes, loc, attn)
        got = func.MSDeformAttnFunction.apply(value, shapes, lsi, loc, attn, 2)
        assert torch.allclose(got, want, rtol=1e-2, atol=1e-3)          # the reference's own criterion
        assert (got - want).abs().max().item() < tol


def test_reference_function_gradcheck(ref):
    """check_gradient_numerical (ops/test.py:60-76), channels 30 and 32."""
    func, _ = ref
    for D in (30, 32):
        N, M, Lq, L, P = 1, 2, 2, 2, 2
        shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
        lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
        S = int(shapes.prod(1).sum())
        torch.manual_seed(3)
        value = (torch.rand(N, S, M, D, dtype=torch.float64) * 0.01).requires_grad_(True)
        loc = torch.rand(N, Lq, M, L, P, 2, dtype=torch.float64).requires_grad_(True)
        attn = torch.rand(N, Lq, M, L, P, dtype=torch.float64) + 1e-5
        attn = (attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)).requires_grad_(True)
        assert torch.autograd.gradcheck(func.MSDeformAttnFunction.apply, (value, shapes, lsi, loc, attn, 2))


def test_reference_module_runs_on_this_library_and_matches_the_mirror(ref):
    """The class the reference's transformer constructs (deformable_transformer_dino.py:338,380) on top of this
    library, against uninext_amd.modules.MSDeformAttn with the same state_dict."""
    _, refmod = ref
    from uninext_amd import workloads
    from uninext_amd.modules import MSDeformAttn
    torch.manual_seed(1)
    theirs = refmod.MSDeformAttn(256, 4, 8, 4)
    ours = MSDeformAttn(256, 4, 8, 4)
    with torch.no_grad():
        theirs.sampling_offsets.weight.normal_(0, 0.02)
        theirs.attention_weights.weight.normal_(0, 0.05)
    assert set(theirs.state_dict()) == set(ours.state_dict())
    ours.load_state_dict(theirs.state_dict())              # reference checkpoints load unchanged
    levels = ((9, 12), (5, 6), (3, 3), (2, 2))
    S = sum(h * w for h, w in levels)
    shapes, lsi = workloads.level_tensors(levels, "cpu")
    src = torch.randn(2, S, 256)
    refpts = workloads.encoder_reference_points(levels, "cpu")[None, :, None, :].expand(2, S, 4, 2).contiguous()
    mask = torch.zeros(2, S, dtype=torch.bool)
    mask[0, :7] = True
    qa, qb = src.clone().requires_grad_(True), src.clone().requires_grad_(True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = theirs(qa, refpts, src, shapes, lsi, mask)
    b = ours(qb, refpts, src, shapes, lsi, mask)
    assert (a - b).abs().max().item() < 1e-5
    a.sum().backward(); b.sum().backward()
    assert (qa.grad - qb.grad).abs().max().item() < 1e-4 * max(1.0, qa.grad.abs().max().item())
    for (n, p), (_, r) in zip(sorted(ours.named_parameters()), sorted(theirs.named_parameters())):
        assert (p.grad - r.grad).abs().max().item() < 1e-4 * max(1.0, r.grad.abs().max().item()), n
    # decoder form: 4-d reference boxes
    ref4 = torch.rand(2, 11, 4, 4) * 0.5 + 0.25
    q = torch.randn(2, 11, 256)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = theirs(q, ref4, src, shapes, lsi, None)
    assert (a - ours(q, ref4, src, shapes, lsi, None)).abs().max().item() < 1e-5
    with pytest.raises(ValueError):
        ours(q, torch.rand(2, 11, 4, 3), src, shapes, lsi, None)
