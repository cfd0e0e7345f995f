"""CPU: host-side mirror of the reference interface (no GPU compute)."""
import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_tensors_take_the_host_variants_or_the_reference_error(monkeypatch):
    """ops/src/ms_deform_attn.h:35-38 raises 'Not implemented on the CPU'.  Here CPU tensors are served by the
    host-pointer variants of the C ABI (msda_host_*); MSDA_HIP_STRICT_DEVICE=1 restores the
    reference's error.  The device-only fused entry point keeps raising."""
    import MultiScaleDeformableAttention as MSDA
    from uninext_amd import ext
    v = torch.ones(1, 2, 1, 4)
    args = (v, torch.tensor([[1, 2]]), torch.tensor([0]), torch.full((1, 1, 1, 1, 1, 2), 0.5), torch.ones(1, 1, 1, 1, 1))
    out = MSDA.ms_deform_attn_forward(*args, 64)
    assert out.shape == (1, 1, 4) and torch.allclose(out, torch.ones(1, 1, 4))
    gv, gl, ga = MSDA.ms_deform_attn_backward(*args, torch.ones(1, 1, 4), 64)
    assert gv.shape == v.shape and abs(float(gv.sum()) - 4.0) < 1e-6 and abs(float(ga) - 4.0) < 1e-6
    monkeypatch.setattr(ext, "STRICT_DEVICE", True)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_forward(*args, 64)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_backward(*args, torch.zeros(1, 1, 4), 64)
    monkeypatch.setattr(ext, "STRICT_DEVICE", False)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        ext.ms_deform_attn_forward_fused(v, args[1], args[2], torch.zeros(1, 1, 1, 2), torch.zeros(1, 1, 2), torch.zeros(1, 1, 1), 1)


def test_module_name_and_exports_match_the_reference_extension():
    """ops/src/vision.cpp:13-16 exports exactly two functions under the module name the reference imports."""
    import MultiScaleDeformableAttention as MSDA
    assert MSDA.__name__ == "MultiScaleDeformableAttention"
    assert sorted(MSDA.__all__) == ["ms_deform_attn_backward", "ms_deform_attn_forward"]


def test_product_package_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under uninext_amd/ (or the drop-in module) may import it."""
    import re
    pkg = os.path.join(ROOT, "uninext_amd")
    files = [os.path.join(ROOT, "MultiScaleDeformableAttention.py")]
    for d, _, names in os.walk(pkg):
        files += [os.path.join(d, n) for n in names if n.endswith(".py")]
    for f in files:
        text = open(f).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
        assert "grid_sample" not in text, f


def test_missing_library_fails_loudly(monkeypatch):
    from uninext_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmsda_hip.so")
    with pytest.raises(RuntimeError, match="There is no fallback"):
        _lib.load()


def test_msdeformattn_constructor_and_init():
    """ops/modules/ms_deform_attn.py:30-76: parameter names/shapes and the initial sampling pattern."""
    from uninext_amd.modules import MSDeformAttn
    m = MSDeformAttn(256, 4, 8, 4)
    sd = m.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == {
        "sampling_offsets.weight": (256, 256), "sampling_offsets.bias": (256,),
        "attention_weights.weight": (128, 256), "attention_weights.bias": (128,),
        "value_proj.weight": (256, 256), "value_proj.bias": (256,),
        "output_proj.weight": (256, 256), "output_proj.bias": (256,)}
    assert m.im2col_step == 64
    assert float(sd["sampling_offsets.weight"].abs().max()) == 0.0
    assert float(sd["attention_weights.weight"].abs().max()) == 0.0 and float(sd["attention_weights.bias"].abs().max()) == 0.0
    bias = sd["sampling_offsets.bias"].view(8, 4, 4, 2)
    for h in range(8):
        th = h * 2 * math.pi / 8
        d = torch.tensor([math.cos(th), math.sin(th)])
        d = d / d.abs().max()
        for p in range(4):
            assert torch.allclose(bias[h, :, p], (d * (p + 1)).expand(4, 2), atol=1e-6)
    with pytest.raises(ValueError, match="divisible"):
        MSDeformAttn(250, 4, 8, 4)


def test_msdeformattn_bad_reference_dim_raises():
    """ops/modules/ms_deform_attn.py:110-112 -- checked before the op is reached, so it runs on CPU."""
    from uninext_amd.modules import MSDeformAttn
    m = MSDeformAttn(32, 1, 2, 1)
    shapes = torch.tensor([[2, 3]])
    with pytest.raises(ValueError, match="Last dim of reference_points"):
        m(torch.zeros(1, 4, 32), torch.zeros(1, 4, 1, 3), torch.zeros(1, 6, 32), shapes, torch.tensor([0]))
    with pytest.raises(AssertionError):
        m(torch.zeros(1, 4, 32), torch.zeros(1, 4, 1, 2), torch.zeros(1, 5, 32), shapes, torch.tensor([0]))


def test_workload_shapes_and_bytes():
    from uninext_amd import workloads as w
    assert sum(h * ww for h, ww in w.R50_LEVELS_INFER) == 22223
    assert sum(h * ww for h, ww in w.R50_LEVELS_TRAIN) == 22323
    x = w.make_inputs("encoder", "model", batch=1, levels=((5, 6), (3, 3)), device="cpu", seed=3)
    assert x["value"].shape == (1, 39, 8, 32) and x["loc"].shape == (1, 39, 8, 2, 4, 2)
    assert x["lsi"].tolist() == [0, 30]
    assert torch.allclose(x["attn"].sum((-1, -2)), torch.ones(1, 39, 8))
    y = w.make_inputs("encoder", "model", batch=1, levels=((5, 6), (3, 3)), device="cpu", seed=3)
    assert torch.equal(x["loc"], y["loc"])
    # 79.65 MB / image encoder forward, 25.06 MB decoder; 136.5 / 49.2 MB backward
    assert abs(w.algorithmic_bytes_forward(1, 22223, 22223) / 1e6 - 79.65) < 0.01
    assert abs(w.algorithmic_bytes_forward(1, 22223, 900) / 1e6 - 25.06) < 0.01
    assert abs(w.algorithmic_bytes_backward(1, 22223, 22223) / 1e6 - 136.5) < 0.1
    assert abs(w.algorithmic_bytes_backward(1, 22223, 900) / 1e6 - 49.2) < 0.01


def test_named_workloads_cover_the_baseline_configs():
    """uninext_amd.workloads.WORKLOADS: one op-level shape set per BASELINE.json config that reaches the op (configs[1..4]); the
    pyramids follow from the image sizes (ceil(size / stride), strides 8 / 16 / 32 / 64) and the level start indices from them."""
    from uninext_amd import workloads
    assert {w["config"] for w in workloads.WORKLOADS.values()} == {1, 2, 3, 4}
    assert workloads.pyramid(800, 1344) == workloads.R50_LEVELS_TRAIN
    assert workloads.WORKLOADS["r50_train_decoder"]["num_query"] == 1100        # 900 matching + 200 denoising queries
    assert workloads.WORKLOADS["ytvis480_clip_encoder"]["batch"] == 5
    for name, w in workloads.WORKLOADS.items():
        x = workloads.make_workload(name, device="cpu") if w["batch"] * sum(h * ww for h, ww in w["levels"]) < 30000 else None
        if x is None:
            continue
        S = sum(h * ww for h, ww in w["levels"])
        assert x["value"].shape == (w["batch"], S, 8, 32)
        assert x["loc"].shape[1] == (w["num_query"] or S)
        assert int(x["lsi"][-1]) + w["levels"][-1][0] * w["levels"][-1][1] == S


def test_call_sites_nest_and_travel_with_the_autograd_function():
    """uninext_amd.ext.call_site: a thread-local, nesting context (the library chooses its encoder kernels per call site,
    include/msda_hip.h); MSDeformAttnFunction records the site of its forward and re-enters it in backward -- on the host
    path here, where the context is not used but the plumbing is the same."""
    import threading

    import torch

    from uninext_amd import ext
    from uninext_amd.functions import MSDeformAttnFunction
    assert ext.current_call_site() == 0
    with ext.call_site(7):
        assert ext.current_call_site() == 7
        with ext.call_site(9):
            assert ext.current_call_site() == 9
        assert ext.current_call_site() == 7
        seen = []
        t = threading.Thread(target=lambda: seen.append(ext.current_call_site()))
        t.start(); t.join()
        assert seen == [0]                                    # per thread
    assert ext.current_call_site() == 0

    shapes = torch.tensor([[4, 5], [2, 3]], dtype=torch.int64)
    lsi = torch.tensor([0, 20], dtype=torch.int64)
    g = torch.Generator().manual_seed(0)
    value = torch.rand(1, 26, 2, 8, generator=g, requires_grad=True)
    loc = torch.rand(1, 3, 2, 2, 2, 2, generator=g, requires_grad=True)
    attn = torch.softmax(torch.rand(1, 3, 2, 4, generator=g), -1).view(1, 3, 2, 2, 2).requires_grad_(True)
    recorded = []
    real_backward = ext.ms_deform_attn_backward

    def spy(*args):
        recorded.append(ext.explicit_call_site())
        return real_backward(*args)

    ext.ms_deform_attn_backward, saved = spy, ext.ms_deform_attn_backward
    try:
        with ext.call_site(5):
            out = MSDeformAttnFunction.apply(value, shapes, lsi, loc, attn, 64)
        out.sum().backward()                                  # outside of the block
    finally:
        ext.ms_deform_attn_backward = saved
    assert recorded == [5]
    assert value.grad is not None and loc.grad is not None and attn.grad is not None
    # a forward outside of every block records NO site: its backward is matched to it by the derived-site rules instead
    value.grad = loc.grad = attn.grad = None
    ext.ms_deform_attn_backward, saved = spy, ext.ms_deform_attn_backward
    try:
        MSDeformAttnFunction.apply(value, shapes, lsi, loc, attn, 64).sum().backward()
    finally:
        ext.ms_deform_attn_backward = saved
    assert recorded == [5, None]


def test_derived_call_sites_for_callers_that_pass_none():
    """VERDICT r04 item 4 (INTEGRATION.md option A): the reference's unmodified MSDeformAttn calls the operator with no notion
    of a call site (ops/modules/ms_deform_attn.py:113).  uninext_amd.ext derives one from the call ordinal since the pass's
    spatial_shapes tensor object was first seen (Deformable-DETR builds it once per forward and hands it to every layer,
    dino.py:338,363): six encoder layers = six sites, the same six in every pass; a backward call finds its forward's site
    through the storage of sampling_loc; decoder-shaped calls take no context and do not count."""
    import torch

    from uninext_amd import ext

    class Lib:
        def __init__(self):
            self.calls = []

        def msda_hip_set_call_context(self, site, flags):
            self.calls.append((site, flags))

    lib, S = Lib(), 2048
    real = ext._geometry_checked
    ext._geometry_checked = lambda *a: True
    ext.reset_auto_sites()
    try:
        def ctx(sh, loc, Lq=S, backward=False):
            ext._set_call_context(lib, torch.float32, sh, None, S, 32, 4, Lq, 4, loc, backward=backward)

        for _ in range(2):                                        # two forward + backward passes of a six-layer encoder
            sh = torch.zeros(4, 2, dtype=torch.long)              # rebuilt per pass, as the reference does
            locs = [torch.zeros(8) for _ in range(6)]             # alive until their backward, as autograd keeps them
            for i, l in enumerate(locs):
                ctx(sh, l)
                assert ext.last_call_site() == ext.AUTO_SITE_BASE + i
                ctx(sh, torch.zeros(3), Lq=900)                   # the decoder's calls in between: no context, no ordinal
            for l in reversed(locs):
                ctx(sh, l, backward=True)
        want = [ext.AUTO_SITE_BASE + i for i in range(6)]
        assert [c[0] for c in lib.calls] == (want + want[::-1]) * 2
        assert all(c[1] == ext.CTX_GEOMETRY_CHECKED for c in lib.calls)
        ctx(sh, torch.zeros(5), backward=True)                    # a backward whose forward is unknown: history-free
        assert lib.calls[-1][0] == -1
        with ext.call_site(5):                                    # an explicit block wins and does not disturb the ordinal
            ctx(sh, locs[0])
        assert lib.calls[-1][0] == 5
        ctx(sh, locs[0])
        assert lib.calls[-1][0] == ext.AUTO_SITE_BASE + 6
        sh._version_bump = sh.add_(1)                             # the same object, modified in place: a new pass
        ctx(sh, locs[0])
        assert lib.calls[-1][0] == ext.AUTO_SITE_BASE
        for i in range(1, 40):                                    # more calls than slots: wraps around
            ctx(sh, locs[0])
        assert lib.calls[-1][0] == ext.AUTO_SITE_BASE + 39 % ext.AUTO_SITES
        # a shapes tensor built under torch.inference_mode() has no version counter (`_version` raises on it) -- the
        # bare operator / the reference's unmodified module is called exactly like that by an inference script
        with torch.inference_mode():
            shi = torch.zeros(4, 2, dtype=torch.long)
            li = torch.zeros(8)
            ctx(shi, li)
            ctx(shi, li)
        assert [c[0] for c in lib.calls[-2:]] == [ext.AUTO_SITE_BASE, ext.AUTO_SITE_BASE + 1]
        # ... and the backward calls that found no forward are counted (they are correct, but take the history-free kernel)
        before = ext.unmatched_backward_calls()
        ctx(sh, torch.zeros(7), backward=True)
        assert ext.unmatched_backward_calls() == before + 1
    finally:
        ext._geometry_checked = real
        ext.reset_auto_sites()
