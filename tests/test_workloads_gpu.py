"""Every named op-level workload of BASELINE.json's configs (uninext_amd.workloads.WORKLOADS: RefCOCO-size pyramid, YouTube-VIS
5-frame clips incl. the ReID-head call shape, the training padding with 1100 decoder queries) at FULL size, EVERY query, forward
and backward, against the C oracle (oracle/msda_oracle.c) -- VERDICT r02 "untested configs".  Bounds as in test_msda_parity_gpu.py:
forward and grad_value |a - b| < 1e-4 against the float64 oracle (decoder-style calls, whose queries pile up on few pixels:
max(1e-4, 2 x the float32 oracle's own error)); grad_sampling_loc 1e-4 * max(W_l, H_l) against the float32 oracle (it jumps at cell
boundaries); grad_attn_weight max(1e-4, 2 x the float32 oracle's own distance from float64)."""
import numpy as np
import pytest
import torch

from golden_util import carried

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def api():
    import MultiScaleDeformableAttention as MSDA
    from uninext_amd import _lib
    _lib.load()
    return MSDA, _lib


def _names():
    from uninext_amd import workloads
    return [n for n in workloads.WORKLOADS if n != "r50_infer_encoder"]      # that one is test_msda_parity_gpu.py's subject


@pytest.mark.parametrize("flavour", ["model", "wide"])
@pytest.mark.parametrize("name", _names())
def test_named_workload_forward_and_backward_every_query(name, flavour, dev, api):
    from oracle import msda_oracle
    from uninext_amd import workloads
    MSDA, lib = api
    w = workloads.WORKLOADS[name]
    x = workloads.make_workload(name, flavour, seed=31 + len(name), device=dev)
    N, S = x["value"].shape[:2]
    Lq = x["loc"].shape[1]
    encoder = w["kind"] == "encoder"
    ref = msda_oracle.forward(x["value"].double(), x["shapes"], x["lsi"], x["loc"].double(), x["attn"].double())
    variants = carried("forward", "auto", "msda_fwd_win", "msda_fwd_win2", "msda_fwd_win3", "msda_fwd_win4", "msda_fwd_winl", "msda_fwd_winp", "msda_fwd_lg3") if encoder else ("auto",)
    for variant in variants:
        lib.set_variant("forward", variant)
        try:
            out = MSDA.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)
        finally:
            lib.set_variant("forward", "auto")
        kernel = lib.last_kernel("forward")
        if variant != "auto":
            assert kernel == variant, (variant, kernel)
        err = float(np.abs(out.cpu().numpy().astype(np.float64) - ref).max())
        print("%-24s %-6s forward  %-20s max |err| %.2e" % (name, flavour, kernel, err))
        assert err < 1e-4, (name, variant, err)

    go = torch.randn(N, Lq, 256, generator=torch.Generator().manual_seed(5)).to(dev)
    gv, gl, ga = MSDA.ms_deform_attn_backward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], go, 64)
    kernel = lib.last_kernel("backward")
    assert kernel in (("msda_bwd_tiled", "msda_bwd_win", "msda_bwd_regions") if encoder else ("msda_bwd_dec",)), kernel
    ogv, ogl, oga = msda_oracle.backward(go, x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"])
    tgv, _, tga = msda_oracle.backward(go.double(), x["value"].double(), x["shapes"], x["lsi"], x["loc"].double(), x["attn"].double())
    e_gv = float(np.abs(gv.cpu().numpy().astype(np.float64) - tgv).max())
    e_ga = float(np.abs(ga.cpu().numpy().astype(np.float64) - tga).max())
    o_gv, o_ga = float(np.abs(ogv - tgv).max()), float(np.abs(oga - tga).max())
    d_gl = np.abs(gl.cpu().numpy().astype(np.float64) - ogl)
    e_gl = [float(d_gl[:, :, :, l].max()) for l in range(4)]
    print("%-24s %-6s backward %-20s grad_value %.2e (f32 oracle %.2e) grad_attn %.2e (f32 oracle %.2e) grad_loc %s" % (
        name, flavour, kernel, e_gv, o_gv, e_ga, o_ga, ["%.1e" % e for e in e_gl]))
    assert e_gv < (1e-4 if encoder else max(1e-4, 2.0 * o_gv)), (name, e_gv, o_gv)
    assert e_ga < max(1e-4, 2.0 * o_ga), (name, e_ga, o_ga)
    for l, (h, ww) in enumerate(w["levels"]):
        assert e_gl[l] < 1e-4 * max(h, ww), (name, l, e_gl[l])
