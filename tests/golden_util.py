"""Helpers shared by the parity tests: golden fixture loading and error metrics."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def golden_names():
    """MSDeformAttn operator fixtures (tests/golden/make_golden.py)."""
    return [n for n in _names() if not n.startswith(("matcher_", "dynmask_", "patch_", "maskhead_", "enclayer_", "encstack_", "encshape_"))]


def dynmask_names():
    """Dynamic mask head fixtures (tests/golden/make_dynmask_golden.py), without the aligned_bilinear one."""
    return [n for n in _names() if n.startswith("dynmask_") and not n.startswith("dynmask_bwd_") and n != "dynmask_aligned_bilinear"]


def dynmask_bwd_names():
    """Gradient fixtures of the dynamic mask head (tests/golden/make_dynmask_bwd_golden.py: the reference under autograd, float64),
    without the aligned_bilinear one."""
    return [n for n in _names() if n.startswith("dynmask_bwd_") and n != "dynmask_bwd_aligned_bilinear"]


def patch_names():
    """Patch-embedding fixtures (tests/golden/make_patch_embed_golden.py)."""
    return [n for n in _names() if n.startswith("patch_")]


def maskhead_names():
    """Static mask head fixtures (tests/golden/make_maskhead_golden.py)."""
    return [n for n in _names() if n.startswith("maskhead_")]


def enclayer_names():
    """Encoder-layer fixtures (tests/golden/make_encoder_layer_golden.py)."""
    return [n for n in _names() if n.startswith("enclayer_")]


def matcher_names():
    """Matcher index fixtures (tests/golden/make_matcher_golden.py)."""
    return [n for n in _names() if n.startswith("matcher_")]


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)))) if a.size else 0.0


def scaled_err(a, b):
    """max |a-b| / max(1, max|b|): the tolerance the tests quote for gradients."""
    b64 = np.asarray(b, dtype=np.float64)
    scale = max(1.0, float(np.max(np.abs(b64)))) if b64.size else 1.0
    return max_abs(a, b) / scale


def grad_loc_err(gl, ref, shapes):
    """Worst ratio of |grad_sampling_loc - ref| to its bound 1e-4 * max(W_l, H_l) over the levels: the derivative with
    respect to a NORMALISED location is the pixel-space derivative times W_l (x) / H_l (y) (cuh:157-158), so its rounding
    error carries that factor.  gl, ref: [N, Lq, M, L, P, 2]; shapes: [L, 2] (H, W).  < 1 passes."""
    d = np.abs(np.asarray(gl, dtype=np.float64) - np.asarray(ref, dtype=np.float64))
    worst = 0.0
    for l, (h, w) in enumerate(np.asarray(shapes).tolist()):
        if d[:, :, :, l].size:
            worst = max(worst, float(d[:, :, :, l].max()) / (1e-4 * max(h, w)))
    return worst


def carried(which, *names):
    """Of the kernel variants `names`, those the loaded library carries.  The default build leaves the kernels that lost
    their A/B out (uninext_amd/csrc/experiments/, names 'exp:...'); with MSDA_HIP_LIB=.../libmsda_hip_exp.so
    (`make -C uninext_amd/csrc experiments`) the same tests cover them again."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from uninext_amd import _lib
    have = set(_lib.variants(which))
    return [n for n in names if n == "auto" or n in have]
