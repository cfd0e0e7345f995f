"""CPU oracle for the 3x3 convolutions of the static mask head.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates in numpy float64 what MaskHeadSmallConv computes with torch.nn.Conv2d(cin, cout, 3, padding=1) + F.relu
(projects/UNINEXT/uninext/models/ddetrs_dn.py:941-953, :991-1025).  The arithmetic lives in PyTorch (aten
convolution); parity is pinned on tests/golden/maskhead_*.npz, minted by executing the reference class itself in
fp64 (tests/golden/make_maskhead_golden.py).
"""
import numpy as np


def conv3x3(x, weight, bias=None, relu=False):
    """x [B, C, H, W], weight [E, C, 3, 3], bias [E] | None -> [B, E, H, W] float64 (zero padding 1, stride 1)."""
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(weight, dtype=np.float64)
    B, C, H, W = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    out = np.zeros((B, w.shape[0], H, W))
    for ky in range(3):
        for kx in range(3):
            out += np.einsum("bchw,ec->behw", xp[:, :, ky:ky + H, kx:kx + W], w[:, :, ky, kx])
    if bias is not None:
        out += np.asarray(bias, dtype=np.float64)[None, :, None, None]
    return np.maximum(out, 0.0) if relu else out


def mask_head_small_conv(x, params):
    """MaskHeadSmallConv.forward(x, fpns=None) (ddetrs_dn.py:975-1031): x = [stride-8, stride-16, stride-32] maps,
    params = {"lay1.weight": ..., ...}."""
    def nearest(t, size):   # F.interpolate(mode="nearest"): src = floor(dst * in / out)
        H, W = t.shape[-2:]
        ys = (np.arange(size[0]) * H) // size[0]
        xs = (np.arange(size[1]) * W) // size[1]
        return t[:, :, ys][:, :, :, xs]
    c = lambda t, n: conv3x3(t, params[n + ".weight"], params[n + ".bias"], relu=True)
    x = [np.asarray(t, dtype=np.float64) for t in x]
    fused = c(x[-1], "lay3")
    fused = c(x[-2] + nearest(fused, x[-2].shape[-2:]), "lay4")
    fused = c(x[-3] + nearest(fused, x[-3].shape[-2:]), "jia_dcn")
    return c(c(fused, "lay1"), "lay2")
