"""CPU oracle for the patch-embedding convolutions.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by
tests/, never by the product path.

Restates, in numpy float64, what the reference computes with nn.Conv2d(kernel_size = stride = k, padding = 0):
  ViT      PatchEmbed.forward   projects/UNINEXT/uninext/backbone/utils.py:182-186  (conv, then permute to B H W C)
  ConvNeXt stem / downsample    projects/UNINEXT/uninext/backbone/convnext.py:80,87 (conv, NCHW)
The arithmetic itself lives in PyTorch (aten convolution), not in the reference; parity is pinned on
tests/golden/patch_*.npz, minted by running the reference's PatchEmbed class / the same nn.Conv2d constructor calls
in fp64 (tests/golden/make_patch_embed_golden.py).
"""
import numpy as np


def forward(x, weight, bias=None, channels_last=True):
    """x [B, C, H, W], weight [E, C, k, k], bias [E] | None -> [B, H//k, W//k, E] or [B, E, H//k, W//k] (float64)."""
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(weight, dtype=np.float64)
    B, C, H, W = x.shape
    E, C2, k, k2 = w.shape
    assert C == C2 and k == k2
    Hp, Wp = H // k, W // k                                   # the remainder rows / columns are ignored
    patches = x[:, :, :Hp * k, :Wp * k].reshape(B, C, Hp, k, Wp, k)
    out = np.einsum("bchywx,ecyx->bhwe", patches, w)           # sum over (c, ky, kx)
    if bias is not None:
        out = out + np.asarray(bias, dtype=np.float64)
    return out if channels_last else np.ascontiguousarray(out.transpose(0, 3, 1, 2))
