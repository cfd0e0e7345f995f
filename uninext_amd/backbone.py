"""Backbone patch-embedding layers with the reference's names and state-dict layout, running on
include/patch_embed_hip.h at inference.

  PatchEmbed        projects/UNINEXT/uninext/backbone/utils.py:160-186 (ViT; constructed at backbone/vit.py:291)
  patch_conv2d      the same convolution with nn.Conv2d's NCHW output: ConvNeXt stem and downsample convolutions
                    (backbone/convnext.py:80,87)

The HIP kernel is forward-only: with autograd recording (training) the layers run the PyTorch-ROCm convolution, which
is the same arithmetic in fp32 and has a backward.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ext
from ._cache import CachedModuleMixin, packed_weight


def _use_hip(x, conv):
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad)
    return (not needs_grad and conv.groups == 1 and tuple(conv.dilation) == (1, 1) and x.is_contiguous()
            and ext.patch_embed_supported(x, conv.weight, conv.stride, conv.padding))


def _packed_weight(conv):
    """Split-bf16 packed copy of conv.weight, cached on the module and rebuilt when the parameter changes."""
    return packed_weight(conv, ext.patch_embed_pack_weight)


def _hip_conv(x, conv, channels_last, exact):
    w = conv.weight
    # short K (ConvNeXt stem, K = 48): the kernel is bound by writing the output and the exact one is faster
    if not exact and w.shape[1] * w.shape[2] * w.shape[3] > 64 and ext.patch_embed_packed_supported(w):
        return ext.patch_embed_packed_forward(x, _packed_weight(conv), w.shape[0], w.shape[2], conv.bias, channels_last)
    return ext.patch_embed_forward(x, w.contiguous(), conv.bias, channels_last=channels_last)


def patch_conv2d(x, conv, exact=True):
    """`conv(x)` for an nn.Conv2d whose kernel equals its stride (no padding): [B, E, H // k, W // k].  Inference on
    the GPU: the exact-fp32 MFMA kernel; exact=False opts into split-bf16 products from cached packed weights (~2e-5 of the output scale)."""
    if _use_hip(x, conv):
        return _hip_conv(x, conv, False, exact)
    return conv(x)


class PatchEmbed(CachedModuleMixin, nn.Module):
    """Image to Patch Embedding (backbone/utils.py:160-186): same constructor, same `proj` parameter names."""

    # True (default): exact-fp32 MFMA kernel (bitwise an fmaf chain, the reference's arithmetic); False -- or env
    # UNINEXT_AMD_SPLIT_BF16=1 -- opts into the split-bf16 path (3 of 4 partial products, ~2e-5 of the output scale, faster)
    exact_fp32 = os.environ.get("UNINEXT_AMD_SPLIT_BF16", "0") != "1"

    def __init__(self, kernel_size=(16, 16), stride=(16, 16), padding=(0, 0), in_chans=3, embed_dim=768):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=kernel_size, stride=stride, padding=padding)

    def forward(self, x):
        if _use_hip(x, self.proj):
            return _hip_conv(x, self.proj, True, self.exact_fp32)
        x = self.proj(x)
        return x.permute(0, 2, 3, 1)   # B C H W -> B H W C (a view, as in the reference)
