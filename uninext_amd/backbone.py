"""Backbone patch-embedding layers with the reference's names and state-dict layout, running on
include/patch_embed_hip.h at inference.

  PatchEmbed        projects/UNINEXT/uninext/backbone/utils.py:160-186 (ViT; constructed at backbone/vit.py:291)
  patch_conv2d      the same convolution with nn.Conv2d's NCHW output: ConvNeXt stem and downsample convolutions
                    (backbone/convnext.py:80,87)

The HIP kernel is forward-only: with autograd recording (training) the layers run the PyTorch-ROCm convolution, which
is the same arithmetic in fp32 and has a backward.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ext


def _use_hip(x, conv):
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad)
    return (not needs_grad and conv.groups == 1 and tuple(conv.dilation) == (1, 1) and x.is_contiguous()
            and ext.patch_embed_supported(x, conv.weight, conv.stride, conv.padding))


def patch_conv2d(x, conv):
    """`conv(x)` for an nn.Conv2d whose kernel equals its stride (no padding): [B, E, H // k, W // k]."""
    if _use_hip(x, conv):
        return ext.patch_embed_forward(x, conv.weight.contiguous(), conv.bias, channels_last=False)
    return conv(x)


class PatchEmbed(nn.Module):
    """Image to Patch Embedding (backbone/utils.py:160-186): same constructor, same `proj` parameter names."""

    def __init__(self, kernel_size=(16, 16), stride=(16, 16), padding=(0, 0), in_chans=3, embed_dim=768):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=kernel_size, stride=stride, padding=padding)

    def forward(self, x):
        if _use_hip(x, self.proj):
            return ext.patch_embed_forward(x, self.proj.weight.contiguous(), self.proj.bias, channels_last=True)
        x = self.proj(x)
        return x.permute(0, 2, 3, 1)   # B C H W -> B H W C (a view, as in the reference)
