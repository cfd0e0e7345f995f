#!/usr/bin/env python
"""Launch the MFMA kernels of this repo a few times each (no PyTorch convolutions) -- the command profiled for
profiles/r01_mfma_kernels_*.txt:

    rocprofv3 --kernel-trace --stats -d out -- python tools/mfma_profile_run.py
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d out -- python tools/mfma_profile_run.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uninext_amd import ext  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    with torch.no_grad():
        # static mask head, bs 2, 256 -> 256 at 100 x 167 (the dominant layer) and 256 -> 64
        x = r(2, 256, 100, 167)
        for cout in (256, 64):
            w, b = r(cout, 256, 3, 3) / 48.0, r(cout)
            packed = ext.conv3x3_pack_weight(w)
            for _ in range(5):
                ext.conv3x3_packed_forward(x, packed, cout, b, relu=True)
            if cout == 256:
                for _ in range(3):
                    ext.conv3x3_forward(x, w, b, relu=True)
        # ViT-Huge patch embedding, bs 2, 800 x 1333
        img, w, b = r(2, 3, 800, 1333), r(1280, 3, 16, 16) / 27.7, r(1280)
        packed = ext.patch_embed_pack_weight(w)
        for _ in range(5):
            ext.patch_embed_packed_forward(img, packed, 1280, 16, b, True)
        for _ in range(3):
            ext.patch_embed_forward(img, w, b, channels_last=True)
        # the layer's projections and the FFN (include/linear_hip.h), 44 446 rows
        xl = r(2, 22223, 256)
        for n in (256, 1024):
            w, b = r(n, 256) / 16.0, r(n)
            packed = ext.linear_pack_weight(w)
            for _ in range(5):
                ext.linear_packed_forward(xl, packed, n, b, relu=(n == 1024))
        xh, w, b = r(2, 22223, 1024), r(256, 1024) / 32.0, r(256)
        packed = ext.linear_pack_weight(w)
        for _ in range(5):
            ext.linear_packed_forward(xh, packed, 256, b)
        # ConvNeXt-L downsample 2
        xd, w, b = r(2, 384, 100, 166), r(768, 384, 2, 2) / 39.2, r(768)
        packed = ext.patch_embed_pack_weight(w)
        for _ in range(5):
            ext.patch_embed_packed_forward(xd, packed, 768, 2, b, False)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
