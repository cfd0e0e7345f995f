"""Mint golden vectors for the patch-embedding convolutions with the REFERENCE's own code (build container only).

    python tests/golden/make_patch_embed_golden.py

`patch_vit_*`: the reference class PatchEmbed (projects/UNINEXT/uninext/backbone/utils.py:160-186; that file imports
only torch, so it is loaded as it is) in fp64, output after its permute (B H W C).
`patch_convnext_*`: backbone/convnext.py needs timm + detectron2 to import, so the two constructor calls on this path
are repeated literally -- nn.Conv2d(in_chans, dims[0], kernel_size=4, stride=4) (:80) and
nn.Conv2d(dims[i], dims[i+1], kernel_size=2, stride=2) (:87) -- and run in fp64; the arithmetic is PyTorch's.
Only inputs, parameters and outputs are stored (tests/golden/patch_*.npz); no reference source is copied.
"""
import importlib.util
import os

import numpy as np
import torch
import torch.nn as nn

REF = os.environ.get("UNINEXT_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_utils():
    path = os.path.join(REF, "projects/UNINEXT/uninext/backbone/utils.py")
    spec = importlib.util.spec_from_file_location("ref_backbone_utils", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def save(name, x, conv, out, channels_last):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), x=x.numpy(), weight=conv.weight.detach().numpy(),
                        bias=conv.bias.detach().numpy(), out=out.detach().numpy(),
                        channels_last=np.array(int(channels_last)))
    print(name, tuple(x.shape), "->", tuple(out.shape))


def main():
    ref = load_reference_utils()
    torch.manual_seed(7)
    # ViT: 16 x 16 patches, sizes that are not multiples of 16 (remainder ignored), a channel count that leaves a
    # partial 128-column tile
    for name, (B, H, W, E) in {"patch_vit_small": (2, 37, 50, 40), "patch_vit_tiles": (1, 64, 160, 136)}.items():
        pe = ref.PatchEmbed(kernel_size=(16, 16), stride=(16, 16), padding=(0, 0), in_chans=3, embed_dim=E).double()
        x = torch.randn(B, 3, H, W, dtype=torch.float64)
        save(name, x, pe.proj, pe(x).contiguous(), True)
    # ConvNeXt stem (convnext.py:80) and one downsample convolution (convnext.py:87)
    stem = nn.Conv2d(3, 24, kernel_size=4, stride=4).double()
    x = torch.randn(2, 3, 22, 35, dtype=torch.float64)
    # K = 3 * 4 * 4 = 48 is a multiple of 16: supported
    save("patch_convnext_stem", x, stem, stem(x), False)
    down = nn.Conv2d(12, 24, kernel_size=2, stride=2).double()
    x = torch.randn(2, 12, 13, 18, dtype=torch.float64)
    save("patch_convnext_down", x, down, down(x), False)


if __name__ == "__main__":
    main()
