"""Mint golden vectors for the static mask head with the REFERENCE's own code (build container only).

    python tests/golden/make_maskhead_golden.py

projects/UNINEXT/uninext/models/ddetrs_dn.py cannot be imported here (detectron2, PIL, ...), so the class
MaskHeadSmallConv (:923-1031) and the helper _expand (:1112-1113) are cut out of the reference source with `ast` and
executed as they are, in fp64: `maskhead_nofpn` is the configuration the model builds (ddetrs_dn.py:83:
MaskHeadSmallConv(hidden_dim, None, hidden_dim), called with fpns=None, :528), `maskhead_fpn` exercises the adapter
branch.  Only inputs, parameters and outputs are stored (tests/golden/maskhead_*.npz).
"""
import ast
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = os.environ.get("UNINEXT_REFERENCE", "/root/reference")
SRC = os.path.join(REF, "projects/UNINEXT/uninext/models/ddetrs_dn.py")
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_class():
    tree = ast.parse(open(SRC).read())
    body = [n for n in tree.body if (isinstance(n, ast.ClassDef) and n.name == "MaskHeadSmallConv")
            or (isinstance(n, ast.FunctionDef) and n.name == "_expand")]
    assert len(body) == 2
    ns = {"torch": torch, "nn": nn, "F": F}
    exec(compile(ast.Module(body=body, type_ignores=[]), SRC, "exec"), ns)
    return ns["MaskHeadSmallConv"]


def main():
    cls = load_reference_class()
    torch.manual_seed(11)
    dim = 32   # dim // 4 = 8, dim // 32 = 1 output channel; 9 * 32 and 9 * 8 are multiples of 16 / 8 ...
    for name, fpn_dims in (("maskhead_nofpn", None), ("maskhead_fpn", [16, 24, 8])):
        head = cls(dim, fpn_dims, dim).double()
        for p in head.parameters():            # the reference zero-initialises the biases: make them count
            if p.dim() == 1:
                torch.nn.init.uniform_(p, -0.3, 0.3)
        sizes = [(13, 18), (7, 9), (4, 5)]     # stride 8 / 16 / 32 of a 100 x 140 image
        x = [torch.randn(2, dim, h, w, dtype=torch.float64) for h, w in sizes]
        fpns = None
        if fpn_dims is not None:
            fpns = [torch.randn(1, fpn_dims[i], *sizes[2 - i], dtype=torch.float64) for i in range(3)]
        out = head(x, fpns)
        arrays = {"x%d" % i: t.numpy() for i, t in enumerate(x)}
        if fpns is not None:
            arrays.update({"fpn%d" % i: t.numpy() for i, t in enumerate(fpns)})
        arrays.update({"p:" + k: v.detach().numpy() for k, v in head.state_dict().items()})
        np.savez_compressed(os.path.join(HERE, name + ".npz"), out=out.detach().numpy(), **arrays)
        print(name, [tuple(t.shape) for t in x], "->", tuple(out.shape))


if __name__ == "__main__":
    main()
