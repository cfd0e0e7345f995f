"""Mint golden vectors for the dynamic mask head with the REFERENCE's own code (build container only).

    python tests/golden/make_dynmask_golden.py

projects/UNINEXT/uninext/models/ddetrs_dn.py cannot be imported here (it pulls detectron2, PIL, ...), so the
functions on this path are cut out of the reference source with `ast` and executed as they are:
module-level `parse_dynamic_params` (:1148-1171), `aligned_bilinear` (:1174-1196), `compute_locations` (:1199-1212) and
the methods `DDETRSegmUniDN.mask_heads_forward` (:734-752) and `.dynamic_mask_with_coords` (:755-844), bound to a
stand-in `self` that carries the attributes the constructor sets (:45-70: 8 dynamic channels, weight/bias sizes,
mask_out_stride, use_raft False).  Nothing of the reference is copied into this repository; only the inputs and the
tensors it returned are stored (tests/golden/dynmask_*.npz).
"""
import ast
import os
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = os.environ.get("UNINEXT_REFERENCE", "/root/reference")
SRC = os.path.join(REF, "projects/UNINEXT/uninext/models/ddetrs_dn.py")
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_functions():
    tree = ast.parse(open(SRC).read())
    wanted_fn = {"parse_dynamic_params", "aligned_bilinear", "compute_locations"}
    wanted_m = {"mask_heads_forward", "dynamic_mask_with_coords"}
    body = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in wanted_fn:
            body.append(node)
        if isinstance(node, ast.ClassDef) and node.name == "DDETRSegmUniDN":
            body += [n for n in node.body if isinstance(n, ast.FunctionDef) and n.name in wanted_m]
    assert len(body) == 5, [n.name for n in body]
    ns = {"torch": torch, "F": F}
    exec(compile(ast.Module(body=body, type_ignores=[]), SRC, "exec"), ns)
    return ns


def make_self(ns, in_channels=8, rel_coord=True, mask_out_stride=4):
    ch = 8
    weight_nums = [(in_channels + 2 if rel_coord else in_channels) * ch, ch * ch, ch * 1]
    bias_nums = [ch, ch, 1]
    me = types.SimpleNamespace(dynamic_mask_channels=ch, weight_nums=weight_nums, bias_nums=bias_nums,
                               mask_out_stride=mask_out_stride, use_raft=False)
    me.mask_heads_forward = types.MethodType(ns["mask_heads_forward"], me)
    me.dynamic_mask_with_coords = types.MethodType(ns["dynamic_mask_with_coords"], me)
    return me


CASES = {
    "dynmask_rel_up2": dict(seed=1, N=2, H=12, W=17, num_insts=[5, 3], rel_coord=True, mask_out_stride=4),
    "dynmask_rel_noup": dict(seed=2, N=2, H=12, W=17, num_insts=[4, 6], rel_coord=True, mask_out_stride=8),
    "dynmask_norel_up2": dict(seed=3, N=1, H=9, W=11, num_insts=[7], rel_coord=False, mask_out_stride=4),
    "dynmask_empty_image_up4": dict(seed=4, N=3, H=7, W=10, num_insts=[2, 0, 3], rel_coord=True, mask_out_stride=2),
}


def main():
    ns = load_reference_functions()
    for name, kw in CASES.items():
        g = torch.Generator().manual_seed(kw["seed"])
        N, H, W = kw["N"], kw["H"], kw["W"]
        n_all = sum(kw["num_insts"])
        me = make_self(ns, rel_coord=kw["rel_coord"], mask_out_stride=kw["mask_out_stride"])
        nparams = sum(me.weight_nums) + sum(me.bias_nums)
        feats = torch.randn(N, 8, H, W, generator=g)
        ref_pts = torch.rand(1, n_all, 2, generator=g) * torch.tensor([W * 8.0, H * 8.0])
        params = torch.randn(1, n_all, nparams, generator=g) * 0.3
        out = me.dynamic_mask_with_coords(feats, ref_pts, params, kw["num_insts"], 8, rel_coord=kw["rel_coord"])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), mask_feats=feats.numpy(), reference_points=ref_pts.numpy(),
                            mask_head_params=params.numpy(), num_insts=np.asarray(kw["num_insts"], dtype=np.int64),
                            rel_coord=np.int64(kw["rel_coord"]), mask_out_stride=np.int64(kw["mask_out_stride"]),
                            out=out.numpy())
        print(name, tuple(out.shape))
    # aligned_bilinear on its own, several factors
    g = torch.Generator().manual_seed(9)
    t = torch.randn(3, 1, 5, 7, generator=g)
    np.savez_compressed(os.path.join(HERE, "dynmask_aligned_bilinear.npz"), x=t.numpy(),
                        **{f"f{f}": ns["aligned_bilinear"](t, f).numpy() for f in (1, 2, 3, 4)})
    print("aligned_bilinear ok")


if __name__ == "__main__":
    main()
