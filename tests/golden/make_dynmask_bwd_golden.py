"""Mint golden GRADIENTS for the dynamic mask head with the REFERENCE's own code under autograd (build container only).

    python tests/golden/make_dynmask_bwd_golden.py

The reference functions are loaded exactly as tests/golden/make_dynmask_golden.py loads them (cut out of
projects/UNINEXT/uninext/models/ddetrs_dn.py with `ast`: mask_heads_forward :734-752, dynamic_mask_with_coords :755-844,
parse_dynamic_params :1148-1171, aligned_bilinear :1174-1196, compute_locations :1199-1212) and run in FLOAT64 on seeded
float32 inputs with a seeded upstream gradient: `torch.autograd.grad` of `(out * upstream).sum()` with respect to the mask
features, the reference points and the controller parameters.  Stored: the float32 inputs, the upstream gradient and the
float64 gradients (tests/golden/dynmask_bwd_*.npz).  Nothing of the reference is copied into this repository.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_dynmask_golden import load_reference_functions, make_self  # noqa: E402

CASES = {
    "dynmask_bwd_rel_up2": dict(seed=11, N=2, H=12, W=17, num_insts=[5, 3], rel_coord=True, mask_out_stride=4),
    "dynmask_bwd_rel_noup": dict(seed=12, N=2, H=10, W=13, num_insts=[4, 6], rel_coord=True, mask_out_stride=8),
    "dynmask_bwd_norel_up2": dict(seed=13, N=1, H=9, W=11, num_insts=[7], rel_coord=False, mask_out_stride=4),
    "dynmask_bwd_empty_image_up4": dict(seed=14, N=3, H=7, W=10, num_insts=[2, 0, 3], rel_coord=True, mask_out_stride=2),
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
        f = 8 // kw["mask_out_stride"]
        upstream = torch.randn(1, n_all, H * f, W * f, generator=g)
        f64, r64, p64 = (t.double().requires_grad_(True) for t in (feats, ref_pts, params))
        out = me.dynamic_mask_with_coords(f64, r64, p64, kw["num_insts"], 8, rel_coord=kw["rel_coord"])
        assert out.shape == upstream.shape and out.dtype == torch.float64
        gf, gr, gp = torch.autograd.grad((out * upstream.double()).sum(), (f64, r64, p64), allow_unused=True)
        if gr is None:
            gr = torch.zeros_like(r64)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), mask_feats=feats.numpy(), reference_points=ref_pts.numpy(),
                            mask_head_params=params.numpy(), num_insts=np.asarray(kw["num_insts"], dtype=np.int64),
                            rel_coord=np.int64(kw["rel_coord"]), mask_out_stride=np.int64(kw["mask_out_stride"]),
                            upstream=upstream.numpy(), out=out.detach().numpy(), grad_mask_feats=gf.numpy(),
                            grad_reference_points=gr.numpy(), grad_mask_head_params=gp.numpy())
        print(name, tuple(out.shape), float(gf.abs().max()), float(gr.abs().max()), float(gp.abs().max()))
    # aligned_bilinear on its own: gradient of sum(out * upstream) for several factors (float64)
    g = torch.Generator().manual_seed(19)
    t = torch.randn(3, 1, 5, 7, generator=g)
    rec = {"x": t.numpy()}
    for f in (2, 3, 4):
        up = torch.randn(3, 1, 5 * f, 7 * f, generator=g)
        t64 = t.double().requires_grad_(True)
        (gx,) = torch.autograd.grad((ns["aligned_bilinear"](t64, f) * up.double()).sum(), (t64,))
        rec[f"up{f}"] = up.numpy()
        rec[f"g{f}"] = gx.numpy()
    np.savez_compressed(os.path.join(HERE, "dynmask_bwd_aligned_bilinear.npz"), **rec)
    print("aligned_bilinear gradients ok")


if __name__ == "__main__":
    main()
