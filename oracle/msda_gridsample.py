"""Torch/grid_sample restatement of the reference's CPU path -- TEST INFRASTRUCTURE.

Follows ops/functions/ms_deform_attn_func.py:43-63 (`ms_deform_attn_core_pytorch`): per
level, view that level's slice of `value` as an image batch [N*M, D, H, W], map the
normalised sampling locations to grid_sample's [-1,1] convention (align_corners=False,
bilinear, zero padding), gather, weight by the attention weights and sum over (L,P).

This is what the reference itself would execute on a GPU-less host, so bench.py times it
on the GPU box's host cores as `cpu_baseline` (kind "port": /root/reference does not exist
on that box).  It is differentiable, so autograd through it is the backward cross-check.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
"""
import torch
import torch.nn.functional as F


def msda_gridsample(value, spatial_shapes, sampling_locations, attention_weights):
    n, _, heads, dim = value.shape
    _, lq, _, levels, points, _ = sampling_locations.shape
    hw = [(int(h), int(w)) for h, w in spatial_shapes]
    grids = sampling_locations * 2.0 - 1.0
    per_level = []
    start = 0
    for lvl, (h, w) in enumerate(hw):
        img = value[:, start:start + h * w]                       # [N, HW, M, D]
        img = img.permute(0, 2, 3, 1).reshape(n * heads, dim, h, w)
        grid = grids[:, :, :, lvl].permute(0, 2, 1, 3, 4).reshape(n * heads, lq, points, 2)
        per_level.append(F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros",
                                       align_corners=False))      # [N*M, D, Lq, P]
        start += h * w
    sampled = torch.cat(per_level, dim=-1)                         # [N*M, D, Lq, L*P]
    weights = attention_weights.permute(0, 2, 1, 3, 4).reshape(n * heads, 1, lq, levels * points)
    out = (sampled * weights).sum(-1)                              # [N*M, D, Lq]
    return out.reshape(n, heads * dim, lq).transpose(1, 2).contiguous()
