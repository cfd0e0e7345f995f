"""Torch restatement of the reference's dynamic mask head -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Follows the reference's own (materialising) algorithm step by step so that it can serve as the checker of the HIP
kernel at sizes the fixtures do not cover: DDETRSegmUniDN.dynamic_mask_with_coords
(projects/UNINEXT/uninext/models/ddetrs_dn.py:755-844) = compute_locations (:1199-1212) -> relative coordinates ->
repeat + cat into [1, n_inst*(C+2), H, W] -> parse_dynamic_params (:1148-1171) -> three grouped 1x1 conv2d with ReLU
(mask_heads_forward, :734-752) -> aligned_bilinear (:1174-1196).  Pinned to fixtures produced by the reference code
itself (tests/golden/dynmask_*.npz, tests/test_dynmask_cpu.py).  Only tests/ may import it.
"""
import torch
import torch.nn.functional as F

DYN_CH = 8


def pixel_locations(h, w, stride, device):
    xs = torch.arange(0, w * stride, step=stride, dtype=torch.float32, device=device)
    ys = torch.arange(0, h * stride, step=stride, dtype=torch.float32, device=device)
    yy, xx = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack((xx.reshape(-1), yy.reshape(-1)), dim=1) + stride // 2


def upsample_aligned(t, factor):
    if factor == 1:
        return t
    h, w = t.shape[2:]
    t = F.pad(t, pad=(0, 1, 0, 1), mode="replicate")
    t = F.interpolate(t, size=(factor * h + 1, factor * w + 1), mode="bilinear", align_corners=True)
    t = F.pad(t, pad=(factor // 2, 0, factor // 2, 0), mode="replicate")
    return t[:, :, :factor * h, :factor * w]


def dynamic_mask_oracle(mask_feats, reference_points, mask_head_params, num_insts, mask_feat_stride, rel_coord=True,
                        mask_out_stride=4):
    n, c, h, w = mask_feats.shape
    n_all = reference_points.shape[1]
    loc = pixel_locations(h, w, mask_feat_stride, mask_feats.device)
    blocks, first = [], 0
    for b, cnt in enumerate(num_insts):
        feats_b = mask_feats[b].reshape(1, c, h * w).unsqueeze(1).repeat(1, cnt, 1, 1)
        if rel_coord:
            rel = reference_points[:, first:first + cnt].reshape(1, cnt, 1, 1, 2) - loc.reshape(1, 1, h, w, 2)
            rel = rel.float().permute(0, 1, 4, 2, 3).flatten(-2, -1)
            feats_b = torch.cat([rel, feats_b], dim=2)
        blocks.append(feats_b.reshape(1, -1, h, w) if not rel_coord else feats_b)
        first += cnt
    x = torch.cat(blocks, dim=1).reshape(1, -1, h, w)
    params = mask_head_params.flatten(0, 1)
    cin = c + 2 if rel_coord else c
    sizes = [cin * DYN_CH, DYN_CH * DYN_CH, DYN_CH, DYN_CH, DYN_CH, 1]
    w0, w1, w2, b0, b1, b2 = torch.split_with_sizes(params, sizes, dim=1)
    layers = [(w0.reshape(n_all * DYN_CH, cin, 1, 1), b0.reshape(-1)), (w1.reshape(n_all * DYN_CH, DYN_CH, 1, 1), b1.reshape(-1)),
              (w2.reshape(n_all, DYN_CH, 1, 1), b2.reshape(-1))]
    for i, (wt, bs) in enumerate(layers):
        x = F.conv2d(x, wt, bias=bs, stride=1, padding=0, groups=n_all)
        if i < 2:
            x = F.relu(x)
    x = upsample_aligned(x.reshape(-1, 1, h, w), int(mask_feat_stride / mask_out_stride))
    return x.reshape(1, -1, x.shape[-2], x.shape[-1])
