"""Synthetic MSDeformAttn workloads at the shapes of the reference's R50 COCO configs
(BASELINE.md section 3).  Used by bench.py, the GPU parity tests and smoke().

One 800x1333 image through the detectron2 ResNet-50 gives feature levels
(100,167) (50,84) (25,42) (13,21) -> S = 22223 tokens; training pads to 800x1344 -> 22323.
Constants of every shipped config: M = 8 heads, D = 32, L = 4 levels, P = 4 points
(projects/UNINEXT/uninext/config.py:156,170-174).

Location flavours
  "uniform"   loc = rand(), as ops/test.py:34 -- no locality at all, worst case.
  "model"     encoder: reference point = centre of the query's own pixel, normalised
              (deformable_transformer_dino.py:289-301) + offsets/(W_l,H_l) with offsets = the MSDeformAttn
              initial bias pattern (ms_deform_attn.py:64-68: head direction x point index) + N(0,1);
              decoder: random reference boxes (cx, cy, w, h), loc = cxcy + offsets / P * wh * 0.5
              (ms_deform_attn.py:107-109).  About 5 % of the encoder points are pushed out of [0,1].
"""
import math

import torch

R50_LEVELS_INFER = ((100, 167), (50, 84), (25, 42), (13, 21))   # 800 x 1333, no padding
R50_LEVELS_TRAIN = ((100, 168), (50, 84), (25, 42), (13, 21))   # padded to 800 x 1344
HEADS, HEAD_DIM, POINTS = 8, 32, 4


def pyramid(height, width, strides=(8, 16, 32, 64)):
    """Feature levels of an image: ceil(size / stride) per level -- the strides-8/16/32 backbone outputs plus the extra
    stride-2 convolution of the deformable transformer's input projection (ddetrs_dn.py: num_feature_levels = 4)."""
    return tuple((-(-height // s), -(-width // s)) for s in strides)


# Named op-level workloads for BASELINE.json's configs; `kind` is what make_inputs() takes, `batch` the
# per-GPU batch of the config, `num_query` the decoder / ReID-head query count (None: the encoder's Lq = S).
#   configs[1]  R50 COCO det+seg inference, bs 2, 800 x 1333 (d2 ResNet: no padding)
#   configs[2]  ConvNeXt-L RefCOCO RES inference: COCO images are 4:3, 800 x 1066 padded to /32 -> 800 x 1088
#   configs[3]  ViT-H YouTube-VIS 5-frame clip: 720p video at MIN_SIZE_TEST 480 (configs/video_joint_r50.yaml:121) -> 480 x 853,
#               padded to /32 -> 480 x 864; and at the training scale 360 (MIN_SIZE_TRAIN_MULTI's 320-640 range) -> 360 x 640.
#               The deformable ReID head (ddetrs_vid_dn.py:37-49, deformable_transformer_dino.py:504-527) adds two decoder-layer
#               calls per frame whose queries are the frame's matched / kept instances (tens, not 900).
#   configs[4]  R50 Objects365 training, bs 2 per GPU, padded to /32 -> 800 x 1344; the DN decoder (ddetrs_dn.py:558-712) runs
#               900 matching queries + up to 200 denoising queries = 1100.
WORKLOADS = {
    "r50_infer_encoder": dict(kind="encoder", levels=R50_LEVELS_INFER, batch=2, num_query=None, config=1),
    "r50_infer_decoder": dict(kind="decoder", levels=R50_LEVELS_INFER, batch=2, num_query=900, config=1),
    "refcoco_encoder": dict(kind="encoder", levels=pyramid(800, 1088), batch=2, num_query=None, config=2),
    "refcoco_decoder": dict(kind="decoder", levels=pyramid(800, 1088), batch=2, num_query=900, config=2),
    "ytvis480_clip_encoder": dict(kind="encoder", levels=pyramid(480, 864), batch=5, num_query=None, config=3),
    "ytvis480_clip_decoder": dict(kind="decoder", levels=pyramid(480, 864), batch=5, num_query=900, config=3),
    "ytvis360_clip_encoder": dict(kind="encoder", levels=pyramid(360, 640), batch=5, num_query=None, config=3),
    "ytvis480_clip_reid": dict(kind="decoder", levels=pyramid(480, 864), batch=5, num_query=64, config=3),
    "r50_train_encoder": dict(kind="encoder", levels=R50_LEVELS_TRAIN, batch=2, num_query=None, config=4),
    "r50_train_decoder": dict(kind="decoder", levels=R50_LEVELS_TRAIN, batch=2, num_query=1100, config=4),
}


def make_workload(name, flavour="model", seed=0, device="cuda", **kw):
    """make_inputs() for a named workload; `wide` = the model-like pattern with sigma = 6 px offsets."""
    w = WORKLOADS[name]
    if flavour == "wide":
        kw.setdefault("offset_sigma", 6.0)
        flavour = "model"
    return make_inputs(w["kind"], flavour, batch=w["batch"], levels=w["levels"], num_query=w["num_query"], seed=seed,
                       device=device, **kw)


def level_tensors(levels, device):
    shapes = torch.as_tensor(levels, dtype=torch.int64, device=device)
    hw = shapes[:, 0] * shapes[:, 1]
    lsi = torch.cat((hw.new_zeros((1,)), hw.cumsum(0)[:-1]))
    return shapes, lsi


def _head_directions(heads):
    theta = torch.arange(heads, dtype=torch.float32) * (2.0 * math.pi / heads)
    d = torch.stack([theta.cos(), theta.sin()], -1)
    return d / d.abs().max(-1, keepdim=True)[0]                      # [M, 2]


def encoder_reference_points(levels, device):
    """Pixel-centre reference points of all S encoder queries, [S, 2] (x, y) normalised."""
    pts = []
    for (h, w) in levels:
        ys = (torch.arange(h, dtype=torch.float32, device=device) + 0.5) / h
        xs = (torch.arange(w, dtype=torch.float32, device=device) + 0.5) / w
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        pts.append(torch.stack([xx.reshape(-1), yy.reshape(-1)], -1))
    return torch.cat(pts, 0)


def make_inputs(kind="encoder", flavour="model", batch=2, levels=R50_LEVELS_INFER, num_query=None, heads=HEADS,
                head_dim=HEAD_DIM, points=POINTS, seed=0, device="cuda", dtype=torch.float32, value_scale=1.0,
                offset_sigma=1.0, far_fraction=0.05):
    """Returns dict(value, shapes, lsi, loc, attn) on `device`; generation is seeded and device-independent
    (drawn on CPU, then moved)."""
    g = torch.Generator().manual_seed(seed)
    L = len(levels)
    S = sum(h * w for h, w in levels)
    if kind == "encoder":
        Lq = S if num_query is None else num_query
    else:
        Lq = 900 if num_query is None else num_query
    value = torch.randn(batch, S, heads, head_dim, generator=g) * value_scale
    attn = torch.softmax(torch.randn(batch, Lq, heads, L * points, generator=g), -1).view(batch, Lq, heads, L, points)
    if flavour == "uniform":
        loc = torch.rand(batch, Lq, heads, L, points, 2, generator=g)
    elif flavour == "model":
        dirs = _head_directions(heads).view(1, 1, heads, 1, 1, 2)
        steps = torch.arange(1, points + 1, dtype=torch.float32).view(1, 1, 1, 1, points, 1)
        offsets = dirs * steps + offset_sigma * torch.randn(batch, Lq, heads, L, points, 2, generator=g)
        if kind == "encoder":
            ref = encoder_reference_points(levels, "cpu")
            if Lq != S:
                ref = ref[torch.randint(0, S, (Lq,), generator=g)]
            wh = torch.tensor([[w, h] for h, w in levels], dtype=torch.float32).view(1, 1, 1, L, 1, 2)
            loc = ref.view(1, Lq, 1, 1, 1, 2) + offsets / wh
            far = torch.rand(batch, Lq, heads, L, points, 1, generator=g) < far_fraction
            loc = torch.where(far, loc + torch.sign(loc - 0.5) * 0.75, loc)
        else:
            cxcy = torch.rand(batch, Lq, 1, 1, 1, 2, generator=g)
            wh = 0.02 + 0.5 * torch.rand(batch, Lq, 1, 1, 1, 2, generator=g) ** 2
            loc = cxcy + offsets / points * wh * 0.5
    else:
        raise ValueError(flavour)
    shapes, lsi = level_tensors(levels, device)
    to = dict(device=device, dtype=dtype)
    return dict(value=value.to(**to).contiguous(), shapes=shapes, lsi=lsi, loc=loc.to(**to).contiguous(),
                attn=attn.to(**to).contiguous())


def algorithmic_bytes_forward(N, S, Lq, M=HEADS, D=HEAD_DIM, L=4, P=POINTS, itemsize=4):
    """read value once + read loc + read attn + write out."""
    return itemsize * N * (S * M * D + Lq * M * L * P * 2 + Lq * M * L * P + Lq * M * D)


def algorithmic_bytes_backward(N, S, Lq, M=HEADS, D=HEAD_DIM, L=4, P=POINTS, itemsize=4):
    """read value, loc, attn, grad_out + write grad_value, grad_loc, grad_attn."""
    return itemsize * N * (2 * S * M * D + Lq * M * (D + 6 * L * P))
