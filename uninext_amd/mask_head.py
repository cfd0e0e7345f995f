"""Dynamic (per-instance) mask head of UNINEXT's CondInst branch -- SURVEY.md 8(f) rank 2.

Host-side mirror of `DDETRSegmUniDN.dynamic_mask_with_coords` (projects/UNINEXT/uninext/models/ddetrs_dn.py:755-844)
and of the helpers it uses (`mask_heads_forward` :734-752, `parse_dynamic_params` :1148-1171, `aligned_bilinear`
:1174-1196, `compute_locations` :1199-1212): same arguments, same output `[1, n_inst_all, H*f, W*f]`.

On the GPU (fp32, 8 mask-feature channels) it runs HIP kernels (include/dynmask_hip.h): the three per-instance 1x1
convolutions with the relative-coordinate channels generated on the fly -- the reference's `[1, n_inst*(C+2), H, W]`
repeat/cat input (1.2 GB at 1800 instances) is never built -- and `aligned_bilinear`; UNDER AUTOGRAD TOO (round 5:
`DynMaskFunction`, `AlignedBilinearFunction` -- BASELINE configs[4] trains this head): the backward kernels recompute
the activations per (instance, pixel), so nothing of size n_inst x 8 x H x W is kept for backward either, and they are
deterministic.  For other geometry / dtypes / CPU tensors a PyTorch composition is used that is algebraically the same
and also non-materialising: the feature part of the first layer is one batched matmul against the shared feature map.  `use_raft` up-sampling is not covered (USE_RAFT is False in
every shipped config, uninext/config.py:178).
"""
import os

import torch
import torch.nn.functional as F

from . import ext as _ext
from ._cache import CachedModuleMixin, packed_weight

DYNAMIC_MASK_CHANNELS = 8   # ddetrs_dn.py:46


def parse_dynamic_params(params, in_channels, rel_coord=True):
    """[n, num_params] -> (w0 [n,8,cin], w1 [n,8,8], w2 [n,1,8], b0 [n,8], b1 [n,8], b2 [n,1]); order of
    ddetrs_dn.py:53-66: all weights first, then all biases."""
    ch = DYNAMIC_MASK_CHANNELS
    cin = in_channels + 2 if rel_coord else in_channels
    n = params.shape[0]
    assert params.dim() == 2 and params.shape[1] == cin * ch + ch * ch + ch + ch + ch + 1
    w0, w1, w2, b0, b1, b2 = torch.split_with_sizes(params, [cin * ch, ch * ch, ch, ch, ch, 1], dim=1)
    return w0.reshape(n, ch, cin), w1.reshape(n, ch, ch), w2.reshape(n, 1, ch), b0, b1, b2


def compute_locations(h, w, device, stride=1):
    xs = torch.arange(0, w * stride, step=stride, dtype=torch.float32, device=device)
    ys = torch.arange(0, h * stride, step=stride, dtype=torch.float32, device=device)
    yy, xx = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack((xx.reshape(-1), yy.reshape(-1)), dim=1) + stride // 2


def _aligned_bilinear_torch(tensor, factor):
    if factor == 1:
        return tensor
    h, w = tensor.shape[2:]
    t = F.pad(tensor, pad=(0, 1, 0, 1), mode="replicate")
    t = F.interpolate(t, size=(factor * h + 1, factor * w + 1), mode="bilinear", align_corners=True)
    t = F.pad(t, pad=(factor // 2, 0, factor // 2, 0), mode="replicate")
    return t[:, :, :factor * h, :factor * w]


class AlignedBilinearFunction(torch.autograd.Function):
    """aligned_bilinear (ddetrs_dn.py:1174-1196) of a [n, 1, h, w] fp32 GPU tensor with its gradient: both are one HIP kernel
    (include/dynmask_hip.h); the backward is a fixed-order gather, bitwise repeatable."""

    @staticmethod
    def forward(ctx, tensor, factor):
        ctx.factor = int(factor)
        return _ext.aligned_bilinear_forward(tensor.contiguous(), ctx.factor)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        return _ext.aligned_bilinear_backward(grad_out.contiguous(), ctx.factor), None


def aligned_bilinear(tensor, factor):
    assert tensor.dim() == 4 and factor >= 1 and int(factor) == factor
    factor = int(factor)
    if factor == 1:
        return tensor
    if tensor.is_cuda and tensor.dtype == torch.float32 and tensor.shape[1] == 1:
        if torch.is_grad_enabled() and tensor.requires_grad:
            return AlignedBilinearFunction.apply(tensor, factor)
        return _ext.aligned_bilinear_forward(tensor.contiguous(), factor)
    return _aligned_bilinear_torch(tensor, factor)


def _dynamic_convs_torch(mask_feats, inst_xy, params, num_insts, stride, rel_coord):
    """Differentiable composition, [n_all, H, W]."""
    n_img, c, h, w = mask_feats.shape
    w0, w1, w2, b0, b1, b2 = parse_dynamic_params(params, c, rel_coord)
    loc = compute_locations(h, w, mask_feats.device, stride)            # [HW, 2]
    outs, first = [], 0
    for b, cnt in enumerate(num_insts):
        sl = slice(first, first + cnt)
        feat = mask_feats[b].reshape(c, h * w)                           # shared by the image's instances
        h0 = torch.matmul(w0[sl, :, (2 if rel_coord else 0):], feat) + b0[sl, :, None]
        if rel_coord:
            rel = (inst_xy[sl, None, :] - loc[None, :, :]).float()        # [cnt, HW, 2]; `.float()` as ddetrs_dn.py:783
            h0 = h0 + torch.matmul(w0[sl, :, :2], rel.transpose(1, 2).to(w0.dtype))
        h1 = torch.relu(torch.bmm(w1[sl], torch.relu(h0)) + b1[sl, :, None])
        outs.append((torch.bmm(w2[sl], h1) + b2[sl, :, None]).reshape(cnt, h, w))
        first += cnt
    return torch.cat(outs, 0) if outs else mask_feats.new_zeros((0, h, w))


class DynMaskFunction(torch.autograd.Function):
    """`apply(mask_feats [N, 8, H, W], inst_xy [n, 2], params [n, P], counts (tuple of ints), stride, rel_coord)` -> mask logits
    [n, H, W]: the three per-instance 1x1 convolutions of mask_heads_forward (ddetrs_dn.py:734-752) on the inputs of :765-808, and
    their gradients with respect to the mask features, the instance parameters and the instance reference points
    (include/dynmask_hip.h: dynmask_hip_backward_f32).  Saved for backward: the three inputs only."""

    @staticmethod
    def forward(ctx, mask_feats, inst_xy, params, counts, stride, rel_coord):
        mask_feats, inst_xy, params = mask_feats.contiguous(), inst_xy.contiguous(), params.contiguous()
        ctx.counts, ctx.stride, ctx.rel_coord = tuple(int(c) for c in counts), int(stride), bool(rel_coord)
        ctx.save_for_backward(mask_feats, inst_xy, params)
        return _ext.dynmask_forward(mask_feats, inst_xy, params, ctx.counts, ctx.stride, ctx.rel_coord)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_logits):
        mask_feats, inst_xy, params = ctx.saved_tensors
        g_feats, g_params, g_xy = _ext.dynmask_backward(mask_feats, inst_xy, params, ctx.counts, ctx.stride, ctx.rel_coord,
                                                        grad_logits.contiguous(), need_xy=ctx.needs_input_grad[1],
                                                        need_feats=ctx.needs_input_grad[0], need_params=ctx.needs_input_grad[2])
        return (g_feats, g_xy, g_params, None, None, None)


def dynamic_mask_logits(mask_feats, reference_points, mask_head_params, num_insts, mask_feat_stride, rel_coord=True):
    """Mask logits at the feature stride, [n_inst_all, 1, H, W] (ddetrs_dn.py:765-822)."""
    n_img, c, h, w = mask_feats.shape
    inst_xy = reference_points.reshape(-1, 2)
    if inst_xy.dtype not in (torch.float32, torch.float64):
        inst_xy = inst_xy.float()
    params = torch.flatten(mask_head_params, 0, 1)
    counts = [int(n) for n in num_insts]
    needs_grad = torch.is_grad_enabled() and any(t.requires_grad for t in (mask_feats, reference_points, mask_head_params))
    if _ext.dynmask_supported(mask_feats) and params.dtype == torch.float32 and inst_xy.dtype == torch.float32 \
            and inst_xy.is_cuda and params.is_cuda \
            and not (needs_grad and (len(counts) > _ext._lib.DYNMASK_BWD_MAX_BATCH or torch.is_autocast_enabled())):
        if needs_grad:
            logits = DynMaskFunction.apply(mask_feats, inst_xy, params, tuple(counts), mask_feat_stride, rel_coord)
        else:
            logits = _ext.dynmask_forward(mask_feats.contiguous(), inst_xy.contiguous(), params.contiguous(), counts,
                                          mask_feat_stride, rel_coord)
    else:
        logits = _dynamic_convs_torch(mask_feats, inst_xy, params, counts, mask_feat_stride, rel_coord)
    return logits.reshape(-1, 1, h, w)


def dynamic_mask_with_coords(mask_feats, reference_points, mask_head_params, num_insts, mask_feat_stride,
                             rel_coord=True, mask_out_stride=4):
    """mask_feats [N, C, H, W]; reference_points [1, n_all, 2] (input pixels); mask_head_params [1, n_all, P];
    num_insts per image.  Returns mask logits [1, n_all, H*f, W*f], f = mask_feat_stride / mask_out_stride."""
    assert mask_feat_stride >= mask_out_stride and mask_feat_stride % mask_out_stride == 0
    n_all = reference_points.shape[1]
    h, w = mask_feats.shape[2:]
    if n_all == 0:   # the reference returns the (empty) inputs plus a zero that keeps the graph connected (:819-821)
        return mask_feats.new_zeros((1, 0, h, w)) + torch.sum(mask_head_params) * 0.0
    logits = dynamic_mask_logits(mask_feats, reference_points, mask_head_params, num_insts, mask_feat_stride, rel_coord)
    logits = aligned_bilinear(logits, int(mask_feat_stride / mask_out_stride))
    return logits.reshape(1, -1, logits.shape[-2], logits.shape[-1])


# ------------------------------------------------------------------------------------------------
# Static mask-feature branch: MaskHeadSmallConv (ddetrs_dn.py:923-1031; the same class in models/ddetrs.py:670)

def _expand(tensor, length):
    return tensor.unsqueeze(1).repeat(1, int(length), 1, 1, 1).flatten(0, 1)   # ddetrs_dn.py:1112-1113


def _packed_weight(conv):
    """Split-bf16 packed copy of conv.weight, cached on the module and rebuilt when the parameter changes."""
    return packed_weight(conv, _ext.conv3x3_pack_weight)


def _hip_conv_ok(x, conv):
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad)
    return (not needs_grad and tuple(conv.kernel_size) == (3, 3) and tuple(conv.padding) == (1, 1)
            and tuple(conv.stride) == (1, 1) and tuple(conv.dilation) == (1, 1) and conv.groups == 1
            and _ext.conv3x3_supported(x, conv.weight))


def conv3x3_relu(x, conv, exact=True, own_exact=True):
    """`F.relu(conv(x))` for a 3x3 / padding 1 nn.Conv2d.
    exact=True (default): fp32 arithmetic as in the reference.  own_exact=True (the default since round 6): through this library's
    own exact-fp32 MFMA convolution (conv3x3_hip_packed_exact_f32, include/conv3x3_hip.h: halo tiles, v_mfma_f32_32x32x2_f32, one
    fixed-order FMA chain per output -- bitwise repeatable; cached packed weights; inference only) -- 713 us for the head's module
    forward at bs 2 against 743 us through MIOpen, ahead on every one of the five layers since the halo travels as 16-byte loads
    (profiles/r06_conv3x3_exact.txt; round 5 had withdrawn the kernel at 929-939 us).  own_exact=False: the PyTorch-ROCm / MIOpen
    convolution, which also takes whatever the kernel does not (input channels not a multiple of 16, autograd, CPU).
    exact=False opts into the split-bf16 MFMA kernels of include/conv3x3_hip.h from cached packed weights (~390 us, ~2e-5 of
    the output scale, inside the 1e-4 parity bound); layers they do not take (input channels not a multiple of 16) go
    through conv3x3_hip_f32.  Training, CPU, other dtypes or geometries: PyTorch."""
    if exact:
        if own_exact and conv.weight.shape[1] % 16 == 0 and _hip_conv_ok(x, conv):
            pe = packed_weight(conv, lambda w: _ext.conv3x3_pack_weight(w, exact=True), slot="_msda_packed_exact")
            return _ext.conv3x3_packed_forward(x.contiguous(), pe, conv.weight.shape[0], conv.bias, relu=True, exact=True)
        return F.relu(conv(x))
    if _hip_conv_ok(x, conv):
        if conv.weight.shape[1] % 16 == 0:
            return _ext.conv3x3_packed_forward(x.contiguous(), _packed_weight(conv), conv.weight.shape[0], conv.bias, relu=True)
        return _ext.conv3x3_forward(x.contiguous(), conv.weight.contiguous(), conv.bias, relu=True)
    return F.relu(conv(x))


class MaskHeadSmallConv(CachedModuleMixin, torch.nn.Module):
    """Simple convolutional head, FPN-style up-sampling (ddetrs_dn.py:923-1031): same constructor arguments, same
    parameter names (lay1..lay4, jia_dcn, adapter1..3), same initialisation, same forward; the five
    `F.relu(self.layN(...))` steps go through conv3x3_relu.  `use_raft` is not covered (False in every shipped
    config, uninext/config.py:178)."""

    # True (default): the reference's fp32 arithmetic (MIOpen convolutions); False -- or env UNINEXT_AMD_SPLIT_BF16=1 -- opts
    # into the split-bf16 MFMA kernels (3 of 4 partial products, ~2e-5 of the output scale; the fast ones)
    exact_fp32 = os.environ.get("UNINEXT_AMD_SPLIT_BF16", "0") != "1"
    # exact fp32 through this library's own MFMA convolution (default since round 6: 4 % ahead of MIOpen on the head, see conv3x3_relu);
    # False: MIOpen.  Inference only either way: under autograd the convolutions are PyTorch's
    own_exact_conv = True

    def __init__(self, dim, fpn_dims, context_dim, use_raft=False, up_rate=4):
        super().__init__()
        if use_raft:
            raise NotImplementedError("MaskHeadSmallConv(use_raft=True) is not part of this path")
        self.use_raft = False
        self.out_stride = 2
        self.up_rate = up_rate
        inter_dims = [dim, context_dim, context_dim, context_dim, context_dim, context_dim]
        self.lay1 = torch.nn.Conv2d(dim, dim // 4, 3, padding=1)
        self.lay2 = torch.nn.Conv2d(dim // 4, dim // 32, 3, padding=1)
        self.lay3 = torch.nn.Conv2d(inter_dims[1], inter_dims[2], 3, padding=1)
        self.lay4 = torch.nn.Conv2d(inter_dims[2], inter_dims[3], 3, padding=1)
        self.jia_dcn = torch.nn.Conv2d(inter_dims[3], inter_dims[4], 3, padding=1)
        self.dim = dim
        if fpn_dims is not None:
            self.adapter1 = torch.nn.Conv2d(fpn_dims[0], inter_dims[1], 1)
            self.adapter2 = torch.nn.Conv2d(fpn_dims[1], inter_dims[2], 1)
            self.adapter3 = torch.nn.Conv2d(fpn_dims[2], inter_dims[3], 1)
        for m in self.modules():
            if isinstance(m, torch.nn.Conv2d):
                torch.nn.init.kaiming_uniform_(m.weight, a=1)
                torch.nn.init.constant_(m.bias, 0)

    def _merge(self, skip, adapter, fpn, fused):
        if fpn is not None:
            cur = adapter(fpn)
            if cur.size(0) != skip.size(0):
                cur = _expand(cur, skip.size(0) // cur.size(0))
            skip = (cur + skip) / 2
        if fused is None:
            return skip
        needs_grad = torch.is_grad_enabled() and (skip.requires_grad or fused.requires_grad)
        if not needs_grad and skip.is_cuda and skip.dtype == torch.float32 and fused.dtype == torch.float32 \
                and fused.is_contiguous():
            return _ext.upsample_add(skip.contiguous(), fused)      # one pass: no interpolate output, no separate add
        return skip + F.interpolate(fused, size=skip.shape[-2:], mode="nearest")

    def forward(self, x, fpns):
        f = fpns if fpns is not None else (None, None, None)
        e, o = self.exact_fp32, self.own_exact_conv
        fused = conv3x3_relu(self._merge(x[-1], getattr(self, "adapter1", None), f[0], None), self.lay3, e, o)
        fused = conv3x3_relu(self._merge(x[-2], getattr(self, "adapter2", None), f[1], fused), self.lay4, e, o)
        fused_fpn = conv3x3_relu(self._merge(x[-3], getattr(self, "adapter3", None), f[2], fused), self.jia_dcn, e, o)
        fused = conv3x3_relu(fused_fpn, self.lay1, e, o)
        return conv3x3_relu(fused, self.lay2, e, o)
