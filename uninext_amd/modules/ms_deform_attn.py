"""`MSDeformAttn` -- the multi-scale deformable attention layer that owns the operator.

Host-side mirror of the reference's ops/modules/ms_deform_attn.py:30-116: same constructor arguments, parameter
names (`sampling_offsets`, `attention_weights`, `value_proj`, `output_proj` -- reference checkpoints load unchanged),
initialisation (:54-76), argument meaning and errors of `forward` (:79-116).  The sampling runs in libmsda_hip.so:

  * when gradients are needed: `MSDeformAttnFusedFunction` -- the fused forward kernel on the raw Linear outputs, and a
    backward that recomputes locations / weights in one kernel, runs the operator's backward kernels and maps the
    gradients back in one more (`fuse_training_prologue`, default on; off or unsupported geometry: softmax + sampling
    locations in PyTorch, then `MSDeformAttnFunction`, exactly the reference's data flow);
  * otherwise (inference): `ms_deform_attn_forward_fused` -- the kernel takes the raw Linear outputs and the
    reference points and does softmax, location arithmetic and sampling in one pass.
    Set `MSDeformAttn.fuse_prologue = False` (or env UNINEXT_AMD_NO_FUSED=1) to force the two-step path.

The four projections are fp32 PyTorch-ROCm GEMMs (hipBLASLt), as in the reference -- the DEFAULT.  Opt-in for inference:
`MSDeformAttn.fast_linear = True` (or env UNINEXT_AMD_SPLIT_BF16=1) runs them through include/linear_hip.h -- split-bf16
products on the matrix cores (3 of the 4 partial products of a bf16 hi/lo split) from weights packed once per module,
~2e-5 of the output scale per layer on unit-scale data (tests/test_linear_gpu.py holds the bound at trained-checkpoint
scales as well), with the padding-mask fill folded into value_proj's epilogue.  Narrower arithmetic than the reference's
is never the out-of-the-box behaviour.
"""
import math
import os
import warnings

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, xavier_uniform_

from .. import ext as MSDA
from .._cache import CachedModuleMixin, CheckedOnce, packed_weight, packed_weight_pair
from ..functions import MSDeformAttnFunction, MSDeformAttnFusedFunction


def _is_power_of_2(n):
    if not isinstance(n, int) or n < 0:
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return n != 0 and (n & (n - 1)) == 0


class MSDeformAttn(CachedModuleMixin, nn.Module):
    fuse_prologue = os.environ.get("UNINEXT_AMD_NO_FUSED", "0") != "1"
    # training: MSDeformAttnFusedFunction (fused forward from the raw Linear outputs; backward recomputes the locations / weights
    # and runs the operator's backward kernels) instead of the PyTorch prologue + MSDeformAttnFunction
    # (class attribute; UNINEXT_AMD_NO_FUSED=1 switches both fusions off, UNINEXT_AMD_NO_FUSED_TRAINING=1 this one alone -- the
    # switch of rounds 3-4 that round 5 had dropped without notice: ADVICE r05)
    fuse_training_prologue = os.environ.get("UNINEXT_AMD_NO_FUSED_TRAINING", "0") != "1"
    fast_linear = os.environ.get("UNINEXT_AMD_SPLIT_BF16", "0") == "1"   # opt-in: split-bf16 projections at inference

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("MSDeformAttn: a per-head dimension that is a power of 2 (and a multiple of 4) takes the "
                          "lane-group HIP kernels; other sizes fall back to the generic kernels.")
        self.im2col_step = 64  # kept for interface parity (ops/modules/ms_deform_attn.py:48); no effect on the HIP path
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points

        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        # ops/modules/ms_deform_attn.py:61-76: zero offset weights, a ring of unit directions (one per head)
        # scaled by the point index as the offset bias, uniform attention, xavier projections.
        constant_(self.sampling_offsets.weight.data, 0.0)
        angle = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        ring = torch.stack([angle.cos(), angle.sin()], -1)
        ring = ring / ring.abs().max(-1, keepdim=True)[0]
        ring = ring.view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        ring = ring * torch.arange(1, self.n_points + 1, dtype=torch.float32).view(1, 1, self.n_points, 1)
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(ring.reshape(-1))
        constant_(self.attention_weights.weight.data, 0.0)
        constant_(self.attention_weights.bias.data, 0.0)
        xavier_uniform_(self.value_proj.weight.data)
        constant_(self.value_proj.bias.data, 0.0)
        xavier_uniform_(self.output_proj.weight.data)
        constant_(self.output_proj.bias.data, 0.0)

    # -- argument check of :93 without a device->host sync per call -----------------------------------------
    _shape_checks = CheckedOnce()
    _instances = 0          # every instance is a call site of its own for the library's forward-kernel choice (sites 1..47;
                            # 48..63 belong to the sites ext.py derives for callers that pass none)

    def _site(self):
        site = self.__dict__.get("_msda_site")
        if site is None:
            cls = MSDeformAttn
            cls._instances += 1
            site = self.__dict__["_msda_site"] = 1 + (cls._instances - 1) % 47
        return site

    @classmethod
    def _check_shapes(cls, spatial_shapes, len_in):
        """`assert (H * W).sum() == Len_in` (ops/modules/ms_deform_attn.py:93).  On a GPU tensor the comparison is a
        host sync: it is done once per shapes TENSOR OBJECT (weak reference + version, so a new tensor that happens to
        reuse a freed address is checked again; inference tensors carry no version and are keyed by identity alone)
        and skipped while a HIP graph is being captured.  A fresh shapes tensor per forward -- what Deformable-DETR
        builds -- is checked on every call, exactly as in the reference."""
        if spatial_shapes.is_cuda:
            if torch.cuda.is_current_stream_capturing():
                return
            if cls._shape_checks.hit(spatial_shapes, int(len_in)):
                return
            assert (spatial_shapes[:, 0] * spatial_shapes[:, 1]).sum() == len_in
            cls._shape_checks.add(spatial_shapes, int(len_in))
            return
        assert (spatial_shapes[:, 0] * spatial_shapes[:, 1]).sum() == len_in

    # -- projections ------------------------------------------------------------------------------------------
    def _fast_ok(self, lin, x):
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or lin.weight.requires_grad)
        return self.fast_linear and not needs_grad and x.is_contiguous() and MSDA.linear_packed_supported(x, lin.weight)

    def _packed(self, lin):
        """Packed copy of lin.weight (include/linear_hip.h), cached on the Linear, rebuilt when the parameter changes
        (uninext_amd/_cache.py: version counter, train()/eval(), load_state_dict, or invalidate_packed())."""
        return packed_weight(lin, MSDA.linear_pack_weight)

    def _project_norm(self, lin, x, residual, norm):
        """`norm(residual + lin(x))`: at inference the add and the LayerNorm run in the Linear's epilogue."""
        if (self._fast_ok(lin, x) and norm.elementwise_affine and residual.is_contiguous()
                and not (torch.is_grad_enabled() and residual.requires_grad)
                and MSDA.linear_packed_ln_supported(x, lin.weight, norm.normalized_shape)):
            return MSDA.linear_packed_ln(x, self._packed(lin), lin.bias, residual, norm.weight, norm.bias, norm.eps)
        return norm(residual + self._project(lin, x))

    def _ffn_norm(self, lin1, lin2, x, norm):
        """`norm(x + lin2(relu(lin1(x))))` in one kernel when include/linear_hip.h covers the shapes (d_model 256,
        d_ffn % 128 == 0, inference), else None."""
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or lin1.weight.requires_grad or lin2.weight.requires_grad)
        if (self.fast_linear and not needs_grad and x.is_contiguous() and norm.elementwise_affine
                and MSDA.ffn_packed_supported(x, lin1.weight, lin2.weight, norm.normalized_shape)):
            return MSDA.ffn_packed(x, self._packed(lin1), lin1.bias, self._packed(lin2), lin2.bias, lin1.weight.shape[0],
                                   x, norm.weight, norm.bias, norm.eps)
        return None

    def _project_offsets_and_logits(self, query, query_pos):
        """sampling_offsets(query) and attention_weights(query) (ops/modules/ms_deform_attn.py:99-100).  Inference on the
        GPU: ONE product with the concatenated weights and two outputs (linear_hip_packed_split_f32) -- the query, and the
        positional embedding added to it, are read once instead of twice; every element is the same sum of products."""
        so, aw = self.sampling_offsets, self.attention_weights
        n_off = so.weight.shape[0]
        if (self._fast_ok(so, query) and self._fast_ok(aw, query) and n_off % 128 == 0
                and (query_pos is None or (query_pos.is_contiguous() and query_pos.shape == query.shape
                                           and query_pos.dtype == query.dtype and not query_pos.requires_grad))):
            packed, bias = packed_weight_pair(self, so, aw, MSDA.linear_pack_weight)
            return MSDA.linear_packed_split_forward(query, packed, n_off, n_off + aw.weight.shape[0], bias, query_pos)
        return self._project(so, query, x_add=query_pos), self._project(aw, query, x_add=query_pos)

    def _project(self, lin, x, row_mask=None, head_major_rows=0, x_add=None, relu=False):
        """`lin(x)` (then zero the rows where row_mask is True).  Inference on the GPU: include/linear_hip.h from a
        packed copy of the weight cached on the Linear and rebuilt when the parameter changes; head_major_rows = S
        writes the result as [N, heads, S, 32] (only requested when _fast_ok); x_add is added to the input while it is
        loaded, relu applied in the epilogue."""
        if self._fast_ok(lin, x) and (x_add is None or (x_add.is_contiguous() and x_add.shape == x.shape
                                                        and x_add.dtype == x.dtype and not x_add.requires_grad)):
            mask = row_mask.contiguous() if row_mask is not None else None
            return MSDA.linear_packed_forward(x, self._packed(lin), lin.weight.shape[0], lin.bias, mask, head_major_rows,
                                              x_add, relu)
        assert head_major_rows == 0
        y = lin(x if x_add is None else x + x_add)
        if relu:
            y = F.relu(y)
        if row_mask is not None:
            y = y.masked_fill(row_mask[..., None], float(0))
        return y

    # -- the two ways to run the sampling ---------------------------------------------------------------------
    def _sample_autograd(self, value, shapes, level_start, reference_points, offsets, logits):
        """Reference data flow (:99-113): PyTorch prologue, then the differentiable operator."""
        N, Lq = offsets.shape[:2]
        M, L, P = self.n_heads, self.n_levels, self.n_points
        offsets = offsets.view(N, Lq, M, L, P, 2)
        weights = F.softmax(logits.view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
        if reference_points.shape[-1] == 2:
            wh = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
            locations = reference_points[:, :, None, :, None, :] + offsets / wh[None, None, None, :, None, :]
        else:
            locations = reference_points[:, :, None, :, None, :2] \
                + offsets / P * reference_points[:, :, None, :, None, 2:] * 0.5
        return MSDeformAttnFunction.apply(value, shapes, level_start, locations, weights, self.im2col_step)

    def _records_grad(self, *tensors):
        """True when autograd would record through this layer: the fused / head-major sampling entry points have no
        backward, so ANY trainable parameter of the layer (e.g. sampling_offsets fine-tuned on a frozen value_proj)
        or any input that carries grad sends the call down the reference's differentiable data flow."""
        return torch.is_grad_enabled() and (any(t is not None and t.requires_grad for t in tensors)
                                            or any(p.requires_grad for p in self.parameters()))

    def _can_fuse(self, value, reference_points, offsets, logits):
        if not self.fuse_prologue or not MSDA.fused_forward_supported(value, reference_points, self.n_levels,
                                                                      self.n_points):
            return False
        if torch.is_grad_enabled() and any(t.requires_grad for t in (value, reference_points, offsets, logits)):
            return False   # the fused entry point has no backward
        return all(t.is_contiguous() for t in (value, reference_points, offsets, logits))

    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None, query_pos=None, residual_norm=None):
        """query (N, Lq, C); reference_points (N, Lq, n_levels, 2|4) in [0,1]; input_flatten (N, sum HW, C);
        input_spatial_shapes (n_levels, 2) int64 (H, W); input_level_start_index (n_levels,) int64;
        input_padding_mask (N, sum HW) bool, True = padding.  Returns (N, Lq, C).
        query_pos (extension, optional): the layer input is query + query_pos (`with_pos_embed`), added inside the
        projections instead of by a separate kernel.
        residual_norm (extension, optional): (residual, LayerNorm) -- return norm(residual + output) with the add and
        the normalisation in output_proj's epilogue."""
        N, Len_in, _ = input_flatten.shape
        self._check_shapes(input_spatial_shapes, Len_in)
        if reference_points.shape[-1] not in (2, 4):
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(
                reference_points.shape[-1]))

        head_dim = self.d_model // self.n_heads
        # encoder-sized inference calls: the value projection writes [N, heads, S, 32] and the fused kernel reads it
        # that way (a head's pixels 128 bytes apart instead of 1 KB: 6-15 % off the sampling kernel)
        head_major = (self.fuse_prologue and self._fast_ok(self.value_proj, input_flatten) and head_dim == 32
                      and MSDA.fused_forward_hm_supported((head_dim,), self.n_levels, self.n_points, query.shape[1])
                      and not self._records_grad(query, reference_points, input_flatten)
                      and reference_points.is_contiguous())
        value = self._project(self.value_proj, input_flatten, input_padding_mask, Len_in if head_major else 0)
        if not head_major:
            value = value.view(N, Len_in, self.n_heads, head_dim)
        offsets, logits = self._project_offsets_and_logits(query, query_pos)          # (N, Lq, M*L*P*2), (N, Lq, M*L*P)

        with MSDA.call_site(self._site()):     # the library picks its encoder-forward kernel per call site
            if head_major:
                sampled = MSDA.ms_deform_attn_forward_fused(value, input_spatial_shapes, input_level_start_index,
                                                            reference_points, offsets.contiguous(), logits.contiguous(),
                                                            self.n_points, value_head_major=True)
            elif self._can_fuse(value, reference_points, offsets, logits):
                sampled = MSDA.ms_deform_attn_forward_fused(value, input_spatial_shapes, input_level_start_index,
                                                            reference_points, offsets, logits, self.n_points)
            elif (self.fuse_prologue and self.fuse_training_prologue
                  and MSDA.fused_forward_supported(value, reference_points, self.n_levels, self.n_points)
                  and query.dtype == torch.float32 and not torch.is_autocast_enabled()):
                # autograd records: the same fused forward, differentiable (training side)
                sampled = MSDeformAttnFusedFunction.apply(value, input_spatial_shapes, input_level_start_index,
                                                          reference_points, offsets, logits, self.n_points, self.im2col_step)
            else:
                sampled = self._sample_autograd(value, input_spatial_shapes, input_level_start_index, reference_points,
                                                offsets, logits)
        if residual_norm is not None:
            return self._project_norm(self.output_proj, sampled, residual_norm[0], residual_norm[1])
        return self._project(self.output_proj, sampled)
