"""Autograd surface of the operator: `MSDeformAttnFunction.apply(value, spatial_shapes,
level_start_index, sampling_locations, attention_weights, im2col_step)`.

Mirror of the reference's ops/functions/ms_deform_attn_func.py:21-40: inputs are cast to
float32 under autocast (`custom_fwd(cast_inputs=torch.float32)`, :23), five tensors are saved
(:28), backward is once-differentiable and returns gradients for value, sampling locations and
attention weights only (:40).  The compute is the gfx950 HIP library via uninext_amd.ext; there
is no PyTorch fallback here (the reference's `ms_deform_attn_core_pytorch`, :43-63, lives on as
test infrastructure under oracle/).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import ext as MSDA


class MSDeformAttnFunction(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        ctx.im2col_step = im2col_step
        ctx.call_site = MSDA.explicit_call_site()       # backward runs on the autograd thread: same site there
        output = MSDA.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                             sampling_locations, attention_weights, ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return output

    @staticmethod
    @once_differentiable
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_output):
        value, shapes, level_start, locations, weights = ctx.saved_tensors
        with MSDA.reenter(ctx.call_site):
            grad_value, grad_loc, grad_attn = MSDA.ms_deform_attn_backward(
                value, shapes, level_start, locations, weights, grad_output.contiguous(), ctx.im2col_step)
        return grad_value, None, None, grad_loc, grad_attn, None


class MSDeformAttnFusedFunction(Function):
    """`apply(value, spatial_shapes, level_start_index, reference_points, sampling_offsets, attention_logits, num_points,
    im2col_step = 64)`: MSDeformAttn's prologue AND the operator as one differentiable function (training side).

    forward   msda_hip_forward_fused_f32: softmax, sampling locations and sampling in one kernel from the RAW Linear outputs --
              `sampling_locations` (45.5 MB per encoder call at N = 2) and the softmaxed weights are never written.
    backward  recomputes them (msda_hip_prologue_f32, the reference's operations in the reference's order,
              ops/modules/ms_deform_attn.py:99-112), runs msda_hip_backward_f32 -- the same kernels MSDeformAttnFunction uses
              -- and maps grad_sampling_loc / grad_attn_weight back onto the raw tensors (msda_hip_prologue_backward_f32:
              softmax backward + the location chain rule, and the reference points' gradient when they require one).
    What is saved for backward: value and the raw tensors, which autograd keeps alive anyway as outputs of the Linears.
    fp32, GPU, 32 channels per head, levels * points == 16 (uninext_amd.ext.fused_forward_supported)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, reference_points, sampling_offsets,
                attention_logits, num_points, im2col_step=64):
        ctx.num_points = int(num_points)
        ctx.im2col_step = int(im2col_step)
        # the reference's host code checks batch % min(batch, im2col_step) in FORWARD (ops/src/cuda/ms_deform_attn_cuda.cu:50-52);
        # the fused kernel has no such step, but the backward operator it pairs with does: fail here, not in backward
        step = min(value.shape[0], ctx.im2col_step)
        if value.shape[0] > 0 and (step <= 0 or value.shape[0] % step != 0):
            raise RuntimeError("batch(%d) must divide im2col_step(%d)" % (value.shape[0], step))
        ctx.call_site = MSDA.explicit_call_site()
        value, reference_points = value.contiguous(), reference_points.contiguous()
        sampling_offsets, attention_logits = sampling_offsets.contiguous(), attention_logits.contiguous()
        output = MSDA.ms_deform_attn_forward_fused(value, value_spatial_shapes, value_level_start_index, reference_points,
                                                   sampling_offsets, attention_logits, ctx.num_points)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, reference_points, sampling_offsets,
                              attention_logits)
        return output

    @staticmethod
    @once_differentiable
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_output):
        value, shapes, level_start, ref, offsets, logits = ctx.saved_tensors
        M = value.shape[2]
        locations, weights = MSDA.msda_prologue(shapes, ref, offsets, logits, M, ctx.num_points)
        with MSDA.reenter(ctx.call_site):
            grad_value, grad_loc, grad_attn = MSDA.ms_deform_attn_backward(
                value, shapes, level_start, locations, weights, grad_output.contiguous(), ctx.im2col_step)
        del locations                                       # 45.5 MB per encoder call: not needed behind the operator's backward
        g_off, g_logits, g_ref = MSDA.msda_prologue_backward(shapes, ref, offsets, weights, grad_loc, grad_attn,
                                                              need_grad_reference=ctx.needs_input_grad[3], inplace=True)
        return (grad_value, None, None, g_ref, g_off, g_logits, None, None)[:len(ctx.needs_input_grad)]   # (im2col_step is optional)
