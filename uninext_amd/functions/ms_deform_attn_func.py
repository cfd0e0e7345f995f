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
        ctx.call_site = MSDA.current_call_site()        # backward runs on the autograd thread: same site there
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
        with MSDA.call_site(ctx.call_site):
            grad_value, grad_loc, grad_attn = MSDA.ms_deform_attn_backward(
                value, shapes, level_start, locations, weights, grad_output.contiguous(), ctx.im2col_step)
        return grad_value, None, None, grad_loc, grad_attn, None
