"""`DeformableTransformerEncoderLayer` -- the caller of the MSDeformAttn path in the encoder.

Host-side mirror of projects/UNINEXT/uninext/models/deformable_detr/deformable_transformer_dino.py:330-370: same
constructor arguments, parameter names (`self_attn`, `norm1`, `linear1`, `linear2`, `norm2`; reference checkpoints load
unchanged) and forward.  When autograd records (training) the layer is the reference's composition of PyTorch ops
around `MSDeformAttn`.  At inference on the GPU (dropout is the identity):

  * `with_pos_embed(src, pos)` is folded into the operand load of the sampling_offsets / attention_weights projections;
  * `src + dropout(src2)` followed by `normN` runs in the epilogue of the Linear that produced src2 (output_proj,
    linear2: `linear_hip_packed_ln_f32`; a stand-alone add + LayerNorm kernel, include/layernorm_hip.h, covers other widths);
  * `linear1` + ReLU + `linear2` + residual + `norm2` run as ONE kernel (`linear_hip_packed_ffn_f32`: the hidden
    activations never leave the CU); `fuse_ffn = False` or other activations / widths take the two Linear kernels.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .. import ext as MSDA
from .._cache import CachedModuleMixin
from .ms_deform_attn import MSDeformAttn


def _get_activation_fn(activation):
    if activation == "relu":
        return F.relu
    if activation == "gelu":
        return F.gelu
    if activation == "glu":
        return F.glu
    raise RuntimeError(F"activation should be relu/gelu, not {activation}.")


class DeformableTransformerEncoderLayer(CachedModuleMixin, nn.Module):
    fuse_ffn = True   # inference: linear1 + ReLU + linear2 + residual + norm2 as one kernel (include/linear_hip.h)

    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation)
        self._relu = activation == "relu"
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def _inference(self, *tensors):
        if self.training and any(d.p > 0 for d in (self.dropout1, self.dropout2, self.dropout3)):
            return False
        if torch.is_grad_enabled() and (any(t is not None and t.requires_grad for t in tensors)
                                        or any(p.requires_grad for p in self.parameters())):
            return False
        return all(t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()) for t in tensors)

    def _add_norm(self, x, residual, norm):
        if MSDA.add_layernorm_supported(x, norm.normalized_shape) and norm.elementwise_affine:
            return MSDA.add_layernorm(x.contiguous(), residual, norm.weight, norm.bias, norm.eps)
        return norm(x + residual)

    def forward_ffn(self, src):
        src2 = self.linear2(self.dropout2(self.activation(self.linear1(src))))
        src = src + self.dropout3(src2)
        return self.norm2(src)

    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None):
        if not self._inference(src, pos):
            src2 = self.self_attn(self.with_pos_embed(src, pos), reference_points, src, spatial_shapes,
                                  level_start_index, padding_mask)
            src = self.norm1(src + self.dropout1(src2))
            return self.forward_ffn(src)
        attn = self.self_attn
        src = attn(src, reference_points, src, spatial_shapes, level_start_index, padding_mask, query_pos=pos,
                   residual_norm=(src, self.norm1))                       # norm1(src + attention) in output_proj's epilogue
        if self._relu and self.fuse_ffn:
            out = attn._ffn_norm(self.linear1, self.linear2, src, self.norm2)   # the whole FFN block in one kernel
            if out is not None:
                return out
        hidden = attn._project(self.linear1, src, relu=self._relu)
        if not self._relu:
            hidden = self.activation(hidden)
        return attn._project_norm(self.linear2, hidden, src, self.norm2)  # norm2(src + ffn) in linear2's epilogue
