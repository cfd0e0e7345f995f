"""Drop-in for the reference's compiled extension of the same name.

The reference does `import MultiScaleDeformableAttention as MSDA` and calls
`MSDA.ms_deform_attn_forward / ms_deform_attn_backward`
(projects/UNINEXT/uninext/models/deformable_detr/ops/functions/ms_deform_attn_func.py:18,26,36).
With this repository on sys.path that import resolves here, i.e. to the gfx950 HIP kernels
behind include/msda_hip.h; the reference's MSDeformAttnFunction / MSDeformAttn / ops/test.py
run unmodified on top.
"""
from uninext_amd.ext import ms_deform_attn_backward, ms_deform_attn_forward  # noqa: F401

__all__ = ["ms_deform_attn_forward", "ms_deform_attn_backward"]
