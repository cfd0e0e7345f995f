from .ms_deform_attn_func import MSDeformAttnFunction, MSDeformAttnFusedFunction  # noqa: F401
