"""uninext_amd -- MI355X (gfx950) native MultiScaleDeformableAttention path for UNINEXT.

Layout (only what the hot path needs):
  csrc/        hand-written HIP kernels + the C ABI (include/msda_hip.h) -> lib/libmsda_hip.so
  _lib.py      ctypes binding of the C ABI (no torch types cross the boundary)
  ext.py       `ms_deform_attn_forward/backward` with the reference pybind signatures
  functions/   MSDeformAttnFunction   (mirror of ops/functions/ms_deform_attn_func.py)
  modules/     MSDeformAttn nn.Module (mirror of ops/modules/ms_deform_attn.py)
The repo-root module `MultiScaleDeformableAttention` re-exports ext.py under the name the
reference imports (ops/functions/ms_deform_attn_func.py:18).
"""
__version__ = "0.1.0"
