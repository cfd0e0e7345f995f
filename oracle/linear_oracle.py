"""CPU oracle for the Linear layers around the operator.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

out = x @ W^T + b in numpy float64, rows with row_mask set to zero: value_proj + masked_fill, sampling_offsets,
attention_weights, output_proj of MSDeformAttn.forward (ops/modules/ms_deform_attn.py:95-100,114).  The arithmetic is
PyTorch's (nn.Linear); there is nothing reference-specific to pin beyond the module-level fixtures.
"""
import numpy as np


def forward(x, weight, bias=None, row_mask=None):
    out = np.asarray(x, dtype=np.float64) @ np.asarray(weight, dtype=np.float64).T
    if bias is not None:
        out = out + np.asarray(bias, dtype=np.float64)
    if row_mask is not None:
        out = np.where(np.asarray(row_mask, dtype=bool)[..., None], 0.0, out)
    return out
