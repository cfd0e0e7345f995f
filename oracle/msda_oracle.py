"""ctypes front-end of oracle/msda_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  It runs the scalar C restatement of the reference operator
(ops/src/cuda/ms_deform_im2col_cuda.cuh:33-159,237-403) on host memory.

Accepts numpy arrays or CPU torch tensors; returns numpy arrays.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmsda_oracle.so")
_lib = None


def build(force=False):
    """Compile oracle/msda_oracle.c with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "msda_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libmsda_oracle.so"])
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        i, p = ctypes.c_int, ctypes.c_void_p
        for suf in ("f32", "f64"):
            f = getattr(_lib, "msda_oracle_forward_" + suf)
            f.argtypes = [p, p, p, p, p, i, i, i, i, i, i, i, p]
            f.restype = i
            g = getattr(_lib, "msda_oracle_backward_" + suf)
            g.argtypes = [p, p, p, p, p, p, i, i, i, i, i, i, i, p, p, p]
            g.restype = i
    return _lib


def _np(x, dtype=None):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    x = np.ascontiguousarray(x)
    if dtype is not None and x.dtype != dtype:
        x = np.ascontiguousarray(x.astype(dtype))
    return x


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _dims(value, shapes, loc):
    N, S, M, D = value.shape
    L = shapes.shape[0]
    Lq, P = loc.shape[1], loc.shape[4]
    assert loc.shape == (N, Lq, M, L, P, 2), loc.shape
    return N, S, M, D, L, Lq, P


def forward(value, shapes, lsi, loc, attn):
    """out[N,Lq,M*D] -- dtype follows `value` (float32 or float64)."""
    value = _np(value)
    dt = value.dtype
    assert dt in (np.float32, np.float64)
    shapes, lsi = _np(shapes, np.int64), _np(lsi, np.int64)
    loc, attn = _np(loc, dt), _np(attn, dt)
    N, S, M, D, L, Lq, P = _dims(value, shapes, loc)
    out = np.empty((N, Lq, M * D), dtype=dt)
    fn = getattr(_load(), "msda_oracle_forward_" + ("f32" if dt == np.float32 else "f64"))
    rc = fn(_ptr(value), _ptr(shapes), _ptr(lsi), _ptr(loc), _ptr(attn), N, S, M, D, L, Lq, P, _ptr(out))
    assert rc == 0
    return out


def backward(grad_out, value, shapes, lsi, loc, attn):
    """(grad_value, grad_loc, grad_attn), deterministic sequential accumulation."""
    value = _np(value)
    dt = value.dtype
    assert dt in (np.float32, np.float64)
    shapes, lsi = _np(shapes, np.int64), _np(lsi, np.int64)
    loc, attn, grad_out = _np(loc, dt), _np(attn, dt), _np(grad_out, dt)
    N, S, M, D, L, Lq, P = _dims(value, shapes, loc)
    gv, gl, ga = np.empty_like(value), np.empty_like(loc), np.empty_like(attn)
    fn = getattr(_load(), "msda_oracle_backward_" + ("f32" if dt == np.float32 else "f64"))
    rc = fn(_ptr(grad_out), _ptr(value), _ptr(shapes), _ptr(lsi), _ptr(loc), _ptr(attn),
            N, S, M, D, L, Lq, P, _ptr(gv), _ptr(gl), _ptr(ga))
    assert rc == 0
    return gv, gl, ga
