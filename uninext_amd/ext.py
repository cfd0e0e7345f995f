"""The two functions of the reference's pybind module `MultiScaleDeformableAttention`
(ops/src/vision.cpp:13-16), same names / argument order / error behaviour, implemented by
calling the C ABI of libmsda_hip.so with raw device pointers and PyTorch's current stream.

Reference host code mirrored: ops/src/ms_deform_attn.h:21-61 (device dispatch: CPU tensors
raise "Not implemented on the CPU") and ops/src/cuda/ms_deform_attn_cuda.cu:28-52
(contiguity / device / im2col_step checks), :54 and :121-123 (output allocation).
"""
import contextlib
import ctypes
import os
import threading

import torch

from . import _lib
from ._cache import CheckedOnce, tensor_version

_SUFFIX = {torch.float32: "f32", torch.float64: "f64"}
# CPU tensors: the reference raises "Not implemented on the CPU" (ops/src/ms_deform_attn.h:35-38).  This library has
# host-pointer variants of the operator (include/msda_hip.h: msda_host_*, csrc/msda_host.cpp), so
# CPU tensors are served by them -- BASELINE configs[0], the model's plumbing on a GPU-less box, then runs through the
# same `MSDeformAttnFunction` instead of ms_deform_attn_core_pytorch.  MSDA_HIP_STRICT_DEVICE=1 restores the
# reference's error.  GPU tensors never take this route: they go to the HIP kernels or fail loudly.
STRICT_DEVICE = os.environ.get("MSDA_HIP_STRICT_DEVICE", "0") == "1"
HOST_THREADS = int(os.environ.get("MSDA_HOST_THREADS", "0"))   # 0: all hardware threads
# Backward workspace (include/msda_hip.h: msda_hip_backward_workspace_bytes / msda_hip_backward_ws_f32): "1" (the default since
# round 6) lends the library a buffer from PyTorch's caching allocator for every fp32 backward call that may need one -- stream-
# ordered, no hipMalloc / device synchronisation inside the library, capturable, and visible to torch.cuda.memory_stats(); the
# block (~0.7 GB at the R50 training shapes) is cached by the allocator and handed out again call after call.  "0" leaves the
# library its own per-device workspace behind the allocator's back, allocated (hipMalloc, grown with a device sync, never freed)
# only when msda_bwd_regions is actually chosen: eight ranks of a node each did that.
TORCH_WORKSPACE = os.environ.get("MSDA_HIP_TORCH_WORKSPACE", "1") != "0"


# ---- call context of the automatic forward-kernel choice (include/msda_hip.h: msda_hip_set_call_context) ----------------
CTX_GEOMETRY_CHECKED, CTX_DETERMINISTIC = 1, 2
_site = threading.local()
_geometry_checks = CheckedOnce()


@contextlib.contextmanager
def call_site(site):
    """Forward calls made inside the block belong to call site `site` (0..63): the library chooses between its two
    encoder-forward kernels PER SITE.  MSDeformAttn modules wrap their operator calls in their own.  Outside of any block
    the site is derived from the call sequence (`_auto_site`)."""
    prev = getattr(_site, "value", None)
    _site.value = int(site)
    try:
        yield
    finally:
        _site.value = prev


def current_call_site():
    """The call site of the enclosing call_site() block (0 outside of one)."""
    v = getattr(_site, "value", None)
    return 0 if v is None else int(v)


def explicit_call_site():
    """The call site of the enclosing call_site() block, None outside of one -- autograd functions record it in forward
    and re-enter it in backward, which runs on another thread (`reenter`)."""
    v = getattr(_site, "value", None)
    return None if v is None else int(v)


def reenter(site):
    """Context manager for a backward: the forward's explicit call site again, or nothing when it had none (the call is
    then matched to its forward by the derived-site rules below)."""
    return call_site(site) if site is not None else contextlib.nullcontext()


# ---- call sites for callers that pass none: the reference's UNMODIFIED module on top (INTEGRATION.md option A) --------------
# ops/modules/ms_deform_attn.py:113 calls MSDeformAttnFunction.apply with no notion of a call site, from six encoder layers
# (dino.py:338,363) that sample differently.  Deformable-DETR builds `spatial_shapes` once per forward pass of the transformer
# and hands the SAME tensor object to every layer, so the ordinal of an encoder-shaped call since that object was first seen
# is the layer index: site = AUTO_SITE_BASE + ordinal % AUTO_SITES.  The backward of such a call arrives on the autograd
# thread with the forward's saved tensors: it is matched to its forward by the storage of `sampling_loc` (a tensor the
# layer made for this call alone and autograd keeps alive until then).  A caller that repeats ONE call in a loop on the same
# shapes object walks through the 16 derived slots (each settles on its kernel at its third visit): wrap such loops in
# call_site().  This repository's own modules pass explicit sites 1..47 from a counter of instances (the derived range 48..63
# is reserved: round 6); unmatched_backward_calls() counts the backward calls that found no forward.
AUTO_SITE_BASE, AUTO_SITES = 48, 16
_AUTO_BWD_KEEP = 256                      # forward calls remembered for their backward (inference never consumes them)


class _AutoSites:
    def __init__(self):
        self.lock = threading.Lock()
        self.shapes_ref = None            # weak reference to the spatial_shapes object of the current pass
        self.shapes_version = -1
        self.ordinal = 0
        self.by_loc = {}                  # (sampling_loc.data_ptr(), its shape) -> site, insertion-ordered
        self.last = -1                    # the site the last context carried (tests, bench)
        self.unmatched = 0                # backward calls whose forward was not found (they take the history-free kernel)

    def forward(self, spatial_shapes, sampling_loc):
        import weakref
        with self.lock:
            cur = self.shapes_ref() if self.shapes_ref is not None else None
            # (tensor_version: -1 for inference tensors -- a shapes tensor built under torch.inference_mode(), as Deformable-DETR
            # does, has no version counter and `_version` raises on it)
            version = tensor_version(spatial_shapes)
            if cur is not spatial_shapes or self.shapes_version != version:
                self.shapes_ref = weakref.ref(spatial_shapes)
                self.shapes_version = version
                self.ordinal = 0
            site = AUTO_SITE_BASE + self.ordinal % AUTO_SITES
            self.ordinal += 1
            if sampling_loc is not None:
                # (address AND shape: the caching allocator hands a freed address out again -- an inference pass's leftover entry must
                # not vouch for a later tensor of another call)
                self.by_loc[(int(sampling_loc.data_ptr()), tuple(sampling_loc.shape))] = site
                while len(self.by_loc) > _AUTO_BWD_KEEP:
                    self.by_loc.pop(next(iter(self.by_loc)))
            return site

    def backward(self, sampling_loc):
        with self.lock:
            site = self.by_loc.pop((int(sampling_loc.data_ptr()), tuple(sampling_loc.shape)), None)
            if site is None:
                self.unmatched += 1
            return site


_auto = _AutoSites()


def reset_auto_sites(forget_history=False):
    """Start a new pass for the derived call sites: the next encoder-shaped call without a call_site() block is layer 0.
    Only needed by a caller that REUSES one spatial_shapes tensor object across forward passes (the reference rebuilds it).
    forget_history=True also clears what the derived slots' earlier calls reported (msda_hip_reset_call_site): for a process
    that puts another model or checkpoint behind them."""
    with _auto.lock:
        _auto.shapes_ref, _auto.shapes_version, _auto.ordinal = None, -1, 0
        _auto.by_loc.clear()
    if forget_history:
        lib = _lib.load()
        for site in range(AUTO_SITE_BASE, AUTO_SITE_BASE + AUTO_SITES):
            lib.msda_hip_reset_call_site(site)


def _auto_site(spatial_shapes, sampling_loc, backward=False):
    """Site of an encoder-shaped call made outside of every call_site() block (see above).  A backward call whose forward is
    unknown gets -1: no history, the library's history-free kernel."""
    if backward:
        site = _auto.backward(sampling_loc)
        return -1 if site is None else site
    return _auto.forward(spatial_shapes, sampling_loc)


def unmatched_backward_calls():
    """How many encoder-shaped backward calls made outside call_site() blocks could not be matched to their forward (a cast or
    contiguous copy of sampling_loc, activation checkpointing, a forward older than the last 256): those calls are correct
    but take the history-free backward kernel.  A count that grows with the steps says the integration should pass sites."""
    return _auto.unmatched


def last_call_site():
    """The call site the last encoder-shaped call of this process carried (explicit or derived); -1 before the first."""
    return _auto.last


def _geometry_checked(spatial_shapes, level_start_index, spatial_size):
    """sum_l H_l * W_l == spatial_size and level_start_index == the prefix sums -- the precondition of the window kernels,
    which the reference operator itself does not require.  A device -> host copy, done once per shapes TENSOR OBJECT (weak
    reference + version, tests/test_cache_cpu.py) and never during a stream capture (a capture only sees earlier verdicts)."""
    extra = (int(spatial_size), id(level_start_index), int(level_start_index.data_ptr()))
    if _geometry_checks.hit(spatial_shapes, extra):
        return True
    if torch.cuda.is_current_stream_capturing():
        return False
    hw = (spatial_shapes[:, 0] * spatial_shapes[:, 1]).cpu()
    lsi = level_start_index.cpu()
    ok = bool((spatial_shapes.cpu() > 0).all()) and int(hw.sum()) == int(spatial_size) and lsi.shape[0] == hw.shape[0] \
        and bool((lsi == torch.cat((hw.new_zeros(1), hw.cumsum(0)[:-1]))).all())
    if ok:
        _geometry_checks.add(spatial_shapes, extra)
    return ok


def _set_call_context(lib, value_dtype, spatial_shapes, level_start_index, S, M_D, L, Lq, P, sampling_loc=None, backward=False):
    """Describe the coming forward or backward call to the library.  Only encoder-shaped fp32 calls have a choice to make; for
    everything else no context is set (and nothing is copied to the host)."""
    if value_dtype != torch.float32 or Lq != S or S < 1024 or L != 4 or P != 4 or M_D != 32:
        return
    flags = CTX_GEOMETRY_CHECKED if _geometry_checked(spatial_shapes, level_start_index, S) else 0
    if torch.are_deterministic_algorithms_enabled():
        flags |= CTX_DETERMINISTIC
    site = getattr(_site, "value", None)
    if site is None:
        site = _auto_site(spatial_shapes, sampling_loc, backward)
    _auto.last = int(site)
    lib.msda_hip_set_call_context(int(site), flags)


def _check(name, t, dev):
    if not t.is_contiguous():
        raise RuntimeError("%s tensor has to be contiguous" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)  # "CUDA" == the HIP device on PyTorch-ROCm
    if t.device != dev:
        raise RuntimeError("%s is on %s but value is on %s" % (name, t.device, dev))


def _dims(value, spatial_shapes, sampling_loc, im2col_step):
    if value.dim() != 4 or sampling_loc.dim() != 6 or spatial_shapes.dim() != 2:
        raise RuntimeError("expected value [N,S,M,D], spatial_shapes [L,2], sampling_loc [N,Lq,M,L,P,2]")
    batch, spatial_size, num_heads, channels = value.shape
    num_levels = spatial_shapes.shape[0]
    num_query, num_point = sampling_loc.shape[1], sampling_loc.shape[4]
    step = min(batch, int(im2col_step))
    if batch > 0 and (step <= 0 or batch % step != 0):
        raise RuntimeError("batch(%d) must divide im2col_step(%d)" % (batch, step))
    return batch, spatial_size, num_heads, channels, num_levels, num_query, num_point


def _host_args(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, extra=()):
    """Validation of an all-CPU call (same contiguity / dtype rules as the device path)."""
    if STRICT_DEVICE:
        raise RuntimeError("Not implemented on the CPU")  # ops/src/ms_deform_attn.h:38
    if value.dtype not in _SUFFIX:
        raise RuntimeError("ms_deform_attn: unsupported dtype %s (float32/float64 only)" % value.dtype)
    named = (("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
             ("sampling_loc", sampling_loc), ("attn_weight", attn_weight)) + tuple(extra)
    for name, t in named:
        if not t.is_contiguous():
            raise RuntimeError("%s tensor has to be contiguous" % name)
        if t.is_cuda:
            raise RuntimeError("%s is on %s but value is on the CPU" % (name, t.device))
    for name, t in (("sampling_loc", sampling_loc), ("attn_weight", attn_weight)) + tuple(extra):
        if t.dtype != value.dtype:
            raise RuntimeError("%s has dtype %s, value has %s" % (name, t.dtype, value.dtype))
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes and level_start_index must be int64")
    return _lib.load(), _SUFFIX[value.dtype]


def _prep(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, extra=()):
    if not value.is_cuda:
        raise RuntimeError("Not implemented on the CPU")  # ops/src/ms_deform_attn.h:38 (fused / device-only entry points)
    if value.dtype not in _SUFFIX:
        raise RuntimeError("ms_deform_attn: unsupported dtype %s (float32/float64 only)" % value.dtype)
    dev = value.device
    _check("value", value, dev)
    _check("spatial_shapes", spatial_shapes, dev)
    _check("level_start_index", level_start_index, dev)
    _check("sampling_loc", sampling_loc, dev)
    _check("attn_weight", attn_weight, dev)
    for name, t in extra:
        _check(name, t, dev)
    for name, t in (("sampling_loc", sampling_loc), ("attn_weight", attn_weight)) + tuple(extra):
        if t.dtype != value.dtype:
            raise RuntimeError("%s has dtype %s, value has %s" % (name, t.dtype, value.dtype))
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes and level_start_index must be int64")
    return _lib.load(), _SUFFIX[value.dtype]


def _raise(rc):
    raise RuntimeError("MultiScaleDeformableAttention (HIP): %s [code %d]" % (_lib.last_error(), rc))


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    if not value.is_cuda:   # host-pointer variant of the C ABI
        lib, suf = _host_args(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
        N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, sampling_loc, im2col_step)
        out = torch.empty((N, Lq, M * D), dtype=value.dtype)
        rc = getattr(lib, "msda_host_forward_" + suf)(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
            attn_weight.data_ptr(), N, S, M, D, L, Lq, P, out.data_ptr(), HOST_THREADS)
        if rc != 0:
            _raise(rc)
        return out
    lib, suf = _prep(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, sampling_loc, im2col_step)
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)  # kernel writes every element
    with torch.cuda.device(value.device):
        stream = torch.cuda.current_stream().cuda_stream
        _set_call_context(lib, value.dtype, spatial_shapes, level_start_index, S, D, L, Lq, P, sampling_loc)
        rc = getattr(lib, "msda_hip_forward_" + suf)(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
            attn_weight.data_ptr(), N, S, M, D, L, Lq, P, out.data_ptr(), ctypes.c_void_p(stream))
    if rc != 0:
        _raise(rc)
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                            im2col_step):
    if not value.is_cuda:   # host-pointer variant of the C ABI
        lib, suf = _host_args(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                              extra=(("grad_output", grad_output),))
        N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, sampling_loc, im2col_step)
        grad_value, grad_loc, grad_attn = torch.zeros_like(value), torch.empty_like(sampling_loc), torch.empty_like(attn_weight)
        rc = getattr(lib, "msda_host_backward_" + suf)(
            grad_output.data_ptr(), value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
            sampling_loc.data_ptr(), attn_weight.data_ptr(), N, S, M, D, L, Lq, P, grad_value.data_ptr(),
            grad_loc.data_ptr(), grad_attn.data_ptr(), HOST_THREADS)
        if rc != 0:
            _raise(rc)
        return [grad_value, grad_loc, grad_attn]
    lib, suf = _prep(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                     extra=(("grad_output", grad_output),))
    N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, sampling_loc, im2col_step)
    grad_value = torch.zeros_like(value)               # accumulated with atomics
    grad_loc = torch.empty_like(sampling_loc)          # every element written by the kernel
    grad_attn = torch.empty_like(attn_weight)
    with torch.cuda.device(value.device):
        stream = torch.cuda.current_stream().cuda_stream
        ws_bytes = int(lib.msda_hip_backward_workspace_bytes(N, S, M, D, L, Lq, P)) if TORCH_WORKSPACE and suf == "f32" else 0
        _set_call_context(lib, value.dtype, spatial_shapes, level_start_index, S, D, L, Lq, P, sampling_loc, backward=True)
        if ws_bytes > 0:
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=value.device)   # freed (to the cache) in stream order
            rc = lib.msda_hip_backward_ws_f32(
                grad_output.data_ptr(), value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                sampling_loc.data_ptr(), attn_weight.data_ptr(), N, S, M, D, L, Lq, P, grad_value.data_ptr(),
                grad_loc.data_ptr(), grad_attn.data_ptr(), ws.data_ptr(), ws_bytes, ctypes.c_void_p(stream))
        else:
            rc = getattr(lib, "msda_hip_backward_" + suf)(
                grad_output.data_ptr(), value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                sampling_loc.data_ptr(), attn_weight.data_ptr(), N, S, M, D, L, Lq, P, grad_value.data_ptr(),
                grad_loc.data_ptr(), grad_attn.data_ptr(), ctypes.c_void_p(stream))
    if rc != 0:
        _raise(rc)
    return [grad_value, grad_loc, grad_attn]


ERR_UNSUPPORTED = -5   # MSDA_ERR_UNSUPPORTED in include/msda_hip.h


def fused_forward_supported(value, reference_points, num_levels, num_points):
    """Geometry the fused-prologue kernel covers (include/msda_hip.h): fp32 on the GPU, 32 channels per head,
    levels * points == 16, 2-d or 4-d reference points."""
    return (value.is_cuda and value.dtype == torch.float32 and value.dim() == 4 and value.shape[3] == 32
            and num_levels * num_points == 16 and num_points % 2 == 0 and reference_points.shape[-1] in (2, 4))


def fused_forward_hm_supported(value_shape, num_levels, num_points, num_query):
    """Geometry msda_hip_forward_fused_hm_f32 covers: 32 channels per head, 4 levels x 4 points, encoder-sized calls."""
    return value_shape[-1] == 32 and num_levels == 4 and num_points == 4 and num_query >= 1024


def ms_deform_attn_forward_fused(value, spatial_shapes, level_start_index, reference_points, sampling_offsets,
                                 attention_logits, num_points, value_head_major=False):
    """MSDeformAttn.forward's softmax + sampling-location arithmetic + sampling in ONE kernel
    (ops/modules/ms_deform_attn.py:99-113).  `sampling_offsets` [N, Lq, M*L*P*2] and `attention_logits`
    [N, Lq, M*L*P] are the raw Linear outputs, `reference_points` [N, Lq, L, 2|4].  Inference only (no autograd).
    value_head_major: `value` is [N, M, S, D] (linear_packed_forward(..., head_major_rows=S)) instead of [N, S, M, D].
    Raises RuntimeError for unsupported geometry -- check fused_forward_supported() first."""
    lib, _ = _prep(value, spatial_shapes, level_start_index, sampling_offsets, attention_logits,
                   extra=(("reference_points", reference_points),))
    if value_head_major:
        N, M, S, D = value.shape
    else:
        N, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    Lq = sampling_offsets.shape[1]
    P = int(num_points)
    if sampling_offsets.shape != (N, Lq, M * L * P * 2) or attention_logits.shape != (N, Lq, M * L * P) \
            or reference_points.shape[:3] != (N, Lq, L):
        raise RuntimeError("ms_deform_attn_forward_fused: inconsistent shapes")
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    with torch.cuda.device(value.device):
        stream = torch.cuda.current_stream().cuda_stream
        _set_call_context(lib, value.dtype, spatial_shapes, level_start_index, S, D, L, Lq, P)
        fn = lib.msda_hip_forward_fused_hm_f32 if value_head_major else lib.msda_hip_forward_fused_f32
        rc = fn(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), reference_points.data_ptr(),
            int(reference_points.shape[-1]), sampling_offsets.data_ptr(), attention_logits.data_ptr(),
            N, S, M, D, L, Lq, P, out.data_ptr(), ctypes.c_void_p(stream))
    if rc != 0:
        _raise(rc)
    return out


# ---- dynamic mask head (include/dynmask_hip.h) -------------------------------------------------------------
def msda_prologue(spatial_shapes, reference_points, sampling_offsets, attention_logits, num_heads, num_points):
    """(sampling_locations [N, Lq, M, L, P, 2], attention_weights [N, Lq, M, L, P]) from the raw Linear outputs and the reference
    points -- ops/modules/ms_deform_attn.py:99-112 in one kernel (include/msda_hip.h: msda_hip_prologue_f32).  fp32, GPU."""
    lib = _lib.load()
    N, Lq = sampling_offsets.shape[:2]
    L, M, P = spatial_shapes.shape[0], int(num_heads), int(num_points)
    for name, t in (("reference_points", reference_points), ("sampling_offsets", sampling_offsets), ("attention_logits", attention_logits)):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.device == sampling_offsets.device):
            raise RuntimeError("msda_prologue: %s must be a contiguous float32 tensor on the GPU" % name)
    if sampling_offsets.numel() != N * Lq * M * L * P * 2 or attention_logits.numel() != N * Lq * M * L * P \
            or tuple(reference_points.shape[:3]) != (N, Lq, L) or reference_points.shape[-1] not in (2, 4):
        raise RuntimeError("msda_prologue: inconsistent shapes")
    loc = torch.empty((N, Lq, M, L, P, 2), dtype=torch.float32, device=sampling_offsets.device)
    attn = torch.empty((N, Lq, M, L, P), dtype=torch.float32, device=sampling_offsets.device)
    with torch.cuda.device(sampling_offsets.device):
        rc = lib.msda_hip_prologue_f32(spatial_shapes.data_ptr(), reference_points.data_ptr(), int(reference_points.shape[-1]),
                                       sampling_offsets.data_ptr(), attention_logits.data_ptr(), N, M, L, Lq, P, loc.data_ptr(),
                                       attn.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        _raise(rc)
    return loc, attn


def msda_prologue_backward(spatial_shapes, reference_points, sampling_offsets, attention_weights, grad_loc, grad_attn,
                           need_grad_reference=False, inplace=False):
    """(grad_sampling_offsets, grad_attention_logits, grad_reference_points | None) -- the backward of msda_prologue in one kernel
    (+ one small one for the reference points).  Shapes as the raw tensors: [N, Lq, M*L*P*2], [N, Lq, M*L*P], [N, Lq, L, 2|4].
    inplace: the results overwrite grad_loc / grad_attn (returned as views of the raw shapes)."""
    lib = _lib.load()
    N, Lq, M, L, P = attention_weights.shape
    for name, t in (("reference_points", reference_points), ("sampling_offsets", sampling_offsets), ("attention_weights", attention_weights),
                    ("grad_loc", grad_loc), ("grad_attn", grad_attn)):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.device == attention_weights.device):
            raise RuntimeError("msda_prologue_backward: %s must be a contiguous float32 tensor on the GPU" % name)
    dev = attention_weights.device
    g_off = grad_loc.view(N, Lq, M * L * P * 2) if inplace else torch.empty((N, Lq, M * L * P * 2), dtype=torch.float32, device=dev)
    g_logits = grad_attn.view(N, Lq, M * L * P) if inplace else torch.empty((N, Lq, M * L * P), dtype=torch.float32, device=dev)
    g_ref = torch.empty_like(reference_points) if need_grad_reference else None
    with torch.cuda.device(dev):
        rc = lib.msda_hip_prologue_backward_f32(
            spatial_shapes.data_ptr(), reference_points.data_ptr(), int(reference_points.shape[-1]), sampling_offsets.data_ptr(),
            attention_weights.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr(), N, M, L, Lq, P, g_off.data_ptr(),
            g_logits.data_ptr(), g_ref.data_ptr() if g_ref is not None else None,
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        _raise(rc)
    return g_off, g_logits, g_ref


def dynmask_supported(mask_feats):
    return mask_feats.is_cuda and mask_feats.dtype == torch.float32 and mask_feats.dim() == 4 and mask_feats.shape[1] == 8


def dynmask_forward(mask_feats, inst_xy, params, num_insts, stride, rel_coord=True):
    """Per-instance 3-layer 1x1 dynamic conv (ddetrs_dn.py:734-752 on the inputs of :765-808) -> [n_inst, H, W]."""
    lib = _lib.load()
    for name, t in (("mask_feats", mask_feats), ("inst_xy", inst_xy), ("params", params)):
        _check(name, t, mask_feats.device)
        if t.dtype != torch.float32:
            raise RuntimeError("%s must be float32" % name)
    N, C, H, W = mask_feats.shape
    counts = [int(n) for n in num_insts]
    n_all = sum(counts)
    want = (C + 2 if rel_coord else C) * 8 + 8 * 8 + 8 + 8 + 8 + 1
    if len(counts) != N or inst_xy.shape != (n_all, 2) or params.shape != (n_all, want):
        raise RuntimeError("dynmask_forward: inconsistent shapes")
    out = torch.empty((n_all, H, W), dtype=torch.float32, device=mask_feats.device)
    arr = (ctypes.c_int * max(N, 1))(*counts)
    with torch.cuda.device(mask_feats.device):
        rc = lib.dynmask_hip_forward_f32(mask_feats.data_ptr(), inst_xy.data_ptr(), params.data_ptr(), arr, N, C, H, W,
                                         int(stride), int(bool(rel_coord)), out.data_ptr(),
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        _raise(rc)
    return out


def aligned_bilinear_forward(x, factor):
    """aligned_bilinear (ddetrs_dn.py:1174-1196) of a [n, 1, h, w] or [n, h, w] fp32 GPU tensor."""
    lib = _lib.load()
    _check("tensor", x, x.device)
    if x.dtype != torch.float32:
        raise RuntimeError("tensor must be float32")
    squeeze = x.dim() == 4
    if squeeze and x.shape[1] != 1:
        raise RuntimeError("aligned_bilinear_forward: expected [n, 1, h, w]")
    n, h, w = x.shape[0], x.shape[-2], x.shape[-1]
    f = int(factor)
    out = torch.empty((n, 1, f * h, f * w) if squeeze else (n, f * h, f * w), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.aligned_bilinear_hip_f32(x.data_ptr(), n, h, w, f, out.data_ptr(),
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        _raise(rc)
    return out


def dynmask_backward(mask_feats, inst_xy, params, num_insts, stride, rel_coord, grad_logits, need_xy=True, need_feats=True,
                     need_params=True):
    """Backward of dynmask_forward (include/dynmask_hip.h: dynmask_hip_backward_f32): grad_logits [n_inst, H, W] ->
    (grad_mask_feats [N, 8, H, W], grad_params [n_inst, P], grad_inst_xy [n_inst, 2] or None).  need_feats / need_params /
    need_xy = False: that gradient is None and its kernels are not launched (round 6).  Deterministic (no float
    atomics); the activations are recomputed, the workspace (partial sums of the pixel slices) comes from PyTorch's
    caching allocator for the duration of the call."""
    lib = _lib.load()
    dev = mask_feats.device
    for name, t in (("mask_feats", mask_feats), ("inst_xy", inst_xy), ("params", params), ("grad_logits", grad_logits)):
        _check(name, t, dev)
        if t.dtype != torch.float32:
            raise RuntimeError("%s must be float32" % name)
    N, C, H, W = mask_feats.shape
    counts = [int(n) for n in num_insts]
    n_all = sum(counts)
    want = (C + 2 if rel_coord else C) * 8 + 8 * 8 + 8 + 8 + 8 + 1
    if len(counts) != N or inst_xy.shape != (n_all, 2) or params.shape != (n_all, want) or grad_logits.shape != (n_all, H, W):
        raise RuntimeError("dynmask_backward: inconsistent shapes")
    if N > _lib.DYNMASK_BWD_MAX_BATCH:
        raise RuntimeError("dynmask_backward: at most %d images per call" % _lib.DYNMASK_BWD_MAX_BATCH)
    g_feats = torch.empty_like(mask_feats) if need_feats else None
    g_params = torch.empty_like(params) if (need_params or need_xy) else None     # (the reference points' gradient comes out of the same pass)
    g_xy = torch.empty_like(inst_xy) if need_xy else None
    ws_bytes = int(lib.dynmask_hip_backward_workspace_bytes(n_all, H, W)) if g_params is not None else 0
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    arr = (ctypes.c_int * max(N, 1))(*counts)
    with torch.cuda.device(dev):
        rc = lib.dynmask_hip_backward_f32(mask_feats.data_ptr(), inst_xy.data_ptr(), params.data_ptr(), arr, N, C, H, W, int(stride),
                                          int(bool(rel_coord)), grad_logits.data_ptr(), g_feats.data_ptr() if g_feats is not None else None,
                                          g_params.data_ptr() if g_params is not None else None,
                                          g_xy.data_ptr() if g_xy is not None else None, ws.data_ptr(), ws_bytes,
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        _raise(rc)
    return g_feats, (g_params if need_params else None), g_xy


def aligned_bilinear_backward(grad_out, factor):
    """Backward of aligned_bilinear_forward: grad_out [n, 1, f h, f w] or [n, f h, f w] -> the input's gradient."""
    lib = _lib.load()
    _check("grad_out", grad_out, grad_out.device)
    if grad_out.dtype != torch.float32:
        raise RuntimeError("grad_out must be float32")
    squeeze = grad_out.dim() == 4
    if squeeze and grad_out.shape[1] != 1:
        raise RuntimeError("aligned_bilinear_backward: expected [n, 1, H, W]")
    f = int(factor)
    n, oh, ow = grad_out.shape[0], grad_out.shape[-2], grad_out.shape[-1]
    if f < 1 or oh % f or ow % f:
        raise RuntimeError("aligned_bilinear_backward: the gradient's size is not a multiple of the factor")
    h, w = oh // f, ow // f
    out = torch.empty((n, 1, h, w) if squeeze else (n, h, w), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        rc = lib.aligned_bilinear_hip_backward_f32(grad_out.data_ptr(), n, h, w, f, out.data_ptr(),
                                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        _raise(rc)
    return out


def patch_embed_supported(x, weight, stride, padding):
    """True when include/patch_embed_hip.h has a kernel for this convolution (fp32, GPU, kernel == stride, no pad)."""
    if not (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4 and weight.dim() == 4):
        return False
    k = weight.shape[2]
    pad = tuple(padding) if isinstance(padding, (tuple, list)) else (padding, padding)
    st = tuple(stride) if isinstance(stride, (tuple, list)) else (stride, stride)
    return (weight.shape[3] == k and st == (k, k) and pad == (0, 0) and k in (2, 4, 8, 16)
            and (weight.shape[1] * k * k) % 16 == 0 and x.shape[1] == weight.shape[1])


def patch_embed_forward(x, weight, bias=None, channels_last=True):
    """Non-overlapping-patch convolution (kernel == stride, no padding) on the matrix cores, exact fp32.

    x [B, C, H, W], weight [E, C, k, k] (nn.Conv2d layout), bias [E] or None ->
    [B, H // k, W // k, E] if channels_last (ViT PatchEmbed.forward, backbone/utils.py:182-186)
    else [B, E, H // k, W // k] (nn.Conv2d, ConvNeXt stem / downsample convs, backbone/convnext.py:80,87).
    Forward only (inference); under autograd use the PyTorch convolution.
    """
    lib = _lib.load()
    _check("x", x, x.device)
    _check("weight", weight, x.device)
    if bias is not None:
        _check("bias", bias, x.device)
    for name, t in (("x", x), ("weight", weight), ("bias", bias)):
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError("%s must be float32" % name)
    if x.dim() != 4 or weight.dim() != 4 or weight.shape[2] != weight.shape[3] or x.shape[1] != weight.shape[1]:
        raise RuntimeError("patch_embed_forward: expected x [B, C, H, W] and weight [E, C, k, k]")
    if bias is not None and bias.shape != (weight.shape[0],):
        raise RuntimeError("patch_embed_forward: bias must be [E]")
    B, C, H, W = x.shape
    E, k = weight.shape[0], weight.shape[2]
    shape = (B, H // k, W // k, E) if channels_last else (B, E, H // k, W // k)
    out = torch.empty(shape, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.patch_embed_hip_f32(x.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                                     B, C, H, W, E, k, int(bool(channels_last)), out.data_ptr(),
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        _raise(rc)
    return out


def conv3x3_supported(x, weight):
    """True when include/conv3x3_hip.h has a kernel: fp32 GPU tensors, weight [cout, cin, 3, 3], 9 * cin % 16 == 0."""
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4
            and weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3) and x.shape[1] == weight.shape[1]
            and (9 * weight.shape[1]) % 16 == 0)


def conv3x3_forward(x, weight, bias=None, relu=False, precision=0):
    """3x3 convolution, stride 1, zero padding 1, + bias (+ ReLU) on the matrix cores: the
    `F.relu(self.layN(x))` steps of MaskHeadSmallConv.forward (ddetrs_dn.py:991-1025).  NCHW in, NCHW out; forward
    only (inference).  precision 0: exact fp32 MFMA; 1: split-bf16 products with fp32 accumulation (~2e-5 of the
    output scale, see include/conv3x3_hip.h)."""
    lib = _lib.load()
    _check("x", x, x.device)
    _check("weight", weight, x.device)
    if bias is not None:
        _check("bias", bias, x.device)
    for name, t in (("x", x), ("weight", weight), ("bias", bias)):
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError("%s must be float32" % name)
    if x.dim() != 4 or weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3) or x.shape[1] != weight.shape[1]:
        raise RuntimeError("conv3x3_forward: expected x [B, C, H, W] and weight [E, C, 3, 3]")
    if bias is not None and bias.shape != (weight.shape[0],):
        raise RuntimeError("conv3x3_forward: bias must be [E]")
    B, C, H, W = x.shape
    E = weight.shape[0]
    out = torch.empty((B, E, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.conv3x3_hip_f32(x.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                                 B, C, H, W, E, int(bool(relu)), int(precision), out.data_ptr(),
                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        _raise(rc)
    return out


def conv3x3_pack_weight(weight, exact=False):
    """Split + re-order a [cout, cin, 3, 3] fp32 GPU weight once for conv3x3_packed_forward (cin % 16 == 0).
    exact: the fp32 re-ordering of the exact kernel (conv3x3_hip_pack_weight_exact_f32) instead of the bf16 hi / lo split.
    Returns an opaque uint8 tensor on the same device."""
    lib = _lib.load()
    if exact:
        _check("weight", weight, weight.device)
        if weight.dtype != torch.float32 or weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3):
            raise RuntimeError("conv3x3_pack_weight: expected a float32 [cout, cin, 3, 3] weight")
        cout, cin = weight.shape[:2]
        nbytes = lib.conv3x3_hip_packed_exact_weight_bytes(cout, cin)
        if nbytes == 0:
            raise RuntimeError("conv3x3_pack_weight: cin must be a multiple of 16")
        packed = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
        with torch.cuda.device(weight.device):
            rc = lib.conv3x3_hip_pack_weight_exact_f32(weight.data_ptr(), cout, cin, packed.data_ptr(),
                                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            _raise(rc)
        return packed
    _check("weight", weight, weight.device)
    if weight.dtype != torch.float32 or weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3):
        raise RuntimeError("conv3x3_pack_weight: expected a float32 [cout, cin, 3, 3] weight")
    cout, cin = weight.shape[:2]
    nbytes = lib.conv3x3_hip_packed_weight_bytes(cout, cin)
    if nbytes == 0:
        raise RuntimeError("conv3x3_pack_weight: cin must be a multiple of 16")
    packed = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    with torch.cuda.device(weight.device):
        rc = lib.conv3x3_hip_pack_weight_f32(weight.data_ptr(), cout, cin, packed.data_ptr(),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        _raise(rc)
    return packed


def conv3x3_packed_forward(x, packed, cout, bias=None, relu=False, exact=False):
    """3x3 / padding 1 convolution + bias (+ ReLU) from weights prepared by conv3x3_pack_weight: split-bf16 products
    with fp32 accumulation (~2e-5 of the output scale), or -- exact=True, weights packed with exact=True -- exact fp32 products
    on v_mfma_f32_32x32x2_f32; both halo-tiled (include/conv3x3_hip.h)."""
    lib = _lib.load()
    nbytes_fn = lib.conv3x3_hip_packed_exact_weight_bytes if exact else lib.conv3x3_hip_packed_weight_bytes
    run = lib.conv3x3_hip_packed_exact_f32 if exact else lib.conv3x3_hip_packed_f32
    _check("x", x, x.device)
    _check("packed", packed, x.device)
    if bias is not None:
        _check("bias", bias, x.device)
        if bias.dtype != torch.float32 or bias.shape != (cout,):
            raise RuntimeError("conv3x3_packed_forward: bias must be float32 [cout]")
    if x.dtype != torch.float32 or x.dim() != 4:
        raise RuntimeError("conv3x3_packed_forward: expected a float32 x [B, C, H, W]")
    B, C, H, W = x.shape
    if packed.dtype != torch.uint8 or packed.numel() != nbytes_fn(int(cout), C):
        raise RuntimeError("conv3x3_packed_forward: `packed` does not belong to a [%d, %d, 3, 3] weight" % (cout, C))
    out = torch.empty((B, int(cout), H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = run(x.data_ptr(), packed.data_ptr(), bias.data_ptr() if bias is not None else None,
                 B, C, H, W, int(cout), int(bool(relu)), out.data_ptr(),
                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        _raise(rc)
    return out


def patch_embed_packed_supported(weight):
    """True when the split-bf16 packed path of include/patch_embed_hip.h covers this [E, C, k, k] weight."""
    return (weight.dim() == 4 and weight.shape[2] == weight.shape[3] and
            _lib.load().patch_embed_hip_packed_weight_bytes(weight.shape[0], weight.shape[1], weight.shape[2]) > 0)


def patch_embed_pack_weight(weight):
    """Split + re-order an [E, C, k, k] fp32 GPU weight once for patch_embed_packed_forward.  Opaque uint8 tensor."""
    lib = _lib.load()
    _check("weight", weight, weight.device)
    if weight.dtype != torch.float32 or weight.dim() != 4 or weight.shape[2] != weight.shape[3]:
        raise RuntimeError("patch_embed_pack_weight: expected a float32 [E, C, k, k] weight")
    E, C, k = weight.shape[0], weight.shape[1], weight.shape[2]
    nbytes = lib.patch_embed_hip_packed_weight_bytes(E, C, k)
    if nbytes == 0:
        raise RuntimeError("patch_embed_pack_weight: needs k in {2, 4, 8, 16} and C * k * k a multiple of 48")
    packed = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    with torch.cuda.device(weight.device):
        rc = lib.patch_embed_hip_pack_weight_f32(weight.data_ptr(), E, C, k, packed.data_ptr(),
                                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        _raise(rc)
    return packed


def patch_embed_packed_forward(x, packed, embed_dim, patch, bias=None, channels_last=True):
    """patch_embed_forward with split-bf16 products (~2e-5 of the output scale) from weights prepared by
    patch_embed_pack_weight."""
    lib = _lib.load()
    _check("x", x, x.device)
    _check("packed", packed, x.device)
    if x.dtype != torch.float32 or x.dim() != 4:
        raise RuntimeError("patch_embed_packed_forward: expected a float32 x [B, C, H, W]")
    B, C, H, W = x.shape
    E, k = int(embed_dim), int(patch)
    if bias is not None:
        _check("bias", bias, x.device)
        if bias.dtype != torch.float32 or bias.shape != (E,):
            raise RuntimeError("patch_embed_packed_forward: bias must be float32 [E]")
    if packed.dtype != torch.uint8 or packed.numel() == 0 or packed.numel() != lib.patch_embed_hip_packed_weight_bytes(E, C, k):
        raise RuntimeError("patch_embed_packed_forward: `packed` does not belong to a [%d, %d, %d, %d] weight" % (E, C, k, k))
    shape = (B, H // k, W // k, E) if channels_last else (B, E, H // k, W // k)
    out = torch.empty(shape, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.patch_embed_hip_packed_f32(x.data_ptr(), packed.data_ptr(), bias.data_ptr() if bias is not None else None,
                                            B, C, H, W, E, k, int(bool(channels_last)), out.data_ptr(),
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        _raise(rc)
    return out


def linear_packed_supported(x, weight):
    """True when include/linear_hip.h covers `F.linear(x, weight)`: fp32 GPU tensors, in_features % 64 == 0."""
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.dim() == 2
            and x.shape[-1] == weight.shape[1] and weight.shape[1] % 64 == 0)


def linear_pack_weight(weight):
    """Split + re-order an [out_features, in_features] fp32 GPU weight once for linear_packed_forward."""
    lib = _lib.load()
    _check("weight", weight, weight.device)
    if weight.dtype != torch.float32 or weight.dim() != 2:
        raise RuntimeError("linear_pack_weight: expected a float32 [out_features, in_features] weight")
    n, k = weight.shape
    nbytes = lib.linear_hip_packed_weight_bytes(n, k)
    if nbytes == 0:
        raise RuntimeError("linear_pack_weight: in_features must be a multiple of 64")
    packed = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    with torch.cuda.device(weight.device):
        rc = lib.linear_hip_pack_weight_f32(weight.data_ptr(), n, k, packed.data_ptr(),
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        _raise(rc)
    return packed


def linear_packed_split_forward(x, packed, split_col, out_features, bias=None, x_add=None):
    """Two Linear layers on the same input as one product (include/linear_hip.h: linear_hip_packed_split_f32): `packed`
    belongs to the row-wise concatenation [W_a; W_b] ([out_features, in_features]), `bias` to the concatenated biases;
    returns (x' W_a^T + b_a  [..., split_col],  x' W_b^T + b_b  [..., out_features - split_col]) with x' = x + x_add."""
    lib = _lib.load()
    _check("x", x, x.device)
    _check("packed", packed, x.device)
    if x.dtype != torch.float32 or x.dim() < 1:
        raise RuntimeError("linear_packed_split_forward: expected a float32 x [..., in_features]")
    k, n, sc = x.shape[-1], int(out_features), int(split_col)
    rows = x.numel() // k if k else 0
    if bias is not None:
        _check("bias", bias, x.device)
        if bias.dtype != torch.float32 or bias.shape != (n,):
            raise RuntimeError("linear_packed_split_forward: bias must be float32 [out_features]")
    if x_add is not None:
        _check("x_add", x_add, x.device)
        if x_add.dtype != torch.float32 or x_add.shape != x.shape:
            raise RuntimeError("linear_packed_split_forward: x_add must be float32 with the shape of x")
    if packed.dtype != torch.uint8 or packed.numel() == 0 or packed.numel() != lib.linear_hip_packed_weight_bytes(n, k):
        raise RuntimeError("linear_packed_split_forward: `packed` does not belong to a [%d, %d] weight" % (n, k))
    out_a = torch.empty(x.shape[:-1] + (sc,), dtype=torch.float32, device=x.device)
    out_b = torch.empty(x.shape[:-1] + (n - sc,), dtype=torch.float32, device=x.device)
    if rows:
        rc = lib.linear_hip_packed_split_f32(x.data_ptr(), x_add.data_ptr() if x_add is not None else None, packed.data_ptr(),
                                             bias.data_ptr() if bias is not None else None, rows, k, n, sc,
                                             out_a.data_ptr(), out_b.data_ptr(),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError("linear_hip_packed_split_f32: " + _lib.last_error())
    return out_a, out_b


def linear_packed_forward(x, packed, out_features, bias=None, row_mask=None, head_major_rows=0, x_add=None, relu=False):
    """`F.linear(x, W, bias)` with split-bf16 products (~2e-5 of the output scale) from weights prepared by
    linear_pack_weight; rows whose `row_mask` entry is True are written as zeros (the masked_fill of
    ops/modules/ms_deform_attn.py:96-97).  x [..., in_features] contiguous -> [..., out_features].
    head_major_rows = S > 0: x is [N, S, in_features] and the result is [N, out_features // 32, S, 32] (head-major
    `value` for ms_deform_attn_forward_fused(..., value_head_major=True)).
    x_add (same shape as x): the layer's input is x + x_add, added while the operand is loaded (`with_pos_embed`);
    relu: ReLU in the epilogue.  Neither combines with head_major_rows."""
    lib = _lib.load()
    _check("x", x, x.device)
    _check("packed", packed, x.device)
    if x.dtype != torch.float32 or x.dim() < 1:
        raise RuntimeError("linear_packed_forward: expected a float32 x [..., in_features]")
    k, n = x.shape[-1], int(out_features)
    rows = x.numel() // k if k else 0
    if bias is not None:
        _check("bias", bias, x.device)
        if bias.dtype != torch.float32 or bias.shape != (n,):
            raise RuntimeError("linear_packed_forward: bias must be float32 [out_features]")
    if row_mask is not None:
        _check("row_mask", row_mask, x.device)
        if row_mask.dtype not in (torch.bool, torch.uint8) or row_mask.numel() != rows:
            raise RuntimeError("linear_packed_forward: row_mask must be bool / uint8 with one entry per row")
    if packed.dtype != torch.uint8 or packed.numel() == 0 or packed.numel() != lib.linear_hip_packed_weight_bytes(n, k):
        raise RuntimeError("linear_packed_forward: `packed` does not belong to a [%d, %d] weight" % (n, k))
    hm = int(head_major_rows)
    if x_add is not None:
        _check("x_add", x_add, x.device)
        if x_add.dtype != torch.float32 or x_add.shape != x.shape:
            raise RuntimeError("linear_packed_forward: x_add must be float32 with the shape of x")
    if hm and (x_add is not None or relu):
        raise RuntimeError("linear_packed_forward: head-major output does not combine with x_add / relu")
    if hm:
        if x.dim() != 3 or x.shape[1] != hm or n % 32:
            raise RuntimeError("linear_packed_forward: head-major output needs x [N, S, in_features] and out_features % 32 == 0")
        out = torch.empty((x.shape[0], n // 32, hm, 32), dtype=torch.float32, device=x.device)
    else:
        out = torch.empty(x.shape[:-1] + (n,), dtype=torch.float32, device=x.device)
    b_ptr = bias.data_ptr() if bias is not None else None
    m_ptr = row_mask.data_ptr() if row_mask is not None else None
    with torch.cuda.device(x.device):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if hm:
            rc = lib.linear_hip_packed_hm_f32(x.data_ptr(), packed.data_ptr(), b_ptr, m_ptr, rows, k, n, hm, out.data_ptr(), st)
        elif x_add is not None or relu:
            rc = lib.linear_hip_packed_ex_f32(x.data_ptr(), x_add.data_ptr() if x_add is not None else None, packed.data_ptr(),
                                              b_ptr, m_ptr, rows, k, n, int(bool(relu)), out.data_ptr(), st)
        else:
            rc = lib.linear_hip_packed_f32(x.data_ptr(), packed.data_ptr(), b_ptr, m_ptr, rows, k, n, out.data_ptr(), st)
    if rc != 0:
        _raise(rc)
    return out


def add_layernorm_supported(x, normalized_shape):
    return (x.is_cuda and x.dtype == torch.float32 and len(normalized_shape) == 1 and x.shape[-1] == normalized_shape[0]
            and x.shape[-1] % 4 == 0 and x.shape[-1] <= 4096)


def add_layernorm(x, residual, weight, bias, eps):
    """LayerNorm(x + residual) over the last dimension in one pass (include/layernorm_hip.h); residual may be None."""
    lib = _lib.load()
    _check("x", x, x.device)
    for name, t in (("residual", residual), ("weight", weight), ("bias", bias)):
        if t is not None:
            _check(name, t, x.device)
            if t.dtype != torch.float32:
                raise RuntimeError("%s must be float32" % name)
    if x.dtype != torch.float32:
        raise RuntimeError("x must be float32")
    d = x.shape[-1]
    if residual is not None and residual.shape != x.shape:
        raise RuntimeError("add_layernorm: residual must have the shape of x")
    for name, t in (("weight", weight), ("bias", bias)):
        if t is not None and t.shape != (d,):
            raise RuntimeError("add_layernorm: %s must be [features]" % name)
    out = torch.empty_like(x)
    rows = x.numel() // d if d else 0
    with torch.cuda.device(x.device):
        rc = lib.add_layernorm_hip_f32(x.data_ptr(), residual.data_ptr() if residual is not None else None,
                                       weight.data_ptr() if weight is not None else None,
                                       bias.data_ptr() if bias is not None else None, float(eps), rows, d, out.data_ptr(),
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        _raise(rc)
    return out


def matcher_cost(logits, boxes, tgt_boxes, positive_map, w_class, w_bbox, w_giou):
    """[num_pred, num_gt] fp32 cost matrix of HungarianMatcherVL.forward in one kernel (include/matcher_cost_hip.h): the same
    float32 operations in the same order as the PyTorch composition of matcher.py:476-498.  logits [num_pred, T], boxes
    [num_pred, 4], tgt_boxes [num_gt, 4] fp32 on one GPU; positive_map [num_gt, T] bool (or 0 / 1)."""
    lib = _lib.load()
    for name, t in (("logits", logits), ("boxes", boxes), ("tgt_boxes", tgt_boxes)):
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.device == logits.device):
            raise RuntimeError("matcher_cost: %s has to be a 2-d float32 tensor on the GPU of the logits" % name)
    num_pred, T = logits.shape
    G = tgt_boxes.shape[0]
    if boxes.shape != (num_pred, 4) or tgt_boxes.shape[1] != 4 or tuple(positive_map.shape) != (G, T):
        raise RuntimeError("matcher_cost: shapes do not fit together")
    logits, boxes, tgt_boxes = logits.contiguous(), boxes.contiguous(), tgt_boxes.contiguous()
    pm = positive_map.to(device=logits.device) != 0
    # CSR of the positive map, built on the device (nonzero() of a row-major mask lists the tokens of a target in order)
    tok_idx = pm.nonzero()[:, 1].to(torch.int32).contiguous()
    tok_off = torch.zeros(G + 1, dtype=torch.int32, device=logits.device)
    tok_off[1:] = pm.sum(1).cumsum(0).to(torch.int32)
    cost = torch.empty((num_pred, G), dtype=torch.float32, device=logits.device)
    with torch.cuda.device(logits.device):
        rc = lib.matcher_cost_hip_f32(logits.data_ptr(), boxes.data_ptr(), tgt_boxes.data_ptr(), tok_off.data_ptr(),
                                      tok_idx.data_ptr() if tok_idx.numel() else None, num_pred, T, G, float(w_class),
                                      float(w_bbox), float(w_giou), cost.data_ptr(),
                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        _raise(rc)
    return cost


def ota_assign(class_table, boxes, tgt_boxes, positive_map, sizes, max_rounds=10000):
    """simOTA assignment of a whole batch on the device (include/ota_hip.h): two kernels, nothing returns to the host.
    class_table [bs, Q, T] fp32 (the focal table pos - neg of matcher.py:327-330), boxes [bs, Q, 4] cxcywh, tgt_boxes
    [G_total, 4], positive_map [G_total, T] bool / uint8 -- targets of all images concatenated, `sizes` their counts per image
    (Python ints).  Returns device tensors (sel_query [bs, Q] int64, sel_gt [bs, Q] int64, matched_query [G_total] int64,
    num_selected [bs] int32, status [bs] int32): image b's assignment is sel_query[b, :num_selected[b]] (ascending) with the
    target sel_gt[b, :num_selected[b]] each, and matched_query[off_b : off_b + G_b] the cheapest own query of every target."""
    lib = _lib.load()
    dev = class_table.device
    bs, Q, T = class_table.shape
    if bs > _lib.OTA_MAX_BATCH:
        raise RuntimeError("ota_assign: at most %d images per call" % _lib.OTA_MAX_BATCH)
    for name, t, shape in (("class_table", class_table, (bs, Q, T)), ("boxes", boxes, (bs, Q, 4)), ("tgt_boxes", tgt_boxes, (sum(sizes), 4))):
        if not (t.is_cuda and t.device == dev and t.dtype == torch.float32 and tuple(t.shape) == shape):
            raise RuntimeError("ota_assign: %s has to be a float32 tensor of shape %s on the GPU of the table" % (name, (shape,)))
    if tuple(positive_map.shape) != (sum(sizes), T) or positive_map.dtype not in (torch.bool, torch.uint8) or positive_map.device != dev:
        raise RuntimeError("ota_assign: positive_map has to be [G_total, T] bool / uint8 on the same GPU")
    class_table, boxes, tgt_boxes = class_table.contiguous(), boxes.contiguous(), tgt_boxes.contiguous()
    pm = positive_map.contiguous().view(torch.uint8)
    off = [0]
    for n in sizes:
        off.append(off[-1] + int(n))
    gt_off = (ctypes.c_int32 * (bs + 1))(*off)
    G = off[-1]
    cost = torch.empty(Q * G, dtype=torch.float32, device=dev)
    iou = torch.empty(Q * G, dtype=torch.float32, device=dev)
    flags = torch.empty(Q * G, dtype=torch.uint8, device=dev)
    matching = torch.empty(Q * G, dtype=torch.uint8, device=dev)
    sel_q = torch.empty((bs, Q), dtype=torch.int64, device=dev)
    sel_g = torch.empty((bs, Q), dtype=torch.int64, device=dev)
    matched = torch.empty(G, dtype=torch.int64, device=dev)
    count = torch.empty(bs, dtype=torch.int32, device=dev)
    status = torch.empty(bs, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = lib.ota_cost_hip_f32(class_table.data_ptr(), boxes.data_ptr(), tgt_boxes.data_ptr() if G else None,
                                  pm.data_ptr() if G else None, gt_off, bs, Q, T, cost.data_ptr() if G else None,
                                  iou.data_ptr() if G else None, flags.data_ptr() if G else None, stream)
        if rc != 0:
            _raise(rc)
        rc = lib.ota_dynamic_k_hip(cost.data_ptr() if G else None, iou.data_ptr() if G else None, flags.data_ptr() if G else None,
                                   matching.data_ptr() if G else None, gt_off, bs, Q, int(max_rounds), sel_q.data_ptr(),
                                   sel_g.data_ptr(), matched.data_ptr() if G else None, count.data_ptr(), status.data_ptr(), stream)
        if rc != 0:
            _raise(rc)
    return sel_q, sel_g, matched, count, status


def lsap_batch(costs, check=True):
    """Linear sum assignment of each 2-d fp32 GPU cost matrix in `costs` (row-major views with any row stride, e.g.
    column slices of one big matrix) on the device, with SciPy's result index for index (include/lsap_hip.h).
    Returns a list of (row_ind, col_ind) int64 GPU tensors.  check=True synchronises once at the end and raises
    ValueError where SciPy would (NaN / -inf entries, infeasible matrix); check=False never leaves the stream."""
    lib = _lib.load()
    if not costs:
        return []
    dev = costs[0].device
    for c in costs:
        if not (c.is_cuda and c.device == dev and c.dtype == torch.float32 and c.dim() == 2):
            raise RuntimeError("lsap_batch: expected 2-d float32 tensors on one GPU")
        if c.numel() and (c.stride(1) != 1 or c.stride(0) < c.shape[1]):
            raise RuntimeError("lsap_batch: rows must be contiguous (row stride >= cols, column stride 1)")
    results, keep = [], []
    status = torch.full((len(costs),), -1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for lo in range(0, len(costs), _lib.LSAP_MAX_BATCH):
            chunk = costs[lo:lo + _lib.LSAP_MAX_BATCH]
            n = len(chunk)
            P, LL, I = ctypes.c_void_p * n, ctypes.c_longlong * n, ctypes.c_int * n
            rows = [c.shape[0] for c in chunk]
            cols = [c.shape[1] for c in chunk]
            outs = [(torch.empty(min(r, k), dtype=torch.int64, device=dev), torch.empty(min(r, k), dtype=torch.int64, device=dev))
                    for r, k in zip(rows, cols)]
            ws = [torch.empty(lib.lsap_hip_workspace_bytes(r, k), dtype=torch.uint8, device=dev) for r, k in zip(rows, cols)]
            keep.append(ws)
            rc = lib.lsap_hip_batch_f32(
                n, P(*[c.data_ptr() if c.numel() else None for c in chunk]),
                LL(*[c.stride(0) if c.numel() else max(k, 1) for c, k in zip(chunk, cols)]), I(*rows), I(*cols),
                P(*[o[0].data_ptr() if o[0].numel() else None for o in outs]),
                P(*[o[1].data_ptr() if o[1].numel() else None for o in outs]), P(*[w.data_ptr() for w in ws]),
                P(*[status[lo + k:].data_ptr() for k in range(n)]), stream)
            if rc != 0:
                _raise(rc)
            results += outs
    if check:
        st = status.cpu().tolist()      # also keeps the workspaces alive until the kernels are done
        if any(s == 1 for s in st):
            raise ValueError("matrix contains invalid numeric entries")
        if any(s == 2 for s in st):
            raise ValueError("cost matrix is infeasible")
    else:
        for ws in keep:                 # the caching allocator may hand the workspace out again: tie it to the stream
            for w in ws:
                w.record_stream(torch.cuda.current_stream())
    return results


def lsap(cost, check=True):
    """scipy.optimize.linear_sum_assignment(cost) for one 2-d fp32 GPU matrix, on the device."""
    return lsap_batch([cost], check)[0]


def upsample_add(skip, low):
    """`skip + F.interpolate(low, size=skip.shape[-2:], mode="nearest")` in one pass (include/conv3x3_hip.h)."""
    lib = _lib.load()
    _check("skip", skip, skip.device)
    _check("low", low, skip.device)
    if skip.dtype != torch.float32 or low.dtype != torch.float32 or skip.dim() != 4 or low.dim() != 4 \
            or skip.shape[:2] != low.shape[:2]:
        raise RuntimeError("upsample_add: expected float32 skip [B, C, H, W] and low [B, C, h, w]")
    B, C, H, W = skip.shape
    out = torch.empty_like(skip)
    with torch.cuda.device(skip.device):
        rc = lib.upsample_add_hip_f32(skip.data_ptr(), low.data_ptr(), B, C, H, W, low.shape[2], low.shape[3],
                                      out.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        _raise(rc)
    return out


def linear_packed_ln_supported(x, weight, norm_shape):
    return linear_packed_supported(x, weight) and weight.shape[0] == 256 and tuple(norm_shape) == (256,)


def linear_packed_ln(x, packed, bias, residual, ln_weight, ln_bias, eps):
    """LayerNorm(residual + F.linear(x, W, bias)) * ln_weight + ln_bias with the add and the normalisation in the
    Linear's epilogue (include/linear_hip.h; out_features == 256).  x [..., in_features] -> [..., 256]."""
    lib = _lib.load()
    _check("x", x, x.device)
    _check("packed", packed, x.device)
    k, n = x.shape[-1], 256
    rows = x.numel() // k if k else 0
    for name, t, shape in (("bias", bias, (n,)), ("ln_weight", ln_weight, (n,)), ("ln_bias", ln_bias, (n,)),
                           ("residual", residual, x.shape[:-1] + (n,))):
        if t is not None:
            _check(name, t, x.device)
            if t.dtype != torch.float32 or tuple(t.shape) != tuple(shape):
                raise RuntimeError("linear_packed_ln: %s must be float32 %s" % (name, tuple(shape)))
    if x.dtype != torch.float32 or packed.dtype != torch.uint8 or packed.numel() != lib.linear_hip_packed_weight_bytes(n, k) \
            or packed.numel() == 0:
        raise RuntimeError("linear_packed_ln: expected float32 x and the packed copy of a [256, %d] weight" % k)
    out = torch.empty(x.shape[:-1] + (n,), dtype=torch.float32, device=x.device)
    ptr = lambda t: t.data_ptr() if t is not None else None
    with torch.cuda.device(x.device):
        rc = lib.linear_hip_packed_ln_f32(x.data_ptr(), packed.data_ptr(), ptr(bias), ptr(residual), ptr(ln_weight),
                                          ptr(ln_bias), float(eps), rows, k, n, out.data_ptr(),
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        _raise(rc)
    return out


def ffn_packed_supported(x, weight1, weight2, norm_shape=None):
    """True when linear_hip_packed_ffn_f32 covers `linear2(relu(linear1(x)))` (+ residual, + LayerNorm): fp32 GPU
    tensors, d_model == 256, d_ffn a multiple of 128."""
    return (x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == 256
            and weight1.dtype == torch.float32 and weight2.dtype == torch.float32
            and weight1.dim() == 2 and weight2.dim() == 2 and weight1.shape[1] == 256 and weight2.shape[0] == 256
            and weight1.shape[0] == weight2.shape[1] and weight1.shape[0] % 128 == 0
            and (norm_shape is None or tuple(norm_shape) == (256,)))


def ffn_packed(x, packed1, bias1, packed2, bias2, d_ffn, residual=None, ln_weight=None, ln_bias=None, eps=1e-5,
               layer_norm=True):
    """[LayerNorm](residual + F.linear(relu(F.linear(x, W1, bias1)), W2, bias2)) in one kernel: the hidden activations
    stay on the CU (include/linear_hip.h).  packed1 / packed2 = linear_pack_weight(W1 [d_ffn, 256]) / (W2 [256, d_ffn]).
    Bitwise equal to linear_packed_forward(..., relu=True) followed by linear_packed_ln / linear_packed_forward."""
    lib = _lib.load()
    _check("x", x, x.device)
    _check("packed1", packed1, x.device)
    _check("packed2", packed2, x.device)
    k, n = x.shape[-1], 256
    rows = x.numel() // k if k else 0
    for name, t, shape in (("bias1", bias1, (d_ffn,)), ("bias2", bias2, (n,)), ("ln_weight", ln_weight, (n,)),
                           ("ln_bias", ln_bias, (n,)), ("residual", residual, tuple(x.shape))):
        if t is not None:
            _check(name, t, x.device)
            if t.dtype != torch.float32 or tuple(t.shape) != tuple(shape):
                raise RuntimeError("ffn_packed: %s must be float32 %s" % (name, tuple(shape)))
    if (x.dtype != torch.float32 or k != n or d_ffn <= 0 or d_ffn % 128 != 0 or packed1.dtype != torch.uint8
            or packed2.dtype != torch.uint8 or packed1.numel() != lib.linear_hip_packed_weight_bytes(d_ffn, k)
            or packed2.numel() != lib.linear_hip_packed_weight_bytes(n, d_ffn)):
        raise RuntimeError("ffn_packed: expected float32 x [..., 256] and the packed copies of [%d, 256] and [256, %d] "
                           "weights (d_ffn a multiple of 128)" % (d_ffn, d_ffn))
    out = torch.empty_like(x)
    ptr = lambda t: t.data_ptr() if t is not None else None
    with torch.cuda.device(x.device):
        rc = lib.linear_hip_packed_ffn_f32(x.data_ptr(), packed1.data_ptr(), ptr(bias1), packed2.data_ptr(), ptr(bias2),
                                           ptr(residual), ptr(ln_weight), ptr(ln_bias), float(eps), 1 if layer_norm else 0,
                                           rows, k, d_ffn, out.data_ptr(),
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        _raise(rc)
    return out
