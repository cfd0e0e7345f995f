"""ctypes binding of include/msda_hip.h.  Loading never falls back to anything: if the HIP
library is absent the import of the op fails loudly (there is no CPU or eager path)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MSDA_HIP_LIB points the binding at another build of the SAME library (A/B builds of experimental kernels)
LIB_PATH = os.environ.get("MSDA_HIP_LIB") or os.path.join(_HERE, "lib", "libmsda_hip.so")

ABI_VERSION = 2   # 2 (round 6): dynmask backward exports, msda_hip_reset_call_site, OTA flags as a bit field (bit 1 = degenerate box) and status bit 2, msda_bwd_regions accumulates -- all of round 5, which had left the number at 1
EXPORTS = (
    "msda_hip_abi_version", "msda_hip_last_error",
    "msda_hip_forward_f32", "msda_hip_forward_f64", "msda_hip_backward_f32", "msda_hip_backward_f64",
    "msda_hip_forward_fused_f32", "msda_hip_forward_fused_hm_f32",
    "msda_host_forward_f32", "msda_host_forward_f64", "msda_host_backward_f32", "msda_host_backward_f64",
    "msda_hip_set_variant", "msda_hip_get_variant", "msda_hip_variant_name", "msda_hip_last_kernel",
    "msda_hip_forward_locality", "msda_hip_set_call_context", "msda_hip_reset_call_site",
    "msda_hip_backward_workspace_bytes", "msda_hip_backward_ws_f32", "msda_host_last_num_threads",
    "msda_hip_prologue_f32", "msda_hip_prologue_backward_f32",
)

DYNMASK_EXPORTS = ("dynmask_hip_forward_f32", "aligned_bilinear_hip_f32", "dynmask_hip_set_variant",
                   "dynmask_hip_last_kernel", "dynmask_hip_backward_workspace_bytes", "dynmask_hip_backward_parts",
                   "dynmask_hip_backward_f32", "aligned_bilinear_hip_backward_f32")   # include/dynmask_hip.h
DYNMASK_BWD_MAX_BATCH = 64
PATCH_EMBED_EXPORTS = ("patch_embed_hip_f32", "patch_embed_hip_packed_weight_bytes", "patch_embed_hip_pack_weight_f32",
                       "patch_embed_hip_packed_f32")                           # include/patch_embed_hip.h
LINEAR_EXPORTS = ("linear_hip_packed_weight_bytes", "linear_hip_pack_weight_f32", "linear_hip_packed_f32",
                  "linear_hip_packed_hm_f32", "linear_hip_packed_ex_f32", "linear_hip_packed_split_f32", "linear_hip_packed_ln_f32",
                  "linear_hip_packed_ffn_f32")   # include/linear_hip.h
LAYERNORM_EXPORTS = ("add_layernorm_hip_f32",)                                 # include/layernorm_hip.h
LSAP_EXPORTS = ("lsap_hip_workspace_bytes", "lsap_hip_f32", "lsap_hip_batch_f32")   # include/lsap_hip.h
MATCHER_COST_EXPORTS = ("matcher_cost_hip_f32",)                                # include/matcher_cost_hip.h
OTA_EXPORTS = ("ota_cost_hip_f32", "ota_dynamic_k_hip")                         # include/ota_hip.h
OTA_MAX_BATCH = 64
LSAP_MAX_BATCH = 32
CONV3X3_EXPORTS = ("conv3x3_hip_f32", "conv3x3_hip_packed_weight_bytes", "conv3x3_hip_pack_weight_f32",
                   "conv3x3_hip_packed_f32", "conv3x3_hip_packed_exact_weight_bytes", "conv3x3_hip_pack_weight_exact_f32",
                   "conv3x3_hip_packed_exact_f32", "upsample_add_hip_f32")     # include/conv3x3_hip.h

_lib = None


def load():
    """Return the loaded library (cached).  Raises RuntimeError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "MultiScaleDeformableAttention: %s not found. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C uninext_amd/csrc` "
            "(hipcc, gfx950). There is no fallback implementation." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    i, p, s = ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p
    lib.msda_hip_abi_version.argtypes, lib.msda_hip_abi_version.restype = [], i
    lib.msda_hip_last_error.argtypes, lib.msda_hip_last_error.restype = [], s
    for suf in ("f32", "f64"):
        f = getattr(lib, "msda_hip_forward_" + suf)
        f.argtypes, f.restype = [p, p, p, p, p, i, i, i, i, i, i, i, p, p], i
        g = getattr(lib, "msda_hip_backward_" + suf)
        g.argtypes, g.restype = [p, p, p, p, p, p, i, i, i, i, i, i, i, p, p, p, p], i
        fh = getattr(lib, "msda_host_forward_" + suf)     # host pointers; last argument: number of threads
        fh.argtypes, fh.restype = [p, p, p, p, p, i, i, i, i, i, i, i, p, i], i
        gh = getattr(lib, "msda_host_backward_" + suf)
        gh.argtypes, gh.restype = [p, p, p, p, p, p, i, i, i, i, i, i, i, p, p, p, i], i
    lib.msda_host_last_num_threads.argtypes, lib.msda_host_last_num_threads.restype = [], i
    lib.msda_hip_prologue_f32.argtypes, lib.msda_hip_prologue_f32.restype = [p, p, i, p, p, i, i, i, i, i, p, p, p], i
    lib.msda_hip_prologue_backward_f32.argtypes = [p, p, i, p, p, p, p, i, i, i, i, i, p, p, p, p]
    lib.msda_hip_prologue_backward_f32.restype = i
    lib.msda_hip_backward_workspace_bytes.argtypes, lib.msda_hip_backward_workspace_bytes.restype = [i] * 7, ctypes.c_size_t
    lib.msda_hip_backward_ws_f32.argtypes = [p, p, p, p, p, p, i, i, i, i, i, i, i, p, p, p, p, ctypes.c_size_t, p]
    lib.msda_hip_backward_ws_f32.restype = i
    lib.msda_hip_forward_fused_f32.argtypes = [p, p, p, p, i, p, p, i, i, i, i, i, i, i, p, p]
    lib.msda_hip_forward_fused_f32.restype = i
    lib.msda_hip_forward_fused_hm_f32.argtypes = lib.msda_hip_forward_fused_f32.argtypes
    lib.msda_hip_forward_fused_hm_f32.restype = i
    lib.linear_hip_packed_hm_f32.argtypes = [p, p, p, p, ctypes.c_longlong, i, i, i, p, p]
    lib.linear_hip_packed_hm_f32.restype = i
    lib.linear_hip_packed_ex_f32.argtypes = [p, p, p, p, p, ctypes.c_longlong, i, i, i, p, p]
    lib.linear_hip_packed_ex_f32.restype = i
    lib.linear_hip_packed_split_f32.argtypes = [p, p, p, p, ctypes.c_longlong, i, i, i, p, p, p]
    lib.linear_hip_packed_split_f32.restype = i
    lib.linear_hip_packed_ln_f32.argtypes = [p, p, p, p, p, p, ctypes.c_float, ctypes.c_longlong, i, i, p, p]
    lib.linear_hip_packed_ln_f32.restype = i
    lib.linear_hip_packed_ffn_f32.argtypes = [p, p, p, p, p, p, p, p, ctypes.c_float, i, ctypes.c_longlong, i, i, p, p]
    lib.linear_hip_packed_ffn_f32.restype = i
    lib.add_layernorm_hip_f32.argtypes = [p, p, p, p, ctypes.c_float, ctypes.c_longlong, i, p, p]
    lib.add_layernorm_hip_f32.restype = i
    lib.lsap_hip_workspace_bytes.argtypes, lib.lsap_hip_workspace_bytes.restype = [i, i], ctypes.c_size_t
    lib.lsap_hip_f32.argtypes, lib.lsap_hip_f32.restype = [p, ctypes.c_longlong, i, i, p, p, p, p, p], i
    lib.lsap_hip_batch_f32.argtypes, lib.lsap_hip_batch_f32.restype = [i, p, p, p, p, p, p, p, p, p], i
    f = ctypes.c_float
    lib.matcher_cost_hip_f32.argtypes, lib.matcher_cost_hip_f32.restype = [p, p, p, p, p, i, i, i, f, f, f, p, p], i
    lib.ota_cost_hip_f32.argtypes, lib.ota_cost_hip_f32.restype = [p, p, p, p, p, i, i, i, p, p, p, p], i
    lib.ota_dynamic_k_hip.argtypes, lib.ota_dynamic_k_hip.restype = [p, p, p, p, p, i, i, i, p, p, p, p, p, p], i
    lib.dynmask_hip_forward_f32.argtypes = [p, p, p, p, i, i, i, i, i, i, p, p]
    lib.dynmask_hip_forward_f32.restype = i
    lib.aligned_bilinear_hip_f32.argtypes, lib.aligned_bilinear_hip_f32.restype = [p, i, i, i, i, p, p], i
    lib.dynmask_hip_set_variant.argtypes, lib.dynmask_hip_set_variant.restype = [i], i
    lib.dynmask_hip_backward_workspace_bytes.argtypes, lib.dynmask_hip_backward_workspace_bytes.restype = [i, i, i], ctypes.c_size_t
    lib.dynmask_hip_backward_parts.argtypes, lib.dynmask_hip_backward_parts.restype = [i, i, i], i
    lib.dynmask_hip_backward_f32.argtypes = [p, p, p, p, i, i, i, i, i, i, p, p, p, p, p, ctypes.c_size_t, p]
    lib.dynmask_hip_backward_f32.restype = i
    lib.aligned_bilinear_hip_backward_f32.argtypes, lib.aligned_bilinear_hip_backward_f32.restype = [p, i, i, i, i, p, p], i
    lib.dynmask_hip_last_kernel.argtypes, lib.dynmask_hip_last_kernel.restype = [], s
    lib.patch_embed_hip_f32.argtypes, lib.patch_embed_hip_f32.restype = [p, p, p, i, i, i, i, i, i, i, p, p], i
    lib.patch_embed_hip_packed_weight_bytes.argtypes = [i, i, i]
    lib.patch_embed_hip_packed_weight_bytes.restype = ctypes.c_size_t
    lib.patch_embed_hip_pack_weight_f32.argtypes, lib.patch_embed_hip_pack_weight_f32.restype = [p, i, i, i, p, p], i
    lib.patch_embed_hip_packed_f32.argtypes, lib.patch_embed_hip_packed_f32.restype = [p, p, p, i, i, i, i, i, i, i, p, p], i
    lib.linear_hip_packed_weight_bytes.argtypes, lib.linear_hip_packed_weight_bytes.restype = [i, i], ctypes.c_size_t
    lib.linear_hip_pack_weight_f32.argtypes, lib.linear_hip_pack_weight_f32.restype = [p, i, i, p, p], i
    lib.linear_hip_packed_f32.argtypes = [p, p, p, p, ctypes.c_longlong, i, i, p, p]
    lib.linear_hip_packed_f32.restype = i
    lib.conv3x3_hip_f32.argtypes, lib.conv3x3_hip_f32.restype = [p, p, p, i, i, i, i, i, i, i, p, p], i
    lib.conv3x3_hip_packed_weight_bytes.argtypes, lib.conv3x3_hip_packed_weight_bytes.restype = [i, i], ctypes.c_size_t
    lib.conv3x3_hip_pack_weight_f32.argtypes, lib.conv3x3_hip_pack_weight_f32.restype = [p, i, i, p, p], i
    lib.conv3x3_hip_packed_f32.argtypes, lib.conv3x3_hip_packed_f32.restype = [p, p, p, i, i, i, i, i, i, p, p], i
    lib.conv3x3_hip_packed_exact_weight_bytes.argtypes, lib.conv3x3_hip_packed_exact_weight_bytes.restype = [i, i], ctypes.c_size_t
    lib.conv3x3_hip_pack_weight_exact_f32.argtypes, lib.conv3x3_hip_pack_weight_exact_f32.restype = [p, i, i, p, p], i
    lib.conv3x3_hip_packed_exact_f32.argtypes, lib.conv3x3_hip_packed_exact_f32.restype = [p, p, p, i, i, i, i, i, i, p, p], i
    lib.upsample_add_hip_f32.argtypes, lib.upsample_add_hip_f32.restype = [p, p, i, i, i, i, i, i, p, p], i
    lib.msda_hip_set_variant.argtypes, lib.msda_hip_set_variant.restype = [i, i], i
    lib.msda_hip_get_variant.argtypes, lib.msda_hip_get_variant.restype = [i], i
    lib.msda_hip_variant_name.argtypes, lib.msda_hip_variant_name.restype = [i, i], s
    lib.msda_hip_last_kernel.argtypes, lib.msda_hip_last_kernel.restype = [i], s
    lib.msda_hip_forward_locality.argtypes, lib.msda_hip_forward_locality.restype = [ctypes.POINTER(ctypes.c_double)], i
    lib.msda_hip_reset_call_site.argtypes, lib.msda_hip_reset_call_site.restype = [i], None
    lib.msda_hip_set_call_context.argtypes, lib.msda_hip_set_call_context.restype = [i, ctypes.c_uint], None
    got = lib.msda_hip_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError("libmsda_hip.so ABI version %d, binding expects %d: rebuild" % (got, ABI_VERSION))
    _lib = lib
    return lib


def last_error():
    return load().msda_hip_last_error().decode()


def set_variant(which, variant):
    """which: 'forward' | 'backward'; variant: int or name (see variants())."""
    w = {"forward": 0, "backward": 1}[which]
    if isinstance(variant, str):
        variant = variants(which).index(variant)
    if load().msda_hip_set_variant(w, int(variant)) != 0:
        raise ValueError(last_error())


def variants(which):
    w = {"forward": 0, "backward": 1}[which]
    out, k = [], 0
    while True:
        n = load().msda_hip_variant_name(w, k)
        if n is None:
            return out
        out.append(n.decode())
        k += 1


def forward_locality():
    """(reports consumed so far, far fraction of the latest) on the call site used last (include/msda_hip.h); waits for the
    launches made so far on that site."""
    frac = ctypes.c_double(0.0)
    n = load().msda_hip_forward_locality(ctypes.byref(frac))
    return n, frac.value


def last_kernel(which):
    return load().msda_hip_last_kernel({"forward": 0, "backward": 1}[which]).decode()
