// extern "C" surface of libmsda_hip.so -- see include/msda_hip.h for the contract.
// Mirrors the argument checks of the reference host launcher
// (ops/src/cuda/ms_deform_attn_cuda.cu:28-52) that are expressible on raw pointers.
#include "../../include/msda_hip.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "msda_common.hpp"

namespace {

thread_local char g_err[512] = "";
// process-wide (autograd runs backward on its own thread); last writer wins
std::atomic<const char*> g_last_kernel[2] = {{""}, {""}};
std::atomic<int> g_variant[2] = {{-1}, {-1}};  // -1: not initialised (read env once)

// Variant numbers are stable across builds (tools and profiles quote them).  The default library carries the kernels that
// won their A/B; the others (uninext_amd/csrc/experiments/) are compiled in by `make experiments` (-DMSDA_EXPERIMENTS) only --
// in the default build their numbers keep a name ("exp:...") for listings, and selecting one is MSDA_ERR_BAD_VARIANT.
#ifdef MSDA_EXPERIMENTS
#define MSDA_EXP(name) name
constexpr bool kHaveExperiments = true;
#else
#define MSDA_EXP(name) "exp:" name
constexpr bool kHaveExperiments = false;
#endif
constexpr int kNumFwdVariants = msda::kNumVariants, kNumBwdVariants = 9;
const char* const kFwdNames[kNumFwdVariants] = {
    "auto", "msda_fwd_generic", "msda_fwd_lanegroup", MSDA_EXP("msda_fwd_tiled"), MSDA_EXP("msda_fwd_tiled_l0"),
    MSDA_EXP("msda_fwd_tiled_l0big"), MSDA_EXP("msda_fwd_lgcl"), "msda_fwd_lg3", MSDA_EXP("msda_fwd_lgp"), "msda_fwd_win",
    MSDA_EXP("msda_fwd_win2"), MSDA_EXP("msda_fwd_win3"), MSDA_EXP("msda_fwd_win4"), MSDA_EXP("msda_fwd_winl"), MSDA_EXP("msda_fwd_winp")};
const char* const kBwdNames[kNumBwdVariants] = {"auto", "msda_bwd_generic", "msda_bwd_lanegroup", "msda_bwd_tiled",
                                                "msda_bwd_win", "msda_bwd_dec", "msda_bwd_regions", MSDA_EXP("msda_bwd_win2"), "msda_bwd_dst"};
#undef MSDA_EXP

int num_variants(int which) { return which == 0 ? kNumFwdVariants : kNumBwdVariants; }
const char* variant_name(int which, int v) { return which == 0 ? kFwdNames[v] : kBwdNames[v]; }
bool variant_available(int which, int v) {
  if (v < 0 || v >= num_variants(which)) return false;
  return kHaveExperiments || std::strncmp(variant_name(which, v), "exp:", 4) != 0;
}

// the thread's call context (msda_hip_set_call_context) describes ONE call.  Whatever path that call takes --
// an argument error, an empty batch, the fp64 kernels, a pinned variant -- it is gone when the entry point returns, so a
// stale GEOMETRY_CHECKED can never vouch for the next, unrelated call of the thread.
struct ContextScope {
  ~ContextScope() { msda::drop_call_context(); msda::set_call_workspace(nullptr, 0); }
};

int current_variant(int which) {
  int v = g_variant[which].load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = std::getenv(which == 0 ? "MSDA_HIP_FWD_VARIANT" : "MSDA_HIP_BWD_VARIANT");
    v = e ? std::atoi(e) : 0;
    if (!variant_available(which, v)) v = 0;
    g_variant[which].store(v, std::memory_order_relaxed);
  }
  return v;
}

int fail(int code, const char* what) {
  std::snprintf(g_err, sizeof(g_err), "%s", what);
  return code;
}

int check_dims(const msda::Dims& d) {
  if (d.N < 0 || d.Lq < 0) return fail(MSDA_ERR_BAD_DIMS, "batch and num_query must be >= 0");
  if (d.S <= 0 || d.M <= 0 || d.D <= 0 || d.L <= 0 || d.P <= 0)
    return fail(MSDA_ERR_BAD_DIMS, "spatial_size, num_heads, channels, num_levels, num_point must be > 0");
  if ((int64_t)d.Lq * d.M * d.L * d.P * 2 >= (int64_t)1 << 40 || (int64_t)d.S * d.M * d.D >= (int64_t)1 << 40)
    return fail(MSDA_ERR_TOO_LARGE, "per-image extent too large");
  return 0;
}

int finish(int rc, const char* what) {
  if (rc == 0) return 0;
  std::snprintf(g_err, sizeof(g_err), "%s: launch failed: %s (hipError_t %d)", what,
                hipGetErrorString((hipError_t)rc), rc);
  return rc;
}

template <typename T>
int forward_impl(const T* value, const int64_t* shapes, const int64_t* lsi, const T* loc, const T* attn,
                 const msda::Dims& d, T* out, void* stream) {
  const ContextScope scope;
  if (int rc = check_dims(d)) return rc;
  if (d.N == 0 || d.Lq == 0) return 0;
  if (!value || !shapes || !lsi || !loc || !attn || !out) return fail(MSDA_ERR_NULL_POINTER, "null pointer argument");
  const char* name = "";
  const int rc = msda::launch_forward<T>(current_variant(0), value, shapes, lsi, loc, attn, d, out,
                                         (hipStream_t)stream, &name);
  g_last_kernel[0].store(name, std::memory_order_relaxed);
  return finish(rc, "msda_hip_forward");
}

template <typename T>
int backward_impl(const T* grad_out, const T* value, const int64_t* shapes, const int64_t* lsi, const T* loc,
                  const T* attn, const msda::Dims& d, T* grad_value, T* grad_loc, T* grad_attn, void* stream) {
  const ContextScope scope;
  if (int rc = check_dims(d)) return rc;
  if (d.N == 0 || d.Lq == 0) return 0;
  if (!grad_out || !value || !shapes || !lsi || !loc || !attn || !grad_value || !grad_loc || !grad_attn)
    return fail(MSDA_ERR_NULL_POINTER, "null pointer argument");
  const char* name = "";
  const int rc = msda::launch_backward<T>(current_variant(1), grad_out, value, shapes, lsi, loc, attn, d, grad_value,
                                          grad_loc, grad_attn, (hipStream_t)stream, &name);
  g_last_kernel[1].store(name, std::memory_order_relaxed);
  return finish(rc, "msda_hip_backward");
}

}  // namespace

extern "C" {

// shared error slot for the other entry points of the library (dynmask.hip)
int dynmask_set_error(int code, const char* what) { return fail(code, what); }

int msda_hip_abi_version(void) { return MSDA_HIP_ABI_VERSION; }
const char* msda_hip_last_error(void) { return g_err; }

int msda_hip_forward_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                         const float* sampling_loc, const float* attn_weight, int batch, int spatial_size,
                         int num_heads, int channels, int num_levels, int num_query, int num_point, float* output,
                         void* stream) {
  const msda::Dims d{batch, spatial_size, num_heads, channels, num_levels, num_query, num_point};
  return forward_impl<float>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, d, output, stream);
}

int msda_hip_forward_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                         const double* sampling_loc, const double* attn_weight, int batch, int spatial_size,
                         int num_heads, int channels, int num_levels, int num_query, int num_point, double* output,
                         void* stream) {
  const msda::Dims d{batch, spatial_size, num_heads, channels, num_levels, num_query, num_point};
  return forward_impl<double>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, d, output, stream);
}

int msda_hip_backward_f32(const float* grad_output, const float* value, const int64_t* spatial_shapes,
                          const int64_t* level_start_index, const float* sampling_loc, const float* attn_weight,
                          int batch, int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                          int num_point, float* grad_value, float* grad_sampling_loc, float* grad_attn_weight,
                          void* stream) {
  const msda::Dims d{batch, spatial_size, num_heads, channels, num_levels, num_query, num_point};
  return backward_impl<float>(grad_output, value, spatial_shapes, level_start_index, sampling_loc, attn_weight, d,
                              grad_value, grad_sampling_loc, grad_attn_weight, stream);
}

size_t msda_hip_backward_workspace_bytes(int batch, int spatial_size, int num_heads, int channels, int num_levels,
                                         int num_query, int num_point) {
  const msda::Dims d{batch, spatial_size, num_heads, channels, num_levels, num_query, num_point};
  if (check_dims(d) || d.N == 0 || d.Lq == 0) return 0;
  return msda::regions_workspace_bytes(d);
}

int msda_hip_backward_ws_f32(const float* grad_output, const float* value, const int64_t* spatial_shapes,
                             const int64_t* level_start_index, const float* sampling_loc, const float* attn_weight,
                             int batch, int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                             int num_point, float* grad_value, float* grad_sampling_loc, float* grad_attn_weight,
                             void* workspace, size_t workspace_bytes, void* stream) {
  const msda::Dims d{batch, spatial_size, num_heads, channels, num_levels, num_query, num_point};
  msda::set_call_workspace(workspace, workspace_bytes);   // consumed by this call (ContextScope clears what is left)
  return backward_impl<float>(grad_output, value, spatial_shapes, level_start_index, sampling_loc, attn_weight, d,
                              grad_value, grad_sampling_loc, grad_attn_weight, stream);
}

int msda_hip_backward_f64(const double* grad_output, const double* value, const int64_t* spatial_shapes,
                          const int64_t* level_start_index, const double* sampling_loc, const double* attn_weight,
                          int batch, int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                          int num_point, double* grad_value, double* grad_sampling_loc, double* grad_attn_weight,
                          void* stream) {
  const msda::Dims d{batch, spatial_size, num_heads, channels, num_levels, num_query, num_point};
  return backward_impl<double>(grad_output, value, spatial_shapes, level_start_index, sampling_loc, attn_weight, d,
                               grad_value, grad_sampling_loc, grad_attn_weight, stream);
}

static int forward_fused_impl(const float* value, int head_major, const int64_t* spatial_shapes,
                              const int64_t* level_start_index, const float* reference_points, int ref_dim,
                              const float* sampling_offsets, const float* attn_logits, const msda::Dims& d, float* output,
                              void* stream) {
  const ContextScope scope;
  if (int rc = check_dims(d)) return rc;
  if (!msda::fused_forward_ok(d, ref_dim))
    return fail(MSDA_ERR_UNSUPPORTED, "fused forward needs channels == 32, num_levels * num_point == 16, ref_dim 2 or 4");
  if (head_major && !msda::fused_forward_hm_ok(d, ref_dim))
    return fail(MSDA_ERR_UNSUPPORTED, "fused forward on head-major value needs num_levels == num_point == 4 and num_query >= 1024");
  if (d.N == 0 || d.Lq == 0) return 0;
  if (!value || !spatial_shapes || !level_start_index || !reference_points || !sampling_offsets || !attn_logits || !output)
    return fail(MSDA_ERR_NULL_POINTER, "null pointer argument");
  const char* name = "";
  const int rc = msda::launch_forward_fused(current_variant(0), value, head_major, spatial_shapes, level_start_index, reference_points,
                                            ref_dim, sampling_offsets, attn_logits, d, output, (hipStream_t)stream, &name);
  g_last_kernel[0].store(name, std::memory_order_relaxed);
  return finish(rc, "msda_hip_forward_fused");
}

int msda_hip_forward_fused_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                               const float* reference_points, int ref_dim, const float* sampling_offsets,
                               const float* attn_logits, int batch, int spatial_size, int num_heads, int channels,
                               int num_levels, int num_query, int num_point, float* output, void* stream) {
  const msda::Dims d{batch, spatial_size, num_heads, channels, num_levels, num_query, num_point};
  return forward_fused_impl(value, 0, spatial_shapes, level_start_index, reference_points, ref_dim, sampling_offsets,
                            attn_logits, d, output, stream);
}

int msda_hip_forward_fused_hm_f32(const float* value_head_major, const int64_t* spatial_shapes,
                                  const int64_t* level_start_index, const float* reference_points, int ref_dim,
                                  const float* sampling_offsets, const float* attn_logits, int batch, int spatial_size,
                                  int num_heads, int channels, int num_levels, int num_query, int num_point,
                                  float* output, void* stream) {
  const msda::Dims d{batch, spatial_size, num_heads, channels, num_levels, num_query, num_point};
  return forward_fused_impl(value_head_major, 1, spatial_shapes, level_start_index, reference_points, ref_dim,
                            sampling_offsets, attn_logits, d, output, stream);
}

int msda_hip_set_variant(int which, int variant) {
  if (which < 0 || which > 1 || !variant_available(which, variant))
    return fail(MSDA_ERR_BAD_VARIANT, "unknown kernel variant (or an experiment this build does not carry)");
  g_variant[which].store(variant, std::memory_order_relaxed);
  return 0;
}

int msda_hip_get_variant(int which) { return (which < 0 || which > 1) ? MSDA_ERR_BAD_VARIANT : current_variant(which); }

const char* msda_hip_variant_name(int which, int variant) {
  if (which < 0 || which > 1 || variant < 0 || variant >= num_variants(which)) return nullptr;
  return variant_name(which, variant);
}

int msda_hip_forward_locality(double* far_fraction) { return msda::forward_locality(far_fraction); }
void msda_hip_reset_call_site(int call_site) { msda::reset_call_site(call_site); }

void msda_hip_set_call_context(int call_site, unsigned flags) { msda::set_call_context(call_site, flags); }

const char* msda_hip_last_kernel(int which) {
  return (which < 0 || which > 1) ? "" : g_last_kernel[which].load(std::memory_order_relaxed);
}

}  // extern "C"
