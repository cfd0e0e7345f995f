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

const char* const kVariantNames[2][msda::kNumVariants] = {
    {"auto", "msda_fwd_generic", "msda_fwd_lanegroup", "msda_fwd_tiled", "msda_fwd_tiled_l0", "msda_fwd_tiled_l0big", "msda_fwd_lgcl", "msda_fwd_lg3", "msda_fwd_lgp", "msda_fwd_win", "msda_fwd_win2", "msda_fwd_win3", "msda_fwd_win4"},
    {"auto", "msda_bwd_generic", "msda_bwd_lanegroup", "msda_bwd_tiled", "msda_bwd_win", "msda_bwd_dec", "msda_bwd_tiled", "msda_bwd_tiled", "msda_bwd_tiled", "msda_bwd_tiled", "msda_bwd_tiled", "msda_bwd_tiled", "msda_bwd_tiled"},
};

int current_variant(int which) {
  int v = g_variant[which].load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = std::getenv(which == 0 ? "MSDA_HIP_FWD_VARIANT" : "MSDA_HIP_BWD_VARIANT");
    v = e ? std::atoi(e) : 0;
    if (v < 0 || v >= msda::kNumVariants) v = 0;
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
  if (which < 0 || which > 1 || variant < 0 || variant >= msda::kNumVariants)
    return fail(MSDA_ERR_BAD_VARIANT, "unknown kernel variant");
  g_variant[which].store(variant, std::memory_order_relaxed);
  return 0;
}

int msda_hip_get_variant(int which) { return (which < 0 || which > 1) ? MSDA_ERR_BAD_VARIANT : current_variant(which); }

const char* msda_hip_variant_name(int which, int variant) {
  if (which < 0 || which > 1 || variant < 0 || variant >= msda::kNumVariants) return nullptr;
  return kVariantNames[which][variant];
}

int msda_hip_forward_locality(double* far_fraction) { return msda::forward_locality(far_fraction); }

void msda_hip_set_call_context(int call_site, unsigned flags) { msda::set_call_context(call_site, flags); }

const char* msda_hip_last_kernel(int which) {
  return (which < 0 || which > 1) ? "" : g_last_kernel[which].load(std::memory_order_relaxed);
}

}  // extern "C"
