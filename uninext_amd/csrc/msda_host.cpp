// Host-pointer (CPU) variants of the operator's C ABI: msda_host_{forward,backward}_{f32,f64} (include/msda_hip.h).
//
// SURVEY.md 8(b)(i) asks for them next to the device entry points: the reference has no CPU implementation at all
// (ops/src/cpu/ms_deform_attn_cpu.cpp:17-41 are AT_ERROR stubs, ops/src/ms_deform_attn.h:35-38 raises for CPU tensors),
// so BASELINE configs[0] -- the model's plumbing on a GPU-less box -- has to go through the slow grid_sample
// composition ms_deform_attn_core_pytorch (ops/functions/ms_deform_attn_func.py:43-63).  These functions compute the
// same operator (semantics of ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299 forward and :87-159 backward: bilinear
// taps with corner-wise zero padding, locations normalised to [0,1], x first) directly on host memory, threaded with
// std::thread.  Plain C++, no HIP call: the library can serve them on a box without a GPU.
//
// Work split: forward -- contiguous ranges of (image, query) rows per thread; backward -- one (image, head) slice of
// grad_value per work item, so no two threads ever add into the same element (deterministic, no atomics).
#include "../../include/msda_hip.h"

#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

extern "C" int dynmask_set_error(int code, const char* what);   // msda_capi.hip: the library's error slot

namespace {

struct HostDims {
  int N, S, M, D, L, Lq, P;
};

int check(const HostDims& d, bool ptrs_ok) {
  if (d.N < 0 || d.Lq < 0) return dynmask_set_error(MSDA_ERR_BAD_DIMS, "batch and num_query must be >= 0");
  if (d.S <= 0 || d.M <= 0 || d.D <= 0 || d.L <= 0 || d.P <= 0)
    return dynmask_set_error(MSDA_ERR_BAD_DIMS, "spatial_size, num_heads, channels, num_levels, num_point must be > 0");
  if (d.N == 0 || d.Lq == 0) return 0;
  if (!ptrs_ok) return dynmask_set_error(MSDA_ERR_NULL_POINTER, "null pointer argument");
  return 0;
}

// The host pointers are dereferenced at value[(b * S + lsi[l] + pixel) * M * D ...]: levels that do not lie inside
// [0, spatial_size) would read -- and the backward WRITE -- outside the tensors (the GPU kernels are clamped by their
// buffer descriptors; plain host memory is not).
int check_geometry(const HostDims& d, const int64_t* shapes, const int64_t* lsi) {
  for (int l = 0; l < d.L; ++l) {
    const int64_t H = shapes[2 * l], W = shapes[2 * l + 1], st = lsi[l];
    // H, W <= S first: H * W <= S * S then fits int64, and the start is compared WITHOUT an addition (a start near INT64_MAX
    // would wrap negative and pass)
    if (H <= 0 || W <= 0 || H > (int64_t)d.S || W > (int64_t)d.S || st < 0 || H * W > (int64_t)d.S ||
        st > (int64_t)d.S - H * W)
      return dynmask_set_error(MSDA_ERR_BAD_DIMS, "spatial_shapes / level_start_index do not fit spatial_size");
  }
  return 0;
}

// items = independent units the call can be split into; work = the call's size in cheap units (output rows / samples
// rows): the backward's units are whole (image, head) slices of Lq rows each, so its thread count must not be derived
// from the NUMBER of slices (16 slices / 256 = 0 -> one thread, an N * M-fold slow-down of the default CPU backward).
int thread_count(int requested, int64_t items, int64_t work) {
  int n = requested > 0 ? requested : (int)std::thread::hardware_concurrency();
  // threads are created per call: not more than one per ~256 rows of work (a 600-row decoder call gets 2-3, not 256)
  if (requested <= 0) n = (int)std::min<int64_t>(n, std::max<int64_t>(1, work / 256));
  if (n < 1) n = 1;
  if ((int64_t)n > items) n = (int)std::max<int64_t>(items, 1);
  return n;
}

thread_local int t_last_threads = 0;   // msda_host_last_num_threads(): what the last call of this thread ran on

template <typename F>
void parallel_for(int64_t items, int threads, F&& body) {   // body(begin, end)
  t_last_threads = threads <= 1 ? 1 : threads;
  if (threads <= 1) { body((int64_t)0, items); return; }
  std::vector<std::thread> pool;
  pool.reserve(threads);
  for (int t = 0; t < threads; ++t) {
    const int64_t lo = items * t / threads, hi = items * (t + 1) / threads;
    if (lo < hi) pool.emplace_back([=, &body] { body(lo, hi); });
  }
  for (auto& th : pool) th.join();
}

// One bilinear tap: corner indices / validity / fractions of location (lx, ly) on an H x W level.
template <typename T>
struct Tap {
  bool in_range, ok[4];
  int64_t pix[4];          // pixel index inside the level for corners TL, TR, BL, BR
  T lh, lw, hh, hw;
};

template <typename T>
inline Tap<T> make_tap(T lx, T ly, int H, int W) {
  Tap<T> t;
  const T h_im = ly * (T)H - (T)0.5, w_im = lx * (T)W - (T)0.5;
  t.in_range = (h_im > (T)-1) && (w_im > (T)-1) && (h_im < (T)H) && (w_im < (T)W);   // false for NaN
  if (!t.in_range) return t;
  const T hf = std::floor(h_im), wf = std::floor(w_im);
  const int h0 = (int)hf, w0 = (int)wf;
  t.lh = h_im - hf; t.lw = w_im - wf; t.hh = (T)1 - t.lh; t.hw = (T)1 - t.lw;
  const bool top = h0 >= 0, bot = h0 + 1 <= H - 1, lft = w0 >= 0, rgt = w0 + 1 <= W - 1;
  t.ok[0] = top && lft; t.ok[1] = top && rgt; t.ok[2] = bot && lft; t.ok[3] = bot && rgt;
  const int64_t p0 = (int64_t)h0 * W + w0;
  t.pix[0] = p0; t.pix[1] = p0 + 1; t.pix[2] = p0 + W; t.pix[3] = p0 + W + 1;
  return t;
}

template <typename T>
int forward_host(const T* value, const int64_t* shapes, const int64_t* lsi, const T* loc, const T* attn,
                 const HostDims& d, T* out, int num_threads) {
  if (int rc = check(d, value && shapes && lsi && loc && attn && out)) return rc;
  if (d.N > 0 && d.Lq > 0)
    if (int rc = check_geometry(d, shapes, lsi)) return rc;
  if (d.N == 0 || d.Lq == 0) return 0;
  const int64_t rows = (int64_t)d.N * d.Lq;
  const int64_t pix_stride = (int64_t)d.M * d.D;
  const int LP = d.L * d.P;
  parallel_for(rows, thread_count(num_threads, rows, rows), [&](int64_t lo, int64_t hi) {
    std::vector<T> acc((size_t)d.D);
    for (int64_t row = lo; row < hi; ++row) {
      const int64_t b = row / d.Lq;
      for (int m = 0; m < d.M; ++m) {
        const int64_t pair = row * d.M + m;
        std::fill(acc.begin(), acc.end(), (T)0);
        for (int l = 0; l < d.L; ++l) {
          const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
          const T* lvl = value + (b * d.S + lsi[l]) * pix_stride + (int64_t)m * d.D;
          for (int p = 0; p < d.P; ++p) {
            const int64_t s = pair * LP + l * d.P + p;
            const Tap<T> t = make_tap<T>(loc[2 * s], loc[2 * s + 1], H, W);
            if (!t.in_range) continue;
            const T a = attn[s];
            const T w[4] = {t.hh * t.hw, t.hh * t.lw, t.lh * t.hw, t.lh * t.lw};
            const T* v[4];
            for (int k = 0; k < 4; ++k) v[k] = t.ok[k] ? lvl + t.pix[k] * pix_stride : nullptr;
            for (int c = 0; c < d.D; ++c) {
              T smp = 0;
              for (int k = 0; k < 4; ++k)
                if (v[k]) smp += w[k] * v[k][c];
              acc[c] += smp * a;
            }
          }
        }
        std::copy(acc.begin(), acc.end(), out + pair * d.D);
      }
    }
  });
  return 0;
}

template <typename T>
int backward_host(const T* grad_out, const T* value, const int64_t* shapes, const int64_t* lsi, const T* loc,
                  const T* attn, const HostDims& d, T* grad_value, T* grad_loc, T* grad_attn, int num_threads) {
  if (int rc = check(d, grad_out && value && shapes && lsi && loc && attn && grad_value && grad_loc && grad_attn)) return rc;
  if (d.N > 0 && d.Lq > 0)
    if (int rc = check_geometry(d, shapes, lsi)) return rc;
  if (d.N == 0 || d.Lq == 0) return 0;
  const int64_t slices = (int64_t)d.N * d.M;     // a thread owns grad_value[b, :, m, :]
  const int64_t pix_stride = (int64_t)d.M * d.D;
  const int LP = d.L * d.P;
  parallel_for(slices, thread_count(num_threads, slices, slices * (int64_t)d.Lq), [&](int64_t lo, int64_t hi) {
    for (int64_t sl = lo; sl < hi; ++sl) {
      const int64_t b = sl / d.M;
      const int m = (int)(sl % d.M);
      for (int q = 0; q < d.Lq; ++q) {
        const int64_t pair = (b * d.Lq + q) * d.M + m;
        const T* g = grad_out + pair * d.D;
        for (int l = 0; l < d.L; ++l) {
          const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
          const int64_t lvl_off = (b * d.S + lsi[l]) * pix_stride + (int64_t)m * d.D;
          for (int p = 0; p < d.P; ++p) {
            const int64_t s = pair * LP + l * d.P + p;
            const Tap<T> t = make_tap<T>(loc[2 * s], loc[2 * s + 1], H, W);
            T ga = 0, gw = 0, gh = 0;
            if (t.in_range) {
              const T a = attn[s];
              const T w[4] = {t.hh * t.hw, t.hh * t.lw, t.lh * t.hw, t.lh * t.lw};
              const T* v[4];
              T* gv[4];
              for (int k = 0; k < 4; ++k) {
                v[k] = t.ok[k] ? value + lvl_off + t.pix[k] * pix_stride : nullptr;
                gv[k] = t.ok[k] ? grad_value + lvl_off + t.pix[k] * pix_stride : nullptr;
              }
              for (int c = 0; c < d.D; ++c) {
                const T tgv = g[c] * a;
                T vv[4];
                for (int k = 0; k < 4; ++k) {
                  vv[k] = v[k] ? v[k][c] : (T)0;
                  if (gv[k]) gv[k][c] += w[k] * tgv;
                }
                ga += g[c] * (w[0] * vv[0] + w[1] * vv[1] + w[2] * vv[2] + w[3] * vv[3]);
                gw += tgv * (t.hh * (vv[1] - vv[0]) + t.lh * (vv[3] - vv[2]));
                gh += tgv * (t.hw * (vv[2] - vv[0]) + t.lw * (vv[3] - vv[1]));
              }
            }
            grad_attn[s] = ga;
            grad_loc[2 * s] = (T)W * gw;
            grad_loc[2 * s + 1] = (T)H * gh;
          }
        }
      }
    }
  });
  return 0;
}

}  // namespace

extern "C" {

int msda_host_forward_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                          const float* sampling_loc, const float* attn_weight, int batch, int spatial_size, int num_heads,
                          int channels, int num_levels, int num_query, int num_point, float* output, int num_threads) {
  const HostDims d{batch, spatial_size, num_heads, channels, num_levels, num_query, num_point};
  return forward_host<float>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, d, output, num_threads);
}

int msda_host_forward_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                          const double* sampling_loc, const double* attn_weight, int batch, int spatial_size,
                          int num_heads, int channels, int num_levels, int num_query, int num_point, double* output,
                          int num_threads) {
  const HostDims d{batch, spatial_size, num_heads, channels, num_levels, num_query, num_point};
  return forward_host<double>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, d, output, num_threads);
}

int msda_host_backward_f32(const float* grad_output, const float* value, const int64_t* spatial_shapes,
                           const int64_t* level_start_index, const float* sampling_loc, const float* attn_weight,
                           int batch, int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                           int num_point, float* grad_value, float* grad_sampling_loc, float* grad_attn_weight,
                           int num_threads) {
  const HostDims d{batch, spatial_size, num_heads, channels, num_levels, num_query, num_point};
  return backward_host<float>(grad_output, value, spatial_shapes, level_start_index, sampling_loc, attn_weight, d,
                              grad_value, grad_sampling_loc, grad_attn_weight, num_threads);
}

int msda_host_backward_f64(const double* grad_output, const double* value, const int64_t* spatial_shapes,
                           const int64_t* level_start_index, const double* sampling_loc, const double* attn_weight,
                           int batch, int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                           int num_point, double* grad_value, double* grad_sampling_loc, double* grad_attn_weight,
                           int num_threads) {
  const HostDims d{batch, spatial_size, num_heads, channels, num_levels, num_query, num_point};
  return backward_host<double>(grad_output, value, spatial_shapes, level_start_index, sampling_loc, attn_weight, d,
                               grad_value, grad_sampling_loc, grad_attn_weight, num_threads);
}


int msda_host_last_num_threads(void) { return t_last_threads; }

}  // extern "C"
