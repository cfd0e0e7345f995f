/*
 * oracle/msda_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's MultiScaleDeformableAttention
 * operator (forward + backward), used ONLY as the checker by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product path
 * (uninext_amd/csrc HIP sources behind include/msda_hip.h) never links, loads or
 * calls anything in this file.
 *
 * Reference semantics followed (paths relative to the UNINEXT checkout,
 * ops/ = projects/UNINEXT/uninext/models/deformable_detr/ops/):
 *   forward : ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299 (per-output loop)
 *             + :33-84 (bilinear sample with per-corner zero padding)
 *             == ops/functions/ms_deform_attn_func.py:43-63 (grid_sample,
 *             bilinear, zeros, align_corners=False).
 *   backward: ops/src/cuda/ms_deform_im2col_cuda.cuh:87-159 (per-sample
 *             gradients) accumulated as in :301-403; outputs zero-initialised
 *             as in ops/src/cuda/ms_deform_attn_cuda.cu:121-123.
 *
 * Parity pin: tests/test_oracle_golden.py checks this file against fixtures in
 * tests/golden/ that were produced by importing the reference's own
 * ms_deform_attn_core_pytorch (+ autograd through it) -- see
 * tests/golden/make_golden.py.
 *
 * Layouts (all contiguous, row-major):
 *   value [N,S,M,D]   shapes [L,2] (H,W) int64   lsi [L] int64
 *   loc   [N,Lq,M,L,P,2] (x = along W first, y = along H second, in [0,1])
 *   attn  [N,Lq,M,L,P]   out [N,Lq,M*D]
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/*
 * Pixel coordinate of a sample: loc * size - 0.5 as ONE fused multiply-add.  The reference's CUDA source writes
 * `loc_h * spatial_h - 0.5` (cuh:285-286), which nvcc contracts into an FMA (default -fmad=true); floor() of the result
 * decides the bilinear cell, so for a sample within an ulp of a cell edge one rounding or two pick different -- equally
 * valid -- one-sided derivatives (about one sample in a million; found on the RefCOCO-size pyramids).  The restatement
 * follows the compiled reference.  float: the double product of two floats is exact and so is the sum, one rounding to
 * float at the end; double: fma() is exact by definition.
 */
static inline float msda_coord_f32(float loc, int size) { return (float)((double)loc * (double)size - 0.5); }
static inline double msda_coord_f64(double loc, int size) { return fma(loc, (double)size, -0.5); }

#define MSDA_ORACLE_DEFINE(T, SUFFIX)                                                     \
                                                                                          \
  int msda_oracle_forward_##SUFFIX(const T* value, const int64_t* shapes,                 \
                                   const int64_t* lsi, const T* loc, const T* attn,       \
                                   int N, int S, int M, int D, int L, int Lq, int P,      \
                                   T* out) {                                              \
    const int64_t pix_stride = (int64_t)M * D;                                            \
    for (int b = 0; b < N; ++b)                                                           \
      for (int q = 0; q < Lq; ++q)                                                        \
        for (int m = 0; m < M; ++m) {                                                     \
          const int64_t pair = ((int64_t)b * Lq + q) * M + m;                             \
          const T* a_ptr = attn + pair * L * P;                                           \
          const T* l_ptr = loc + pair * L * P * 2;                                        \
          T* o_ptr = out + pair * D;                                                      \
          for (int c = 0; c < D; ++c) o_ptr[c] = (T)0;                                    \
          for (int l = 0; l < L; ++l) {                                                   \
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                 \
            const T* v_lvl = value + ((int64_t)b * S + lsi[l]) * pix_stride + m * D;      \
            for (int p = 0; p < P; ++p) {                                                 \
              const T loc_w = l_ptr[(l * P + p) * 2];                                     \
              const T loc_h = l_ptr[(l * P + p) * 2 + 1];                                 \
              const T weight = a_ptr[l * P + p];                                          \
              const T h_im = msda_coord_##SUFFIX(loc_h, H);                               \
              const T w_im = msda_coord_##SUFFIX(loc_w, W);                               \
              if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue;            \
              const int h_low = (int)floor(h_im), w_low = (int)floor(w_im);               \
              const int h_high = h_low + 1, w_high = w_low + 1;                           \
              const T lh = h_im - h_low, lw = w_im - w_low;                               \
              const T hh = 1 - lh, hw = 1 - lw;                                           \
              const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;             \
              const int ok1 = h_low >= 0 && w_low >= 0;                                   \
              const int ok2 = h_low >= 0 && w_high <= W - 1;                              \
              const int ok3 = h_high <= H - 1 && w_low >= 0;                              \
              const int ok4 = h_high <= H - 1 && w_high <= W - 1;                         \
              for (int c = 0; c < D; ++c) {                                               \
                const T v1 = ok1 ? v_lvl[((int64_t)h_low * W + w_low) * pix_stride + c] : (T)0;   \
                const T v2 = ok2 ? v_lvl[((int64_t)h_low * W + w_high) * pix_stride + c] : (T)0;  \
                const T v3 = ok3 ? v_lvl[((int64_t)h_high * W + w_low) * pix_stride + c] : (T)0;  \
                const T v4 = ok4 ? v_lvl[((int64_t)h_high * W + w_high) * pix_stride + c] : (T)0; \
                o_ptr[c] += (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4) * weight;             \
              }                                                                           \
            }                                                                             \
          }                                                                               \
        }                                                                                 \
    return 0;                                                                             \
  }                                                                                       \
                                                                                          \
  int msda_oracle_backward_##SUFFIX(const T* grad_out, const T* value,                    \
                                    const int64_t* shapes, const int64_t* lsi,            \
                                    const T* loc, const T* attn, int N, int S, int M,     \
                                    int D, int L, int Lq, int P, T* grad_value,           \
                                    T* grad_loc, T* grad_attn) {                          \
    const int64_t pix_stride = (int64_t)M * D;                                            \
    memset(grad_value, 0, sizeof(T) * (size_t)N * S * M * D);                             \
    memset(grad_loc, 0, sizeof(T) * (size_t)N * Lq * M * L * P * 2);                      \
    memset(grad_attn, 0, sizeof(T) * (size_t)N * Lq * M * L * P);                         \
    for (int b = 0; b < N; ++b)                                                           \
      for (int q = 0; q < Lq; ++q)                                                        \
        for (int m = 0; m < M; ++m) {                                                     \
          const int64_t pair = ((int64_t)b * Lq + q) * M + m;                             \
          const T* a_ptr = attn + pair * L * P;                                           \
          const T* l_ptr = loc + pair * L * P * 2;                                        \
          const T* g_ptr = grad_out + pair * D;                                           \
          for (int l = 0; l < L; ++l) {                                                   \
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                 \
            const int64_t lvl_off = ((int64_t)b * S + lsi[l]) * pix_stride + m * D;       \
            const T* v_lvl = value + lvl_off;                                             \
            T* gv_lvl = grad_value + lvl_off;                                             \
            for (int p = 0; p < P; ++p) {                                                 \
              const int s = l * P + p;                                                    \
              const T loc_w = l_ptr[s * 2], loc_h = l_ptr[s * 2 + 1];                     \
              const T weight = a_ptr[s];                                                  \
              const T h_im = msda_coord_##SUFFIX(loc_h, H);                               \
              const T w_im = msda_coord_##SUFFIX(loc_w, W);                               \
              if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue;            \
              const int h_low = (int)floor(h_im), w_low = (int)floor(w_im);               \
              const int h_high = h_low + 1, w_high = w_low + 1;                           \
              const T lh = h_im - h_low, lw = w_im - w_low;                               \
              const T hh = 1 - lh, hw = 1 - lw;                                           \
              const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;             \
              const int ok1 = h_low >= 0 && w_low >= 0;                                   \
              const int ok2 = h_low >= 0 && w_high <= W - 1;                              \
              const int ok3 = h_high <= H - 1 && w_low >= 0;                              \
              const int ok4 = h_high <= H - 1 && w_high <= W - 1;                         \
              const int64_t o1 = ((int64_t)h_low * W + w_low) * pix_stride;               \
              const int64_t o2 = ((int64_t)h_low * W + w_high) * pix_stride;              \
              const int64_t o3 = ((int64_t)h_high * W + w_low) * pix_stride;              \
              const int64_t o4 = ((int64_t)h_high * W + w_high) * pix_stride;             \
              T acc_w = 0, acc_h = 0, acc_a = 0;                                          \
              for (int c = 0; c < D; ++c) {                                               \
                const T top_grad = g_ptr[c];                                              \
                const T tgv = top_grad * weight;                                          \
                T gh = 0, gw = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;                         \
                if (ok1) { v1 = v_lvl[o1 + c]; gh -= hw * v1; gw -= hh * v1; gv_lvl[o1 + c] += w1 * tgv; } \
                if (ok2) { v2 = v_lvl[o2 + c]; gh -= lw * v2; gw += hh * v2; gv_lvl[o2 + c] += w2 * tgv; } \
                if (ok3) { v3 = v_lvl[o3 + c]; gh += hw * v3; gw -= lh * v3; gv_lvl[o3 + c] += w3 * tgv; } \
                if (ok4) { v4 = v_lvl[o4 + c]; gh += lw * v4; gw += lh * v4; gv_lvl[o4 + c] += w4 * tgv; } \
                acc_a += top_grad * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);              \
                acc_w += W * gw * tgv;                                                    \
                acc_h += H * gh * tgv;                                                    \
              }                                                                           \
              grad_attn[pair * L * P + s] = acc_a;                                        \
              grad_loc[(pair * L * P + s) * 2] = acc_w;                                   \
              grad_loc[(pair * L * P + s) * 2 + 1] = acc_h;                               \
            }                                                                             \
          }                                                                               \
        }                                                                                 \
    return 0;                                                                             \
  }

MSDA_ORACLE_DEFINE(float, f32)
MSDA_ORACLE_DEFINE(double, f64)
