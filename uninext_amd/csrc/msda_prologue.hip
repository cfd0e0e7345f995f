// msda_hip_prologue_f32 / msda_hip_prologue_backward_f32 (include/msda_hip.h): the elementwise prologue of MSDeformAttn.forward
// -- softmax over a head's L * P attention logits and the sampling-location arithmetic, ops/modules/ms_deform_attn.py:99-112 -- and
// its backward, as one kernel each.  They are what lets the TRAINING path use the fused forward entry points
// (msda_hip_forward_fused[_hm]_f32 take the raw Linear outputs): the backward recomputes sampling_loc / attn_weight from the raw
// tensors here instead of keeping them alive from the forward (45.5 + 22.8 MB per encoder layer at N = 2), runs
// msda_hip_backward_f32, and maps grad_sampling_loc / grad_attn_weight back onto the raw tensors here -- two kernels where autograd
// runs ~20 elementwise ones with their intermediates.  gfx950; HBM-bound elementwise work: one thread per (image, query, head),
// 16-byte accesses, the 64 lanes of a wave cover 64 consecutive (query, head) pairs = contiguous memory in every tensor.
//
// Arithmetic: the reference's operations in the reference's order, contraction off (this file forms the tensors the backward kernel
// differentiates through; they should be the ones PyTorch would have formed to the last bit where the operations are the same):
//   attn   = exp(x - max) / sum                                              (F.softmax over the L * P logits of a head, :100-101)
//   loc    = ref[l] + off / (W_l, H_l)                                       (2-d reference points, :104-106)
//   loc    = ref_xy[l] + ((off / P) * ref_wh[l]) * 0.5                       (4-d reference boxes, :107-109)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/msda_hip.h"
#include "msda_common.hpp"

#pragma clang fp contract(off)

namespace msda {
namespace {

constexpr int kMaxLP = 64;        // logits of one head held in registers

// n floats (n % 4 == 0, 16-byte aligned: every per-head row of the tensors here starts at a multiple of L * P * 4 bytes) as 16-byte
// accesses -- the compiler does not merge dword accesses through a float* it cannot prove aligned, and 16 dword loads per lane with
// the lanes 64 bytes apart are 16 instructions of 64 cache lines each
template <int N4>
__device__ __forceinline__ void load4(const float* __restrict__ p, float* __restrict__ r) {
#pragma unroll
  for (int j = 0; j < N4; ++j) {
    const float4 v = reinterpret_cast<const float4*>(p)[j];
    r[4 * j] = v.x; r[4 * j + 1] = v.y; r[4 * j + 2] = v.z; r[4 * j + 3] = v.w;
  }
}
template <int N4>
__device__ __forceinline__ void store4(float* __restrict__ p, const float* __restrict__ r) {
#pragma unroll
  for (int j = 0; j < N4; ++j) reinterpret_cast<float4*>(p)[j] = make_float4(r[4 * j], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]);
}

// FAST: L == P == 4 at compile time (every shipped config): the head's 16 logits and 32 offsets stay in registers and move as
// 16-byte accesses; otherwise run-time loops over (L, P) with dword accesses (private memory for the logits).
template <int REFD>
__device__ __forceinline__ void locate(float ox, float oy, const float* rr, float W, float H, float fP, float& lx, float& ly) {
  if constexpr (REFD == 2) {
    lx = rr[0] + ox / W;
    ly = rr[1] + oy / H;
  } else {
    lx = rr[0] + ((ox / fP) * rr[2]) * 0.5f;
    ly = rr[1] + ((oy / fP) * rr[3]) * 0.5f;
  }
}

template <int REFD, bool FAST>
__global__ void __launch_bounds__(256)
prologue_fwd(const int64_t* __restrict__ shapes, const float* __restrict__ ref, const float* __restrict__ off,
             const float* __restrict__ logits, int64_t pairs, int M, int L, int P, float* __restrict__ loc,
             float* __restrict__ attn) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;        // (image, query, head)
  if (i >= pairs) return;
  const int64_t nq = i / M;
  if constexpr (FAST) {
    float x[16], o[32];
    load4<4>(logits + i * 16, x);
    load4<8>(off + i * 32, o);
    float mx = x[0];
#pragma unroll
    for (int j = 1; j < 16; ++j) mx = fmaxf(mx, x[j]);
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { x[j] = expf(x[j] - mx); sum = sum + x[j]; }
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = x[j] / sum;
    store4<4>(attn + i * 16, x);
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      float rr[4] = {0.f, 0.f, 0.f, 0.f};
      const float* r = ref + (nq * 4 + l) * REFD;
      if constexpr (REFD == 2) { const float2 v = *reinterpret_cast<const float2*>(r); rr[0] = v.x; rr[1] = v.y; }
      else load4<1>(r, rr);
      const float W = (float)shapes[2 * l + 1], H = (float)shapes[2 * l];
#pragma unroll
      for (int p = 0; p < 4; ++p) locate<REFD>(o[(l * 4 + p) * 2], o[(l * 4 + p) * 2 + 1], rr, W, H, 4.0f, o[(l * 4 + p) * 2], o[(l * 4 + p) * 2 + 1]);
    }
    store4<8>(loc + i * 32, o);
  } else {
    const int LP = L * P;
    const float* lg = logits + i * LP;
    float x[kMaxLP];
    float mx = -INFINITY;
    for (int j = 0; j < LP; ++j) { x[j] = lg[j]; mx = fmaxf(mx, x[j]); }
    float sum = 0.0f;
    for (int j = 0; j < LP; ++j) { x[j] = expf(x[j] - mx); sum = sum + x[j]; }
    float* aw = attn + i * LP;
    for (int j = 0; j < LP; ++j) aw[j] = x[j] / sum;
    const float* of = off + i * LP * 2;
    float* lc = loc + i * LP * 2;
    for (int l = 0; l < L; ++l) {
      float rr[4] = {0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < REFD; ++c) rr[c] = ref[(nq * L + l) * REFD + c];
      const float W = (float)shapes[2 * l + 1], H = (float)shapes[2 * l];
      for (int p = 0; p < P; ++p) {
        float lx, ly;
        locate<REFD>(of[(l * P + p) * 2], of[(l * P + p) * 2 + 1], rr, W, H, (float)P, lx, ly);
        lc[(l * P + p) * 2] = lx;
        lc[(l * P + p) * 2 + 1] = ly;
      }
    }
  }
}

// backward: softmax g_x_j = w_j (g_j - sum_k w_k g_k); off enters as off / (W, H)  or  ((off / P) * ref_wh) * 0.5
template <int REFD>
__device__ __forceinline__ void locate_bwd(float gx, float gy, const float* rr, float W, float H, float fP, float& ox, float& oy) {
  if constexpr (REFD == 2) {
    ox = gx / W;
    oy = gy / H;
  } else {
    ox = ((gx * 0.5f) * rr[2]) / fP;
    oy = ((gy * 0.5f) * rr[3]) / fP;
  }
}

template <int REFD, bool FAST>
__global__ void __launch_bounds__(256)
prologue_bwd(const int64_t* __restrict__ shapes, const float* __restrict__ ref, const float* __restrict__ attn,
             const float* g_loc, const float* g_attn, int64_t pairs, int M, int L, int P, float* g_off, float* g_logits) {
  // (g_off may BE g_loc and g_logits may BE g_attn: a thread reads its head's values before it writes them, and nobody else's)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pairs) return;
  const int64_t nq = i / M;
  if constexpr (FAST) {
    float w[16], g[16], o[32];
    load4<4>(attn + i * 16, w);
    load4<4>(g_attn + i * 16, g);
    load4<8>(g_loc + i * 32, o);
    float dot = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) dot = dot + w[j] * g[j];
#pragma unroll
    for (int j = 0; j < 16; ++j) g[j] = w[j] * (g[j] - dot);
    store4<4>(g_logits + i * 16, g);
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      float rr[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (REFD == 4) load4<1>(ref + (nq * 4 + l) * 4, rr);
      const float W = (float)shapes[2 * l + 1], H = (float)shapes[2 * l];
#pragma unroll
      for (int p = 0; p < 4; ++p) locate_bwd<REFD>(o[(l * 4 + p) * 2], o[(l * 4 + p) * 2 + 1], rr, W, H, 4.0f, o[(l * 4 + p) * 2], o[(l * 4 + p) * 2 + 1]);
    }
    store4<8>(g_off + i * 32, o);
  } else {
    const int LP = L * P;
    const float* w = attn + i * LP;
    const float* g = g_attn + i * LP;
    float dot = 0.0f;
    for (int j = 0; j < LP; ++j) dot = dot + w[j] * g[j];
    float* gx = g_logits + i * LP;
    for (int j = 0; j < LP; ++j) gx[j] = w[j] * (g[j] - dot);
    const float* gl = g_loc + i * LP * 2;
    float* go = g_off + i * LP * 2;
    for (int l = 0; l < L; ++l) {
      float rr[4] = {0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < REFD; ++c) rr[c] = ref[(nq * L + l) * REFD + c];
      const float W = (float)shapes[2 * l + 1], H = (float)shapes[2 * l];
      for (int p = 0; p < P; ++p) {
        float ox, oy;
        locate_bwd<REFD>(gl[(l * P + p) * 2], gl[(l * P + p) * 2 + 1], rr, W, H, (float)P, ox, oy);
        go[(l * P + p) * 2] = ox;
        go[(l * P + p) * 2 + 1] = oy;
      }
    }
  }
}

// grad of the reference points: one thread per (image, query, level), a fixed summation order over heads and points
template <int REFD>
__global__ void __launch_bounds__(256)
prologue_bwd_ref(const float* __restrict__ off, const float* __restrict__ g_loc, int64_t nql, int M, int L, int P,
                 float* __restrict__ g_ref) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;        // (image, query, level)
  if (i >= nql) return;
  const int64_t nq = i / L;
  const int l = (int)(i - nq * L);
  float sx = 0.0f, sy = 0.0f, sw = 0.0f, sh = 0.0f;
  for (int m = 0; m < M; ++m) {
    const int64_t base = (((nq * M + m) * L + l) * P) * 2;
    for (int p = 0; p < P; ++p) {
      const float gx = g_loc[base + 2 * p], gy = g_loc[base + 2 * p + 1];
      sx = sx + gx; sy = sy + gy;
      if constexpr (REFD == 4) {
        sw = sw + (gx * 0.5f) * (off[base + 2 * p] / (float)P);
        sh = sh + (gy * 0.5f) * (off[base + 2 * p + 1] / (float)P);
      }
    }
  }
  g_ref[i * REFD] = sx;
  g_ref[i * REFD + 1] = sy;
  if constexpr (REFD == 4) { g_ref[i * REFD + 2] = sw; g_ref[i * REFD + 3] = sh; }
}

}  // namespace
}  // namespace msda

extern "C" int dynmask_set_error(int code, const char* what);   // msda_capi.hip (shared last-error slot)

namespace {
int check(int batch, int M, int L, int Lq, int P, int ref_dim, const char* who) {
  if (batch < 0 || Lq < 0 || M <= 0 || L <= 0 || P <= 0) return dynmask_set_error(MSDA_ERR_BAD_DIMS, who);
  if (ref_dim != 2 && ref_dim != 4) return dynmask_set_error(MSDA_ERR_UNSUPPORTED, "msda prologue: ref_dim must be 2 or 4");
  if (L * P > msda::kMaxLP) return dynmask_set_error(MSDA_ERR_UNSUPPORTED, "msda prologue: num_levels * num_point > 64");
  return 0;
}
}  // namespace

extern "C" int msda_hip_prologue_f32(const int64_t* spatial_shapes, const float* reference_points, int ref_dim,
                                     const float* sampling_offsets, const float* attn_logits, int batch, int num_heads,
                                     int num_levels, int num_query, int num_point, float* sampling_loc, float* attn_weight,
                                     void* stream) {
  if (int rc = check(batch, num_heads, num_levels, num_query, num_point, ref_dim, "msda_hip_prologue_f32: bad dimensions")) return rc;
  const int64_t pairs = (int64_t)batch * num_query * num_heads;
  if (pairs == 0) return 0;
  if (!spatial_shapes || !reference_points || !sampling_offsets || !attn_logits || !sampling_loc || !attn_weight)
    return dynmask_set_error(MSDA_ERR_NULL_POINTER, "msda_hip_prologue_f32: null pointer");
  const int64_t blocks = (pairs + 255) / 256;
  if (blocks > 0x7fffffffLL) return dynmask_set_error(MSDA_ERR_TOO_LARGE, "msda_hip_prologue_f32: too many (query, head) pairs");
  const bool fast = num_levels == 4 && num_point == 4;
  auto k = ref_dim == 2 ? (fast ? msda::prologue_fwd<2, true> : msda::prologue_fwd<2, false>) : (fast ? msda::prologue_fwd<4, true> : msda::prologue_fwd<4, false>);
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), spatial_shapes, reference_points,
                     sampling_offsets, attn_logits, pairs, num_heads, num_levels, num_point, sampling_loc, attn_weight);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : dynmask_set_error((int)e, hipGetErrorString(e));
}

extern "C" int msda_hip_prologue_backward_f32(const int64_t* spatial_shapes, const float* reference_points, int ref_dim,
                                              const float* sampling_offsets, const float* attn_weight,
                                              const float* grad_sampling_loc, const float* grad_attn_weight, int batch,
                                              int num_heads, int num_levels, int num_query, int num_point,
                                              float* grad_sampling_offsets, float* grad_attn_logits,
                                              float* grad_reference_points, void* stream) {
  if (int rc = check(batch, num_heads, num_levels, num_query, num_point, ref_dim, "msda_hip_prologue_backward_f32: bad dimensions")) return rc;
  const int64_t pairs = (int64_t)batch * num_query * num_heads;
  if (pairs == 0) return 0;
  if (!spatial_shapes || !reference_points || !sampling_offsets || !attn_weight || !grad_sampling_loc || !grad_attn_weight ||
      !grad_sampling_offsets || !grad_attn_logits)
    return dynmask_set_error(MSDA_ERR_NULL_POINTER, "msda_hip_prologue_backward_f32: null pointer");
  const int64_t blocks = (pairs + 255) / 256;
  if (blocks > 0x7fffffffLL) return dynmask_set_error(MSDA_ERR_TOO_LARGE, "msda_hip_prologue_backward_f32: too many (query, head) pairs");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool fast = num_levels == 4 && num_point == 4;
  if (grad_reference_points) {                       // first: it reads grad_sampling_loc, which the next kernel may overwrite in place
    const int64_t nql = (int64_t)batch * num_query * num_levels;
    auto kr = ref_dim == 2 ? msda::prologue_bwd_ref<2> : msda::prologue_bwd_ref<4>;
    hipLaunchKernelGGL(kr, dim3((unsigned)((nql + 255) / 256)), dim3(256), 0, s, sampling_offsets, grad_sampling_loc, nql, num_heads,
                       num_levels, num_point, grad_reference_points);
  }
  auto k = ref_dim == 2 ? (fast ? msda::prologue_bwd<2, true> : msda::prologue_bwd<2, false>) : (fast ? msda::prologue_bwd<4, true> : msda::prologue_bwd<4, false>);
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), 0, s, spatial_shapes, reference_points, attn_weight, grad_sampling_loc,
                     grad_attn_weight, pairs, num_heads, num_levels, num_point, grad_sampling_offsets, grad_attn_logits);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : dynmask_set_error((int)e, hipGetErrorString(e));
}
