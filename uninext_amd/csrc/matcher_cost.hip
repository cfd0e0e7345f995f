// matcher_cost_hip_f32 (include/matcher_cost_hip.h): the cost matrix of HungarianMatcherVL.forward in one kernel --
// projects/UNINEXT/uninext/models/deformable_detr/matcher.py:476-498, util/box_ops.py:17-85.  One thread per
// (prediction, target) pair, target fastest: coalesced stores, a wave shares a handful of prediction rows.  Every float
// operation is the one the PyTorch composition performs, in its order: contraction is OFF for the whole file (an FMA would
// round once where two kernels round twice), divisions are IEEE, exp / log are the library functions PyTorch calls.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/matcher_cost_hip.h"
#include "msda_common.hpp"

#pragma clang fp contract(off)

namespace msda {
namespace {

constexpr float kAlpha = 0.25f, kEps = 1e-8f;               // matcher.py:480-481 (gamma = 2: PyTorch evaluates x ** 2.0 as x * x)

__device__ __forceinline__ float focal_term(float logit) {   // pos - neg of one token (matcher.py:482-484)
  const float p = 1.0f / (1.0f + expf(-logit));              // sigmoid as PyTorch writes it
  const float q = 1.0f - p;
  const float neg = ((1.0f - kAlpha) * (p * p)) * (-logf(q + kEps));
  const float pos = (kAlpha * (q * q)) * (-logf(p + kEps));
  return pos - neg;
}

__global__ void __launch_bounds__(256)
matcher_cost_kernel(const float* __restrict__ logits, const float* __restrict__ boxes, const float* __restrict__ tgt_boxes,
                    const int32_t* __restrict__ tok_off, const int32_t* __restrict__ tok_idx, int num_pred, int T, int G,
                    float w_class, float w_bbox, float w_giou, float* __restrict__ cost) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)num_pred * G) return;
  const int q = (int)(i / G), k = (int)(i - (int64_t)q * G);
  // ---- classification: mean of (pos - neg) over the target's positive tokens (matcher.py:485-488) --------------------
  const int t0 = tok_off[k], t1 = tok_off[k + 1];
  const float* row = logits + (int64_t)q * T;
  float sum = 0.0f;
  for (int t = t0; t < t1; ++t) sum = sum + focal_term(row[tok_idx[t]]);
  // the mean of PyTorch: the sum times float(#outputs) / float(#inputs) = fl(1 / count); an empty selection gives 0 * inf = NaN
  const float cls = sum * (1.0f / (float)(t1 - t0));
  // ---- L1 distance of the cxcywh boxes (torch.cdist, p = 1; matcher.py:491) ---------------------------------------------
  const float4 b = *reinterpret_cast<const float4*>(boxes + (int64_t)q * 4);
  const float4 g = *reinterpret_cast<const float4*>(tgt_boxes + (int64_t)k * 4);
  const float l1 = (fabsf(b.x - g.x) + fabsf(b.z - g.z)) + (fabsf(b.y - g.y) + fabsf(b.w - g.w));
  // ---- generalised IoU of the xyxy boxes (box_ops.py:17-23, :62-85) ------------------------------------------------------
  const float bx0 = b.x - 0.5f * b.z, by0 = b.y - 0.5f * b.w, bx1 = b.x + 0.5f * b.z, by1 = b.y + 0.5f * b.w;
  const float gx0 = g.x - 0.5f * g.z, gy0 = g.y - 0.5f * g.w, gx1 = g.x + 0.5f * g.z, gy1 = g.y + 0.5f * g.w;
  const float area1 = (bx1 - bx0) * (by1 - by0), area2 = (gx1 - gx0) * (gy1 - gy0);
  const float iw = fmaxf(fminf(bx1, gx1) - fmaxf(bx0, gx0), 0.0f), ih = fmaxf(fminf(by1, gy1) - fmaxf(by0, gy0), 0.0f);
  const float inter = iw * ih;
  const float uni = (area1 + area2) - inter;
  const float iou = inter / uni;
  const float hw = fmaxf(fmaxf(bx1, gx1) - fminf(bx0, gx0), 0.0f), hh = fmaxf(fmaxf(by1, gy1) - fminf(by0, gy0), 0.0f);
  const float hull = hw * hh;
  const float giou = iou - (hull - uni) / (hull + 1e-7f);
  cost[i] = ((w_bbox * l1) + (w_class * cls)) + (w_giou * (-giou));
}

}  // namespace
}  // namespace msda

extern "C" int dynmask_set_error(int code, const char* what);   // msda_capi.hip (shared last-error slot)

extern "C" int matcher_cost_hip_f32(const float* logits, const float* boxes, const float* tgt_boxes, const int32_t* tok_off,
                                    const int32_t* tok_idx, int num_pred, int num_tokens, int num_gt, float w_class,
                                    float w_bbox, float w_giou, float* cost, void* stream) {
  if (num_pred < 0 || num_gt < 0 || num_tokens < 0) {
    return dynmask_set_error(MATCHER_COST_ERR_BAD_DIMS, "matcher_cost_hip_f32: negative dimension");
  }
  if (num_pred == 0 || num_gt == 0) return 0;
  if (!logits || !boxes || !tgt_boxes || !tok_off || !cost || (!tok_idx && num_tokens > 0)) {
    return dynmask_set_error(MATCHER_COST_ERR_NULL_POINTER, "matcher_cost_hip_f32: null pointer");
  }
  const int64_t n = (int64_t)num_pred * num_gt;
  const int64_t blocks = (n + 255) / 256;
  if (blocks > 0x7fffffffLL) {
    return dynmask_set_error(MATCHER_COST_ERR_BAD_DIMS, "matcher_cost_hip_f32: num_pred * num_gt too large");
  }
  hipLaunchKernelGGL(msda::matcher_cost_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), logits,
                     boxes, tgt_boxes, tok_off, tok_idx, num_pred, num_tokens, num_gt, w_class, w_bbox, w_giou, cost);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : dynmask_set_error((int)e, hipGetErrorString(e));
}

