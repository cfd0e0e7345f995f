// ota_cost_hip_f32 / ota_dynamic_k_hip (include/ota_hip.h): the simOTA assignment of HungarianMatcherVL.forward_ota on the
// device -- projects/UNINEXT/uninext/models/deformable_detr/matcher.py:313-447, util/box_ops.py:17-85.  gfx950.
//
// The outputs are integer index lists; every float that decides one is the float the reference's PyTorch composition
// forms: contraction is OFF for the whole file (an FMA rounds once where two kernels round twice), divisions are IEEE, the
// in-place penalties are float32 adds on the stored matrix in the reference's sequence (adding 100000 to a row can MERGE
// two nearby costs into a tie, which the lowest-index rule then decides -- so the adds cannot be folded away).
// Selections (top-k, argmin, argmax) break ties towards the lowest index, PyTorch's documented rule for min / max with
// indices and what its sort-based top-k does for equal keys.
//
// Mapping: the cost kernel is one thread per (query, target) pair, target fastest (coalesced stores; the threads of a wave
// share a few query rows and the target rows are a few hundred bytes).  The assignment is ONE 1024-thread workgroup per
// image: row operations (a query's claims, its cheapest target, penalties) are a thread per query, column operations (a
// target's top-k, its cheapest query) a wave per target with the lanes striding over the queries; phases are separated by
// workgroup barriers, nothing leaves the CU but the matrices themselves (L2-resident: 900 x G floats).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/ota_hip.h"
#include "msda_common.hpp"

#pragma clang fp contract(off)

namespace msda {
namespace {

struct GtOffsets { int32_t off[OTA_HIP_MAX_BATCH + 1]; };

constexpr float kGiouWeight = 3.0f, kPriorPenalty = 100.0f, kBgPenalty = 10000.0f, kTakenPenalty = 100000.0f;   // matcher.py:338,340,415
constexpr float kCentreHalf = 2.5f / 32.0f;     // 1 * center_radius / expanded_strides (matcher.py:323,367): 0.078125, exact
constexpr int kTopIou = 10;                     // matcher.py:391

__global__ void __launch_bounds__(256)
ota_cost_kernel(const float* __restrict__ table, const float* __restrict__ boxes, const float* __restrict__ tgt_boxes,
                const uint8_t* __restrict__ posmap, GtOffsets go, int Q, int T, float* __restrict__ cost,
                float* __restrict__ iou_out, uint8_t* __restrict__ flags) {
  const int b = blockIdx.y;
  const int g0 = go.off[b], G = go.off[b + 1] - g0;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)Q * G) return;
  const int q = (int)(i / G), k = (int)(i - (int64_t)q * G);
  // ---- classification: mean of the focal table over the target's positive tokens (matcher.py:332-334).  PyTorch's mean
  // is the sum times fl(1 / count); an empty selection gives 0 * inf = NaN, as there
  const float* row = table + ((int64_t)b * Q + q) * T;
  const uint8_t* pm = posmap + (int64_t)(g0 + k) * T;
  float sum = 0.0f;
  int cnt = 0;
  int t = 0;
  if ((T & 3) == 0 && (((uintptr_t)pm) & 3) == 0) {             // four map bytes at a time: most words are zero
    for (; t < T; t += 4) {
      const uint32_t w = *reinterpret_cast<const uint32_t*>(pm + t);
      if (w == 0u) continue;
      if (w & 0x000000ffu) { sum = sum + row[t]; ++cnt; }
      if (w & 0x0000ff00u) { sum = sum + row[t + 1]; ++cnt; }
      if (w & 0x00ff0000u) { sum = sum + row[t + 2]; ++cnt; }
      if (w & 0xff000000u) { sum = sum + row[t + 3]; ++cnt; }
    }
  } else {
    for (; t < T; ++t)
      if (pm[t]) { sum = sum + row[t]; ++cnt; }
  }
  const float cls = sum * (1.0f / (float)cnt);
  // ---- IoU and generalised IoU of the xyxy boxes (box_ops.py:17-23, torchvision box_iou, box_ops.py:62-85) --------------
  const float4 bq = *reinterpret_cast<const float4*>(boxes + ((int64_t)b * Q + q) * 4);
  const float4 g = *reinterpret_cast<const float4*>(tgt_boxes + (int64_t)(g0 + k) * 4);
  const float bx0 = bq.x - 0.5f * bq.z, by0 = bq.y - 0.5f * bq.w, bx1 = bq.x + 0.5f * bq.z, by1 = bq.y + 0.5f * bq.w;
  const float gx0 = g.x - 0.5f * g.z, gy0 = g.y - 0.5f * g.w, gx1 = g.x + 0.5f * g.z, gy1 = g.y + 0.5f * g.w;
  const float area1 = (bx1 - bx0) * (by1 - by0), area2 = (gx1 - gx0) * (gy1 - gy0);
  const float iw = fmaxf(fminf(bx1, gx1) - fmaxf(bx0, gx0), 0.0f), ih = fmaxf(fminf(by1, gy1) - fmaxf(by0, gy0), 0.0f);
  const float inter = iw * ih;
  const float uni = (area1 + area2) - inter;
  const float iou = inter / uni;
  const float hw = fmaxf(fmaxf(bx1, gx1) - fminf(bx0, gx0), 0.0f), hh = fmaxf(fmaxf(by1, gy1) - fminf(by0, gy0), 0.0f);
  const float hull = hw * hh;
  const float giou = iou - (hull - uni) / (hull + 1e-7f);
  // ---- centre prior (matcher.py:344-385): the query's centre strictly inside the box / inside the centre square --------
  const float cx = bq.x, cy = bq.y;
  const bool in_box = cx > gx0 && cx < gx1 && cy > gy0 && cy < gy1;
  const bool in_ctr = cx > g.x - kCentreHalf && cx < g.x + kCentreHalf && cy > g.y - kCentreHalf && cy < g.y + kCentreHalf;
  const int64_t o = (int64_t)Q * g0 + i;
  cost[o] = (cls + kGiouWeight * (-giou)) + kPriorPenalty * ((in_box && in_ctr) ? 0.0f : 1.0f);
  iou_out[o] = iou;
  // bit 1: a box of the pair fails the reference's `(boxes[:, 2:] >= boxes[:, :2]).all()` (util/box_ops.py:76-77; NaN fails it too):
  // the reference aborts the step with an AssertionError; the assignment kernel turns the bit into status 4 of the image
  const bool degenerate = !(bx1 >= bx0) || !(by1 >= by0) || !(gx1 >= gx0) || !(gy1 >= gy0);
  flags[o] = (uint8_t)(((in_box || in_ctr) ? 1 : 0) | (degenerate ? 2 : 0));
}

// ---- wave-level (value, index) reductions: the smallest / largest value, lowest index among equals ---------------------
__device__ __forceinline__ void wave_argmin(float& v, int& i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(i, o, 64);
    if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(i, o, 64);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}
// NaN costs (0 / 0 IoU of two zero-area boxes).  PyTorch's top-k sorts NaN behind everything (nan_to_inf); its min / argmin
// PROPAGATE NaN -- the first NaN of a row or column is "the minimum" (nan_first; a real -inf beside a NaN ties with it here and
// the lower index wins: the one place this file does not follow PyTorch, on inputs no detector produces).
__device__ __forceinline__ float nan_to_inf(float v) { return v != v ? INFINITY : v; }
__device__ __forceinline__ float nan_first(float v) { return v != v ? -INFINITY : v; }

constexpr int kOT = 1024, kOW = kOT / 64;
constexpr int kMaxGt = 4096;                    // targets of one image (unmatched flags live in LDS)
constexpr uint8_t kMatch = 1, kStale = 2;       // bit 0 of matching[q, g]; bit 1 of matching[q, 0]: the row was multiply claimed before the repair loop

__global__ void __launch_bounds__(kOT)
ota_dynamic_k_kernel(float* __restrict__ cost, const float* __restrict__ iou, const uint8_t* __restrict__ flags,
                     uint8_t* __restrict__ matching, GtOffsets go, int Q, int max_rounds, int64_t* __restrict__ sel_query,
                     int64_t* __restrict__ sel_gt, int64_t* __restrict__ matched_query, int32_t* __restrict__ num_selected,
                     int32_t* __restrict__ status) {
  __shared__ uint8_t s_unmatched[kMaxGt];
  __shared__ int s_count[2];                    // [0] unmatched targets of this round, [1] rows holding more than one target
  __shared__ int s_wave[kOW];
  const int b = blockIdx.x;
  const int g0 = go.off[b], G = go.off[b + 1] - g0;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (G <= 0) {
    if (tid == 0) { num_selected[b] = 0; status[b] = 0; }
    return;
  }
  float* const C = cost + (int64_t)Q * g0;
  const float* const I = iou + (int64_t)Q * g0;
  const uint8_t* const F = flags + (int64_t)Q * g0;
  uint8_t* const M = matching + (int64_t)Q * g0;

  // a row's cheapest target (torch.min(cost[rows], dim=1): first minimum) replaces everything the row holds
  auto keep_cheapest = [&](int q) {
    float best = nan_first(C[(int64_t)q * G]);
    int arg = 0;
    for (int g = 1; g < G; ++g) {
      const float v = nan_first(C[(int64_t)q * G + g]);
      if (v < best) { best = v; arg = g; }
    }
    const uint8_t stale = M[(int64_t)q * G] & kStale;
    for (int g = 0; g < G; ++g) M[(int64_t)q * G + g] = 0;
    M[(int64_t)q * G + arg] = kMatch;
    M[(int64_t)q * G] |= stale;
  };
  auto row_sum = [&](int q) {
    int n = 0;
    for (int g = 0; g < G; ++g) n += M[(int64_t)q * G + g] & kMatch;
    return n;
  };

  // ---- (1) background penalty (matcher.py:340; fg_mask of :374: inside ANY box or ANY centre square), matching = 0 ------
  int bad = 0;
  for (int q = tid; q < Q; q += kOT) {
    bool fg = false;
    for (int g = 0; g < G; ++g) {
      const uint8_t f = F[(int64_t)q * G + g];
      fg = fg || (f & 1) != 0;
      bad |= f & 2;
    }
    for (int g = 0; g < G; ++g) {
      if (!fg) C[(int64_t)q * G + g] = C[(int64_t)q * G + g] + kBgPenalty;
      M[(int64_t)q * G + g] = 0;
    }
  }
  const int degenerate = __syncthreads_or(bad) ? 4 : 0;      // (also the barrier behind this phase)

  // ---- (2) dynamic k per target and its k cheapest queries (matcher.py:390-402) -------------------------------------------
  const int ncand = min(Q, kTopIou);
  for (int g = wv; g < G; g += kOW) {
    // the ncand largest IoUs of the column, in descending order; their sum in that order
    float prev_v = INFINITY, sum = 0.0f;
    int prev_i = -1;
    bool has_nan = false;
    for (int r = 0; r < ncand; ++r) {
      float bv = -INFINITY;
      int bi = 0x7fffffff;
      for (int q = lane; q < Q; q += 64) {
        const float v = I[(int64_t)q * G + g];
        if (!(v == v)) { has_nan = true; continue; }           // (0 / 0 of two zero-area boxes; see k below)
        const bool after = v < prev_v || (v == prev_v && q > prev_i);
        if (after && (v > bv || (v == bv && q < bi))) { bv = v; bi = q; }
      }
      wave_argmax(bv, bi);
      if (bi == 0x7fffffff) break;
      sum = sum + bv;
      prev_v = bv; prev_i = bi;
    }
    // torch.clamp(topk_ious.sum(0).int(), min=1).  torch.topk ranks NaN LARGEST: a column with a NaN IoU has it among its ten,
    // the sum is NaN, .int() of NaN is 0 (GPU) or INT_MIN (CPU), the clamp makes it 1
    const int k = __any(has_nan) ? 1 : max((int)sum, 1);
    // the k cheapest queries of the column claim the target (torch.topk(cost[:, g], k, largest=False))
    float pv = -INFINITY;
    int pi = -1;
    for (int r = 0; r < k; ++r) {
      float bv = INFINITY;
      int bi = 0x7fffffff;
      for (int q = lane; q < Q; q += 64) {
        const float v = nan_to_inf(C[(int64_t)q * G + g]);
        const bool after = v > pv || (v == pv && q > pi);
        if (after && (v < bv || (v == bv && q < bi))) { bv = v; bi = q; }
      }
      wave_argmin(bv, bi);
      if (bi == 0x7fffffff) break;                              // k > Q cannot happen (k <= 10 <= ncand terms <= 1 each), kept for safety
      if (lane == 0) M[(int64_t)bi * G + g] = kMatch;
      pv = bv; pi = bi;
    }
  }
  __syncthreads();

  // ---- (3) a query claimed by several targets keeps its cheapest one; the set of such rows is remembered (matcher.py:406-411)
  for (int q = tid; q < Q; q += kOT) {
    if (row_sum(q) > 1) {
      keep_cheapest(q);
      M[(int64_t)q * G] |= kStale;
    }
  }
  __syncthreads();

  // ---- (4) repair loop (matcher.py:417-435) ---------------------------------------------------------------------------------
  int st = 0;
  for (int round = 0;; ++round) {
    if (tid < 2) s_count[tid] = 0;
    __syncthreads();
    for (int g = wv; g < G; g += kOW) {                        // targets without a query
      bool any = false;
      for (int q = lane; q < Q; q += 64) any = any || (M[(int64_t)q * G + g] & kMatch);
      const bool un = __ballot(any) == 0ull;
      if (lane == 0) {
        s_unmatched[g] = un ? 1 : 0;
        if (un) atomicAdd(&s_count[0], 1);
      }
    }
    __syncthreads();
    if (s_count[0] == 0) break;
    if (round >= max_rounds) { st = 2; break; }
    for (int q = tid; q < Q; q += kOT) {                       // cost[matched_query_id] += 100000.0
      if (row_sum(q) > 0)
        for (int g = 0; g < G; ++g) C[(int64_t)q * G + g] = C[(int64_t)q * G + g] + kTakenPenalty;
    }
    __syncthreads();
    for (int g = wv; g < G; g += kOW) {                        // every unmatched target takes its cheapest query (torch.argmin: first minimum)
      if (!s_unmatched[g]) continue;
      float bv = INFINITY;
      int bi = 0x7fffffff;
      for (int q = lane; q < Q; q += 64) {
        const float v = nan_first(C[(int64_t)q * G + g]);
        if (v < bv || (v == bv && q < bi)) { bv = v; bi = q; }
      }
      wave_argmin(bv, bi);
      if (lane == 0 && bi != 0x7fffffff) M[(int64_t)bi * G + g] |= kMatch;
    }
    __syncthreads();
    for (int q = tid; q < Q; q += kOT)
      if (row_sum(q) > 1) atomicAdd(&s_count[1], 1);
    __syncthreads();
    if (s_count[1] > 0) {
      // the reference indexes with the mask computed BEFORE the loop (`anchor_matching_gt > 1`, never refreshed): those rows,
      // and only those, are reset to their cheapest target -- whether or not they are the rows in conflict now
      for (int q = tid; q < Q; q += kOT)
        if (M[(int64_t)q * G] & kStale) keep_cheapest(q);
    }
    __syncthreads();
  }

  // ---- (5) results (matcher.py:437-447): selected queries ascending with the first target of their row ------------------
  int base = 0;
  for (int q0 = 0; q0 < Q; q0 += kOT) {
    const int q = q0 + tid;
    int first = -1;
    if (q < Q)
      for (int g = G - 1; g >= 0; --g)
        if (M[(int64_t)q * G + g] & kMatch) first = g;          // matching[selected].max(1)[1]: the first maximum
    const unsigned long long bal = __ballot(first >= 0);
    if (lane == 0) s_wave[wv] = __popcll(bal);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < kOW; ++w) {
      const int c = s_wave[w];
      if (w < wv) before += c;
      total += c;
    }
    if (first >= 0) {
      const int pos = base + before + __popcll(bal & ((1ull << lane) - 1ull));
      sel_query[(int64_t)b * Q + pos] = q;
      sel_gt[(int64_t)b * Q + pos] = first;
    }
    base += total;
    __syncthreads();
  }
  // per target the cheapest query among its own (cost[matching == 0] += inf; torch.min(cost, dim=0)[1])
  for (int g = wv; g < G; g += kOW) {
    float bv = INFINITY;
    int bi = 0x7fffffff;
    for (int q = lane; q < Q; q += 64) {
      // (an entry outside the matching is cost + inf: +inf -- or NaN, which torch.min then returns even though it is not matched)
      const float c = C[(int64_t)q * G + g];
      const float v = nan_first((M[(int64_t)q * G + g] & kMatch) ? c : c + INFINITY);
      if (v < bv || (v == bv && q < bi)) { bv = v; bi = q; }
    }
    wave_argmin(bv, bi);
    if (lane == 0) matched_query[g0 + g] = bi == 0x7fffffff ? 0 : bi;   // (a column of +inf only: torch's argmin of all-inf is 0)
  }
  if (tid == 0) { num_selected[b] = base; status[b] = st | degenerate; }
}

}  // namespace
}  // namespace msda

extern "C" int dynmask_set_error(int code, const char* what);   // msda_capi.hip (shared last-error slot)

namespace {
int fill_offsets(const int32_t* gt_off, int batch, msda::GtOffsets* go, int* max_g, const char* who) {
  if (batch < 0 || batch > OTA_HIP_MAX_BATCH) return dynmask_set_error(OTA_ERR_BAD_DIMS, "ota: batch out of range (<= OTA_HIP_MAX_BATCH)");
  if (batch > 0 && !gt_off) return dynmask_set_error(OTA_ERR_NULL_POINTER, "ota: gt_off is null");
  *max_g = 0;
  for (int b = 0; b <= batch; ++b) go->off[b] = batch ? gt_off[b] : 0;
  if (batch && go->off[0] != 0) return dynmask_set_error(OTA_ERR_BAD_DIMS, "ota: gt_off[0] must be 0");
  for (int b = 0; b < batch; ++b) {
    const int g = go->off[b + 1] - go->off[b];
    if (g < 0) return dynmask_set_error(OTA_ERR_BAD_DIMS, "ota: gt_off must be non-decreasing");
    if (g > *max_g) *max_g = g;
  }
  (void)who;
  return 0;
}
}  // namespace

extern "C" int ota_cost_hip_f32(const float* class_table, const float* boxes, const float* tgt_boxes, const uint8_t* positive_map,
                                const int32_t* gt_off, int batch, int num_queries, int num_tokens, float* cost, float* iou,
                                uint8_t* flags, void* stream) {
  msda::GtOffsets go;
  int max_g = 0;
  if (num_queries < 0 || num_tokens < 0) return dynmask_set_error(OTA_ERR_BAD_DIMS, "ota_cost_hip_f32: negative dimension");
  if (int rc = fill_offsets(gt_off, batch, &go, &max_g, "ota_cost_hip_f32")) return rc;
  if (batch == 0 || num_queries == 0 || max_g == 0) return 0;
  if (!class_table || !boxes || !tgt_boxes || !positive_map || !cost || !iou || !flags)
    return dynmask_set_error(OTA_ERR_NULL_POINTER, "ota_cost_hip_f32: null pointer");
  const int64_t n = (int64_t)num_queries * max_g, blocks = (n + 255) / 256;
  if (blocks > 0x7fffffffLL || (int64_t)num_queries * go.off[batch] > 0x7fffffffLL)
    return dynmask_set_error(OTA_ERR_BAD_DIMS, "ota_cost_hip_f32: num_queries * targets too large");
  hipLaunchKernelGGL(msda::ota_cost_kernel, dim3((unsigned)blocks, (unsigned)batch), dim3(256), 0, static_cast<hipStream_t>(stream),
                     class_table, boxes, tgt_boxes, positive_map, go, num_queries, num_tokens, cost, iou, flags);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : dynmask_set_error((int)e, hipGetErrorString(e));
}

extern "C" int ota_dynamic_k_hip(float* cost, const float* iou, const uint8_t* flags, uint8_t* matching, const int32_t* gt_off,
                                 int batch, int num_queries, int max_rounds, int64_t* sel_query, int64_t* sel_gt,
                                 int64_t* matched_query, int32_t* num_selected, int32_t* status, void* stream) {
  msda::GtOffsets go;
  int max_g = 0;
  if (num_queries < 0) return dynmask_set_error(OTA_ERR_BAD_DIMS, "ota_dynamic_k_hip: negative dimension");
  if (int rc = fill_offsets(gt_off, batch, &go, &max_g, "ota_dynamic_k_hip")) return rc;
  if (batch == 0) return 0;
  if (max_g > msda::kMaxGt) return dynmask_set_error(OTA_ERR_BAD_DIMS, "ota_dynamic_k_hip: more than 4096 targets in one image");
  if (!num_selected || !status) return dynmask_set_error(OTA_ERR_NULL_POINTER, "ota_dynamic_k_hip: null pointer");
  if (max_g > 0 && num_queries > 0 && (!cost || !iou || !flags || !matching || !sel_query || !sel_gt || !matched_query))
    return dynmask_set_error(OTA_ERR_NULL_POINTER, "ota_dynamic_k_hip: null pointer");
  if (num_queries == 0) max_g = 0;
  msda::GtOffsets run = go;
  if (max_g == 0)                      // nothing to assign anywhere: the kernel only writes the zero counts
    for (int b = 0; b <= batch; ++b) run.off[b] = 0;
  hipLaunchKernelGGL(msda::ota_dynamic_k_kernel, dim3((unsigned)batch), dim3(msda::kOT), 0, static_cast<hipStream_t>(stream), cost,
                     iou, flags, matching, run, num_queries, max_rounds, sel_query, sel_gt, matched_query, num_selected, status);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : dynmask_set_error((int)e, hipGetErrorString(e));
}
