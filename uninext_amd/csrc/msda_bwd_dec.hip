// msda_bwd_dec -- MSDeformAttn backward for decoder-style calls (few queries, many pixels): fp32, D = 32, L = P = 4.
// gfx950 only.  Replaces, for these calls, the work of ops/src/cuda/ms_deform_im2col_cuda.cuh:301-403 / :406-920.
//
// The decoder backward is bound by its value-gradient atomics: 1100 queries x 8 heads x 16 samples x 4 corners = one
// full-line L2 atomic each (msda_bwd_generic: 1.13 M per call, 116 us).  Half of them go to the two COARSE levels, whose
// pixels are few (R50: 1050 + 273 per head) and hit again and again.  Here a workgroup owns (image, head, slice of the
// queries) and keeps int32 accumulators for as many whole rows of level 3, then level 2, as fit 150 KB of LDS: the
// corner adds of those rows are ds_add_u32 (fixed point with a per-workgroup power-of-two scale from a bound that cannot
// overflow, as msda_bwd_tiled), and every touched accumulator row leaves once, as one full-line float atomic.  Levels 0
// and 1 (and rows that did not fit) take the direct atomics of msda_bwd_generic; grad_sampling_loc / grad_attn_weight
// are computed as there (half a wave per pair, lane = channel, DPP sums), every element written once.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "msda_common.hpp"

namespace msda {
namespace {

constexpr int MSDA_BWD_DEC_THREADS = 1024;
constexpr int kDT = MSDA_BWD_DEC_THREADS;                                  // threads per workgroup = 32 half-waves = 32 pairs in flight
constexpr int MSDA_BWD_DEC_SLOTS = 1200;
constexpr int kAccSlots = MSDA_BWD_DEC_SLOTS;              // accumulator slots (pixels) of 32 int32: 150 KB
struct DMeta { unsigned gmax_bits, amax_bits; };
constexpr int kDecLds = kAccSlots * 128 + 16;

__device__ __forceinline__ float half_sum(float f) {       // over the 32 lanes of a half wave; every lane gets the total
  f += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(f), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  f += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(f), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  f += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(f), 0x141, 0xF, 0xF, true));   // row_half_mirror
  f += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(f), 0x140, 0xF, 0xF, true));   // row_mirror
  f += __shfl_xor(f, 16, 64);
  return f;
}
__device__ __forceinline__ float abs_or_inf(float v) {     // |v|, +inf for NaN: non-finite inputs must reach the bound
  const float a = fabsf(v);
  return a == a ? a : __builtin_inff();
}

}  // namespace

__global__ void __launch_bounds__(kDT)
msda_bwd_dec(const float* __restrict__ grad_out, const float* __restrict__ value, const int64_t* __restrict__ shapes,
             const int64_t* __restrict__ lsi, const float* __restrict__ loc, const float* __restrict__ attn, Dims d,
             float* __restrict__ grad_value, float* __restrict__ grad_loc, float* __restrict__ grad_attn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* const acc = reinterpret_cast<int*>(smem);                         // [kAccSlots][32]
  DMeta& mt = *reinterpret_cast<DMeta*>(smem + kAccSlots * 128);
  const int tid = threadIdx.x, lane = tid & 31, hw = tid >> 5;           // half wave hw of 32
  const int m = blockIdx.x, sl = blockIdx.y, nsl = gridDim.y, b = blockIdx.z;
  const int M = d.M;
  int H[4], W[4], S0[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) { H[l] = (int)shapes[2 * l]; W[l] = (int)shapes[2 * l + 1]; S0[l] = (int)lsi[l]; }
  // whole rows of level 3, then of level 2, that fit the accumulator slots (rows 0 .. rows - 1 of each)
  const int rows3 = min(H[3], kAccSlots / max(W[3], 1));
  const int base2 = rows3 * W[3];
  const int rows2 = min(H[2], (kAccSlots - base2) / max(W[2], 1));
  const int nslots = base2 + rows2 * W[2];
  // this workgroup's queries
  const int qper = (d.Lq + nsl - 1) / nsl;
  const int q0 = sl * qper, q1 = min(d.Lq, q0 + qper);

  for (int o = tid * 4; o < nslots * 32; o += kDT * 4) *reinterpret_cast<int4*>(acc + o) = make_int4(0, 0, 0, 0);
  if (tid == 0) { mt.gmax_bits = 0u; mt.amax_bits = 0u; }
  __syncthreads();
  // ---- the bound of the fixed-point scale: max |grad_out| and max |attn| over the workgroup's pairs -----------------------
  {
    float gm = 0.f, am = 0.f;
    for (int q = q0 + hw; q < q1; q += kDT / 32) {
      const int64_t pair = ((int64_t)b * d.Lq + q) * M + m;
      gm = fmaxf(gm, abs_or_inf(grad_out[pair * 32 + lane]));
      if (lane < 16) am = fmaxf(am, abs_or_inf(attn[pair * 16 + lane]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { gm = fmaxf(gm, __shfl_xor(gm, o, 64)); am = fmaxf(am, __shfl_xor(am, o, 64)); }
    if ((tid & 63) == 0) {   // non-negative floats order like their bit patterns
      atomicMax(&mt.gmax_bits, __float_as_uint(gm));
      atomicMax(&mt.amax_bits, __float_as_uint(am));
    }
  }
  __syncthreads();
  // an accumulator receives at most one corner of each level-l sample of the slice: (q1 - q0) * 4 adds of at most
  // max |grad_out| * max |attn| each (bilinear weights <= 1)
  const float bound = (float)(max(q1 - q0, 1) * 4) * __uint_as_float(mt.gmax_bits) * __uint_as_float(mt.amax_bits);
  const bool use_lds = bound < 0x1p120f && nslots > 0;   // false for NaN / Inf and for bounds the clamped exponent below cannot scale into int32 (>= 2^120): float atomics
  float scale = 1.f, inv_scale = 1.f;
  if (use_lds && bound > 0.f) {
    int e;
    (void)frexpf(bound, &e);                                             // bound < 2^e
    e = max(-90, min(90, 30 - e));
    scale = ldexpf(1.f, e);
    inv_scale = ldexpf(1.f, -e);
  }

  const int64_t pix_stride = (int64_t)M * 32;
  // Work unit = (query, level): a half wave takes the slice's units u = hw, hw + 32, ... in order (u = 4 x query + level).
  // Round 4 walked a half wave through the FOUR levels of its query -- 69 queries on 32 half waves = 3 queries x 4 dependent
  // level steps for the slowest half wave, 12 memory round trips in a chain that the launch waits for (66.8 % of the wave
  // cycles waiting, profiles/r04_sq_pmc.txt).  Units balance the same work to ceil(276 / 32) = 9 steps.  A unit's 8 location
  // floats, 4 attention weights and 32 upstream gradients arrive LANE-PARALLEL (one load each per lane), one unit AHEAD of
  // their use, so that a step is ONE memory round trip: the 16 corner loads.
  const int nunits = (q1 - q0) * 4;
  const uint32_t ps32 = (uint32_t)M * 32u;
  const __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(value), 0, (int)((uint32_t)d.N * (uint32_t)d.S * ps32 * 4u), 0x00020000);
  float locv_n = 0.f, attv_n = 0.f, g_n = 0.f;             // the NEXT unit's
  auto fetch_unit = [&](int u) __attribute__((always_inline)) {
    if (u < nunits) {
      const int64_t pr = ((int64_t)b * d.Lq + q0 + (u >> 2)) * M + m;
      const int l = u & 3;
      locv_n = loc[pr * 32 + l * 8 + (lane & 7)];
      attv_n = attn[pr * 16 + l * 4 + (lane & 3)];
      g_n = grad_out[pr * 32 + lane];
    } else {
      locv_n = 0.f; attv_n = 0.f; g_n = 0.f;
    }
  };
  fetch_unit(hw);
  for (int u = hw; ; u += kDT / 32) {
    // the two halves of a wave run in lock step: a half past the end idles through the loop with `live` off
    const bool live = u < nunits;
    if (!__ballot(live)) break;
    const int l = u & 3;                                     // (per HALF wave: the halves of a wave work on two levels)
    const bool l0 = (l & 1) != 0, l1 = (l & 2) != 0;
    const int Hl = l1 ? (l0 ? H[3] : H[2]) : (l0 ? H[1] : H[0]), Wl = l1 ? (l0 ? W[3] : W[2]) : (l0 ? W[1] : W[0]);
    const int Sl = l1 ? (l0 ? S0[3] : S0[2]) : (l0 ? S0[1] : S0[0]);
    const int64_t pair = ((int64_t)b * d.Lq + q0 + (live ? (u >> 2) : 0)) * M + m;
    const float locv = locv_n, attv = attv_n, g = g_n;
    fetch_unit(u + kDT / 32);
    const uint32_t lvl_off = ((uint32_t)b * (uint32_t)d.S + (uint32_t)Sl) * ps32 + (uint32_t)m * 32u + (uint32_t)lane;
    // accumulator rows of this level (none on levels 0 and 1)
    const int rows_l = use_lds ? (l == 3 ? rows3 : l == 2 ? rows2 : 0) : 0;
    const int slot0 = l == 3 ? 0 : base2;
    // ---- issue: the four samples of the unit from the broadcast locations, their 16 corner loads ------------------------
    float lh[4], lw[4], a[4], v[4][4];
    int h_low[4], w_low[4];
    bool in[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // (half-wave broadcasts: lanes 2 k, 2 k + 1 of the half hold point k's x, y; lane k its weight)
      a[k] = __shfl(attv, k, 32);
      const Sample<float> t = make_sample<float>(__shfl(locv, 2 * k, 32), __shfl(locv, 2 * k + 1, 32), Hl, Wl);
      in[k] = live && t.in_range;
      lh[k] = t.lh; lw[k] = t.lw; h_low[k] = t.h_low; w_low[k] = t.w_low;
      // raw buffer loads: one 32-bit byte offset per corner, dead corners at an out-of-range offset (they return 0)
      const uint32_t o1 = (lvl_off + (uint32_t)(t.h_low * Wl + t.w_low) * ps32) * 4u;   // (h_low, w_low = 0 for samples out of range)
      const uint32_t rowb = (uint32_t)Wl * ps32 * 4u, pxb = ps32 * 4u;
      v[k][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vsrc, (in[k] && t.ok1) ? o1 : kOobOffset, 0, 0));
      v[k][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vsrc, (in[k] && t.ok2) ? o1 + pxb : kOobOffset, 0, 0));
      v[k][2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vsrc, (in[k] && t.ok3) ? o1 + rowb : kOobOffset, 0, 0));
      v[k][3] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vsrc, (in[k] && t.ok4) ? o1 + rowb + pxb : kOobOffset, 0, 0));
    }
    // ---- consume: the gradients and the adds ----------------------------------------------------------------------------
    float* const ga_p = grad_attn + pair * 16 + l * 4;        // this unit's 4 + 8 outputs
    float* const gl_p = grad_loc + pair * 32 + l * 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float pa = 0.f, pw = 0.f, ph = 0.f;
      if (__ballot(in[k])) {                                 // wave-uniform (the sums need every lane of the half)
        if (in[k]) {
          const float hh = 1.f - lh[k], hw_ = 1.f - lw[k];
          const bool tp = h_low[k] >= 0, bt = h_low[k] + 1 <= Hl - 1, lf = w_low[k] >= 0, rt = w_low[k] + 1 <= Wl - 1;   // (make_sample's rule)
          const bool ok1 = tp && lf, ok2 = tp && rt, ok3 = bt && lf, ok4 = bt && rt;
          const float tgv = g * a[k];
          const float w1 = hh * hw_, w2 = hh * lw[k], w3 = lh[k] * hw_, w4 = lh[k] * lw[k];
          // (all four in 32-bit arithmetic: with h_low or w_low = -1 the top-left offset wraps and its neighbours wrap back)
          const uint32_t o1 = lvl_off + (uint32_t)(h_low[k] * Wl + w_low[k]) * ps32, o2 = o1 + ps32, o3 = o1 + (uint32_t)Wl * ps32, o4 = o3 + ps32;
          // a corner on an accumulator row: fixed-point LDS add; otherwise the direct full-line atomic
          const bool top_acc = h_low[k] >= 0 && h_low[k] < rows_l, bot_acc = h_low[k] + 1 < rows_l;
          int* const at = acc + (slot0 + h_low[k] * Wl + w_low[k]) * 32 + lane;
          if (ok1) { if (top_acc) atomicAdd(at, __float2int_rn(w1 * tgv * scale)); else atomic_add(grad_value + o1, w1 * tgv); }
          if (ok2) { if (top_acc) atomicAdd(at + 32, __float2int_rn(w2 * tgv * scale)); else atomic_add(grad_value + o2, w2 * tgv); }
          if (ok3) { if (bot_acc) atomicAdd(at + Wl * 32, __float2int_rn(w3 * tgv * scale)); else atomic_add(grad_value + o3, w3 * tgv); }
          if (ok4) { if (bot_acc) atomicAdd(at + Wl * 32 + 32, __float2int_rn(w4 * tgv * scale)); else atomic_add(grad_value + o4, w4 * tgv); }
          pa = g * (w1 * v[k][0] + w2 * v[k][1] + w3 * v[k][2] + w4 * v[k][3]);
          pw = tgv * (hh * (v[k][1] - v[k][0]) + lh[k] * (v[k][3] - v[k][2]));
          ph = tgv * (hw_ * (v[k][2] - v[k][0]) + lw[k] * (v[k][3] - v[k][1]));
        }
        pa = half_sum(pa);
        pw = half_sum(pw);
        ph = half_sum(ph);
      }
      if (live && lane == 0) {
        ga_p[k] = pa;
        gl_p[2 * k] = (float)Wl * pw;
        gl_p[2 * k + 1] = (float)Hl * ph;
      }
    }
  }
  __syncthreads();
  // ---- flush: every touched accumulator row of 32 channels leaves as one full-line float atomic --------------------------
  if (use_lds) {
    for (int slot = hw; slot < nslots; slot += kDT / 32) {
      const int raw = acc[slot * 32 + lane];
      const unsigned long long any = __ballot(raw != 0) >> (tid & 32) & 0xffffffffull;
      if (any) {
        const int pix = slot < base2 ? S0[3] + slot : S0[2] + (slot - base2);
        atomic_add(grad_value + ((int64_t)b * d.S + pix) * pix_stride + (int64_t)m * 32 + lane, (float)raw * inv_scale);
      }
    }
  }
}

// Slices of the queries per (image, head): enough workgroups to cover the chip once, at most 16 (every slice flushes its own
// accumulators) -- and at most kDecSliceQueries queries per slice, which is what the documented step of the fixed-point
// accumulators rests on (include/msda_hip.h: the scale of a slice comes from 4 x its queries x max |grad_out| x max |attn|; a
// coarse level's pixel collects a rounding per add, and thousands of queries per slice would push their sum past 1e-4).
// MSDA_BWD_DEC_SLICES=n: A/B switch for the first rule.
constexpr int kDecSliceQueries = 256, kDecMaxSlices = 64;
static int dec_slices(const Dims& d) {
  static const int env = ab_env_int("MSDA_BWD_DEC_SLICES", 0);
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int64_t heads = std::max<int64_t>(1, (int64_t)d.M * d.N);
  int nsl = env > 0 ? env : (int)std::min<int64_t>(16, (cus + heads - 1) / heads);
  nsl = std::max(1, std::min(nsl, env > 0 ? kDecMaxSlices : 16));
  nsl = std::min(nsl, (d.Lq + 31) / 32);
  return std::max(nsl, (d.Lq + kDecSliceQueries - 1) / kDecSliceQueries);
}

bool dec_backward_ok(const Dims& d) {
  return d.D == 32 && d.L == 4 && d.P == 4 && d.Lq >= 64 && d.M <= 65535 && d.N <= 65535 && dec_slices(d) <= kDecMaxSlices &&
         (int64_t)d.N * d.S * d.M * 128 < (int64_t)kOobOffset;   // (32-bit byte offsets into value / grad_value)
}

int launch_backward_dec(const float* grad_out, const float* value, const int64_t* shapes, const int64_t* lsi,
                        const float* loc, const float* attn, const Dims& d, float* grad_value, float* grad_loc,
                        float* grad_attn, hipStream_t stream) {
  static std::atomic<uint64_t> lds_opted_in{0};
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(msda_bwd_dec), kDecLds, lds_opted_in)) return rc;
  const int nsl = dec_slices(d);
  hipLaunchKernelGGL(msda_bwd_dec, dim3((unsigned)d.M, (unsigned)nsl, (unsigned)d.N), dim3(kDT), kDecLds, stream, grad_out, value,
                     shapes, lsi, loc, attn, d, grad_value, grad_loc, grad_attn);
  return (int)hipGetLastError();
}

}  // namespace msda
