// msda_bwd_q -- the QUERY side of the MSDeformAttn backward alone: grad_sampling_loc and grad_attn_weight (cuh:113-158), nothing
// that concerns grad_value.  fp32, D = 32, L = P = 4, encoder-sized calls.  gfx950 only.
//
// msda_bwd_regions sums grad_value on the destination side and needs the other two gradients from a query-side pass.  That
// pass was msda_bwd_tiled with its grad_value half compiled out (msda_bwd_tiled_nogv: 222 us per encoder call -- LDS tiles,
// placement, sample records of a kernel built around accumulators it no longer has).  This one is the forward gather kernel
// msda_fwd_lg3 (msda_fwd.hip) run for gradients: 8 lanes per (query, head) pair, lane j = the pair's channels 4 j .. 4 j + 3 and
// the preparer of samples j and 8 + j; levels 0..2 through the L1, level 3 from a copy in LDS; the sample's three sums are formed
// per corner row (linear form: A_r = sum_c g_c (left_c + lw (right_c - left_c)), D_r = sum_c g_c (right_c - left_c)), reduced
// over the pair's 8 lanes with three DPP steps each and kept by the lane that prepared the sample.
//   768-thread workgroups (96 pairs), 62 KB of LDS, two per CU; 45 registers.  512 / 768 / 1024 threads measure the same
//   (MSDA_BWD_Q_THREADS).  msda_bwd_regions with this pass instead of msda_bwd_tiled_nogv: 692 -> 644 us (`wide`), 893 -> 826
//   (`uniform`), 766 -> 739 (`model`); 22 / 22 cases of tools/bwin_check.py (profiles/r04_backward_regions_q.txt).
#include <cstdlib>

#include "msda_common.hpp"

namespace msda {
namespace {

constexpr int MSDA_BWD_Q_THREADS = 768;
constexpr int kQT = MSDA_BWD_Q_THREADS, kQG = 8, kQPairs = kQT / kQG;
constexpr int kQSlots = 280;                                     // resident pixels of the last level (35 KB) + one all-zero slot
constexpr int kQRecPair = 8 * 32 + 16;                           // eight 32-byte sample records per pair (+ 16: bank skew)
constexpr int kQLdsBytes = kLevelTableBytes + (kQSlots + 1) * 128 + kQPairs * kQRecPair;
static_assert(kQLdsBytes <= 80 * 1024, "two workgroups per CU");

typedef float v2f __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float qdpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float group8_sum(float v) {   // over the 8 lanes of a pair; every lane gets the total
  v += qdpp<0xB1>(v);                                    // quad_perm [1,0,3,2]
  v += qdpp<0x4E>(v);                                    // quad_perm [2,3,0,1]
  v += qdpp<0x141>(v);                                   // row_half_mirror
  return v;
}

}  // namespace

__global__ void __launch_bounds__(kQT, kQT == 1024 ? 8 : 6)
msda_bwd_q(const float* __restrict__ grad_out, const float* __restrict__ value, const int64_t* __restrict__ shapes,
           const int64_t* __restrict__ lsi, const float* __restrict__ loc, const float* __restrict__ attn, Dims d,
           float* __restrict__ grad_loc, float* __restrict__ grad_attn) {
  constexpr int LPT = 16, P = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* smp_H = reinterpret_cast<int*>(smem);
  int* smp_W = smp_H + kMaxLP;
  int* smp_start = smp_W + kMaxLP;
  char* cl_base = smem + kLevelTableBytes;
  char* rec_base = cl_base + (kQSlots + 1) * 128;
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x;
  if (tid < LPT) {
    const int l = tid / P;
    smp_H[tid] = (int)shapes[2 * l];
    smp_W[tid] = (int)shapes[2 * l + 1];
    smp_start[tid] = (int)lsi[l];
  }
  __syncthreads();
  const int res_pix0 = __builtin_amdgcn_readfirstlane(smp_start[LPT - 1]);   // first pixel of the last level
  const int nres_all = d.S - res_pix0;
  const bool fits = nres_all <= kQSlots;           // uniform; otherwise the last level also takes the L1 path
  const int nres = fits ? nres_all : 0;

  const int b = blockIdx.y;
  const int m = blockIdx.x % d.M;
  const uint32_t pix_bytes = (uint32_t)d.M * 128u;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(value) + (int64_t)b * d.S * d.M * 32, 0, (int)((uint32_t)d.S * (uint32_t)d.M * 128u), 0x00020000);
  const uint32_t head_off = (uint32_t)m * 128u;
  for (int i = tid >> 3; i <= nres && fits; i += kQT / 8) {    // slot `nres` stays zero: dead corners read it
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (i < nres) v = buffer_load_f32x4(rsrc, (uint32_t)(res_pix0 + i) * pix_bytes + (uint32_t)(tid & 7) * 16u, head_off);
    *reinterpret_cast<f32x4*>(cl_base + i * 128 + (tid & 7) * 16) = v;
  }

  const int g = tid / kQG, j = tid % kQG;
  const int q = (blockIdx.x / d.M) * kQPairs + g;
  const bool live = q < d.Lq;
  const int64_t pair = ((int64_t)b * d.Lq + (live ? q : 0)) * d.M + m;
  char* rec = rec_base + g * kQRecPair;
  const uint32_t lane_off = (uint32_t)j * 16u;
  const uint32_t cl_addr0 = smem_base + kLevelTableBytes;
  const uint32_t zero_slot = cl_addr0 + (uint32_t)nres * 128u;

  // lane j prepares sample j of pass 0 (levels 0, 1) and sample 8 + j of pass 1 (levels 2, 3), and keeps their gradients
  float2 lc0 = make_float2(0.f, 0.f), lc1 = make_float2(0.f, 0.f);
  float at0 = 0.f, at1 = 0.f;
  f32x4 go = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    lc0 = *reinterpret_cast<const float2*>(loc + pair * (2 * LPT) + 2 * j);
    lc1 = *reinterpret_cast<const float2*>(loc + pair * (2 * LPT) + 2 * (8 + j));
    at0 = attn[pair * LPT + j];
    at1 = attn[pair * LPT + 8 + j];
    go = *reinterpret_cast<const f32x4*>(grad_out + pair * 32 + 4 * j);
  }
  // record of a sample: lw, lh, a W, a H (what turns the reduced d/dx, d/dy sums into grad_sampling_loc, cuh:157-158) and the
  // four corner offsets -- dead corners (and every corner of an out-of-range sample) at an out-of-range offset / the zero slot:
  // they read 0, which is what the reference's `if (h_low >= 0 && ...)` guards amount to in the linear form
  auto prepare = [&](int s, int slot, float lx, float ly, float a, bool resident) {
    const int H = smp_H[s], W = smp_W[s];
    const Sample<float> sm = make_sample<float>(lx, ly, H, W);
    const bool in = live && sm.in_range;
    const int pix1 = smp_start[s] + sm.h_low * W + sm.w_low;
    u32x4 o;
    if (resident) {
      const uint32_t a1 = cl_addr0 + (uint32_t)(pix1 - res_pix0) * 128u;
      o[0] = (in && sm.ok1) ? a1 : zero_slot;
      o[1] = (in && sm.ok2) ? a1 + 128u : zero_slot;
      o[2] = (in && sm.ok3) ? a1 + (uint32_t)W * 128u : zero_slot;
      o[3] = (in && sm.ok4) ? a1 + (uint32_t)(W + 1) * 128u : zero_slot;
    } else {
      const uint32_t o1 = (uint32_t)pix1 * pix_bytes;
      o[0] = (in && sm.ok1) ? o1 : kOobOffset;
      o[1] = (in && sm.ok2) ? o1 + pix_bytes : kOobOffset;
      o[2] = (in && sm.ok3) ? o1 + (uint32_t)W * pix_bytes : kOobOffset;
      o[3] = (in && sm.ok4) ? o1 + (uint32_t)(W + 1) * pix_bytes : kOobOffset;
    }
    // (poisoned locations: lw / lh of an out-of-range sample may be NaN; its corners read 0 and 0 x NaN would be NaN)
    const float lw = in ? sm.lw : 0.f, lh = in ? sm.lh : 0.f, aa = in ? a : 0.f;
    *reinterpret_cast<float4*>(rec + slot * 32) = make_float4(lw, lh, aa * (float)W, aa * (float)H);
    *reinterpret_cast<u32x4*>(rec + slot * 32 + 16) = o;
  };
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  float ga = 0.f, gx = 0.f, gy = 0.f;                          // gradients of the sample this lane prepared in the current pass
  auto finish = [&](int slot, const float4 w, const f32x4 r1, const f32x4 r2, const f32x4 r3, const f32x4 r4) {
    const float lw = w.x, lh = w.y, hh = 1.f - lh;
    const v2f LW = {lw, lw};
    v2f At = {0.f, 0.f}, Ab = {0.f, 0.f}, Dt = {0.f, 0.f}, Db = {0.f, 0.f};
#pragma unroll
    for (int cp = 0; cp < 2; ++cp) {
      const v2f V1 = {r1[2 * cp], r1[2 * cp + 1]}, V2 = {r2[2 * cp], r2[2 * cp + 1]};
      const v2f V3 = {r3[2 * cp], r3[2 * cp + 1]}, V4 = {r4[2 * cp], r4[2 * cp + 1]};
      const v2f G = {go[2 * cp], go[2 * cp + 1]};
      const v2f tt = V2 - V1, tb = V4 - V3;
      const v2f top = __builtin_elementwise_fma(LW, tt, V1), bot = __builtin_elementwise_fma(LW, tb, V3);
      At = __builtin_elementwise_fma(G, top, At);
      Ab = __builtin_elementwise_fma(G, bot, Ab);
      Dt = __builtin_elementwise_fma(G, tt, Dt);
      Db = __builtin_elementwise_fma(G, tb, Db);
    }
    const float at_ = At.x + At.y, ab_ = Ab.x + Ab.y, dt_ = Dt.x + Dt.y, db_ = Db.x + Db.y;
    const float ra = group8_sum(fmaf(lh, ab_, hh * at_));
    const float rw = group8_sum(fmaf(lh, db_, hh * dt_)) * w.z;
    const float rh = group8_sum(ab_ - at_) * w.w;
    const bool mine = j == slot;
    ga = mine ? ra : ga; gx = mine ? rw : gx; gy = mine ? rh : gy;
  };
  auto gather_global = [&](int slot) {
    const float4 w = *reinterpret_cast<const float4*>(rec + slot * 32);
    const u32x4 o = *reinterpret_cast<const u32x4*>(rec + slot * 32 + 16);
    const f32x4 r1 = buffer_load_f32x4(rsrc, o[0] + lane_off, head_off);
    const f32x4 r2 = buffer_load_f32x4(rsrc, o[1] + lane_off, head_off);
    const f32x4 r3 = buffer_load_f32x4(rsrc, o[2] + lane_off, head_off);
    const f32x4 r4 = buffer_load_f32x4(rsrc, o[3] + lane_off, head_off);
    finish(slot, w, r1, r2, r3, r4);
  };
  auto gather_lds = [&](int slot) {
    typedef const f32x4 __attribute__((address_space(3)))* lp;
    const float4 w = *reinterpret_cast<const float4*>(rec + slot * 32);
    const u32x4 o = *reinterpret_cast<const u32x4*>(rec + slot * 32 + 16);
    const f32x4 r1 = *reinterpret_cast<lp>((uintptr_t)(o[0] + lane_off));
    const f32x4 r2 = *reinterpret_cast<lp>((uintptr_t)(o[1] + lane_off));
    const f32x4 r3 = *reinterpret_cast<lp>((uintptr_t)(o[2] + lane_off));
    const f32x4 r4 = *reinterpret_cast<lp>((uintptr_t)(o[3] + lane_off));
    finish(slot, w, r1, r2, r3, r4);
  };
  auto store = [&](int s) {                                    // sample s of the pair: one weight gradient, one (x, y) pair
    if (live) {
      grad_attn[pair * LPT + s] = ga;
      *reinterpret_cast<float2*>(grad_loc + pair * (2 * LPT) + 2 * s) = make_float2(gx, gy);
    }
  };

  // pass 0: samples 0..7 (levels 0 and 1), all through the L1 path; overlaps the other waves' staging
  prepare(j, j, lc0.x, lc0.y, at0, false);
  wave_sync();
#pragma unroll
  for (int s = 0; s < 8; ++s) gather_global(s);
  store(j);
  wave_sync();
  // pass 1: samples 8..15 (levels 2 and 3); level 3 (slots 4..7) comes from the LDS copy
  prepare(8 + j, j, lc1.x, lc1.y, at1, fits && j >= 4);
  __syncthreads();   // the level copy is complete (and the records of this wave are visible)
#pragma unroll
  for (int s = 0; s < 4; ++s) gather_global(s);
  if (fits) {
#pragma unroll
    for (int s = 4; s < 8; ++s) gather_lds(s);
  } else {
#pragma unroll
    for (int s = 4; s < 8; ++s) gather_global(s);
  }
  store(8 + j);
}

bool q_backward_ok(const Dims& d) {
  return d.D == 32 && d.P == 4 && d.L == 4 && (int64_t)d.S * d.M * 128 < (int64_t)kOobOffset && d.N <= 65535 && d.Lq >= 1024;
}

int launch_backward_q(const float* grad_out, const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc,
                      const float* attn, const Dims& d, float* grad_loc, float* grad_attn, hipStream_t stream) {
  static std::atomic<uint64_t> lds_opted_in{0};
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(msda_bwd_q), kQLdsBytes, lds_opted_in)) return rc;
  dim3 grid((unsigned)(d.M * ((d.Lq + kQPairs - 1) / kQPairs)), (unsigned)d.N);
  hipLaunchKernelGGL(msda_bwd_q, grid, dim3(kQT), kQLdsBytes, stream, grad_out, value, shapes, lsi, loc, attn, d, grad_loc,
                     grad_attn);
  return (int)hipGetLastError();
}

}  // namespace msda
