// MSDeformAttn forward kernels for gfx950 (MI355X).  Hand-written HIP; replaces
// ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299 (+ :33-84, :923-954) of the reference.
//
// Kernels
//   msda_fwd_generic<T>      any D/L/P, float or double.  One thread per output element, the
//                            reference's own decomposition -- the un-tuned path ops/test.py's tiny
//                            fp64 fixtures run on.
//   msda_fwd_lanegroup<G,LP> fp32, D = 4*G.  A group of G lanes owns one (query, head) pair and
//                            each lane 4 channels, so a sampled corner is ONE 16-byte load per lane
//                            and a full 128-byte line per group (D = 32).  The L*P sampling
//                            locations / weights of a pair are loaded once, coalesced, by the
//                            group's lanes (2 samples per lane at L*P = 16), turned into 4 corner
//                            weights (attention weight and zero-padding folded in) + 4 byte offsets,
//                            and exchanged inside the wave through padded LDS records (broadcast
//                            ds_read_b128, bank-conflict free).  Value rows are fetched with raw
//                            buffer loads: invalid corners use an out-of-range offset and return 0
//                            without touching memory.  Workgroups are ordered head-minor so that
//                            (observed, not required) XCD x only ever touches head x's 1/8 slice of
//                            `value` (2.8 MB / image at the R50 shapes, inside its 4 MiB L2).
#include <type_traits>

#include <cstdlib>

#include "msda_common.hpp"

namespace msda {


// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kBlock)
msda_fwd_generic(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                 const int64_t* __restrict__ lsi, const T* __restrict__ loc,
                 const T* __restrict__ attn, Dims d, T* __restrict__ out) {
  const int64_t total = (int64_t)d.N * d.Lq * d.M * d.D;
  const int64_t pix_stride = (int64_t)d.M * d.D;
  for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * kBlock) {
    const int c = (int)(idx % d.D);
    const int64_t pair = idx / d.D;  // (b*Lq + q)*M + m
    const int m = (int)(pair % d.M);
    const int64_t b = pair / ((int64_t)d.M * d.Lq);
    const T* l_ptr = loc + pair * d.L * d.P * 2;
    const T* a_ptr = attn + pair * d.L * d.P;
    T col = 0;
    for (int l = 0; l < d.L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const T* v_lvl = value + (b * d.S + lsi[l]) * pix_stride + (int64_t)m * d.D + c;
      for (int p = 0; p < d.P; ++p) {
        const Sample<T> s = make_sample<T>(l_ptr[(l * d.P + p) * 2], l_ptr[(l * d.P + p) * 2 + 1], H, W);
        if (!s.in_range) continue;
        const int64_t o1 = ((int64_t)s.h_low * W + s.w_low) * pix_stride;
        const T v1 = s.ok1 ? v_lvl[o1] : (T)0;
        const T v2 = s.ok2 ? v_lvl[o1 + pix_stride] : (T)0;
        const T v3 = s.ok3 ? v_lvl[o1 + (int64_t)W * pix_stride] : (T)0;
        const T v4 = s.ok4 ? v_lvl[o1 + (int64_t)(W + 1) * pix_stride] : (T)0;
        col += (s.hh * s.hw * v1 + s.hh * s.lw * v2 + s.lh * s.hw * v3 + s.lh * s.lw * v4) * a_ptr[l * d.P + p];
      }
    }
    out[idx] = col;
  }
}

// ------------------------------------------------------------------------------------------------
// LDS record of one sample of one pair: 4 corner weights then 4 byte offsets (32 bytes).
// Records of a pair are contiguous; pairs are padded by 16 bytes so that the four pairs a
// ds_read_b128 lane-group serves fall into disjoint bank quads.
__device__ __forceinline__ int rec_pair_stride(int LP) { return LP * 32 + 16; }

template <int G, int LPT>  // LPT: compile-time L*P (0 = runtime)
__global__ void __launch_bounds__(kBlock, 4)
msda_fwd_lanegroup(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                   const int64_t* __restrict__ lsi, const float* __restrict__ loc,
                   const float* __restrict__ attn, Dims d, float* __restrict__ out) {
  constexpr int kPairs = kBlock / G;  // (query, head) pairs per workgroup, all of the same head
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* smp_H = reinterpret_cast<int*>(smem);  // per-sample level table: no division in the hot path
  int* smp_W = smp_H + kMaxLP;
  int* smp_start = smp_W + kMaxLP;
  char* rec_base = smem + kLevelTableBytes;

  const int LP = LPT ? LPT : d.L * d.P;
  const int tid = threadIdx.x;
  if (tid < LP) {
    const int l = tid / d.P;
    smp_H[tid] = (int)shapes[2 * l];
    smp_W[tid] = (int)shapes[2 * l + 1];
    smp_start[tid] = (int)lsi[l];
  }
  __syncthreads();

  const int b = blockIdx.y;
  const int m = blockIdx.x % d.M;
  const int g = tid / G, j = tid % G;
  const int q = (blockIdx.x / d.M) * kPairs + g;
  if (q >= d.Lq) return;  // whole lane-groups drop out; the exchange below is group-local

  const int64_t pair = ((int64_t)b * d.Lq + q) * d.M + m;
  const uint32_t pix_bytes = (uint32_t)d.M * d.D * 4u;
  char* rec = rec_base + g * rec_pair_stride(LP);

  // ---- phase 1: each lane prepares its share of the pair's samples -------------------------------
  auto prepare = [&](int s, float lx, float ly, float a) {
    const int H = smp_H[s], W = smp_W[s];
    const Sample<float> sm = make_sample<float>(lx, ly, H, W);
    const float wa = sm.hh * a, wb = sm.lh * a;
    float4 w;
    w.x = sm.ok1 ? wa * sm.hw : 0.f;
    w.y = sm.ok2 ? wa * sm.lw : 0.f;
    w.z = sm.ok3 ? wb * sm.hw : 0.f;
    w.w = sm.ok4 ? wb * sm.lw : 0.f;
    const uint32_t o1 = (uint32_t)(smp_start[s] + sm.h_low * W + sm.w_low) * pix_bytes;
    u32x4 o;
    o[0] = sm.ok1 ? o1 : kOobOffset;
    o[1] = sm.ok2 ? o1 + pix_bytes : kOobOffset;
    o[2] = sm.ok3 ? o1 + (uint32_t)W * pix_bytes : kOobOffset;
    o[3] = sm.ok4 ? o1 + (uint32_t)(W + 1) * pix_bytes : kOobOffset;
    *reinterpret_cast<float4*>(rec + s * 32) = w;
    *reinterpret_cast<u32x4*>(rec + s * 32 + 16) = o;
  };

  if constexpr (LPT == 2 * G) {
    // two consecutive samples per lane: one 16-byte and one 8-byte coalesced load
    const float4 lc = *reinterpret_cast<const float4*>(loc + pair * (2 * LPT) + 4 * j);
    const float2 at = *reinterpret_cast<const float2*>(attn + pair * LPT + 2 * j);
    prepare(2 * j, lc.x, lc.y, at.x);
    prepare(2 * j + 1, lc.z, lc.w, at.y);
  } else {
    for (int s = j; s < LP; s += G) {
      const float2 lc = *reinterpret_cast<const float2*>(loc + (pair * LP + s) * 2);
      prepare(s, lc.x, lc.y, attn[pair * LP + s]);
    }
  }
  // Producer and consumer lanes are in the same wave (G divides 64): order the LDS traffic only.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- phase 2: gather.  16 bytes per lane per corner, 128 contiguous bytes per group at D=32 -----
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(value) + (int64_t)b * d.S * d.M * d.D, 0, (int)((uint32_t)d.S * pix_bytes), 0x00020000);
  const uint32_t head_off = (uint32_t)m * d.D * 4u;  // wave-uniform -> soffset
  const uint32_t lane_off = (uint32_t)j * 16u;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);

  auto gather = [&](int s) {
    const float4 w = *reinterpret_cast<const float4*>(rec + s * 32);
    const u32x4 o = *reinterpret_cast<const u32x4*>(rec + s * 32 + 16);
    const f32x4 r1 = buffer_load_f32x4(rsrc, o[0] + lane_off, head_off);
    const f32x4 r2 = buffer_load_f32x4(rsrc, o[1] + lane_off, head_off);
    const f32x4 r3 = buffer_load_f32x4(rsrc, o[2] + lane_off, head_off);
    const f32x4 r4 = buffer_load_f32x4(rsrc, o[3] + lane_off, head_off);
    acc.x = fmaf(w.w, r4[0], fmaf(w.z, r3[0], fmaf(w.y, r2[0], fmaf(w.x, r1[0], acc.x))));
    acc.y = fmaf(w.w, r4[1], fmaf(w.z, r3[1], fmaf(w.y, r2[1], fmaf(w.x, r1[1], acc.y))));
    acc.z = fmaf(w.w, r4[2], fmaf(w.z, r3[2], fmaf(w.y, r2[2], fmaf(w.x, r1[2], acc.z))));
    acc.w = fmaf(w.w, r4[3], fmaf(w.z, r3[3], fmaf(w.y, r2[3], fmaf(w.x, r1[3], acc.w))));
  };

  if constexpr (LPT != 0) {
#pragma unroll
    for (int s = 0; s < LPT; ++s) gather(s);
  } else {
#pragma unroll 4
    for (int s = 0; s < LP; ++s) gather(s);
  }

  const f32x4 accv = {acc.x, acc.y, acc.z, acc.w};
  __builtin_nontemporal_store(accv, reinterpret_cast<f32x4*>(out + pair * d.D + 4 * j));
}

// ------------------------------------------------------------------------------------------------
// msda_fwd_fused: msda_fwd_lanegroup<8,16> with MSDeformAttn.forward's elementwise prologue folded in
// (ops/modules/ms_deform_attn.py:99-112): the kernel reads the RAW outputs of the sampling_offsets and
// attention_weights Linear layers plus the reference points and computes, in the lanes that prepare the samples,
//     attn = softmax over the 16 (level, point) logits of the (query, head) pair       (:100-101)
//     loc  = ref_xy + offset / (W_l, H_l)                         (2-d reference points, :103-106)
//     loc  = ref_xy + offset / P * ref_wh * 0.5                   (4-d reference boxes,  :107-109)
// The softmax runs across the 8 lanes of the pair with DPP (2 logits per lane).  This removes the separate
// softmax and location kernels and the write + re-read of the 45.5 MB `sampling_locations` tensor per encoder
// call.  Inference path only: no gradients flow through this entry point.
template <int CTRL>
__device__ __forceinline__ float fdpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float group8_max(float v) {
  v = fmaxf(v, fdpp<0xB1>(v));
  v = fmaxf(v, fdpp<0x4E>(v));
  v = fmaxf(v, fdpp<0x141>(v));
  return v;
}
__device__ __forceinline__ float group8_add(float v) {
  v += fdpp<0xB1>(v);
  v += fdpp<0x4E>(v);
  v += fdpp<0x141>(v);
  return v;
}

template <int REFD>   // 2 or 4
__global__ void __launch_bounds__(kBlock, 4)
msda_fwd_fused(const float* __restrict__ value, const int64_t* __restrict__ shapes,
               const int64_t* __restrict__ lsi, const float* __restrict__ ref_points,
               const float* __restrict__ offsets, const float* __restrict__ logits, Dims d,
               float* __restrict__ out) {
  constexpr int G = 8, LPT = 16, kPairs = kBlock / G;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* smp_H = reinterpret_cast<int*>(smem);
  int* smp_W = smp_H + kMaxLP;
  int* smp_start = smp_W + kMaxLP;
  char* rec_base = smem + kLevelTableBytes;
  const int tid = threadIdx.x;
  if (tid < LPT) {
    const int l = tid / d.P;
    smp_H[tid] = (int)shapes[2 * l];
    smp_W[tid] = (int)shapes[2 * l + 1];
    smp_start[tid] = (int)lsi[l];
  }
  __syncthreads();

  const int b = blockIdx.y;
  const int m = blockIdx.x % d.M;
  const int g = tid / G, j = tid % G;
  const int q = (blockIdx.x / d.M) * kPairs + g;
  if (q >= d.Lq) return;   // whole lane-groups drop out; softmax and exchange are group-local

  const int64_t pair = ((int64_t)b * d.Lq + q) * d.M + m;
  const uint32_t pix_bytes = (uint32_t)d.M * 128u;
  char* rec = rec_base + g * (LPT * 32 + 16);

  // raw Linear outputs of this lane's two samples (2j, 2j+1), their level and its reference point
  const float4 off = *reinterpret_cast<const float4*>(offsets + pair * (2 * LPT) + 4 * j);
  const float2 lg = *reinterpret_cast<const float2*>(logits + pair * LPT + 2 * j);
  const int s0 = 2 * j;
  const int lvl = s0 / d.P;                                       // both samples share it when P is even
  const int lvl1 = (s0 + 1) / d.P;
  const float* rp = ref_points + ((int64_t)b * d.Lq + q) * d.L * REFD;
  // softmax over the pair's 16 logits
  const float mx = group8_max(fmaxf(lg.x, lg.y));
  const float e0 = __expf(lg.x - mx), e1 = __expf(lg.y - mx);
  const float inv = 1.0f / group8_add(e0 + e1);
  const float a0 = e0 * inv, a1 = e1 * inv;

  auto location = [&](int level, float ox, float oy, float& lx, float& ly) {
    if constexpr (REFD == 2) {
      const float2 r = *reinterpret_cast<const float2*>(rp + level * 2);
      lx = r.x + ox / (float)smp_W[level * d.P];
      ly = r.y + oy / (float)smp_H[level * d.P];
    } else {
      const float4 r = *reinterpret_cast<const float4*>(rp + level * 4);
      lx = r.x + ox / (float)d.P * r.z * 0.5f;
      ly = r.y + oy / (float)d.P * r.w * 0.5f;
    }
  };

  auto prepare = [&](int s, float lx, float ly, float a) {
    const int H = smp_H[s], W = smp_W[s];
    const Sample<float> sm = make_sample<float>(lx, ly, H, W);
    const float wa = sm.hh * a, wb = sm.lh * a;
    float4 w;
    w.x = sm.ok1 ? wa * sm.hw : 0.f;
    w.y = sm.ok2 ? wa * sm.lw : 0.f;
    w.z = sm.ok3 ? wb * sm.hw : 0.f;
    w.w = sm.ok4 ? wb * sm.lw : 0.f;
    const uint32_t o1 = (uint32_t)(smp_start[s] + sm.h_low * W + sm.w_low) * pix_bytes;
    u32x4 o;
    o[0] = sm.ok1 ? o1 : kOobOffset;
    o[1] = sm.ok2 ? o1 + pix_bytes : kOobOffset;
    o[2] = sm.ok3 ? o1 + (uint32_t)W * pix_bytes : kOobOffset;
    o[3] = sm.ok4 ? o1 + (uint32_t)(W + 1) * pix_bytes : kOobOffset;
    *reinterpret_cast<float4*>(rec + s * 32) = w;
    *reinterpret_cast<u32x4*>(rec + s * 32 + 16) = o;
  };
  float lx, ly;
  location(lvl, off.x, off.y, lx, ly);
  prepare(s0, lx, ly, a0);
  location(lvl1, off.z, off.w, lx, ly);
  prepare(s0 + 1, lx, ly, a1);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(value) + (int64_t)b * d.S * d.M * 32, 0, (int)((uint32_t)d.S * pix_bytes), 0x00020000);
  const uint32_t head_off = (uint32_t)m * 128u, lane_off = (uint32_t)j * 16u;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < LPT; ++s) {
    const float4 w = *reinterpret_cast<const float4*>(rec + s * 32);
    const u32x4 o = *reinterpret_cast<const u32x4*>(rec + s * 32 + 16);
    const f32x4 r1 = buffer_load_f32x4(rsrc, o[0] + lane_off, head_off);
    const f32x4 r2 = buffer_load_f32x4(rsrc, o[1] + lane_off, head_off);
    const f32x4 r3 = buffer_load_f32x4(rsrc, o[2] + lane_off, head_off);
    const f32x4 r4 = buffer_load_f32x4(rsrc, o[3] + lane_off, head_off);
#pragma unroll
    for (int c = 0; c < 4; ++c)
      acc[c] = fmaf(w.w, r4[c], fmaf(w.z, r3[c], fmaf(w.y, r2[c], fmaf(w.x, r1[c], acc[c]))));
  }
  __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(out + pair * 32 + 4 * j));
}

bool fused_forward_ok(const Dims& d, int ref_dim) {
  return d.D == 32 && d.L * d.P == 16 && d.P % 2 == 0 && (ref_dim == 2 || ref_dim == 4) &&
         (int64_t)d.S * d.M * 128 < (int64_t)kOobOffset && d.N <= 65535;
}

static inline bool lg3_ok(const Dims& d);
template <int REFD>
static int launch_lg3_t(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc,
                        const float* attn, const float* ref_points, const Dims& d, int head_major, float* out,
                        hipStream_t stream);

bool fused_forward_hm_ok(const Dims& d, int ref_dim) { return fused_forward_ok(d, ref_dim) && lg3_ok(d); }

int launch_forward_fused(int variant, const float* value, int head_major, const int64_t* shapes, const int64_t* lsi,
                         const float* ref_points, int ref_dim, const float* offsets, const float* logits, const Dims& d,
                         float* out, hipStream_t stream, const char** kernel_name) {
  static const bool use_lg3 = ab_env_int("MSDA_HIP_FUSED_LG3", 1) != 0;
  // encoder-shaped calls: the LDS-window kernel with the prologue folded in while the samples are local (variant 0
  // follows the locality report like the operator does; variants 9 / 7 pin the window / the gather kernel)
  if (variant != kAuto) drop_call_context();
  if (win_forward_ok(d) && (variant == kWin || (variant == kAuto && win_forward_auto(d, stream)))) {
    *kernel_name = "msda_fwd_win_fused";
    return launch_forward_win_fused(value, head_major, shapes, lsi, ref_points, ref_dim, offsets, logits, d, out, stream);
  }
  if (lg3_ok(d) && (use_lg3 || head_major)) {   // encoder-sized calls: the lg3 structure with the prologue folded in
    *kernel_name = "msda_fwd_lg3_fused";
    return ref_dim == 2 ? launch_lg3_t<2>(value, shapes, lsi, offsets, logits, ref_points, d, head_major, out, stream)
                        : launch_lg3_t<4>(value, shapes, lsi, offsets, logits, ref_points, d, head_major, out, stream);
  }
  if (head_major) return -5;   // MSDA_ERR_UNSUPPORTED: the small-call fused kernel reads the reference layout only
  *kernel_name = "msda_fwd_fused";
  constexpr int kPairs = kBlock / 8;
  const size_t lds = kLevelTableBytes + (size_t)kPairs * (16 * 32 + 16);
  dim3 grid((unsigned)(d.M * ((d.Lq + kPairs - 1) / kPairs)), (unsigned)d.N);
  if (ref_dim == 2)
    hipLaunchKernelGGL(msda_fwd_fused<2>, grid, dim3(kBlock), lds, stream, value, shapes, lsi, ref_points, offsets, logits,
                       d, out);
  else
    hipLaunchKernelGGL(msda_fwd_fused<4>, grid, dim3(kBlock), lds, stream, value, shapes, lsi, ref_points, offsets, logits,
                       d, out);
  return (int)hipGetLastError();
}

// (msda_fwd_lgcl -- the lane-group kernel with the coarsest level resident in LDS, variant 6 -- and msda_fwd_lgp -- msda_fwd_lg3
// made persistent, variant 8 -- lost their A/B and live in experiments/msda_fwd_lg_variants.inc, built by `make experiments`.)
constexpr int kClSlots = 280;                                   // resident pixels (35 KB) + one all-zero slot
// ------------------------------------------------------------------------------------------------
// msda_fwd_lg3: one-shot (non-persistent) 1024-thread workgroups of 128 queries x 1 head, last pyramid level
// resident in LDS.  Lessons of msda_fwd_lgcl (profiles/): the gather needs ~32 waves per CU in flight;
// a persistent 4-wave workgroup that owns 54 KB of LDS leaves 12 and becomes latency bound (68 % of wave cycles in
// s_waitcnt).  Here 16 waves share one 35 KB copy and the sample records are built 8 samples at a time (2.1 KB per
// wave instead of 4.2), so two workgroups = 32 waves fit a CU and the kernel must stay within 64 VGPRs.
constexpr int kL3Threads = 1024;
constexpr int kL3RecPair = 8 * 32 + 16;
constexpr int kL3LdsBytes = kLevelTableBytes + (kClSlots + 1) * 128 + (kL3Threads / 8) * kL3RecPair;

// REFD 0: `loc` / `attn` are normalised sampling locations and softmaxed weights (the operator).  REFD 2 / 4: they are
// the RAW Linear outputs (offsets, logits) and `ref_points` [N, Lq, L, REFD] the reference points: softmax over the
// pair's 16 logits and the location arithmetic of ops/modules/ms_deform_attn.py:99-112 run in the kernel
// (msda_fwd_fused's prologue on this kernel's sample assignment: lane j holds samples j and 8 + j).
// head_major != 0: `value` is [N, M, S, 32] (written that way by the value projection, include/linear_hip.h) -- a
// head's pixels are then 128 bytes apart instead of 1 KB.
template <int REFD>
__global__ void __launch_bounds__(kL3Threads, 8)
msda_fwd_lg3(const float* __restrict__ value, const int64_t* __restrict__ shapes,
             const int64_t* __restrict__ lsi, const float* __restrict__ loc,
             const float* __restrict__ attn, const float* __restrict__ ref_points, Dims d, int head_major,
             float* __restrict__ out) {
  constexpr int G = 8, LPT = 16, P = 4, kPairs = kL3Threads / G;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* smp_H = reinterpret_cast<int*>(smem);
  int* smp_W = smp_H + kMaxLP;
  int* smp_start = smp_W + kMaxLP;
  char* cl_base = smem + kLevelTableBytes;
  char* rec_base = cl_base + (kClSlots + 1) * 128;
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
  const bool fits = nres_all <= kClSlots;          // uniform; otherwise the last level also takes the L1 path
  const int nres = fits ? nres_all : 0;

  const int b = blockIdx.y;
  const int m = blockIdx.x % d.M;
  const uint32_t pix_bytes = head_major ? 128u : (uint32_t)d.M * 128u;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(value) + (int64_t)b * d.S * d.M * 32, 0, (int)((uint32_t)d.S * (uint32_t)d.M * 128u), 0x00020000);
  const uint32_t head_off = head_major ? (uint32_t)m * (uint32_t)d.S * 128u : (uint32_t)m * 128u;
  for (int i = tid >> 3; i <= nres && fits; i += kL3Threads / 8) {   // slot `nres` stays zero: dead corners read it
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (i < nres) v = buffer_load_f32x4(rsrc, (uint32_t)(res_pix0 + i) * pix_bytes + (uint32_t)(tid & 7) * 16u, head_off);
    *reinterpret_cast<f32x4*>(cl_base + i * 128 + (tid & 7) * 16) = v;
  }

  const int g = tid / G, j = tid % G;
  const int q = (blockIdx.x / d.M) * kPairs + g;
  const bool live = q < d.Lq;
  const int64_t pair = ((int64_t)b * d.Lq + (live ? q : 0)) * d.M + m;
  char* rec = rec_base + g * kL3RecPair;
  const uint32_t lane_off = (uint32_t)j * 16u;
  const uint32_t cl_addr0 = smem_base + kLevelTableBytes;
  const uint32_t zero_slot = cl_addr0 + (uint32_t)nres * 128u;

  // lane j prepares sample j of pass 0 (levels 0, 1) and sample 8 + j of pass 1 (levels 2, 3)
  float2 lc0 = make_float2(0.f, 0.f), lc1 = make_float2(0.f, 0.f);
  float at0 = 0.f, at1 = 0.f;
  if (live) {
    lc0 = *reinterpret_cast<const float2*>(loc + pair * (2 * LPT) + 2 * j);
    lc1 = *reinterpret_cast<const float2*>(loc + pair * (2 * LPT) + 2 * (8 + j));
    at0 = attn[pair * LPT + j];
    at1 = attn[pair * LPT + 8 + j];
  }
  if constexpr (REFD != 0) {   // raw offsets / logits -> locations / softmax weights; dead pairs carry zeros through
    auto gmax = [](float v) {
      v = fmaxf(v, fdpp<0xB1>(v)); v = fmaxf(v, fdpp<0x4E>(v)); v = fmaxf(v, fdpp<0x141>(v));
      return v;
    };
    const float mx = gmax(fmaxf(at0, at1));
    const float e0 = __expf(at0 - mx), e1 = __expf(at1 - mx);
    // this prologue runs in all 16 waves of the workgroup at once, before any gather is issued: every instruction
    // here is on the critical path of the CU.  v_rcp_f32 (1 ulp) instead of IEEE divisions: the sampling location
    // moves by < 1e-6 of a pixel, the weights by 1 ulp
    const float inv = __builtin_amdgcn_rcpf(group8_add(e0 + e1));
    at0 = e0 * inv;
    at1 = e1 * inv;
    const float* rp = ref_points + ((int64_t)b * d.Lq + (live ? q : 0)) * LPT / P * REFD;
    const int l0 = j / P, l1 = 2 + j / P;               // levels of samples j and 8 + j (L = P = 4)
    if constexpr (REFD == 2) {
      const float2 r0 = *reinterpret_cast<const float2*>(rp + l0 * 2), r1 = *reinterpret_cast<const float2*>(rp + l1 * 2);
      lc0.x = fmaf(lc0.x, __builtin_amdgcn_rcpf((float)smp_W[j]), r0.x);
      lc0.y = fmaf(lc0.y, __builtin_amdgcn_rcpf((float)smp_H[j]), r0.y);
      lc1.x = fmaf(lc1.x, __builtin_amdgcn_rcpf((float)smp_W[8 + j]), r1.x);
      lc1.y = fmaf(lc1.y, __builtin_amdgcn_rcpf((float)smp_H[8 + j]), r1.y);
    } else {
      const float4 r0 = *reinterpret_cast<const float4*>(rp + l0 * 4), r1 = *reinterpret_cast<const float4*>(rp + l1 * 4);
      lc0.x = r0.x + lc0.x / (float)P * r0.z * 0.5f;
      lc0.y = r0.y + lc0.y / (float)P * r0.w * 0.5f;
      lc1.x = r1.x + lc1.x / (float)P * r1.z * 0.5f;
      lc1.y = r1.y + lc1.y / (float)P * r1.w * 0.5f;
    }
  }
  auto prepare = [&](int s, int slot, float lx, float ly, float a, bool resident) {
    const int H = smp_H[s], W = smp_W[s];
    const Sample<float> sm = make_sample<float>(lx, ly, H, W);
    const float wa = sm.hh * a, wb = sm.lh * a;
    float4 w;
    w.x = (live && sm.ok1) ? wa * sm.hw : 0.f;
    w.y = (live && sm.ok2) ? wa * sm.lw : 0.f;
    w.z = (live && sm.ok3) ? wb * sm.hw : 0.f;
    w.w = (live && sm.ok4) ? wb * sm.lw : 0.f;
    const int pix1 = smp_start[s] + sm.h_low * W + sm.w_low;
    u32x4 o;
    if (resident) {
      const uint32_t a1 = cl_addr0 + (uint32_t)(pix1 - res_pix0) * 128u;
      o[0] = sm.ok1 ? a1 : zero_slot;
      o[1] = sm.ok2 ? a1 + 128u : zero_slot;
      o[2] = sm.ok3 ? a1 + (uint32_t)W * 128u : zero_slot;
      o[3] = sm.ok4 ? a1 + (uint32_t)(W + 1) * 128u : zero_slot;
      // (round 3, tried and taken out again: the two pairs of a 16-lane ds_read_b128 service group read two pixel slots, which
      // share their banks when the slots have the same parity -- r02 counters: conflicts on 79 % of this kernel's LDS-active
      // cycles.  Letting the even pair read the even slot of each corner row first and the odd pair the odd one brought that
      // to 45 %, but the two conditional swaps cost three registers the kernel does not have at 64: 12 bytes of scratch per
      // lane, HBM traffic 213 -> 255 MB and 126 -> 138-143 us on the wide / uniform flavours.  LDS-active cycles are 1 % of
      // the kernel's wave cycles.)
    } else {
      const uint32_t o1 = (uint32_t)pix1 * pix_bytes;
      o[0] = sm.ok1 ? o1 : kOobOffset;
      o[1] = sm.ok2 ? o1 + pix_bytes : kOobOffset;
      o[2] = sm.ok3 ? o1 + (uint32_t)W * pix_bytes : kOobOffset;
      o[3] = sm.ok4 ? o1 + (uint32_t)(W + 1) * pix_bytes : kOobOffset;
    }
    *reinterpret_cast<float4*>(rec + slot * 32) = w;
    *reinterpret_cast<u32x4*>(rec + slot * 32 + 16) = o;
  };
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  auto gather_global = [&](int slot) {
    const float4 w = *reinterpret_cast<const float4*>(rec + slot * 32);
    const u32x4 o = *reinterpret_cast<const u32x4*>(rec + slot * 32 + 16);
    const f32x4 r1 = buffer_load_f32x4(rsrc, o[0] + lane_off, head_off);
    const f32x4 r2 = buffer_load_f32x4(rsrc, o[1] + lane_off, head_off);
    const f32x4 r3 = buffer_load_f32x4(rsrc, o[2] + lane_off, head_off);
    const f32x4 r4 = buffer_load_f32x4(rsrc, o[3] + lane_off, head_off);
#pragma unroll
    for (int c = 0; c < 4; ++c)
      acc[c] = fmaf(w.w, r4[c], fmaf(w.z, r3[c], fmaf(w.y, r2[c], fmaf(w.x, r1[c], acc[c]))));
  };
  auto gather_lds = [&](int slot) {
    typedef const f32x4 __attribute__((address_space(3)))* lp;
    const float4 w = *reinterpret_cast<const float4*>(rec + slot * 32);
    const u32x4 o = *reinterpret_cast<const u32x4*>(rec + slot * 32 + 16);
    const f32x4 r1 = *reinterpret_cast<lp>((uintptr_t)(o[0] + lane_off));
    const f32x4 r2 = *reinterpret_cast<lp>((uintptr_t)(o[1] + lane_off));
    const f32x4 r3 = *reinterpret_cast<lp>((uintptr_t)(o[2] + lane_off));
    const f32x4 r4 = *reinterpret_cast<lp>((uintptr_t)(o[3] + lane_off));
#pragma unroll
    for (int c = 0; c < 4; ++c)
      acc[c] = fmaf(w.w, r4[c], fmaf(w.z, r3[c], fmaf(w.y, r2[c], fmaf(w.x, r1[c], acc[c]))));
  };

  // pass 0: samples 0..7 (levels 0 and 1), all through the L1 path; overlaps the other waves' staging
  prepare(j, j, lc0.x, lc0.y, at0, false);
  wave_sync();
#pragma unroll
  for (int s = 0; s < 8; ++s) gather_global(s);
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
  if (live) __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(out + pair * 32 + 4 * j));
}

static inline bool lg3_ok(const Dims& d) {
  return d.D == 32 && d.P == 4 && d.L == 4 && (int64_t)d.S * d.M * 128 < (int64_t)kOobOffset && d.N <= 65535 &&
         d.Lq >= 1024;   // a 128-query workgroup copies 35 KB: pointless for a handful of queries
}

template <int REFD>
static int launch_lg3_t(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc,
                        const float* attn, const float* ref_points, const Dims& d, int head_major, float* out,
                        hipStream_t stream) {
  static std::atomic<uint64_t> lds_opted_in{0};
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(msda_fwd_lg3<REFD>), kL3LdsBytes, lds_opted_in)) return rc;
  constexpr int kPairs = kL3Threads / 8;
  dim3 grid((unsigned)(d.M * ((d.Lq + kPairs - 1) / kPairs)), (unsigned)d.N);
  hipLaunchKernelGGL(msda_fwd_lg3<REFD>, grid, dim3(kL3Threads), kL3LdsBytes, stream, value, shapes, lsi, loc, attn,
                     ref_points, d, head_major, out);
  return (int)hipGetLastError();
}

static int launch_lg3(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc,
                      const float* attn, const Dims& d, float* out, hipStream_t stream) {
  return launch_lg3_t<0>(value, shapes, lsi, loc, attn, nullptr, d, 0, out, stream);
}

#ifdef MSDA_EXPERIMENTS
#include "experiments/msda_fwd_lg_variants.inc"
#endif

// ------------------------------------------------------------------------------------------------
static inline bool lanegroup_ok(const Dims& d, int* G_out) {
  if (d.D % 4 != 0) return false;
  const int G = d.D / 4;
  if (G < 1 || G > 64 || (G & (G - 1)) != 0) return false;
  if (d.L * d.P > kMaxLP) return false;
  if ((int64_t)d.S * d.M * d.D * 4 >= (int64_t)kOobOffset) return false;  // 32-bit byte offsets per image
  if (d.N > 65535) return false;
  const int64_t lds = kLevelTableBytes + (int64_t)(kBlock / G) * (d.L * d.P * 32 + 16);
  if (lds > 64 * 1024) return false;
  *G_out = G;
  return true;
}

template <int G, int LPT>
static int launch_lanegroup(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc,
                            const float* attn, const Dims& d, float* out, hipStream_t stream) {
  constexpr int kPairs = kBlock / G;
  const int LP = d.L * d.P;
  const size_t lds = kLevelTableBytes + (size_t)kPairs * (LP * 32 + 16);
  dim3 grid((unsigned)(d.M * ((d.Lq + kPairs - 1) / kPairs)), (unsigned)d.N);
  hipLaunchKernelGGL((msda_fwd_lanegroup<G, LPT>), grid, dim3(kBlock), lds, stream, value, shapes, lsi, loc, attn,
                     d, out);
  return (int)hipGetLastError();
}

template <>
int launch_forward<float>(int variant, const float* value, const int64_t* shapes, const int64_t* lsi,
                          const float* loc, const float* attn, const Dims& d, float* out, hipStream_t stream,
                          const char** kernel_name) {
  int G = 0;
  const bool lg = lanegroup_ok(d, &G);
#ifdef MSDA_EXPERIMENTS
  const bool tl = tiled_forward_ok(d);
#else
  const bool tl = false;   // variants 3..5 are not in this build: they resolve to the lane-group kernel
#endif
  // kbench A/B (profiles/): msda_fwd_lg3 beats msda_fwd_lanegroup by 14-20 % from ~1000 queries on; the tiled and
  // lgcl kernels lose and stay opt-in
  // ... and the LDS-window kernel beats msda_fwd_lg3 on the encoder shape while the samples stay near their queries:
  // win_forward_auto follows the locality the window kernel itself reported for the latest launches
  if (variant == kAuto) variant = win_forward_auto(d, stream) ? kWin : (lg3_ok(d) ? kLaneGroupL3 : (lg ? kLaneGroup : kGeneric));
  else drop_call_context();
#ifdef MSDA_EXPERIMENTS   // the generations of the window kernel that lost their A/B (experiments/, `make experiments`)
  if (variant == kWinP && !winp_forward_ok(d)) variant = lg3_ok(d) ? kLaneGroupL3 : kLaneGroup;
  if (variant == kWinP) {
    *kernel_name = "msda_fwd_winp";
    return launch_forward_winp(value, shapes, lsi, loc, attn, d, out, stream);
  }
  if (variant == kWinL && !winl_forward_ok(d)) variant = lg3_ok(d) ? kLaneGroupL3 : kLaneGroup;
  if (variant == kWinL) {
    *kernel_name = "msda_fwd_winl";
    return launch_forward_winl(value, shapes, lsi, loc, attn, d, out, stream);
  }
  if (variant == kWin4 && !win4_forward_ok(d)) variant = lg3_ok(d) ? kLaneGroupL3 : kLaneGroup;
  if (variant == kWin4) {
    *kernel_name = "msda_fwd_win4";
    return launch_forward_win4(value, shapes, lsi, loc, attn, d, out, stream);
  }
  if (variant == kWin3 && !win3_forward_ok(d)) variant = lg3_ok(d) ? kLaneGroupL3 : kLaneGroup;
  if (variant == kWin3) {
    *kernel_name = "msda_fwd_win3";
    return launch_forward_win3(value, shapes, lsi, loc, attn, d, out, stream);
  }
  if (variant == kWin2 && !win2_forward_ok(d)) variant = lg3_ok(d) ? kLaneGroupL3 : kLaneGroup;
  if (variant == kWin2) {
    *kernel_name = "msda_fwd_win2";
    return launch_forward_win2(value, shapes, lsi, loc, attn, d, out, stream);
  }
#endif
  if (variant == kWin && !win_forward_ok(d)) variant = lg3_ok(d) ? kLaneGroupL3 : kLaneGroup;
  if (variant == kWin) {
    *kernel_name = "msda_fwd_win";
    return launch_forward_win(value, shapes, lsi, loc, attn, d, out, stream);
  }
  if (variant == kLaneGroupL3 && !lg3_ok(d)) variant = kLaneGroup;
  if (variant == kLaneGroupL3) {
    *kernel_name = "msda_fwd_lg3";
    return launch_lg3(value, shapes, lsi, loc, attn, d, out, stream);
  }
#ifdef MSDA_EXPERIMENTS
  if (variant == kLaneGroupP && !lg3_ok(d)) variant = kLaneGroup;
  if (variant == kLaneGroupP) {
    *kernel_name = "msda_fwd_lgp";
    return launch_lgp(value, shapes, lsi, loc, attn, d, out, stream);
  }
  if (variant == kLaneGroupCL && !lgcl_ok(d)) variant = kLaneGroup;
  if (variant == kLaneGroupCL) {
    *kernel_name = "msda_fwd_lgcl";
    return launch_lgcl(value, shapes, lsi, loc, attn, d, out, stream);
  }
#endif
  if (variant >= kTiled && !tl) variant = lg ? kLaneGroup : kGeneric;
  if (variant == kLaneGroup && !lg) variant = kGeneric;
#ifdef MSDA_EXPERIMENTS
  if (variant >= kTiled) {
    *kernel_name = "msda_fwd_tiled";
    return launch_forward_tiled(variant - kTiled, value, shapes, lsi, loc, attn, d, out, stream);
  }
#endif
  if (variant == kLaneGroup) {
    const int LP = d.L * d.P;
    *kernel_name = "msda_fwd_lanegroup";
    switch (G) {
      case 1: return launch_lanegroup<1, 0>(value, shapes, lsi, loc, attn, d, out, stream);
      case 2: return launch_lanegroup<2, 0>(value, shapes, lsi, loc, attn, d, out, stream);
      case 4: return launch_lanegroup<4, 0>(value, shapes, lsi, loc, attn, d, out, stream);
      case 8:
        if (LP == 16) return launch_lanegroup<8, 16>(value, shapes, lsi, loc, attn, d, out, stream);
        return launch_lanegroup<8, 0>(value, shapes, lsi, loc, attn, d, out, stream);
      case 16: return launch_lanegroup<16, 0>(value, shapes, lsi, loc, attn, d, out, stream);
      case 32: return launch_lanegroup<32, 0>(value, shapes, lsi, loc, attn, d, out, stream);
      default: return launch_lanegroup<64, 0>(value, shapes, lsi, loc, attn, d, out, stream);
    }
  }
  *kernel_name = "msda_fwd_generic";
  const int64_t total = (int64_t)d.N * d.Lq * d.M * d.D;
  const unsigned blocks = (unsigned)((total + kBlock - 1) / kBlock < 65536 * 16 ? (total + kBlock - 1) / kBlock : 65536 * 16);
  hipLaunchKernelGGL(msda_fwd_generic<float>, dim3(blocks), dim3(kBlock), 0, stream, value, shapes, lsi, loc, attn, d,
                     out);
  return (int)hipGetLastError();
}

template <>
int launch_forward<double>(int /*variant*/, const double* value, const int64_t* shapes, const int64_t* lsi,
                           const double* loc, const double* attn, const Dims& d, double* out, hipStream_t stream,
                           const char** kernel_name) {
  *kernel_name = "msda_fwd_generic";
  const int64_t total = (int64_t)d.N * d.Lq * d.M * d.D;
  const unsigned blocks = (unsigned)((total + kBlock - 1) / kBlock < 65536 * 16 ? (total + kBlock - 1) / kBlock : 65536 * 16);
  hipLaunchKernelGGL(msda_fwd_generic<double>, dim3(blocks), dim3(kBlock), 0, stream, value, shapes, lsi, loc, attn, d,
                     out);
  return (int)hipGetLastError();
}

}  // namespace msda
