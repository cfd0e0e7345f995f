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
#include "msda_common.hpp"

namespace msda {

constexpr bool kTiledIsDefault = false;

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
  const bool tl = tiled_forward_ok(d);
  // the tiled kernel is selected automatically only once it beats the lane-group kernel (kbench A/B)
  if (variant == kAuto) variant = (kTiledIsDefault && tl && d.S >= 4096) ? kTiled : (lg ? kLaneGroup : kGeneric);
  if (variant >= kTiled && !tl) variant = lg ? kLaneGroup : kGeneric;
  if (variant == kLaneGroup && !lg) variant = kGeneric;
  if (variant >= kTiled) {
    *kernel_name = "msda_fwd_tiled";
    return launch_forward_tiled(variant - kTiled, value, shapes, lsi, loc, attn, d, out, stream);
  }
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
