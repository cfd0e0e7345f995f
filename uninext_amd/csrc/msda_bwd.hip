// MSDeformAttn backward kernels for gfx950 (MI355X).  Hand-written HIP; replaces the reference's
// col2im family ops/src/cuda/ms_deform_im2col_cuda.cuh:87-234 (per-sample gradient math),
// :301-920 (six kernel variants keyed on D) and the dispatcher :956-1327.
//
// Gradient formulas (ms_deform_im2col_cuda.cuh:113-158), per sample with bilinear fractions
// (lh, lw), corner values v1..v4 (0 where the corner is outside), upstream g_c, weight a:
//   grad_value[corner k, c] += w_k * a * g_c
//   grad_attn               = sum_c g_c * (w1 v1 + w2 v2 + w3 v3 + w4 v4)
//   grad_loc.x              = W * a * sum_c g_c * (hh (v2 - v1) + lh (v4 - v3))
//   grad_loc.y              = H * a * sum_c g_c * (hw (v3 - v1) + lw (v4 - v2))
//
// Kernels
//   msda_bwd_generic<T,HALF> any D/L/P, float or double: one wave64 (D > 32) or half a wave (D <= 32) per (query,
//                            head) pair, lanes stride the channels, shuffle reduction.  One kernel instead of the
//                            reference's six D-specific ones; this is what fp64 gradcheck runs on.
//   msda_bwd_lanegroup<G,LP> fp32, D = 4*G: same work split as msda_fwd_lanegroup (G lanes per pair,
//                            4 channels per lane, setup shared through LDS records); the three
//                            per-sample reductions run over the G lanes with DPP, the results go back
//                            to the lane that owns the sample and leave as coalesced 16/8-byte stores.
#include "msda_common.hpp"

namespace msda {

// HALF = 1: D <= 32, a wave carries TWO pairs (32 lanes each) -- at the UNINEXT head size a whole wave per pair would
// leave half the lanes idle; the three reductions then stay inside a half-wave.
template <typename T, int HALF>
__global__ void __launch_bounds__(kBlock)
msda_bwd_generic(const T* __restrict__ grad_out, const T* __restrict__ value,
                 const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                 const T* __restrict__ loc, const T* __restrict__ attn, Dims d,
                 T* __restrict__ grad_value, T* __restrict__ grad_loc, T* __restrict__ grad_attn) {
  constexpr int kLanes = HALF ? 32 : 64;                   // lanes per pair
  constexpr int kPairsPerBlock = kBlock / kLanes;
  const int lane = threadIdx.x & (kLanes - 1);
  const int64_t n_pairs = (int64_t)d.N * d.Lq * d.M;
  const int64_t pix_stride = (int64_t)d.M * d.D;
  const int LP = d.L * d.P;
  auto group_sum = [](T v) {
    if constexpr (sizeof(T) == 4) {
      // float: four DPP butterfly steps inside a row of 16 and one cross-row exchange per 32 lanes (the shuffle loop is a
      // chain of five or six dependent ds_bpermute round trips per value, three values per sample)
      float f = (float)v;
      f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
      f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
      f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x141, 0xF, 0xF, true));   // row_half_mirror
      f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x140, 0xF, 0xF, true));   // row_mirror
      f += __shfl_xor(f, 16, 64);
      if constexpr (kLanes == 64) f += __shfl_xor(f, 32, 64);
      return (T)f;
    } else {
#pragma unroll
      for (int o = kLanes / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      return v;
    }
  };
  for (int64_t pair0 = (int64_t)blockIdx.x * kPairsPerBlock + threadIdx.x / kLanes; ; pair0 += (int64_t)gridDim.x * kPairsPerBlock) {
    // the two halves of a wave run in lock step: a half past the end idles through the loop with `live` off
    const bool live = pair0 < n_pairs;
    if (!__ballot(live)) break;
    const int64_t pair = live ? pair0 : 0;
    const int m = (int)(pair % d.M);
    const int64_t b = pair / ((int64_t)d.M * d.Lq);
    const T* g_ptr = grad_out + pair * d.D;
    if constexpr (HALF == 1 && sizeof(T) == 4) {
      // D <= 32, float: one channel per lane.  The points of a level are taken four at a time -- all four are prepared
      // and their 16 corner loads issued before the first one is consumed, so a pair pays one memory round trip per
      // four samples instead of one per sample (this path is what the decoder-call backward runs; it is bound by the
      // L2 atomic rate, the round trips were the part above that bound)
      const bool ch = lane < d.D;
      const T g = ch ? g_ptr[lane] : (T)0;
      for (int l = 0; l < d.L; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const int64_t lvl_off = (b * d.S + lsi[l]) * pix_stride + (int64_t)m * d.D + lane;
        for (int p0 = 0; p0 < d.P; p0 += 4) {
          Sample<T> s[4];
          T a[4], v[4][4];
          int64_t o1[4];
          bool in[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const bool have = p0 + k < d.P;
            const int64_t si = pair * LP + l * d.P + (have ? p0 + k : 0);
            a[k] = attn[si];
            s[k] = make_sample<T>(loc[si * 2], loc[si * 2 + 1], H, W);
            in[k] = live && have && s[k].in_range;
            o1[k] = lvl_off + ((int64_t)s[k].h_low * W + s[k].w_low) * pix_stride;
            const bool ld = in[k] && ch;
            v[k][0] = (ld && s[k].ok1) ? value[o1[k]] : (T)0;
            v[k][1] = (ld && s[k].ok2) ? value[o1[k] + pix_stride] : (T)0;
            v[k][2] = (ld && s[k].ok3) ? value[o1[k] + (int64_t)W * pix_stride] : (T)0;
            v[k][3] = (ld && s[k].ok4) ? value[o1[k] + (int64_t)W * pix_stride + pix_stride] : (T)0;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (p0 + k >= d.P) break;                       // uniform
            const int64_t si = pair * LP + l * d.P + p0 + k;
            T pa = 0, pw = 0, ph = 0;
            if (__ballot(in[k])) {                           // wave-uniform (the reductions need every lane of the group)
              if (in[k] && ch) {
                const Sample<T>& q = s[k];
                const T tgv = g * a[k];
                const T w1 = q.hh * q.hw, w2 = q.hh * q.lw, w3 = q.lh * q.hw, w4 = q.lh * q.lw;
                const int64_t o3 = o1[k] + (int64_t)W * pix_stride;
                if (q.ok1) atomic_add(grad_value + o1[k], w1 * tgv);
                if (q.ok2) atomic_add(grad_value + o1[k] + pix_stride, w2 * tgv);
                if (q.ok3) atomic_add(grad_value + o3, w3 * tgv);
                if (q.ok4) atomic_add(grad_value + o3 + pix_stride, w4 * tgv);
                pa = g * (w1 * v[k][0] + w2 * v[k][1] + w3 * v[k][2] + w4 * v[k][3]);
                pw = tgv * (q.hh * (v[k][1] - v[k][0]) + q.lh * (v[k][3] - v[k][2]));
                ph = tgv * (q.hw * (v[k][2] - v[k][0]) + q.lw * (v[k][3] - v[k][1]));
              }
              pa = group_sum(pa);
              pw = group_sum(pw);
              ph = group_sum(ph);
            }
            if (live && lane == 0) {
              grad_attn[si] = pa;
              grad_loc[si * 2] = (T)W * pw;
              grad_loc[si * 2 + 1] = (T)H * ph;
            }
          }
        }
      }
      continue;
    }
    for (int l = 0; l < d.L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const int64_t lvl_off = (b * d.S + lsi[l]) * pix_stride + (int64_t)m * d.D;
      for (int p = 0; p < d.P; ++p) {
        const int64_t si = pair * LP + l * d.P + p;
        const T a = attn[si];
        const Sample<T> s = make_sample<T>(loc[si * 2], loc[si * 2 + 1], H, W);
        T pa = 0, pw = 0, ph = 0;
        const bool in = live && s.in_range;
        if (__ballot(in)) {  // wave-uniform branch (the reductions below need every lane of the group)
          if (in) {
            const int64_t o1 = lvl_off + ((int64_t)s.h_low * W + s.w_low) * pix_stride;
            const int64_t o2 = o1 + pix_stride, o3 = o1 + (int64_t)W * pix_stride, o4 = o3 + pix_stride;
            const T w1 = s.hh * s.hw, w2 = s.hh * s.lw, w3 = s.lh * s.hw, w4 = s.lh * s.lw;
            for (int c = lane; c < d.D; c += kLanes) {
              const T g = g_ptr[c];
              const T tgv = g * a;
              const T v1 = s.ok1 ? value[o1 + c] : (T)0;
              const T v2 = s.ok2 ? value[o2 + c] : (T)0;
              const T v3 = s.ok3 ? value[o3 + c] : (T)0;
              const T v4 = s.ok4 ? value[o4 + c] : (T)0;
              if (s.ok1) atomic_add(grad_value + o1 + c, w1 * tgv);
              if (s.ok2) atomic_add(grad_value + o2 + c, w2 * tgv);
              if (s.ok3) atomic_add(grad_value + o3 + c, w3 * tgv);
              if (s.ok4) atomic_add(grad_value + o4 + c, w4 * tgv);
              pa += g * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
              pw += tgv * (s.hh * (v2 - v1) + s.lh * (v4 - v3));
              ph += tgv * (s.hw * (v3 - v1) + s.lw * (v4 - v2));
            }
          }
          pa = group_sum(pa);
          pw = group_sum(pw);
          ph = group_sum(ph);
        }
        if (live && lane == 0) {
          grad_attn[si] = pa;
          grad_loc[si * 2] = (T)W * pw;
          grad_loc[si * 2 + 1] = (T)H * ph;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Sum over the G lanes of a lane-group; every lane ends up with the total.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

template <int G>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (G >= 2) v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  if constexpr (G >= 4) v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  if constexpr (G >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror: lane i <- 7-i within 8
  if constexpr (G >= 16) v += dpp_mov<0x140>(v);  // row_mirror: lane i <- 15-i within 16
  if constexpr (G >= 32) v += __shfl_xor(v, 16, 64);
  if constexpr (G >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

template <int G, int LPT>
__global__ void __launch_bounds__(kBlock, 4)
msda_bwd_lanegroup(const float* __restrict__ grad_out, const float* __restrict__ value,
                   const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                   const float* __restrict__ loc, const float* __restrict__ attn, Dims d,
                   float* __restrict__ grad_value, float* __restrict__ grad_loc,
                   float* __restrict__ grad_attn) {
  constexpr int kPairs = kBlock / G;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* smp_H = reinterpret_cast<int*>(smem);
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
  if (q >= d.Lq) return;

  const int64_t pair = ((int64_t)b * d.Lq + q) * d.M + m;
  const uint32_t pix_bytes = (uint32_t)d.M * d.D * 4u;
  char* rec = rec_base + g * (LP * 32 + 16);

  // record: {lw, lh, a, unused} {o1, o2, o3, o4}
  auto prepare = [&](int s, float lx, float ly, float a) {
    const int W = smp_W[s];
    const Sample<float> sm = make_sample<float>(lx, ly, smp_H[s], W);
    const uint32_t o1 = (uint32_t)(smp_start[s] + sm.h_low * W + sm.w_low) * pix_bytes;
    u32x4 o;
    o[0] = sm.ok1 ? o1 : kOobOffset;
    o[1] = sm.ok2 ? o1 + pix_bytes : kOobOffset;
    o[2] = sm.ok3 ? o1 + (uint32_t)W * pix_bytes : kOobOffset;
    o[3] = sm.ok4 ? o1 + (uint32_t)(W + 1) * pix_bytes : kOobOffset;
    // out-of-range samples (incl. NaN/inf locations) contribute exact zeros, as the reference's skip does
    *reinterpret_cast<float4*>(rec + s * 32) =
        sm.in_range ? make_float4(sm.lw, sm.lh, a, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<u32x4*>(rec + s * 32 + 16) = o;
  };

  if constexpr (LPT == 2 * G) {
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
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const int64_t img_off = (int64_t)b * d.S * d.M * d.D;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(value) + img_off, 0, (int)((uint32_t)d.S * pix_bytes), 0x00020000);
  const uint32_t head_off = (uint32_t)m * d.D * 4u;
  const uint32_t lane_off = (uint32_t)j * 16u;
  char* gv_base = reinterpret_cast<char*>(grad_value + img_off) + head_off + lane_off;
  const float4 go = *reinterpret_cast<const float4*>(grad_out + pair * d.D + 4 * j);

  // One sample: loads, value-gradient atomics, partial sums reduced over the group.
  auto sample = [&](int s, float& ra, float& rw, float& rh) {
    const float4 f = *reinterpret_cast<const float4*>(rec + s * 32);
    const u32x4 o = *reinterpret_cast<const u32x4*>(rec + s * 32 + 16);
    const float lw = f.x, lh = f.y, a = f.z;
    const float hw = 1.f - lw, hh = 1.f - lh;
    const f32x4 r1 = buffer_load_f32x4(rsrc, o[0] + lane_off, head_off);
    const f32x4 r2 = buffer_load_f32x4(rsrc, o[1] + lane_off, head_off);
    const f32x4 r3 = buffer_load_f32x4(rsrc, o[2] + lane_off, head_off);
    const f32x4 r4 = buffer_load_f32x4(rsrc, o[3] + lane_off, head_off);
    const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    float pa = 0.f, pw = 0.f, ph = 0.f;
    const float gq[4] = {go[0], go[1], go[2], go[3]};
    float t1[4], t2[4], t3[4], t4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float v1 = r1[c], v2 = r2[c], v3 = r3[c], v4 = r4[c];
      const float tgv = gq[c] * a;
      pa += gq[c] * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
      pw += tgv * (hh * (v2 - v1) + lh * (v4 - v3));
      ph += tgv * (hw * (v3 - v1) + lw * (v4 - v2));
      t1[c] = w1 * tgv; t2[c] = w2 * tgv; t3[c] = w3 * tgv; t4[c] = w4 * tgv;
    }
    if (o[0] != kOobOffset) {
      float* p = reinterpret_cast<float*>(gv_base + o[0]);
#pragma unroll
      for (int c = 0; c < 4; ++c) atomic_add(p + c, t1[c]);
    }
    if (o[1] != kOobOffset) {
      float* p = reinterpret_cast<float*>(gv_base + o[1]);
#pragma unroll
      for (int c = 0; c < 4; ++c) atomic_add(p + c, t2[c]);
    }
    if (o[2] != kOobOffset) {
      float* p = reinterpret_cast<float*>(gv_base + o[2]);
#pragma unroll
      for (int c = 0; c < 4; ++c) atomic_add(p + c, t3[c]);
    }
    if (o[3] != kOobOffset) {
      float* p = reinterpret_cast<float*>(gv_base + o[3]);
#pragma unroll
      for (int c = 0; c < 4; ++c) atomic_add(p + c, t4[c]);
    }
    ra = group_sum<G>(pa);
    rw = group_sum<G>(pw) * (float)smp_W[s];
    rh = group_sum<G>(ph) * (float)smp_H[s];
  };

  if constexpr (LPT == 2 * G) {
    // lane j owns samples 2j and 2j+1: keep their reduced gradients, store 16 + 8 bytes coalesced
    float4 gl = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 ga = make_float2(0.f, 0.f);
#pragma unroll
    for (int s = 0; s < LPT; ++s) {
      float ra, rw, rh;
      sample(s, ra, rw, rh);
      const bool mine = (j == (s >> 1));
      if ((s & 1) == 0) {
        gl.x = mine ? rw : gl.x; gl.y = mine ? rh : gl.y; ga.x = mine ? ra : ga.x;
      } else {
        gl.z = mine ? rw : gl.z; gl.w = mine ? rh : gl.w; ga.y = mine ? ra : ga.y;
      }
    }
    *reinterpret_cast<float4*>(grad_loc + pair * (2 * LPT) + 4 * j) = gl;
    *reinterpret_cast<float2*>(grad_attn + pair * LPT + 2 * j) = ga;
  } else {
    for (int s = 0; s < LP; ++s) {
      float ra, rw, rh;
      sample(s, ra, rw, rh);
      if (j == s % G) {
        *reinterpret_cast<float2*>(grad_loc + (pair * LP + s) * 2) = make_float2(rw, rh);
        grad_attn[pair * LP + s] = ra;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
static inline bool lanegroup_ok(const Dims& d, int* G_out) {
  if (d.D % 4 != 0) return false;
  const int G = d.D / 4;
  if (G < 1 || G > 64 || (G & (G - 1)) != 0) return false;
  if (d.L * d.P > kMaxLP) return false;
  if ((int64_t)d.S * d.M * d.D * 4 >= (int64_t)kOobOffset) return false;
  if (d.N > 65535) return false;
  const int64_t lds = kLevelTableBytes + (int64_t)(kBlock / G) * (d.L * d.P * 32 + 16);
  if (lds > 64 * 1024) return false;
  *G_out = G;
  return true;
}

template <int G, int LPT>
static int launch_lanegroup(const float* grad_out, const float* value, const int64_t* shapes, const int64_t* lsi,
                            const float* loc, const float* attn, const Dims& d, float* grad_value, float* grad_loc,
                            float* grad_attn, hipStream_t stream) {
  constexpr int kPairs = kBlock / G;
  const size_t lds = kLevelTableBytes + (size_t)kPairs * (d.L * d.P * 32 + 16);
  dim3 grid((unsigned)(d.M * ((d.Lq + kPairs - 1) / kPairs)), (unsigned)d.N);
  hipLaunchKernelGGL((msda_bwd_lanegroup<G, LPT>), grid, dim3(kBlock), lds, stream, grad_out, value, shapes, lsi, loc,
                     attn, d, grad_value, grad_loc, grad_attn);
  return (int)hipGetLastError();
}

template <typename T>
static int launch_generic(const T* grad_out, const T* value, const int64_t* shapes, const int64_t* lsi, const T* loc,
                          const T* attn, const Dims& d, T* grad_value, T* grad_loc, T* grad_attn,
                          hipStream_t stream) {
  const int64_t n_pairs = (int64_t)d.N * d.Lq * d.M;
  const bool half = d.D <= 32;                              // two pairs per wave
  const int per_block = half ? kBlock / 32 : kBlock / 64;
  const int64_t want = (n_pairs + per_block - 1) / per_block;
  const unsigned blocks = (unsigned)(want < 65536 * 16 ? want : 65536 * 16);
  if (half)
    hipLaunchKernelGGL((msda_bwd_generic<T, 1>), dim3(blocks), dim3(kBlock), 0, stream, grad_out, value, shapes, lsi, loc,
                       attn, d, grad_value, grad_loc, grad_attn);
  else
    hipLaunchKernelGGL((msda_bwd_generic<T, 0>), dim3(blocks), dim3(kBlock), 0, stream, grad_out, value, shapes, lsi, loc,
                       attn, d, grad_value, grad_loc, grad_attn);
  return (int)hipGetLastError();
}

template <>
int launch_backward<float>(int variant, const float* grad_out, const float* value, const int64_t* shapes,
                           const int64_t* lsi, const float* loc, const float* attn, const Dims& d,
                           float* grad_value, float* grad_loc, float* grad_attn, hipStream_t stream,
                           const char** kernel_name) {
  int G = 0;
  const bool lg = lanegroup_ok(d, &G);
  // Measured on MI355X (profiles/): the value-gradient atomics dominate and cost one L2 operation per
  // (instruction, cache line) pair.  msda_bwd_generic adds 128 contiguous bytes per instruction (2.05 ms per
  // encoder call), msda_bwd_lanegroup 8 lines of 8 scattered dwords (8.2 ms), msda_bwd_tiled combines the adds
  // in LDS first (0.48 ms).  `auto`: tiled for encoder-style calls, generic otherwise.
  const bool tl = tiled_backward_ok(d);
  constexpr int kBwdWin = 4;            // backward variant 4: msda_bwd_win (value + gradient windows in LDS)
  constexpr int kBwdDec = 5;            // backward variant 5: msda_bwd_dec (decoder-style calls)
  constexpr int kBwdRegions = 6;        // backward variant 6: msda_bwd_regions (destination-side sums, no global atomics)
  constexpr int kBwdDst = 8;            // backward variant 8: msda_bwd_dst (decoder-style calls, destination-side sums in LDS tiles; round 6)
  if (variant == kAuto) {
    // encoder-style calls: msda_bwd_win where the forward calls of the call site have reported near samples
    // (backward_site_choice, msda_fwd_win.hip), msda_bwd_tiled otherwise; generic for everything else
    // decoder-style calls (fp32, D = 32, L = P = 4): msda_bwd_dec, LDS accumulators for the coarse levels
    // ... and msda_bwd_regions where they have reported far ones (backward_site_choice, msda_fwd_win.hip)
    if (tl && d.S >= 1024) {
      const int site = backward_site_choice(d);
      variant = site == 1 ? kBwdWin : site == 2 ? kBwdRegions : kTiled;
    } else {
      variant = dec_backward_ok(d) ? kBwdDec : kGeneric;
    }
  }
  drop_call_context();
  if (variant == kBwdDst && dst_backward_ok(d)) {
    *kernel_name = "msda_bwd_dst";
    return launch_backward_dst(grad_out, value, shapes, lsi, loc, attn, d, grad_value, grad_loc, grad_attn, stream);
  }
  if (variant == kBwdDst) variant = dec_backward_ok(d) ? kBwdDec : kGeneric;
  if (variant == kBwdDec && dec_backward_ok(d)) {
    *kernel_name = "msda_bwd_dec";
    return launch_backward_dec(grad_out, value, shapes, lsi, loc, attn, d, grad_value, grad_loc, grad_attn, stream);
  }
  if (variant == kBwdDec) variant = kGeneric;
  if (variant == kBwdRegions && regions_backward_ok(d)) {
    return launch_backward_regions(grad_out, value, shapes, lsi, loc, attn, d, grad_value, grad_loc, grad_attn, stream, kernel_name);
  }
  if (variant == kBwdRegions) variant = tl ? kTiled : kGeneric;
  constexpr int kBwdWin2 = 7;           // backward variant 7: msda_bwd_win2 (one window set, two workgroups per CU; experiments/)
#ifdef MSDA_EXPERIMENTS
  if (variant == kBwdWin2 && win2_backward_ok(d)) {
    *kernel_name = "msda_bwd_win2";
    return launch_backward_win2(grad_out, value, shapes, lsi, loc, attn, d, grad_value, grad_loc, grad_attn, stream);
  }
#endif
  if (variant == kBwdWin2) variant = kBwdWin;
  if (variant == kBwdWin && win_backward_ok(d)) {
    *kernel_name = "msda_bwd_win";
    return launch_backward_win(grad_out, value, shapes, lsi, loc, attn, d, grad_value, grad_loc, grad_attn, stream);
  }
  if (variant >= kTiled && !tl) variant = kGeneric;
  if (variant >= kTiled) {
    *kernel_name = "msda_bwd_tiled";
    return launch_backward_tiled(grad_out, value, shapes, lsi, loc, attn, d, grad_value, grad_loc, grad_attn, stream);
  }
  if (variant == kLaneGroup && !lg) variant = kGeneric;
  if (variant == kLaneGroup) {
    *kernel_name = "msda_bwd_lanegroup";
#define MSDA_BWD_CASE(GG, LL) \
  return launch_lanegroup<GG, LL>(grad_out, value, shapes, lsi, loc, attn, d, grad_value, grad_loc, grad_attn, stream)
    switch (G) {
      case 1: MSDA_BWD_CASE(1, 0);
      case 2: MSDA_BWD_CASE(2, 0);
      case 4: MSDA_BWD_CASE(4, 0);
      case 8:
        if (d.L * d.P == 16) MSDA_BWD_CASE(8, 16);
        MSDA_BWD_CASE(8, 0);
      case 16: MSDA_BWD_CASE(16, 0);
      case 32: MSDA_BWD_CASE(32, 0);
      default: MSDA_BWD_CASE(64, 0);
    }
#undef MSDA_BWD_CASE
  }
  *kernel_name = "msda_bwd_generic";
  return launch_generic<float>(grad_out, value, shapes, lsi, loc, attn, d, grad_value, grad_loc, grad_attn, stream);
}

template <>
int launch_backward<double>(int /*variant*/, const double* grad_out, const double* value, const int64_t* shapes,
                            const int64_t* lsi, const double* loc, const double* attn, const Dims& d,
                            double* grad_value, double* grad_loc, double* grad_attn, hipStream_t stream,
                            const char** kernel_name) {
  drop_call_context();
  *kernel_name = "msda_bwd_generic";
  return launch_generic<double>(grad_out, value, shapes, lsi, loc, attn, d, grad_value, grad_loc, grad_attn, stream);
}

}  // namespace msda
