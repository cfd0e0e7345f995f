// Dynamic (per-instance) mask head of UNINEXT's CondInst branch for gfx950 -- see include/dynmask_hip.h.
//
// Work split: a workgroup owns 1024 pixels of one image (4 per thread, coalesced along x) and keeps their 8
// mask-feature channels in registers (as 16 channel pairs); it then walks a strided subset of the image's
// instances.  The 169 controller parameters of the current instance sit in LDS (double-buffered, one float per
// thread, transposed to [in][out]) and are read with broadcast ds_read_b128 -- a weight row (8 output channels of one
// input) is fetched once and used for the 4 pixels.  Per (pixel, instance): 152 FMAs ((8+2)*8 + 8*8 + 8), two ReLUs,
// one 4-byte store; nothing of size n_inst x (C+2) x H x W is ever materialised.  The two hidden layers run on
// OUTPUT-CHANNEL PAIRS with v_pk_fma_f32 (72 packed + 8 scalar FMAs per pixel and instance instead of 152 scalar
// ones; each output still sums its inputs in the reference order, so the result is bitwise the scalar kernel's).
// fp32 on the VALU: the f32 MFMA runs at the same 157 TF vector rate (MI355X_MICROARCH.md); the roofline that binds
// is fp32 VALU issue (4.6 G FMA at 1800 instances x 100x167).
#include "../../include/dynmask_hip.h"

#include <atomic>
#include <cstddef>
#include <cstdlib>

#include "msda_common.hpp"

namespace dynmask {

constexpr int kThreads = 256, kPx = 4;      // pixels per thread
constexpr int kC = 8, kCh = 8;              // mask-feature channels, dynamic channels

// LDS copy of one instance's parameters, input-major ([in][out]) so that a weight row holds the 8 output channels of
// one input: the layers then run on OUTPUT-CHANNEL PAIRS with v_pk_fma_f32 (the weight pair comes straight out of the
// ds_read_b128, the input is an op_sel broadcast) -- half the VALU instructions, same registers, and every output
// still accumulates its inputs in the reference order (bitwise equal to the scalar form).
struct __attribute__((aligned(16))) InstParams {
  float w0[kC + 2][kCh];   // in 0,1 = relative x, y (unused without rel_coord), 2.. = feature channels
  float w1[kCh][kCh];
  float w2[8];
  float b0[8], b1[8];
  float b2, ix, iy, pad;
};

typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <bool REL>
__device__ __forceinline__ void pkfma_body(const float* __restrict__ feats, const float* __restrict__ inst_xy,
                                           const float* __restrict__ params, int inst_first, int inst_count, int H, int W,
                                           int stride, float* __restrict__ out, const int bx, const int by, const int gy) {
  constexpr int kIn = REL ? kC + 2 : kC;
  constexpr int kNumParams = kIn * kCh + kCh * kCh + kCh + kCh + kCh + 1;
  __shared__ InstParams sp[2];
  const int tid = threadIdx.x;
  const int HW = H * W;
  const int p0 = bx * (kThreads * kPx) + tid;

  // this thread's pixels: features in registers, pixel-centre coordinates in input pixels
  // kept as channel PAIRS: a v_pk_fma_f32 source is a 64-bit register tuple, the broadcast picks a half with op_sel
  f32x2v f2[kPx][kC / 2], lxy[kPx];
#pragma unroll
  for (int k = 0; k < kPx; ++k) {
    const int p = p0 + k * kThreads;
    const int pc = p < HW ? p : HW - 1;
#pragma unroll
    for (int c = 0; c < kC; ++c) f2[k][c >> 1][c & 1] = feats[(size_t)c * HW + pc];
    const int py = pc / W, px = pc - py * W;
    lxy[k] = f32x2v{(float)(px * stride + stride / 2), (float)(py * stride + stride / 2)};
  }
  // the feature loads complete HERE: vmcnt retires in order, so a wait for them inside the instance loop would also
  // wait for the parameter load of the next instance issued at the top of every iteration
#pragma unroll
  for (int k = 0; k < kPx; ++k)
#pragma unroll
    for (int c = 0; c < kC / 2; ++c) asm volatile("" : "+v"(f2[k][c]));

  // Staging of an instance's parameters, one float per thread, reference layout ([out][in]) -> [in][out]: the slot a
  // thread fills does not depend on the instance, so it is resolved once; the global load of the NEXT instance is
  // issued at the top of an iteration and its LDS write sits at the bottom (the wave never waits on the load).
  int slot = -1;
  if (tid < kNumParams) {
    int t = tid;
    if (t < kIn * kCh) slot = (int)(offsetof(InstParams, w0) / 4) + ((REL ? 0 : 2) + t % kIn) * kCh + t / kIn;
    else if ((t -= kIn * kCh) < kCh * kCh) slot = (int)(offsetof(InstParams, w1) / 4) + (t % kCh) * kCh + t / kCh;
    else if ((t -= kCh * kCh) < kCh) slot = (int)(offsetof(InstParams, w2) / 4) + t;
    else if ((t -= kCh) < kCh) slot = (int)(offsetof(InstParams, b0) / 4) + t;
    else if ((t -= kCh) < kCh) slot = (int)(offsetof(InstParams, b1) / 4) + t;
    else slot = (int)(offsetof(InstParams, b2) / 4);
  }
  float pv = 0.f, pix = 0.f, piy = 0.f;
  auto fetch = [&](int inst) {
    if (slot >= 0) pv = params[(size_t)inst * kNumParams + tid];
    if (tid == kThreads - 1) { pix = inst_xy[(size_t)inst * 2]; piy = inst_xy[(size_t)inst * 2 + 1]; }
  };
  auto commit = [&](InstParams& dst) {
    if (slot >= 0) reinterpret_cast<float*>(&dst)[slot] = pv;
    if (tid == kThreads - 1) { dst.ix = pix; dst.iy = piy; }
  };
  // one input row of a layer: acc[k][c] (channel pair c of pixel k) = w[c] * x[k] + (first ? bias[c] : acc[k][c]).
  // The weights of row r + 1 are requested before row r is multiplied (Wrow ring of two); sched_barriers keep the
  // scheduler from hoisting all 43 ds_read_b128 of an instance to the top (172 VGPRs -> spills).
  struct Wrow { f32x4v a, b; };
  auto fma_row = [](const Wrow& wr, const f32x2v (&xp)[kPx], int half, f32x2v (&acc)[kPx][4], const Wrow* bias, bool relu) {
    const f32x2v w[4] = {wr.a.xy, wr.a.zw, wr.b.xy, wr.b.zw};
    if (bias) {
      const f32x2v b[4] = {bias->a.xy, bias->a.zw, bias->b.xy, bias->b.zw};
#pragma unroll
      for (int k = 0; k < kPx; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[k][c] = __builtin_elementwise_fma(w[c], (half ? xp[k].yy : xp[k].xx), b[c]);
    } else {
#pragma unroll
      for (int k = 0; k < kPx; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[k][c] = __builtin_elementwise_fma(w[c], (half ? xp[k].yy : xp[k].xx), acc[k][c]);
    }
    // the IR-level code sinking ignores sched_barrier: pin the row's results where they are computed (the ReLU of
    // the last row sits before the pin, where the compiler still knows the value is an fma result: no canonicalize)
#pragma unroll
    for (int k = 0; k < kPx; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (relu) acc[k][c] = __builtin_elementwise_max(acc[k][c], f32x2v{0.f, 0.f});
        asm volatile("" : "+v"(acc[k][c]));
      }
  };

  int i = by;
  if (i < inst_count) { fetch(inst_first + i); commit(sp[0]); }
  int buf = 0;
  for (; i < inst_count; i += gy) {
    __syncthreads();                                   // sp[buf] is complete, sp[buf^1] is free
    const bool more = i + gy < inst_count;
    if (more) fetch(inst_first + i + gy);
    const InstParams& P = sp[buf];
    constexpr int kFirst = REL ? 0 : 2;
    const float* wbase = &P.w0[0][0];                  // rows 0..9 = w0, 10..17 = w1 (contiguous)
    auto ld = [&](int r) { return Wrow{*reinterpret_cast<const f32x4v*>(wbase + 8 * r), *reinterpret_cast<const f32x4v*>(wbase + 8 * r + 4)}; };
    const Wrow bias0 = Wrow{*reinterpret_cast<const f32x4v*>(&P.b0[0]), *reinterpret_cast<const f32x4v*>(&P.b0[4])};
    Wrow wcur = ld(kFirst), wnxt;
    f32x2v h0[kPx][4];
#pragma unroll
    for (int r = kFirst; r < kC + 2; ++r) {
      wnxt = ld(r + 1);                                // r + 1 == 10 is the first row of w1
      f32x2v xp[kPx];
      if (r < 2) {
        const f32x2v ixy = f32x2v{P.ix, P.iy};
#pragma unroll
        for (int k = 0; k < kPx; ++k) xp[k] = ixy - lxy[k];
      } else {
#pragma unroll
        for (int k = 0; k < kPx; ++k) xp[k] = f2[k][(r - 2) >> 1];
      }
      __builtin_amdgcn_sched_barrier(0);
      fma_row(wcur, xp, r & 1, h0, r == kFirst ? &bias0 : nullptr, r == kC + 1);
      __builtin_amdgcn_sched_barrier(0);
      wcur = wnxt;
    }
    const Wrow bias1 = Wrow{*reinterpret_cast<const f32x4v*>(&P.b1[0]), *reinterpret_cast<const f32x4v*>(&P.b1[4])};
    f32x2v h1[kPx][4];
#pragma unroll
    for (int c = 0; c < kCh; ++c) {
      if (c + 1 < kCh) wnxt = ld(kC + 2 + c + 1);
      f32x2v xp[kPx];
#pragma unroll
      for (int k = 0; k < kPx; ++k) xp[k] = h0[k][c >> 1];
      __builtin_amdgcn_sched_barrier(0);
      fma_row(wcur, xp, c & 1, h1, c == 0 ? &bias1 : nullptr, c == kCh - 1);
      __builtin_amdgcn_sched_barrier(0);
      wcur = wnxt;
    }
    const f32x4v w2a = *reinterpret_cast<const f32x4v*>(&P.w2[0]), w2b = *reinterpret_cast<const f32x4v*>(&P.w2[4]);
    const float w2[8] = {w2a.x, w2a.y, w2a.z, w2a.w, w2b.x, w2b.y, w2b.z, w2b.w};
    float y[kPx];
#pragma unroll
    for (int k = 0; k < kPx; ++k) {
      y[k] = P.b2;
#pragma unroll
      for (int o = 0; o < kCh; ++o) y[k] = fmaf(w2[o], h1[k][o >> 1][o & 1], y[k]);
    }
    float* o_ptr = out + (size_t)(inst_first + i) * HW;
#pragma unroll
    for (int k = 0; k < kPx; ++k) {
      const int p = p0 + k * kThreads;
      if (p < HW) __builtin_nontemporal_store(y[k], o_ptr + p);
    }
    if (more) commit(sp[buf ^ 1]);
    buf ^= 1;
  }
}

// ---- MFMA form of the same chain ---------------------------------------------------------------------------------
// v_mfma_f32_4x4x1_16b_f32 multiplies sixteen independent 4x1 by 1x4 blocks per instruction; block = lane / 4.  Laid
// out with OUTPUT CHANNELS as the M rows (two half-chains: channels 0-3 and 4-7), the lane's own pixel as the N column
// (n = lane % 4, so lane = pixel within the wave: loads and stores stay coalesced) and one input channel per
// instruction (K = 1), nothing is padded: 36 MFMAs per 64 (pixel, instance) pairs = (10 + 8) * 8 * 64 MACs exactly.
//   A  (4x1, lane m = lane % 4 of block 0, broadcast to the other blocks with CBSZ = 4): W[out = m (+4)][in = k]
//   B  (1x4, lane n): the lane's pixel's input k -- a feature register, or the ReLU'd accumulator of layer 0
//   C/D (4 VGPRs): row i of the block = output channel i (+4) of the lane's pixel; initialised with the bias
// so the accumulators of layer 0 ARE the B operands of layer 1 (no cross-lane traffic), and every output sums its
// inputs in the reference order.  The 8 -> 1 layer has one useful M row: it stays on the VALU (8 FMAs per pixel).
// The 32x32x2 shape of the same instruction family would carry the 8 channels in a 32-wide N (or M): a quarter of
// the array at best, on a unit whose fp32 peak equals the packed VALU's.
struct __attribute__((aligned(16))) InstParamsM {
  float w0[kCh][12];       // [out][in], rows padded to 48 bytes: in 0,1 = relative x, y; 2..9 = feature channels
  float w1[kCh][kCh];      // [out][in]
  float b0[8], b1[8], w2[8];
  float b2, ix, iy, pad;
};

template <bool REL, int Q>
__device__ __forceinline__ void mfma_body(const float* __restrict__ feats, const float* __restrict__ inst_xy,
                                          const float* __restrict__ params, int inst_first, int inst_count, int H, int W,
                                          int stride, float* __restrict__ out, const int bx, const int by, const int gy) {
  constexpr int kIn = REL ? kC + 2 : kC;
  constexpr int kNumParams = kIn * kCh + kCh * kCh + kCh + kCh + kCh + 1;
  __shared__ InstParamsM sp[2];
  const int tid = threadIdx.x;
  const int HW = H * W;
  const int p0 = bx * (kThreads * Q) + tid;
  const int m = tid & 3;

  float f[Q][kC], lx[Q], ly[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int p = p0 + q * kThreads;
    const int pc = p < HW ? p : HW - 1;
#pragma unroll
    for (int c = 0; c < kC; ++c) f[q][c] = feats[(size_t)c * HW + pc];
    const int py = pc / W, px = pc - py * W;
    lx[q] = (float)(px * stride + stride / 2);
    ly[q] = (float)(py * stride + stride / 2);
  }
#pragma unroll
  for (int q = 0; q < Q; ++q)       // feature loads complete before the loop (see dynmask_fwd)
#pragma unroll
    for (int c = 0; c < kC; ++c) asm volatile("" : "+v"(f[q][c]));

  // staging as in dynmask_fwd (slot resolved once, load at the top of an iteration, LDS write at the bottom);
  // the reference layout ([out][in]) is kept
  int slot = -1;
  if (tid < kNumParams) {
    int t = tid;
    if (t < kIn * kCh) slot = (int)(offsetof(InstParamsM, w0) / 4) + (t / kIn) * 12 + (REL ? 0 : 2) + t % kIn;
    else if ((t -= kIn * kCh) < kCh * kCh) slot = (int)(offsetof(InstParamsM, w1) / 4) + t;
    else if ((t -= kCh * kCh) < kCh) slot = (int)(offsetof(InstParamsM, w2) / 4) + t;
    else if ((t -= kCh) < kCh) slot = (int)(offsetof(InstParamsM, b0) / 4) + t;
    else if ((t -= kCh) < kCh) slot = (int)(offsetof(InstParamsM, b1) / 4) + t;
    else slot = (int)(offsetof(InstParamsM, b2) / 4);
  }
  float pv = 0.f, pix = 0.f, piy = 0.f;
  auto fetch = [&](int inst) {
    if (slot >= 0) pv = params[(size_t)inst * kNumParams + tid];
    if (tid == kThreads - 1) { pix = inst_xy[(size_t)inst * 2]; piy = inst_xy[(size_t)inst * 2 + 1]; }
  };
  auto commit = [&](InstParamsM& dst) {
    if (slot >= 0) reinterpret_cast<float*>(&dst)[slot] = pv;
    if (tid == kThreads - 1) { dst.ix = pix; dst.iy = piy; }
  };
  auto ld4 = [](const float* p) { return *reinterpret_cast<const f32x4v*>(p); };

  int i = by;
  if (i < inst_count) { fetch(inst_first + i); commit(sp[0]); }
  int buf = 0;
  for (; i < inst_count; i += gy) {
    __syncthreads();
    const bool more = i + gy < inst_count;
    if (more) fetch(inst_first + i + gy);
    const InstParamsM& P = sp[buf];
    // A operands: weight rows m and m + 4 of both layers (only block 0's lanes are read by the MFMA)
    f32x4v a0[2][3], a1[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int j = 0; j < 3; ++j) a0[h][j] = ld4(&P.w0[m + 4 * h][4 * j]);
#pragma unroll
      for (int j = 0; j < 2; ++j) a1[h][j] = ld4(&P.w1[m + 4 * h][4 * j]);
    }
    f32x4v h0[Q][2];
    {
      const f32x4v blo = ld4(&P.b0[0]), bhi = ld4(&P.b0[4]);
#pragma unroll
      for (int q = 0; q < Q; ++q) { h0[q][0] = blo; h0[q][1] = bhi; }
    }
    const float ix = P.ix, iy = P.iy;
#pragma unroll
    for (int k = REL ? 0 : 2; k < kC + 2; ++k) {
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const float x = k == 0 ? ix - lx[q] : (k == 1 ? iy - ly[q] : f[q][k - 2]);
#pragma unroll
        for (int h = 0; h < 2; ++h)
          h0[q][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[h][k >> 2][k & 3], x, h0[q][h], 4, 0, 0);
      }
    }
    f32x4v h1[Q][2];
    {
      const f32x4v blo = ld4(&P.b1[0]), bhi = ld4(&P.b1[4]);
#pragma unroll
      for (int q = 0; q < Q; ++q) { h1[q][0] = blo; h1[q][1] = bhi; }
    }
#pragma unroll
    for (int c = 0; c < kCh; ++c) {
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const float x = fmaxf(h0[q][c >> 2][c & 3], 0.f);
#pragma unroll
        for (int h = 0; h < 2; ++h)
          h1[q][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[h][c >> 2][c & 3], x, h1[q][h], 4, 0, 0);
      }
    }
    const f32x4v w2a = ld4(&P.w2[0]), w2b = ld4(&P.w2[4]);
    const float b2 = P.b2;
    float* o_ptr = out + (size_t)(inst_first + i) * HW;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      float y = b2;
#pragma unroll
      for (int o = 0; o < kCh; ++o) y = fmaf((o < 4 ? w2a : w2b)[o & 3], fmaxf(h1[q][o >> 2][o & 3], 0.f), y);
      const int p = p0 + q * kThreads;
      if (p < HW) __builtin_nontemporal_store(y, o_ptr + p);
    }
    if (more) commit(sp[buf ^ 1]);
    buf ^= 1;
  }
}

template <bool REL>
__global__ void __launch_bounds__(kThreads, 2)
dynmask_fwd(const float* __restrict__ feats, const float* __restrict__ inst_xy, const float* __restrict__ params,
            int inst_first, int inst_count, int H, int W, int stride, float* __restrict__ out) {
  pkfma_body<REL>(feats, inst_xy, params, inst_first, inst_count, H, W, stride, out, blockIdx.x, blockIdx.y, gridDim.y);
}

template <bool REL, int Q>
__global__ void __launch_bounds__(kThreads, Q <= 2 ? 2 : 1)
dynmask_fwd_mfma(const float* __restrict__ feats, const float* __restrict__ inst_xy, const float* __restrict__ params,
                 int inst_first, int inst_count, int H, int W, int stride, float* __restrict__ out) {
  mfma_body<REL, Q>(feats, inst_xy, params, inst_first, inst_count, H, W, stride, out, blockIdx.x, blockIdx.y, gridDim.y);
}

// aligned_bilinear (ddetrs_dn.py:1174-1196): replicate-pad by one, interpolate to (f*h+1, f*w+1) with
// align_corners=True (source coordinate = i / f), replicate-pad f/2 on the top/left, crop to (f*h, f*w).
// One workgroup per (image, band of kRows output rows): row index arithmetic is uniform, the source rows stay in
// L1, stores are coalesced.
constexpr int kRows = 32;
__global__ void __launch_bounds__(kThreads)
aligned_bilinear_kernel(const float* __restrict__ in, int h, int w, int factor, float* __restrict__ out) {
  const int oh = factor * h, ow = factor * w;
  const int bands = (oh + kRows - 1) / kRows;
  const unsigned img = blockIdx.x / (unsigned)bands;
  const int y_first = (int)(blockIdx.x - img * (unsigned)bands) * kRows;
  const float inv = 1.0f / (float)factor;
  const float* src = in + (size_t)img * h * w;
  float* dst = out + (size_t)img * oh * ow;
  // the band is walked as one flat list of x-pairs (full lanes whatever the row length; 8-byte stores)
  const int ow2 = (ow + 1) / 2;
  const int rows = min(kRows, oh - y_first);
  for (int p = threadIdx.x; p < rows * ow2; p += kThreads) {
    const int yy = p / ow2, x = 2 * (p - yy * ow2);
    const int y = y_first + yy;
    const int iy = max(y - factor / 2, 0);
    const float sy = (float)iy * inv;
    const int y0 = (int)sy;
    const float fy = sy - (float)y0;
    const float* r0 = src + (size_t)min(y0, h - 1) * w;
    const float* r1 = src + (size_t)min(y0 + 1, h - 1) * w;
    float res[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int ix = max(x + e - factor / 2, 0);
      const float sx = (float)ix * inv;
      const int x0 = (int)sx;
      const float fx = sx - (float)x0;
      const int x0c = min(x0, w - 1), x1c = min(x0 + 1, w - 1);
      const float top = r0[x0c] + (r0[x1c] - r0[x0c]) * fx, bot = r1[x0c] + (r1[x1c] - r1[x0c]) * fx;
      res[e] = top + (bot - top) * fy;
    }
    float* o = dst + (size_t)y * ow + x;
    if (x + 1 < ow && (((size_t)y * ow + x) & 1) == 0 && ((reinterpret_cast<uintptr_t>(dst) & 7) == 0)) {
      __builtin_nontemporal_store(msda::f32x2{res[0], res[1]}, reinterpret_cast<msda::f32x2*>(o));
    } else {
      __builtin_nontemporal_store(res[0], o);
      if (x + 1 < ow) __builtin_nontemporal_store(res[1], o + 1);
    }
  }
}

// kernel choice of dynmask_hip_forward_f32: 0 = auto, 1 = packed-FMA VALU kernel, 2 / 3 = MFMA kernel with 2 / 4
// pixels per lane.  DYNMASK_HIP_VARIANT seeds it; dynmask_hip_set_variant overrides.
constexpr int kNumVariants = 4, kAutoVariant = 1;
std::atomic<int> g_variant{-1};
std::atomic<const char*> g_last_kernel{""};
int current_variant() {
  int v = g_variant.load(std::memory_order_relaxed);
  if (v < 0) {
    v = msda::ab_env_int("DYNMASK_HIP_VARIANT", 0);
    if (v < 0 || v >= kNumVariants) v = 0;
    g_variant.store(v, std::memory_order_relaxed);
  }
  return v == 0 ? kAutoVariant : v;
}

}  // namespace dynmask

extern "C" {

int dynmask_set_error(int code, const char* what);   // msda_capi.hip

int dynmask_hip_set_variant(int variant) {
  if (variant < 0 || variant >= dynmask::kNumVariants) return dynmask_set_error(DYNMASK_ERR_BAD_DIMS, "dynmask: unknown kernel variant");
  dynmask::g_variant.store(variant, std::memory_order_relaxed);
  return 0;
}

const char* dynmask_hip_last_kernel(void) { return dynmask::g_last_kernel.load(std::memory_order_relaxed); }

int dynmask_hip_forward_f32(const float* mask_feats, const float* inst_xy, const float* params, const int* num_insts,
                            int batch, int channels, int H, int W, int stride, int rel_coord, float* out_logits,
                            void* stream) {
  if (batch < 0 || H <= 0 || W <= 0 || stride <= 0) return dynmask_set_error(DYNMASK_ERR_BAD_DIMS, "dynmask: bad dimensions");
  if (channels != dynmask::kC) return dynmask_set_error(DYNMASK_ERR_UNSUPPORTED, "dynmask: only 8 mask-feature channels");
  if (batch == 0) return 0;
  if (!num_insts) return dynmask_set_error(DYNMASK_ERR_NULL_POINTER, "dynmask: null pointer argument");
  const int HW = H * W;
  const unsigned chunks = (unsigned)((HW + dynmask::kThreads * dynmask::kPx - 1) / (dynmask::kThreads * dynmask::kPx));
  int first = 0;
  for (int b = 0; b < batch; ++b) {
    const int n = num_insts[b];
    if (n < 0) return dynmask_set_error(DYNMASK_ERR_BAD_DIMS, "dynmask: negative instance count");
    if (n > 0) {
      if (!mask_feats || !inst_xy || !params || !out_logits)
        return dynmask_set_error(DYNMASK_ERR_NULL_POINTER, "dynmask: null pointer argument");
      const float* f = mask_feats + (size_t)b * dynmask::kC * HW;
      const int variant = dynmask::current_variant();
      if (variant >= 2) {
        const int Q = variant == 3 ? 4 : 2;
        const unsigned mchunks = (unsigned)((HW + dynmask::kThreads * Q - 1) / (dynmask::kThreads * Q));
        unsigned mgroups = 2048u / mchunks;
        mgroups = mgroups < 1u ? 1u : (mgroups > (unsigned)n ? (unsigned)n : mgroups);
        const dim3 grid(mchunks, mgroups), block(dynmask::kThreads);
#define DYNMASK_LAUNCH_MFMA(REL, QQ) hipLaunchKernelGGL((dynmask::dynmask_fwd_mfma<REL, QQ>), grid, block, 0, (hipStream_t)stream, \
                                                        f, inst_xy, params, first, n, H, W, stride, out_logits)
        if (rel_coord) { if (Q == 4) DYNMASK_LAUNCH_MFMA(true, 4); else DYNMASK_LAUNCH_MFMA(true, 2); }
        else { if (Q == 4) DYNMASK_LAUNCH_MFMA(false, 4); else DYNMASK_LAUNCH_MFMA(false, 2); }
#undef DYNMASK_LAUNCH_MFMA
        dynmask::g_last_kernel.store(Q == 4 ? "dynmask_fwd_mfma_q4" : "dynmask_fwd_mfma_q2", std::memory_order_relaxed);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return dynmask_set_error((int)e, hipGetErrorString(e));
        first += n;
        continue;
      }
      dynmask::g_last_kernel.store("dynmask_fwd_pkfma", std::memory_order_relaxed);
      // ~2048 workgroups in flight; every workgroup handles ceil(n / groups) instances of its pixel chunk
      unsigned groups = 2048u / chunks;
      groups = groups < 1u ? 1u : (groups > (unsigned)n ? (unsigned)n : groups);
      if (rel_coord)
        hipLaunchKernelGGL(dynmask::dynmask_fwd<true>, dim3(chunks, groups), dim3(dynmask::kThreads), 0,
                           (hipStream_t)stream, f, inst_xy, params, first, n, H, W, stride, out_logits);
      else
        hipLaunchKernelGGL(dynmask::dynmask_fwd<false>, dim3(chunks, groups), dim3(dynmask::kThreads), 0,
                           (hipStream_t)stream, f, inst_xy, params, first, n, H, W, stride, out_logits);
      const hipError_t e = hipGetLastError();
      if (e != hipSuccess) return dynmask_set_error((int)e, hipGetErrorString(e));
    }
    first += n;
  }
  return 0;
}

int aligned_bilinear_hip_f32(const float* in, int n, int h, int w, int factor, float* out, void* stream) {
  if (n < 0 || h <= 0 || w <= 0 || factor < 1) return dynmask_set_error(DYNMASK_ERR_BAD_DIMS, "aligned_bilinear: bad dimensions");
  if (n == 0) return 0;
  if (!in || !out) return dynmask_set_error(DYNMASK_ERR_NULL_POINTER, "aligned_bilinear: null pointer argument");
  const long long blocks = (long long)n * ((factor * h + dynmask::kRows - 1) / dynmask::kRows);
  if (blocks >= (1ll << 31)) return dynmask_set_error(DYNMASK_ERR_BAD_DIMS, "aligned_bilinear: too many rows");
  hipLaunchKernelGGL(dynmask::aligned_bilinear_kernel, dim3((unsigned)blocks), dim3(dynmask::kThreads), 0,
                     (hipStream_t)stream, in, h, w, factor, out);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : dynmask_set_error((int)e, hipGetErrorString(e));
}

}  // extern "C"
