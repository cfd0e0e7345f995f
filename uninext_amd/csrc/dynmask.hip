// Dynamic (per-instance) mask head of UNINEXT's CondInst branch for gfx950 -- see include/dynmask_hip.h.
//
// Work split: a workgroup owns 1024 pixels of one image (4 per thread, coalesced along x) and keeps their 8
// mask-feature channels in registers (32 VGPRs); it then walks a strided subset of the image's instances.  The
// 169 controller parameters of the current instance sit in LDS (double-buffered, one float per thread) and are
// read with broadcast ds_read_b128 -- a weight row is fetched once and used for the 4 pixels.  Per (pixel,
// instance): 152 FMAs ((8+2)*8 + 8*8 + 8), two ReLUs, one 4-byte store; nothing of size n_inst x (C+2) x H x W is
// ever materialised.  fp32 FMAs on the VALU: the f32 MFMA runs at the same 157 TF vector rate
// (MI355X_MICROARCH.md), so there is nothing to gain from the matrix core at this precision; the roofline that
// binds is fp32 VALU issue (4.6 G FMA at 1800 instances x 100x167).
#include "../../include/dynmask_hip.h"

#include "msda_common.hpp"

namespace dynmask {

constexpr int kThreads = 256, kPx = 4;      // pixels per thread
constexpr int kC = 8, kCh = 8;              // mask-feature channels, dynamic channels

// LDS copy of one instance's parameters, rows padded to 12 floats so every row starts 16-byte aligned
struct __attribute__((aligned(16))) InstParams {
  float w0[kCh][12];   // [out][in]: in 0,1 = relative x, y (or unused), 2.. = feature channels
  float w1[kCh][8];
  float w2[8];
  float b0[8], b1[8];
  float b2, ix, iy, pad;
};

template <bool REL>
__global__ void __launch_bounds__(kThreads, 2)
dynmask_fwd(const float* __restrict__ feats, const float* __restrict__ inst_xy, const float* __restrict__ params,
            int inst_first, int inst_count, int H, int W, int stride, float* __restrict__ out) {
  constexpr int kIn = REL ? kC + 2 : kC;
  constexpr int kNumParams = kIn * kCh + kCh * kCh + kCh + kCh + kCh + 1;
  __shared__ InstParams sp[2];
  const int tid = threadIdx.x;
  const int HW = H * W;
  const int p0 = blockIdx.x * (kThreads * kPx) + tid;

  // this thread's pixels: features in registers, pixel-centre coordinates in input pixels
  float f[kPx][kC], lx[kPx], ly[kPx];
#pragma unroll
  for (int k = 0; k < kPx; ++k) {
    const int p = p0 + k * kThreads;
    const int pc = p < HW ? p : HW - 1;
#pragma unroll
    for (int c = 0; c < kC; ++c) f[k][c] = feats[(size_t)c * HW + pc];
    const int py = pc / W, px = pc - py * W;
    lx[k] = (float)(px * stride + stride / 2);
    ly[k] = (float)(py * stride + stride / 2);
  }

  auto stage = [&](int inst, InstParams& dst) {   // one float per thread, reference layout -> padded rows
    if (tid < kNumParams) {
      const float v = params[(size_t)inst * kNumParams + tid];
      int t = tid;
      if (t < kIn * kCh) { dst.w0[t / kIn][(REL ? 0 : 2) + t % kIn] = v; }
      else if ((t -= kIn * kCh) < kCh * kCh) { dst.w1[t / kCh][t % kCh] = v; }
      else if ((t -= kCh * kCh) < kCh) { dst.w2[t] = v; }
      else if ((t -= kCh) < kCh) { dst.b0[t] = v; }
      else if ((t -= kCh) < kCh) { dst.b1[t] = v; }
      else { dst.b2 = v; }
    }
    if (tid == kThreads - 1) { dst.ix = inst_xy[(size_t)inst * 2]; dst.iy = inst_xy[(size_t)inst * 2 + 1]; }
  };

  int i = blockIdx.y;
  if (i < inst_count) stage(inst_first + i, sp[0]);
  int buf = 0;
  for (; i < inst_count; i += gridDim.y) {
    __syncthreads();                                   // sp[buf] is complete, sp[buf^1] is free
    if (i + (int)gridDim.y < inst_count) stage(inst_first + i + gridDim.y, sp[buf ^ 1]);
    const InstParams& P = sp[buf];
    float h0[kPx][kCh];
#pragma unroll
    for (int o = 0; o < kCh; ++o) {
      const float4 wa = *reinterpret_cast<const float4*>(&P.w0[o][0]);
      const float4 wb = *reinterpret_cast<const float4*>(&P.w0[o][4]);
      const float2 wc = *reinterpret_cast<const float2*>(&P.w0[o][8]);
      const float bo = P.b0[o];
#pragma unroll
      for (int k = 0; k < kPx; ++k) {
        float a = bo;
        if (REL) { a = fmaf(wa.x, P.ix - lx[k], a); a = fmaf(wa.y, P.iy - ly[k], a); }
        a = fmaf(wa.z, f[k][0], a); a = fmaf(wa.w, f[k][1], a);
        a = fmaf(wb.x, f[k][2], a); a = fmaf(wb.y, f[k][3], a); a = fmaf(wb.z, f[k][4], a); a = fmaf(wb.w, f[k][5], a);
        a = fmaf(wc.x, f[k][6], a); a = fmaf(wc.y, f[k][7], a);
        h0[k][o] = fmaxf(a, 0.f);
      }
    }
    float y[kPx];
#pragma unroll
    for (int k = 0; k < kPx; ++k) y[k] = P.b2;
#pragma unroll
    for (int o = 0; o < kCh; ++o) {
      const float4 wa = *reinterpret_cast<const float4*>(&P.w1[o][0]);
      const float4 wb = *reinterpret_cast<const float4*>(&P.w1[o][4]);
      const float bo = P.b1[o], w2o = P.w2[o];
#pragma unroll
      for (int k = 0; k < kPx; ++k) {
        float a = bo;
        a = fmaf(wa.x, h0[k][0], a); a = fmaf(wa.y, h0[k][1], a); a = fmaf(wa.z, h0[k][2], a); a = fmaf(wa.w, h0[k][3], a);
        a = fmaf(wb.x, h0[k][4], a); a = fmaf(wb.y, h0[k][5], a); a = fmaf(wb.z, h0[k][6], a); a = fmaf(wb.w, h0[k][7], a);
        y[k] = fmaf(w2o, fmaxf(a, 0.f), y[k]);
      }
    }
    float* o_ptr = out + (size_t)(inst_first + i) * HW;
#pragma unroll
    for (int k = 0; k < kPx; ++k) {
      const int p = p0 + k * kThreads;
      if (p < HW) __builtin_nontemporal_store(y[k], o_ptr + p);
    }
    buf ^= 1;
  }
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

}  // namespace dynmask

extern "C" {

int dynmask_set_error(int code, const char* what);   // msda_capi.hip

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
      // ~2048 workgroups in flight; every workgroup handles ceil(n / groups) instances of its pixel chunk
      unsigned groups = 2048u / chunks;
      groups = groups < 1u ? 1u : (groups > (unsigned)n ? (unsigned)n : groups);
      const float* f = mask_feats + (size_t)b * dynmask::kC * HW;
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
