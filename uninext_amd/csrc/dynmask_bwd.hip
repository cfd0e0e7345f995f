// Backward of the dynamic (per-instance) mask head and of aligned_bilinear -- include/dynmask_hip.h, gfx950.
//
// What is differentiated: DDETRSegmUniDN.dynamic_mask_with_coords (projects/UNINEXT/uninext/models/ddetrs_dn.py:755-844) between
// "build mask_head_inputs" and "upsample" -- relative coordinates (:765-784), the three grouped 1x1 convolutions of
// mask_heads_forward (:734-752) on the parameters of parse_dynamic_params (:1148-1171) -- and aligned_bilinear (:1174-1196).
// BASELINE configs[4] trains this head (BoxInst on): a few dozen matched instances per image at 100 x 168, i.e. a small
// problem (~1 GFLOP) whose PyTorch composition is ~40 launches and materialises [n, 8, H W] activations three times over.
//
// Per (instance i, pixel p) with g = grad_logits[i, p] and x = (rel_x, rel_y, f_0 .. f_7):
//     h0 = relu(W0 x + b0)   h1 = relu(W1 h0 + b1)   y = w2 . h1 + b2                                  (recomputed)
//     g_b2 += g       g_w2 += g h1       g_h1 = g w2 [h1 > 0]
//     g_b1 += g_h1    g_W1 += g_h1 (x) h0    g_h0 = W1^T g_h1 [h0 > 0]
//     g_b0 += g_h0    g_W0 += g_h0 (x) x     g_x  = W0^T g_h0:  grad_feats[p, c] += g_x[2 + c],  grad_xy[i] += (g_x[0], g_x[1])
// The two families of sums run over different axes, so there are two kernels, both DETERMINISTIC (no float atomics):
//     dynmask_bwd_params   instance-major: a workgroup owns (instance, slice of the pixels); every thread keeps the parameter
//                          gradients of its pixels in registers (two launches: first layer / the other two), the workgroup reduces them in a fixed order into a partial row;
//                          dynmask_bwd_params_reduce adds the slices' rows in order (and forms grad_xy from g_b0 and W0).
//     dynmask_bwd_feats    pixel-major: a thread owns a pixel and walks the instances of its image in order (parameters staged in
//                          LDS, double-buffered), summing g_x[2..9] in registers; one store per channel.
//     aligned_bilinear_bwd gather: every input pixel sums, in a fixed order, the output pixels whose interpolation reads it
//                          (<= (2 f + f / 2)^2 of them), with the forward's own weights.
#include "../../include/dynmask_hip.h"

#include <cstddef>

#include "msda_common.hpp"

namespace dynmask {

constexpr int kBT = 256;                    // threads of dynmask_bwd_params
constexpr int kFT = 128;                    // threads of dynmask_bwd_feats
constexpr int kCf = 8, kHid = 8;            // mask-feature channels, dynamic channels
constexpr int kRowPad = 176;                // a partial row: 169 parameter gradients (reference order) + padding
constexpr int kMaxBatch = DYNMASK_HIP_BWD_MAX_BATCH;
struct InstOffsets { int off[kMaxBatch + 1]; };   // first instance of every image (kernel argument: no host-to-device copy)

template <bool REL>
struct Layout {                             // offsets inside a parameter row, reference order (ddetrs_dn.py:53-66): weights, then biases
  static constexpr int kIn = REL ? kCf + 2 : kCf;
  static constexpr int w0 = 0, w1 = kIn * kHid, w2 = w1 + kHid * kHid, b0 = w2 + kHid, b1 = b0 + kHid, b2 = b1 + kHid, n = b2 + 1;
};

// forward of one (instance, pixel) from parameters in LDS (reference layout [out][in]); returns h0, h1 (post-ReLU)
template <bool REL>
__device__ __forceinline__ void forward_pair(const float* __restrict__ P, const float (&x)[Layout<REL>::kIn], float (&h0)[kHid],
                                             float (&h1)[kHid]) {
  using L = Layout<REL>;
#pragma unroll
  for (int o = 0; o < kHid; ++o) {
    float a = P[L::b0 + o];
#pragma unroll
    for (int i = 0; i < L::kIn; ++i) a = fmaf(P[L::w0 + o * L::kIn + i], x[i], a);
    h0[o] = fmaxf(a, 0.f);
  }
#pragma unroll
  for (int o = 0; o < kHid; ++o) {
    float a = P[L::b1 + o];
#pragma unroll
    for (int i = 0; i < kHid; ++i) a = fmaf(P[L::w1 + o * kHid + i], h0[i], a);
    h1[o] = fmaxf(a, 0.f);
  }
}

// PART 0: the first layer's gradients (g_W0, g_b0: 88 sums per thread); PART 1: the other two layers' (g_W1, g_w2, g_b1, g_b2: 81).
// One kernel with all 169 accumulators needs 256 registers and still spills 700 bytes; the halves recompute the cheap forward.
template <bool REL, int PART>
__global__ void __launch_bounds__(kBT, 2)
dynmask_bwd_params(const float* __restrict__ feats, const float* __restrict__ inst_xy, const float* __restrict__ params,
                   const float* __restrict__ grad_logits, InstOffsets io, int batch, int H, int W, int stride, int parts,
                   float* __restrict__ partial) {
  using L = Layout<REL>;
  __shared__ float P[kRowPad];
  __shared__ float red[kBT / 64][kRowPad];
  const int inst = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
  const int HW = H * W;
  int img = 0;                                               // the image of this instance (uniform: scalar compares)
  for (int b = 1; b < batch; ++b) img += inst >= io.off[b] ? 1 : 0;
  const float* f = feats + (size_t)img * kCf * HW;
  if (tid < L::n) P[tid] = params[(size_t)inst * L::n + tid];
  __syncthreads();
  const float ix = inst_xy[(size_t)inst * 2], iy = inst_xy[(size_t)inst * 2 + 1];
  constexpr int kA = PART == 0 ? L::kIn : kHid;              // width of this half's weight-gradient matrix
  float gW[kHid][kA], gb[kHid], gw2[kHid], gb2 = 0.f;
#pragma unroll
  for (int o = 0; o < kHid; ++o) {
    gb[o] = 0.f; gw2[o] = 0.f;
#pragma unroll
    for (int i = 0; i < kA; ++i) gW[o][i] = 0.f;
  }
  // this slice's pixels, strided over the workgroup (coalesced): p = first + tid, + kBT, ...
  const int per = (HW + parts - 1) / parts, first = part * per, last = min(HW, first + per);
  for (int p = first + tid; p < last; p += kBT) {
    // (a compiler barrier: without it every P[...] is a loop invariant, all 169 are hoisted into registers and the kernel spills;
    // re-read per pixel they are broadcast LDS loads)
    asm volatile("" ::: "memory");
    const float g = grad_logits[(size_t)inst * HW + p];
    float x[L::kIn];
    if constexpr (REL) {
      const int py = p / W, px = p - py * W;
      x[0] = ix - (float)(px * stride + stride / 2);
      x[1] = iy - (float)(py * stride + stride / 2);
    }
#pragma unroll
    for (int c = 0; c < kCf; ++c) x[(REL ? 2 : 0) + c] = f[(size_t)c * HW + p];
    float h0[kHid], h1[kHid];
    forward_pair<REL>(P, x, h0, h1);
    float gh1[kHid];
#pragma unroll
    for (int o = 0; o < kHid; ++o) gh1[o] = h1[o] > 0.f ? g * P[L::w2 + o] : 0.f;
    if constexpr (PART == 1) {
      gb2 += g;
#pragma unroll
      for (int o = 0; o < kHid; ++o) {
        gw2[o] = fmaf(g, h1[o], gw2[o]);
        gb[o] += gh1[o];
#pragma unroll
        for (int i = 0; i < kHid; ++i) gW[o][i] = fmaf(gh1[o], h0[i], gW[o][i]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < kHid; ++j) {
        float a = 0.f;
#pragma unroll
        for (int o = 0; o < kHid; ++o) a = fmaf(P[L::w1 + o * kHid + j], gh1[o], a);
        const float gh0 = h0[j] > 0.f ? a : 0.f;
        gb[j] += gh0;
#pragma unroll
        for (int i = 0; i < L::kIn; ++i) gW[j][i] = fmaf(gh0, x[i], gW[j][i]);
      }
    }
  }
  // ---- reduce over the workgroup in a fixed order: lanes of a wave (xor butterfly), then the waves in order ---------------------
  auto put = [&](int slot, float v) __attribute__((always_inline)) {
    v = msda::wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6][slot] = v;
  };
#pragma unroll
  for (int o = 0; o < kHid; ++o) {
#pragma unroll
    for (int i = 0; i < kA; ++i) put((PART == 0 ? L::w0 + o * L::kIn : L::w1 + o * kHid) + i, gW[o][i]);
    put((PART == 0 ? L::b0 : L::b1) + o, gb[o]);
    if constexpr (PART == 1) put(L::w2 + o, gw2[o]);
  }
  if constexpr (PART == 1) put(L::b2, gb2);
  __syncthreads();
  const bool mine = PART == 0 ? (tid < L::w1 || (tid >= L::b0 && tid < L::b1)) : ((tid >= L::w1 && tid < L::b0) || (tid >= L::b1 && tid < L::n));
  if (mine) {
    float s = red[0][tid];
#pragma unroll
    for (int w = 1; w < kBT / 64; ++w) s += red[w][tid];
    partial[((size_t)inst * parts + part) * kRowPad + tid] = s;
  }
}

template <bool REL>
__global__ void __launch_bounds__(kRowPad)
dynmask_bwd_params_reduce(const float* __restrict__ partial, const float* __restrict__ params, int parts,
                          float* __restrict__ grad_params, float* __restrict__ grad_xy) {
  using L = Layout<REL>;
  __shared__ float row[kRowPad];
  const int inst = blockIdx.x, tid = threadIdx.x;
  float s = 0.f;
  if (tid < L::n) {
    for (int p = 0; p < parts; ++p) s += partial[((size_t)inst * parts + p) * kRowPad + tid];
    grad_params[(size_t)inst * L::n + tid] = s;
  }
  row[tid] = s;
  __syncthreads();
  if (grad_xy && tid < 2) {
    // rel = inst_xy - location: d rel / d inst_xy = 1, so grad_xy[d] = sum_p g_x[d] = sum_o W0[o][d] * (sum_p g_h0[o]) = W0[:, d] . g_b0
    float a = 0.f;
    if constexpr (REL) {
#pragma unroll
      for (int o = 0; o < kHid; ++o) a = fmaf(params[(size_t)inst * L::n + L::w0 + o * L::kIn + tid], row[L::b0 + o], a);
    }
    grad_xy[(size_t)inst * 2 + tid] = a;
  }
}

template <bool REL>
__global__ void __launch_bounds__(kFT)
dynmask_bwd_feats(const float* __restrict__ feats, const float* __restrict__ inst_xy, const float* __restrict__ params,
                  const float* __restrict__ grad_logits, int inst_first, int inst_count, int H, int W, int stride,
                  float* __restrict__ grad_feats) {
  using L = Layout<REL>;
  __shared__ float P[2][kRowPad + 2];                       // + the instance's reference point
  const int tid = threadIdx.x, HW = H * W;
  const int p = blockIdx.x * kFT + tid;
  const int pc = p < HW ? p : HW - 1;
  float x[L::kIn], gf[kCf];
#pragma unroll
  for (int c = 0; c < kCf; ++c) { x[(REL ? 2 : 0) + c] = feats[(size_t)c * HW + pc]; gf[c] = 0.f; }
  const int py = pc / W, px = pc - py * W;
  const float lx = (float)(px * stride + stride / 2), ly = (float)(py * stride + stride / 2);
  auto stage = [&](int inst, float* dst) __attribute__((always_inline)) {
    for (int t = tid; t < L::n; t += kFT) dst[t] = params[(size_t)inst * L::n + t];
    if (tid < 2) dst[kRowPad + tid] = inst_xy[(size_t)inst * 2 + tid];
  };
  if (inst_count > 0) stage(inst_first, P[0]);
  for (int i = 0; i < inst_count; ++i) {
    __syncthreads();                                         // P[i & 1] is complete, P[(i + 1) & 1] is free
    if (i + 1 < inst_count) stage(inst_first + i + 1, P[(i + 1) & 1]);
    const float* Q = P[i & 1];
    const float g = p < HW ? grad_logits[(size_t)(inst_first + i) * HW + pc] : 0.f;
    if constexpr (REL) { x[0] = Q[kRowPad] - lx; x[1] = Q[kRowPad + 1] - ly; }
    float h0[kHid], h1[kHid];
    forward_pair<REL>(Q, x, h0, h1);
    float gh1[kHid], gh0[kHid];
#pragma unroll
    for (int o = 0; o < kHid; ++o) gh1[o] = h1[o] > 0.f ? g * Q[L::w2 + o] : 0.f;
#pragma unroll
    for (int j = 0; j < kHid; ++j) {
      float a = 0.f;
#pragma unroll
      for (int o = 0; o < kHid; ++o) a = fmaf(Q[L::w1 + o * kHid + j], gh1[o], a);
      gh0[j] = h0[j] > 0.f ? a : 0.f;
    }
#pragma unroll
    for (int c = 0; c < kCf; ++c) {
      float a = gf[c];                                       // instances in order: a fixed summation order per pixel
#pragma unroll
      for (int o = 0; o < kHid; ++o) a = fmaf(Q[L::w0 + o * L::kIn + (REL ? 2 : 0) + c], gh0[o], a);
      gf[c] = a;
    }
  }
  if (p < HW) {
#pragma unroll
    for (int c = 0; c < kCf; ++c) grad_feats[(size_t)c * HW + p] = gf[c];
  }
}

// aligned_bilinear backward.  Forward (dynmask.hip: aligned_bilinear_kernel), per axis: output index o reads the source position
// s = max(o - f / 2, 0) / f: inputs min(i0, n - 1) and min(i0 + 1, n - 1), i0 = floor(s), with weights 1 - frac and frac.
// Input index j is read by outputs o in [f / 2 + (j - 1) f, f / 2 + (j + 1) f - 1] (and by o < f / 2 when j = 0).
__device__ __forceinline__ float axis_weight(int o, int j, int n, int factor, float inv) {
  const int i = max(o - factor / 2, 0);
  const float s = (float)i * inv;
  const int i0 = (int)s;
  const float fr = s - (float)i0;
  float w = 0.f;
  if (min(i0, n - 1) == j) w += 1.f - fr;
  if (min(i0 + 1, n - 1) == j) w += fr;
  return w;
}

__global__ void __launch_bounds__(256)
aligned_bilinear_bwd_kernel(const float* __restrict__ grad_out, int n_img, int h, int w, int factor, float* __restrict__ grad_in) {
  const int oh = factor * h, ow = factor * w;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)n_img * h * w) return;
  const int jx = (int)(idx % w), jy = (int)((idx / w) % h);
  const long long img = idx / ((long long)w * h);
  const float inv = 1.0f / (float)factor;
  const float* g = grad_out + (size_t)img * oh * ow;
  const int oy0 = jy == 0 ? 0 : max(factor / 2 + (jy - 1) * factor, 0), oy1 = min(oh - 1, factor / 2 + (jy + 1) * factor - 1);
  const int ox0 = jx == 0 ? 0 : max(factor / 2 + (jx - 1) * factor, 0), ox1 = min(ow - 1, factor / 2 + (jx + 1) * factor - 1);
  float acc = 0.f;
  for (int oy = oy0; oy <= oy1; ++oy) {
    const float wy = axis_weight(oy, jy, h, factor, inv);
    if (wy == 0.f) continue;
    float row = 0.f;
    for (int ox = ox0; ox <= ox1; ++ox) {
      const float wx = axis_weight(ox, jx, w, factor, inv);
      row = fmaf(wx, g[(size_t)oy * ow + ox], row);
    }
    acc = fmaf(wy, row, acc);
  }
  grad_in[idx] = acc;
}

}  // namespace dynmask

extern "C" {

int dynmask_set_error(int code, const char* what);   // msda_capi.hip

size_t dynmask_hip_backward_workspace_bytes(int n_inst_all, int H, int W) {
  if (n_inst_all <= 0 || H <= 0 || W <= 0) return 0;
  const int parts = dynmask_hip_backward_parts(n_inst_all, H, W);
  return (size_t)n_inst_all * parts * dynmask::kRowPad * sizeof(float);
}

int dynmask_hip_backward_parts(int n_inst_all, int H, int W) {
  // ~1024 workgroups in flight, at least 256 pixels per slice (a slice is one pass of the workgroup or more)
  const int HW = H * W;
  int parts = n_inst_all > 0 ? 1024 / n_inst_all : 1;
  const int max_parts = (HW + dynmask::kBT - 1) / dynmask::kBT;
  if (parts > max_parts) parts = max_parts;
  if (parts < 1) parts = 1;
  return parts;
}

int dynmask_hip_backward_f32(const float* mask_feats, const float* inst_xy, const float* params, const int* num_insts,
                             int batch, int channels, int H, int W, int stride, int rel_coord, const float* grad_logits,
                             float* grad_feats, float* grad_params, float* grad_xy, void* workspace, size_t workspace_bytes,
                             void* stream) {
  if (batch < 0 || H <= 0 || W <= 0 || stride <= 0) return dynmask_set_error(DYNMASK_ERR_BAD_DIMS, "dynmask backward: bad dimensions");
  if (channels != dynmask::kCf) return dynmask_set_error(DYNMASK_ERR_UNSUPPORTED, "dynmask backward: only 8 mask-feature channels");
  if (batch > dynmask::kMaxBatch) return dynmask_set_error(DYNMASK_ERR_UNSUPPORTED, "dynmask backward: more than DYNMASK_HIP_BWD_MAX_BATCH images");
  if (batch == 0) return 0;
  // grad_feats == NULL / grad_params == NULL (with grad_xy == NULL): that family of kernels is skipped (round 6, ADVICE r05:
  // frozen mask features or detached parameters need not pay for the other half)
  if (!num_insts || (grad_xy && !grad_params)) return dynmask_set_error(DYNMASK_ERR_NULL_POINTER, "dynmask backward: null pointer argument");
  const int HW = H * W;
  int n_all = 0;
  for (int b = 0; b < batch; ++b) {
    if (num_insts[b] < 0) return dynmask_set_error(DYNMASK_ERR_BAD_DIMS, "dynmask backward: negative instance count");
    n_all += num_insts[b];
  }
  hipStream_t st = (hipStream_t)stream;
  if (n_all > 0 && (!mask_feats || !inst_xy || !params || !grad_logits || (grad_params && !workspace)))
    return dynmask_set_error(DYNMASK_ERR_NULL_POINTER, "dynmask backward: null pointer argument");
  if (n_all > 0 && grad_params && workspace_bytes < dynmask_hip_backward_workspace_bytes(n_all, H, W))
    return dynmask_set_error(DYNMASK_ERR_BAD_DIMS, "dynmask backward: workspace too small (dynmask_hip_backward_workspace_bytes)");
  // ---- grad_feats: every pixel of every image is written (zeros for an image without instances) -----------------------------
  if (grad_feats) {
    const unsigned chunks = (unsigned)((HW + dynmask::kFT - 1) / dynmask::kFT);
    int first = 0;
    for (int b = 0; b < batch; ++b) {
      const int n = num_insts[b];
      const float* f = mask_feats ? mask_feats + (size_t)b * dynmask::kCf * HW : nullptr;
      float* gf = grad_feats + (size_t)b * dynmask::kCf * HW;
      if (n == 0) {
        if (hipError_t e = hipMemsetAsync(gf, 0, (size_t)dynmask::kCf * HW * sizeof(float), st); e != hipSuccess)
          return dynmask_set_error((int)e, hipGetErrorString(e));
      } else if (rel_coord) {
        hipLaunchKernelGGL(dynmask::dynmask_bwd_feats<true>, dim3(chunks), dim3(dynmask::kFT), 0, st, f, inst_xy, params,
                           grad_logits, first, n, H, W, stride, gf);
      } else {
        hipLaunchKernelGGL(dynmask::dynmask_bwd_feats<false>, dim3(chunks), dim3(dynmask::kFT), 0, st, f, inst_xy, params,
                           grad_logits, first, n, H, W, stride, gf);
      }
      first += n;
    }
  }
  if (n_all > 0 && grad_params) {
    // ---- grad_params / grad_xy: slices of the pixels per instance, then the slices added in order -----------------------------
    const int parts = dynmask_hip_backward_parts(n_all, H, W);
    float* partial = static_cast<float*>(workspace);
    dynmask::InstOffsets io;
    io.off[0] = 0;
    for (int b = 0; b < batch; ++b) io.off[b + 1] = io.off[b] + num_insts[b];
    if (rel_coord) {
      hipLaunchKernelGGL((dynmask::dynmask_bwd_params<true, 0>), dim3((unsigned)n_all, (unsigned)parts), dim3(dynmask::kBT), 0, st,
                         mask_feats, inst_xy, params, grad_logits, io, batch, H, W, stride, parts, partial);
      hipLaunchKernelGGL((dynmask::dynmask_bwd_params<true, 1>), dim3((unsigned)n_all, (unsigned)parts), dim3(dynmask::kBT), 0, st,
                         mask_feats, inst_xy, params, grad_logits, io, batch, H, W, stride, parts, partial);
      hipLaunchKernelGGL(dynmask::dynmask_bwd_params_reduce<true>, dim3((unsigned)n_all), dim3(dynmask::kRowPad), 0, st,
                         partial, params, parts, grad_params, grad_xy);
    } else {
      hipLaunchKernelGGL((dynmask::dynmask_bwd_params<false, 0>), dim3((unsigned)n_all, (unsigned)parts), dim3(dynmask::kBT), 0, st,
                         mask_feats, inst_xy, params, grad_logits, io, batch, H, W, stride, parts, partial);
      hipLaunchKernelGGL((dynmask::dynmask_bwd_params<false, 1>), dim3((unsigned)n_all, (unsigned)parts), dim3(dynmask::kBT), 0, st,
                         mask_feats, inst_xy, params, grad_logits, io, batch, H, W, stride, parts, partial);
      hipLaunchKernelGGL(dynmask::dynmask_bwd_params_reduce<false>, dim3((unsigned)n_all), dim3(dynmask::kRowPad), 0, st,
                         partial, params, parts, grad_params, grad_xy);
    }
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : dynmask_set_error((int)e, hipGetErrorString(e));
}

int aligned_bilinear_hip_backward_f32(const float* grad_out, int n, int h, int w, int factor, float* grad_in, void* stream) {
  if (n < 0 || h <= 0 || w <= 0 || factor < 1) return dynmask_set_error(DYNMASK_ERR_BAD_DIMS, "aligned_bilinear backward: bad dimensions");
  if (n == 0) return 0;
  if (!grad_out || !grad_in) return dynmask_set_error(DYNMASK_ERR_NULL_POINTER, "aligned_bilinear backward: null pointer argument");
  const long long total = (long long)n * h * w, blocks = (total + 255) / 256;
  if (blocks >= (1ll << 31)) return dynmask_set_error(DYNMASK_ERR_BAD_DIMS, "aligned_bilinear backward: too many pixels");
  hipLaunchKernelGGL(dynmask::aligned_bilinear_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grad_out, n, h, w,
                     factor, grad_in);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : dynmask_set_error((int)e, hipGetErrorString(e));
}

}  // extern "C"
