// Residual add + LayerNorm, one wave per row -- see include/layernorm_hip.h.
//
// Memory bound: read x and residual, write out (136 MB at 44 446 x 256 -> ~20 us at 7 TB/s).  A wave holds its row in
// registers (features / 64 values per lane as float4 chunks, coalesced 16-byte accesses), reduces mean and the
// centred second moment with two butterfly passes (the two-pass form PyTorch's LayerNorm uses: no cancellation), and
// writes the normalised row.  256-thread workgroups = 4 rows.
#include "../../include/layernorm_hip.h"

#include "msda_common.hpp"

namespace layernorm {

using msda::f32x4;
constexpr int kThreads = 256, kMaxChunks = 16;   // 16 float4 per lane = 4096 features

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int CHUNKS>   // float4 chunks per lane: features <= CHUNKS * 256
__global__ void __launch_bounds__(kThreads)
add_layernorm(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma,
              const float* __restrict__ beta, float eps, long long rows, int features, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = features / 4;
  const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * features);
  const f32x4* rr = res ? reinterpret_cast<const f32x4*>(res + row * features) : nullptr;
  f32x4 v[CHUNKS];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    const int i = c * 64 + lane;
    v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i < nvec) {
      v[c] = xr[i];
      if (rr) v[c] += rr[i];
      sum += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
    }
  }
  const float mean = wave_sum(sum) / (float)features;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    const int i = c * 64 + lane;
    if (i < nvec) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dlt = v[c][e] - mean;
        sq += dlt * dlt;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)features + eps);
  f32x4* orow = reinterpret_cast<f32x4*>(out + row * features);
  const f32x4* gv = reinterpret_cast<const f32x4*>(gamma);
  const f32x4* bv = reinterpret_cast<const f32x4*>(beta);
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    const int i = c * 64 + lane;
    if (i < nvec) {
      f32x4 r;
      const f32x4 g = gamma ? gv[i] : f32x4{1.f, 1.f, 1.f, 1.f};
      const f32x4 b = beta ? bv[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = (v[c][e] - mean) * rstd * g[e] + b[e];
      orow[i] = r;
    }
  }
}

}  // namespace layernorm

extern "C" {

int dynmask_set_error(int code, const char* what);   // msda_capi.hip (shared last-error slot)

int add_layernorm_hip_f32(const float* x, const float* residual, const float* gamma, const float* beta, float eps,
                          long long rows, int features, float* out, void* stream) {
  if (rows < 0 || features <= 0) return dynmask_set_error(LAYERNORM_ERR_BAD_DIMS, "add_layernorm: bad dimensions");
  if (features % 4 != 0 || features > layernorm::kMaxChunks * 256)
    return dynmask_set_error(LAYERNORM_ERR_UNSUPPORTED, "add_layernorm: features must be a multiple of 4 and <= 4096");
  if (rows == 0) return 0;
  const long long blocks = (rows + 3) / 4;
  if (blocks >= (1ll << 31)) return dynmask_set_error(LAYERNORM_ERR_BAD_DIMS, "add_layernorm: too many rows");
  if (!x || !out) return dynmask_set_error(LAYERNORM_ERR_NULL_POINTER, "add_layernorm: null pointer argument");
  const dim3 grid((unsigned)blocks), block(layernorm::kThreads);
  hipStream_t st = (hipStream_t)stream;
  const int chunks = (features / 4 + 63) / 64;
#define LN_LAUNCH(C) hipLaunchKernelGGL((layernorm::add_layernorm<C>), grid, block, 0, st, x, residual, gamma, beta, eps, rows, features, out)
  if (chunks <= 1) LN_LAUNCH(1);
  else if (chunks <= 2) LN_LAUNCH(2);
  else if (chunks <= 4) LN_LAUNCH(4);
  else if (chunks <= 8) LN_LAUNCH(8);
  else LN_LAUNCH(16);
#undef LN_LAUNCH
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : dynmask_set_error((int)e, hipGetErrorString(e));
}

}  // extern "C"
