// fp32 Linear layers with split-bf16 products on the gfx950 matrix cores -- see include/linear_hip.h.
//
//   out[m, n] = bias[n] + sum_k x[m, k] * W[n, k]
//
// Same scheme as conv3x3_packed / patch_embed_packed: W is split (hi = upper 16 bits, lo = bf16 of the remainder)
// and re-ordered once into [K / 16 chunks][hi, lo][N padded to 128][16 k] bf16, so a wave's weight fragment of a
// chunk is 1 KB of contiguous memory loaded straight into registers (ring of four register sets, three chunks
// ahead); the activation tile (128 rows x 64 k per step) is split while it is staged into double-buffered LDS as
// [chunk][row][16 bf16] hi / lo, one ds_read_b128 per MFMA operand.  Workgroup: 256 threads = 2 x 2 waves, 128 rows
// x 64 TJ columns; per step a wave issues 12 TJ x 2 v_mfma_f32_32x32x16_bf16.  At K = 256 (the only size the layer
// uses) a tile is 4 steps: the kernel is bound by reading x and writing out, not by the matrix pipe.
#include "../../include/linear_hip.h"

#include "msda_common.hpp"

#include <cstdlib>

namespace linear {

typedef float f32x16 __attribute__((__vector_size__(64)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4v __attribute__((__vector_size__(16)));
using msda::f32x4;

constexpr int kThreads = 256;
constexpr int kChunk = 16, kStepChunks = 4, kStepK = kChunk * kStepChunks;   // 64 k per barrier

__device__ __forceinline__ void split8(const float (&v)[8], u32x4v& hi, u32x4v& lo) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const uint32_t a = __float_as_uint(v[2 * p]), b = __float_as_uint(v[2 * p + 1]);
    const uint32_t ah = a & 0xffff0000u, bh = b & 0xffff0000u;
    const uint32_t al = __float_as_uint(v[2 * p] - __uint_as_float(ah));
    const uint32_t bl = __float_as_uint(v[2 * p + 1] - __uint_as_float(bh));
    hi[p] = (ah >> 16) | bh;
    lo[p] = ((al + 0x8000u) >> 16) | ((bl + 0x8000u) & 0xffff0000u);   // lo rounded to nearest
  }
}

// WM: waves along the rows (1 or 2).  A wave owning ONE 32-row tile re-loads 2 KB of weight fragments per 3 MFMAs, more
// than the CU's 64 B/clk L1 path delivers per MFMA slot; with WM = 1 every wave spans the tile's 64 rows (two row tiles
// per weight fragment) and a quarter of its columns.
struct LnArgs {   // LayerNorm(out + residual) over the N columns in the epilogue (LN kernels only: N == 64 TJ)
  const float* residual;
  const float* gamma;
  const float* beta;
  float eps;
};

template <int CTRL>
__device__ __forceinline__ float msda_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

template <int TJ, int BM, bool ADD, int WM, bool LN = false>   // ADD: the input is x + x2
__global__ void __launch_bounds__(kThreads, 2)
linear_packed(const float* __restrict__ x, const float* __restrict__ x2, const uint32_t* __restrict__ packed,
              const float* __restrict__ bias, const uint8_t* __restrict__ row_mask, long long M, int K, int N, int n_pad,
              int hm_rows, int act, float* __restrict__ out, LnArgs ln = LnArgs{nullptr, nullptr, nullptr, 0.f}) {
  // [buffer][hi / lo][chunk][row (+1 pad row per chunk: staggers the banks of the staging stores)][16 bf16]
  __shared__ __attribute__((aligned(16))) uint32_t As[2][2][kStepChunks][BM + 1][8];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long long m0 = (long long)blockIdx.x * BM;
  const int n0 = blockIdx.y * (64 * TJ);

  // staging: a step is 128 rows x 64 k = 16 pieces of 16 bytes per row; lane t % 16 takes piece t % 16 of rows
  // t / 16 + 16 r, so a wave instruction reads 4 rows x 256 contiguous bytes (full lines)
  const int s_piece = tid & 15, s_row0 = tid >> 4;
  constexpr int kRows = BM / 16;                  // rows staged per thread
  constexpr int WN = 4 / WM, TI = BM / (32 * WM), WJ = 2 * TJ / WN;   // per wave: TI row tiles x WJ column tiles
  static_assert(TI >= 1 && WJ >= 1, "wave layout");
  long long a_off[kRows];                               // element offset of this thread's piece in row r (x and x2 alike)
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    long long mr = m0 + s_row0 + 16 * r;
    mr = mr < M ? mr : M - 1;
    a_off[r] = mr * K + s_piece * 4;
  }
  f32x4 a_reg[kRows];
  auto load_step = [&](int st) {
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      a_reg[r] = *reinterpret_cast<const f32x4*>(x + a_off[r] + st * kStepK);
      if constexpr (ADD) a_reg[r] += *reinterpret_cast<const f32x4*>(x2 + a_off[r] + st * kStepK);
    }
  };
  auto store_step = [&](int buf) {
    const int cc = s_piece >> 2, w2 = (s_piece & 3) * 2;   // chunk of the step, word pair inside the row's 16 bf16
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      uint32_t hi[2], lo[2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const float fa = a_reg[r][2 * p], fb = a_reg[r][2 * p + 1];
        const uint32_t ah = __float_as_uint(fa) & 0xffff0000u, bh = __float_as_uint(fb) & 0xffff0000u;
        const uint32_t al = __float_as_uint(fa - __uint_as_float(ah)), bl = __float_as_uint(fb - __uint_as_float(bh));
        hi[p] = (ah >> 16) | bh;
        lo[p] = ((al + 0x8000u) >> 16) | ((bl + 0x8000u) & 0xffff0000u);   // lo rounded to nearest
      }
      *reinterpret_cast<uint2*>(&As[buf][0][cc][s_row0 + 16 * r][w2]) = make_uint2(hi[0], hi[1]);
      *reinterpret_cast<uint2*>(&As[buf][1][cc][s_row0 + 16 * r][w2]) = make_uint2(lo[0], lo[1]);
    }
  };

  const int wm = (wv / WN) * (BM / WM), wn = (wv % WN) * 32 * WJ;
  const int r32 = lane & 31, half = lane >> 5;
  const uint32_t* w_lane = packed + (long long)(n0 + wn + r32) * 8 + half * 4;
  const long long chunk_stride = (long long)2 * n_pad * 8, part_stride = (long long)n_pad * 8;
  const int nchunks = K / kChunk, nsteps = K / kStepK;
  struct WFrag { u32x4v hi[WJ], lo[WJ]; };
  auto load_w = [&](int ch, WFrag& f) {
    const int cc = ch < nchunks ? ch : nchunks - 1;
    const uint32_t* p = w_lane + cc * chunk_stride;
#pragma unroll
    for (int jn = 0; jn < WJ; ++jn) {
      f.hi[jn] = *reinterpret_cast<const u32x4v*>(p + jn * 32 * 8);
      f.lo[jn] = *reinterpret_cast<const u32x4v*>(p + part_stride + jn * 32 * 8);
    }
  };

  f32x16 acc[TI][WJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int jn = 0; jn < WJ; ++jn)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][jn][v] = 0.f;

  auto chunk_mfma = [&](int buf, int cc, const WFrag& wf) {
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(&As[buf][0][cc][wm + i * 32 + r32][half * 4]));
      const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(&As[buf][1][cc][wm + i * 32 + r32][half * 4]));
#pragma unroll
      for (int jn = 0; jn < WJ; ++jn) {   // rows = x rows, columns = output features; small terms first
        const bf16x8 wh = __builtin_bit_cast(bf16x8, wf.hi[jn]), wl = __builtin_bit_cast(bf16x8, wf.lo[jn]);
        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl, acc[i][jn], 0, 0, 0);
        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh, acc[i][jn], 0, 0, 0);
        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, acc[i][jn], 0, 0, 0);
      }
    }
  };

  WFrag w0, w1, w2, w3;
  load_step(0);
  load_w(0, w0);
  load_w(1, w1);
  load_w(2, w2);
  store_step(0);
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    const int buf = st & 1, ch = st * kStepChunks;
    if (st + 1 < nsteps) load_step(st + 1);
    load_w(ch + 3, w3);
    chunk_mfma(buf, 0, w0);
    load_w(ch + 4, w0);
    chunk_mfma(buf, 1, w1);
    load_w(ch + 5, w1);
    chunk_mfma(buf, 2, w2);
    load_w(ch + 6, w2);
    chunk_mfma(buf, 3, w3);
    if (st + 1 < nsteps) store_step(buf ^ 1);
    __syncthreads();
  }

  if constexpr (LN) {
    // ---- out = LayerNorm(acc + bias + residual) * gamma + beta: the workgroup holds whole rows (N == 64 TJ columns,
    // a quarter per wave).  Two-pass statistics like add_layernorm: row sums are reduced over the 32 lanes of a
    // half-wave with DPP + one xor-16 exchange, then over the four waves through LDS (the operand tile is dead by now).
    static_assert(WM == 1 && BM == 64, "LayerNorm epilogue: every wave spans the 64 rows");
    float* red = reinterpret_cast<float*>(&As[0][0][0][0][0]);   // [4 waves][64 rows] partial sums, then [64] totals at +256
    auto sum32 = [](float v) {
      v += msda_dpp<0xB1>(v); v += msda_dpp<0x4E>(v); v += msda_dpp<0x141>(v); v += msda_dpp<0x140>(v);
      return v + __shfl_xor(v, 16, 64);
    };
    float val[TI][WJ][16];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int jn = 0; jn < WJ; ++jn) {
        const int n = wn + jn * 32 + r32;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          long long m = m0 + wm + i * 32 + 8 * (v / 4) + 4 * half + (v % 4);
          m = m < M ? m : M - 1;
          val[i][jn][v] = acc[i][jn][v] + bv + (ln.residual ? ln.residual[m * N + n] : 0.f);
        }
      }
    float stat[TI][16];
    auto reduce_rows = [&](bool centred, const float (&mean)[TI][16]) {
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          float p = 0.f;
#pragma unroll
          for (int jn = 0; jn < WJ; ++jn) {
            const float t = centred ? val[i][jn][v] - mean[i][v] : val[i][jn][v];
            p += centred ? t * t : t;
          }
          p = sum32(p);
          if (r32 == 0) red[wv * 64 + i * 32 + 8 * (v / 4) + 4 * half + (v % 4)] = p;
        }
      __syncthreads();
      if (tid < 64) red[256 + tid] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) stat[i][v] = red[256 + i * 32 + 8 * (v / 4) + 4 * half + (v % 4)];
      __syncthreads();   // `red` is rewritten by the next pass
    };
    float mean[TI][16];
    reduce_rows(false, mean);
    const float inv_n = 1.0f / (float)N;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int v = 0; v < 16; ++v) mean[i][v] = stat[i][v] * inv_n;
    reduce_rows(true, mean);
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int jn = 0; jn < WJ; ++jn) {
        const int n = wn + jn * 32 + r32;
        const float gmm = ln.gamma ? ln.gamma[n] : 1.f, bt = ln.beta ? ln.beta[n] : 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const long long m = m0 + wm + i * 32 + 8 * (v / 4) + 4 * half + (v % 4);
          const float rstd = rsqrtf(stat[i][v] * inv_n + ln.eps);
          if (m < M) out[m * N + n] = (val[i][jn][v] - mean[i][v]) * rstd * gmm + bt;
        }
      }
    return;
  }
  // epilogue: accumulator register v of lane l is (row 8 (v / 4) + 4 (l / 32) + v % 4, column l % 32)
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    uint32_t zero_rows = 0;                            // bit v: row of register v is masked
    if (row_mask) {
#pragma unroll
      for (int v4 = 0; v4 < 4; ++v4) {                 // rows 8 v4 + 4 half .. + 3 are consecutive: one 4-byte load
        const long long m = m0 + wm + i * 32 + 8 * v4 + 4 * half;
        uint32_t four = 0;
        if (m + 3 < M && ((m & 3) == 0)) four = *reinterpret_cast<const uint32_t*>(row_mask + m);
        else
          for (int e = 0; e < 4; ++e) if (m + e < M && row_mask[m + e]) four |= 0xffu << (8 * e);
#pragma unroll
        for (int e = 0; e < 4; ++e) if ((four >> (8 * e)) & 0xffu) zero_rows |= 1u << (4 * v4 + e);
      }
    }
#pragma unroll
    for (int jn = 0; jn < WJ; ++jn) {
      const int n = n0 + wn + jn * 32 + r32;
      const float bv = (bias && n < N) ? bias[n] : 0.f;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const long long m = m0 + wm + i * 32 + 8 * (v / 4) + 4 * half + (v % 4);
        if (m < M && n < N) {
          float r = ((zero_rows >> v) & 1u) ? 0.f : acc[i][jn][v] + bv;
          if (act == 1) r = fmaxf(r, 0.f);
          if (hm_rows == 0) {
            out[m * N + n] = r;
          } else {   // head-major [image, head = n / 32, row in image, n % 32]: a lane row still writes 128 contiguous bytes
            const long long img = m / hm_rows, sr = m - img * hm_rows;
            out[((img * (N / 32) + (n >> 5)) * hm_rows + sr) * 32 + (n & 31)] = r;
          }
        }
      }
    }
  }
}

__global__ void pack_weight_kernel(const float* __restrict__ w, int N, int K, int n_pad, uint16_t* __restrict__ packed) {
  const long long total = (long long)(K / kChunk) * n_pad * kChunk;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int kl = (int)(idx % kChunk);
    const int n = (int)((idx / kChunk) % n_pad);
    const int chunk = (int)(idx / kChunk / n_pad);
    const float v = n < N ? w[(long long)n * K + chunk * kChunk + kl] : 0.f;
    const uint32_t bits = __float_as_uint(v), hb = bits & 0xffff0000u;
    const uint32_t lb = __float_as_uint(v - __uint_as_float(hb));
    const long long o = ((long long)chunk * 2 * n_pad + n) * kChunk + kl;
    packed[o] = (uint16_t)(hb >> 16);
    packed[o + (long long)n_pad * kChunk] = (uint16_t)((lb + 0x8000u) >> 16);   // lo rounded to nearest
  }
}

static inline int n_padded(int n) { return (n + 127) / 128 * 128; }

}  // namespace linear

extern "C" {

int dynmask_set_error(int code, const char* what);   // msda_capi.hip (shared last-error slot)

size_t linear_hip_packed_weight_bytes(int out_features, int in_features) {
  if (out_features <= 0 || in_features <= 0 || in_features % linear::kStepK != 0) return 0;
  return (size_t)(in_features / linear::kChunk) * 2 * linear::n_padded(out_features) * linear::kChunk * sizeof(uint16_t);
}

int linear_hip_pack_weight_f32(const float* weight, int out_features, int in_features, void* packed, void* stream) {
  if (out_features <= 0 || in_features <= 0) return dynmask_set_error(LINEAR_ERR_BAD_DIMS, "linear: bad dimensions");
  if (in_features % linear::kStepK != 0)
    return dynmask_set_error(LINEAR_ERR_UNSUPPORTED, "linear: in_features must be a multiple of 64");
  if (!weight || !packed) return dynmask_set_error(LINEAR_ERR_NULL_POINTER, "linear: null pointer argument");
  hipLaunchKernelGGL(linear::pack_weight_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, weight, out_features,
                     in_features, linear::n_padded(out_features), static_cast<uint16_t*>(packed));
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : dynmask_set_error((int)e, hipGetErrorString(e));
}

static int linear_impl(const float* x, const float* x2, const void* packed, const float* bias, const uint8_t* row_mask,
                       long long rows, int in_features, int out_features, int hm_rows, int act, float* out, void* stream) {
  if (act != 0 && act != 1) return dynmask_set_error(LINEAR_ERR_BAD_DIMS, "linear: activation must be 0 (none) or 1 (relu)");
  if (rows < 0 || in_features <= 0 || out_features <= 0)
    return dynmask_set_error(LINEAR_ERR_BAD_DIMS, "linear: bad dimensions");
  if (in_features % linear::kStepK != 0)
    return dynmask_set_error(LINEAR_ERR_UNSUPPORTED, "linear: in_features must be a multiple of 64");
  if (rows == 0) return 0;
  // 64-row tiles: a tile is only K / 64 steps long (4 at K = 256), so the kernel lives on the number of workgroups
  // a CU can overlap -- 33 KB of LDS each instead of 66 KB
  constexpr int BM = 64;
  const long long mt = (rows + BM - 1) / BM;
  if (mt >= (1ll << 31) || (long long)(out_features + 63) / 64 > 65535)
    return dynmask_set_error(LINEAR_ERR_BAD_DIMS, "linear: problem too large");
  if (!x || !packed || !out) return dynmask_set_error(LINEAR_ERR_NULL_POINTER, "linear: null pointer argument");
  const int n_pad = linear::n_padded(out_features);
  const uint32_t* pk = static_cast<const uint32_t*>(packed);
  hipStream_t st = (hipStream_t)stream;
  static const int forced_tj = std::getenv("LINEAR_TJ") ? std::atoi(std::getenv("LINEAR_TJ")) : 0;   // A/B hook: 1, 2, 4
  // (the packed weights are padded to 128 columns only: a 256-column workgroup needs out_features % 256 == 0)
  const bool wide_ok = out_features % 256 == 0 && mt * (out_features / 256) >= 512;
  if (wide_ok && (forced_tj == 4 || (forced_tj == 0 && out_features == 256))) {
    // 256 columns per workgroup: with out_features == 256 the activation tile is read, split and staged ONCE
    // (-3.5 % at K = 256, -7 % at K = 1024; no gain for wider outputs, profiles/r01_linear_tiles.txt)
    dim3 grid((unsigned)mt, (unsigned)((out_features + 255) / 256));
    if (x2) hipLaunchKernelGGL((linear::linear_packed<4, BM, true, 1>), grid, dim3(linear::kThreads), 0, st, x, x2, pk, bias, row_mask, rows,
                       in_features, out_features, n_pad, hm_rows, act, out);
    else hipLaunchKernelGGL((linear::linear_packed<4, BM, false, 1>), grid, dim3(linear::kThreads), 0, st, x, x2, pk, bias, row_mask, rows,
                       in_features, out_features, n_pad, hm_rows, act, out);
  } else
  // 128 columns per workgroup unless that leaves CUs idle
  if (forced_tj == 2 || (forced_tj == 0 && out_features > 64 && mt * ((out_features + 127) / 128) >= 512)) {
    dim3 grid((unsigned)mt, (unsigned)((out_features + 127) / 128));
    if (x2) hipLaunchKernelGGL((linear::linear_packed<2, BM, true, 1>), grid, dim3(linear::kThreads), 0, st, x, x2, pk, bias, row_mask, rows,
                       in_features, out_features, n_pad, hm_rows, act, out);
    else hipLaunchKernelGGL((linear::linear_packed<2, BM, false, 1>), grid, dim3(linear::kThreads), 0, st, x, x2, pk, bias, row_mask, rows,
                       in_features, out_features, n_pad, hm_rows, act, out);
  } else {
    dim3 grid((unsigned)mt, (unsigned)((out_features + 63) / 64));
    if (x2) hipLaunchKernelGGL((linear::linear_packed<1, BM, true, 2>), grid, dim3(linear::kThreads), 0, st, x, x2, pk, bias, row_mask, rows,
                       in_features, out_features, n_pad, hm_rows, act, out);
    else hipLaunchKernelGGL((linear::linear_packed<1, BM, false, 2>), grid, dim3(linear::kThreads), 0, st, x, x2, pk, bias, row_mask, rows,
                       in_features, out_features, n_pad, hm_rows, act, out);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : dynmask_set_error((int)e, hipGetErrorString(e));
}


int linear_hip_packed_f32(const float* x, const void* packed, const float* bias, const uint8_t* row_mask,
                          long long rows, int in_features, int out_features, float* out, void* stream) {
  return linear_impl(x, nullptr, packed, bias, row_mask, rows, in_features, out_features, 0, 0, out, stream);
}

int linear_hip_packed_hm_f32(const float* x, const void* packed, const float* bias, const uint8_t* row_mask,
                             long long rows, int in_features, int out_features, int rows_per_image, float* out,
                             void* stream) {
  if (rows_per_image <= 0 || out_features % 32 != 0 || (rows >= 0 && rows % rows_per_image != 0))
    return dynmask_set_error(LINEAR_ERR_BAD_DIMS, "linear (head-major): out_features must be a multiple of 32 and rows a multiple of rows_per_image");
  return linear_impl(x, nullptr, packed, bias, row_mask, rows, in_features, out_features, rows_per_image, 0, out, stream);
}

int linear_hip_packed_ln_f32(const float* x, const void* packed, const float* bias, const float* residual,
                             const float* gamma, const float* beta, float eps, long long rows, int in_features,
                             int out_features, float* out, void* stream) {
  if (rows < 0 || in_features <= 0 || out_features <= 0)
    return dynmask_set_error(LINEAR_ERR_BAD_DIMS, "linear: bad dimensions");
  if (in_features % linear::kStepK != 0 || out_features != 256)
    return dynmask_set_error(LINEAR_ERR_UNSUPPORTED, "linear + LayerNorm: in_features must be a multiple of 64 and out_features 256");
  if (rows == 0) return 0;
  const long long mt = (rows + 63) / 64;
  if (mt >= (1ll << 31)) return dynmask_set_error(LINEAR_ERR_BAD_DIMS, "linear: problem too large");
  if (!x || !packed || !out) return dynmask_set_error(LINEAR_ERR_NULL_POINTER, "linear: null pointer argument");
  hipLaunchKernelGGL((linear::linear_packed<4, 64, false, 1, true>), dim3((unsigned)mt, 1u), dim3(linear::kThreads), 0,
                     (hipStream_t)stream, x, nullptr, static_cast<const uint32_t*>(packed), bias, nullptr, rows, in_features,
                     out_features, linear::n_padded(out_features), 0, 0, out, linear::LnArgs{residual, gamma, beta, eps});
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : dynmask_set_error((int)e, hipGetErrorString(e));
}

int linear_hip_packed_ex_f32(const float* x, const float* x_add, const void* packed, const float* bias,
                             const uint8_t* row_mask, long long rows, int in_features, int out_features, int activation,
                             float* out, void* stream) {
  return linear_impl(x, x_add, packed, bias, row_mask, rows, in_features, out_features, 0, activation, out, stream);
}

}  // extern "C"
