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
#include <type_traits>

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

// out = LayerNorm(acc + bias + residual) * gamma + beta for a workgroup that holds whole rows: NW waves x (2 row tiles x WJ
// column tiles), wave `wv` owns columns wn .. wn + 32 WJ - 1 of the N = 32 WJ NW columns, all 64 rows.  Two-pass
// statistics like add_layernorm: row sums are reduced over the 32 lanes of a half-wave with DPP + one xor-16 exchange,
// then over the waves through `red` (>= 64 (NW + 1) floats of LDS that nobody else touches any more).
template <int WJ, int NW = 4>
__device__ __forceinline__ void layernorm_epilogue(const f32x16 (&acc)[2][WJ], const float* __restrict__ bias,
                                                   const LnArgs& ln, float* red, long long m0, long long M, int N, int wn,
                                                   int tid, float* __restrict__ out) {
  constexpr int TI = 2;
  const int lane = tid & 63, wv = tid >> 6, r32 = lane & 31, half = lane >> 5;
  auto sum32 = [](float v) {
    v += msda_dpp<0xB1>(v); v += msda_dpp<0x4E>(v); v += msda_dpp<0x141>(v); v += msda_dpp<0x140>(v);
    return v + __shfl_xor(v, 16, 64);
  };
  const int rows_here = (int)((M - m0) < 64ll ? (M - m0) : 64ll);
  const bool full = rows_here == 64;
  const float* const res_tile = ln.residual ? ln.residual + m0 * N : nullptr;
  float* const out_tile = out + m0 * N;
  const uint32_t lane_off = (uint32_t)(4 * half) * (uint32_t)N;
  float val[TI][WJ][16];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int jn = 0; jn < WJ; ++jn) {
      const int n = wn + jn * 32 + r32;
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        // 32-bit offsets inside the workgroup's 64-row tile (its first row sits in the uniform bases): per element in 64
        // bits this was two quarter-rate multiplies and a 64-bit compare.  Full tiles: lane part + a scalar multiple of N
        const int dr = i * 32 + 8 * (v / 4) + (v % 4);
        uint32_t off = lane_off + (uint32_t)n + (uint32_t)dr * (uint32_t)N;
        if (!full) {
          const int rt = dr + 4 * half;
          off = (uint32_t)(rt < rows_here ? rt : rows_here - 1) * (uint32_t)N + (uint32_t)n;
        }
        val[i][jn][v] = acc[i][jn][v] + bv + (ln.residual ? res_tile[off] : 0.f);
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
    if (tid < 64) {
      float t = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
      if constexpr (NW == 8) t += (red[256 + tid] + red[320 + tid]) + (red[384 + tid] + red[448 + tid]);
      red[64 * NW + tid] = t;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int v = 0; v < 16; ++v) stat[i][v] = red[64 * NW + i * 32 + 8 * (v / 4) + 4 * half + (v % 4)];
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
        const int dr = i * 32 + 8 * (v / 4) + (v % 4);
        const float rstd = rsqrtf(stat[i][v] * inv_n + ln.eps);
        if (full || dr + 4 * half < rows_here)
          out_tile[lane_off + (uint32_t)n + (uint32_t)dr * (uint32_t)N] = (val[i][jn][v] - mean[i][v]) * rstd * gmm + bt;
      }
    }
}

template <int TJ, int BM, bool ADD, int WM, bool LN = false>   // ADD: the input is x + x2
__global__ void __launch_bounds__(kThreads, 2)
linear_packed(const float* __restrict__ x, const float* __restrict__ x2, const uint32_t* __restrict__ packed,
              const float* __restrict__ bias, const uint8_t* __restrict__ row_mask, long long M, int K, int N, int n_pad,
              int hm_rows, int act, float* __restrict__ out, LnArgs ln = LnArgs{nullptr, nullptr, nullptr, 0.f},
              float* __restrict__ out2 = nullptr, int split_col = 0) {
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
    // a quarter per wave); the operand tile is dead by now and lends its LDS to the row statistics
    static_assert(WM == 1 && BM == 64, "LayerNorm epilogue: every wave spans the 64 rows");
    layernorm_epilogue<WJ>(acc, bias, ln, reinterpret_cast<float*>(&As[0][0][0][0][0]), m0, M, N, wn, tid, out);
    return;
  }
  // epilogue: accumulator register v of lane l is (row 8 (v / 4) + 4 (l / 32) + v % 4, column l % 32).
  // All per-element arithmetic is 32-bit and relative to the workgroup's tile: the tile's first row goes into a
  // uniform 64-bit base once (m * N + n per element in 64 bits was two quarter-rate multiplies, a 64-bit compare and --
  // on the head-major path -- a 64-bit DIVISION per stored value: more issue time than the tile's MFMAs).
  const int rows_here = (int)((M - m0) < (long long)BM ? (M - m0) : (long long)BM);   // valid rows of this tile
  // head-major output [image, head = n / 32, row in image, n % 32]: a tile of BM <= hm_rows rows touches at most two images
  long long img0 = 0;
  int first_sr = 0;                                   // row inside image img0 of the tile's first row
  if (hm_rows != 0) {
    img0 = m0 / hm_rows;                              // uniform: once per workgroup
    first_sr = (int)(m0 - img0 * hm_rows);
  }
  // split_col > 0: columns [0, split_col) go to `out` (row pitch split_col), the rest to `out2` (row pitch N - split_col):
  // two Linear layers on the same input as one product.  A column block never straddles the split (host-checked).
  const bool second = split_col > 0 && n0 >= split_col;     // workgroup-uniform
  const int ld = split_col > 0 ? (second ? N - split_col : split_col) : N;   // row pitch = columns of this output
  const int ncol0 = second ? split_col : 0;
  float* const out_tile = hm_rows == 0 ? (second ? out2 : out) + m0 * ld : out + img0 * (long long)(N / 32) * hm_rows * 32;
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    uint32_t zero_rows = 0;                            // bit v: row of register v is masked
    if (row_mask) {
#pragma unroll
      for (int v4 = 0; v4 < 4; ++v4) {                 // rows 8 v4 + 4 half .. + 3 are consecutive: one 4-byte load
        const int rt = wm + i * 32 + 8 * v4 + 4 * half;   // row inside the tile (a multiple of 4; m0 is a multiple of BM)
        uint32_t four = 0;
        if (rt + 3 < rows_here) four = *reinterpret_cast<const uint32_t*>(row_mask + m0 + rt);
        else
          for (int e = 0; e < 4; ++e) if (rt + e < rows_here && row_mask[m0 + rt + e]) four |= 0xffu << (8 * e);
#pragma unroll
        for (int e = 0; e < 4; ++e) if ((four >> (8 * e)) & 0xffu) zero_rows |= 1u << (4 * v4 + e);
      }
    }
    // offset of this lane's first row (4 * half) of row tile i; the other rows are compile-time multiples of N away
    // (scalar multiplies: N is uniform)
    const uint32_t lane_row = (uint32_t)(wm + i * 32 + 4 * half);
    auto store_tile = [&](auto full_tag) __attribute__((always_inline)) {
      constexpr bool FULL = decltype(full_tag)::value;   // the whole tile is inside the matrix: no per-element checks
#pragma unroll
      for (int jn = 0; jn < WJ; ++jn) {
        const int n = n0 + wn + jn * 32 + r32;
        const bool n_ok = FULL || n < N;
        const float bv = (bias && n_ok) ? bias[n] : 0.f;
        // element offset of (row, column n) inside the tile's output: row-major, or head-major with the image switch
        const uint32_t col_off = hm_rows == 0 ? (uint32_t)(n - ncol0) : (uint32_t)(n >> 5) * (uint32_t)hm_rows * 32u + (uint32_t)(n & 31);
        uint32_t base_off = hm_rows == 0 ? lane_row * (uint32_t)ld + col_off : col_off;
        asm volatile("" : "+v"(base_off));   // opaque: `base_off + scalar` per element stays an add (see conv3x3.hip)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          constexpr int kDummy = 0;
          (void)kDummy;
          const int dr = 8 * (v / 4) + (v % 4);            // compile-time row distance from lane_row
          const int rt = (int)lane_row + dr;
          if (FULL || (rt < rows_here && n_ok)) {
            float r = ((zero_rows >> v) & 1u) ? 0.f : acc[i][jn][v] + bv;
            if (act == 1) r = fmaxf(r, 0.f);
            uint32_t off;
            if (hm_rows == 0) {
              off = base_off + (uint32_t)dr * (uint32_t)ld;
            } else {
              int sr = first_sr + rt;                     // row inside image img0, or past its end: the next image
              uint32_t img_off = 0;
              if (sr >= hm_rows) { sr -= hm_rows; img_off = (uint32_t)(N / 32) * (uint32_t)hm_rows * 32u; }
              off = img_off + (uint32_t)sr * 32u + base_off;
            }
            out_tile[off] = r;
          }
        }
      }
    };
    if (rows_here == BM && n0 + 64 * TJ <= N) store_tile(std::true_type{});
    else store_tile(std::false_type{});
  }
}

// ------------------------------------------------------------------------------------------------
// Fused feed-forward block: out = [LayerNorm](residual + b2 + relu(x W1^T + b1) W2^T) for d_model = 256.
// The two Linear kernels above spend most of their time moving the [rows, d_ffn] hidden tensor (182 MB written and read
// back at the R50 shapes); here a workgroup keeps its 64 rows on the CU: the split x tile (all 256 k) stays in LDS,
// the hidden activations are produced 128 columns at a time (phase 1, TRANSPOSED accumulator tiles -- weights as the
// A operand -- so that a lane holds 4 consecutive hidden columns of one row and writes them as one 8-byte LDS store,
// split into bf16 hi / lo on the way), and are consumed at once as the K operand of the second product (phase 2),
// whose 64 x 256 fp32 accumulators live in registers for the whole kernel.  Both weight matrices stream from L2 through
// two register rings that stay primed across the phases.  Products, operand order and accumulation order are those of
// linear_packed: the 4-wave form equals the two-kernel path bit for bit, the 8-wave form up to the order of the
// LayerNorm row sums.  135 KB of LDS: one workgroup per CU (phase timings: profiles/r01_ffn_fused.txt).
constexpr int kFfnD = 256, kFfnBM = 64;
// LDS operand tiles are [k chunk][k half (0-7 / 8-15)][row][8 bf16]: with a 16-byte row pitch each 16-lane group that
// the hardware serves per ds_read_b128 pass ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, +32) covers the 64 banks exactly
// once; the [row][16 bf16] layout (32-byte pitch) has 2-way conflicts on every operand read (measured: conflict cycles =
// 65 % of the LDS instruction cycles, profiles/r01_ffn_pmc_before_lds_fix.txt).  8 pad words per plane spread the
// staging stores of the different (chunk, half) over the banks.
constexpr int kPlane = kFfnBM * 4 + 8;                                   // words per [row][8 bf16] plane
constexpr int kFfnXsWords = 2 * (kFfnD / kChunk) * 2 * kPlane;
constexpr int kFfnHsWords = 2 * 2 * (128 / kChunk) * 2 * kPlane;          // two 128-column tiles, or one of 256 columns
constexpr int kFfnLdsBytes = (kFfnXsWords + kFfnHsWords) * 4;

// NW waves (4 or 8): a hidden chunk is 32 NW columns (one 32-column tile per wave in phase 1), phase 2 gives every
// wave 256 / NW output columns.  NW = 8 puts TWO waves on each SIMD -- while one converts its hidden tile on the VALU
// or waits for a weight fragment the other keeps the matrix pipe busy -- at the price of a single hidden buffer (two
// barriers per chunk) and needs d_ffn % 256 == 0.
template <bool LN, int NW>
__global__ void __launch_bounds__(64 * NW, 1)
ffn_packed(const float* __restrict__ x, const uint32_t* __restrict__ packed1, const float* __restrict__ bias1,
           const uint32_t* __restrict__ packed2, const float* __restrict__ bias2, long long M, int F, int f_pad,
           float* __restrict__ out, LnArgs ln) {
  extern __shared__ __attribute__((aligned(16))) uint32_t ffn_smem[];
  constexpr int kFfnHC = 32 * NW, WJ2 = 8 / NW, NBUF = NW == 4 ? 2 : 1;
  constexpr int KC1 = kFfnD / kChunk, KC2 = kFfnHC / kChunk, BM = kFfnBM;
  typedef uint32_t (*XsT)[KC1][2][kPlane];        // [hi / lo][k chunk][k half][row * 4 + word]
  typedef uint32_t (*HsT)[2][KC2][2][kPlane];     // [buffer][hi / lo][k chunk][k half][row * 4 + word]
  XsT Xs = reinterpret_cast<XsT>(ffn_smem);
  HsT Hs = reinterpret_cast<HsT>(ffn_smem + kFfnXsWords);

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r32 = lane & 31, half = lane >> 5;
  const long long m0 = (long long)blockIdx.x * BM;

  // weight fragment streams: g1 = (hidden chunk, k chunk) of W1 in execution order, g2 = k chunk of W2
  struct W1F { u32x4v hi, lo; };
  struct W2F { u32x4v hi[WJ2], lo[WJ2]; };
  const int n1 = (F / kFfnHC) * KC1, n2 = F / kChunk;
  const long long cs1 = (long long)2 * f_pad * 8, ps1 = (long long)f_pad * 8;
  const long long cs2 = (long long)2 * kFfnD * 8, ps2 = (long long)kFfnD * 8;
  const uint32_t* w1_lane = packed1 + (long long)(wv * 32 + r32) * 8 + half * 4;
  const uint32_t* w2_lane = packed2 + (long long)(wv * 32 * WJ2 + r32) * 8 + half * 4;
  auto load_w1 = [&](int g, W1F& f) {
    g = g < n1 ? g : n1 - 1;
    const uint32_t* p = w1_lane + (long long)(g % KC1) * cs1 + (long long)(g / KC1) * (kFfnHC * 8);
    f.hi = *reinterpret_cast<const u32x4v*>(p);
    f.lo = *reinterpret_cast<const u32x4v*>(p + ps1);
  };
  auto load_w2 = [&](int g, W2F& f) {
    g = g < n2 ? g : n2 - 1;
    const uint32_t* p = w2_lane + (long long)g * cs2;
#pragma unroll
    for (int jn = 0; jn < WJ2; ++jn) {
      f.hi[jn] = *reinterpret_cast<const u32x4v*>(p + jn * 32 * 8);
      f.lo[jn] = *reinterpret_cast<const u32x4v*>(p + ps2 + jn * 32 * 8);
    }
  };
  // one wave per SIMD: nothing but the rings hides the L2 latency of the weight fragments (R - 1 chunks ahead)
  constexpr int R1 = 8, R2 = 8;
  static_assert(KC1 % R1 == 0 && KC2 % R2 == 0, "ring slots are compile-time constants");
  W1F ring1[R1];
  W2F ring2[R2];
#pragma unroll
  for (int r = 0; r < R1 - 1; ++r) load_w1(r, ring1[r]);
#pragma unroll
  for (int r = 0; r < R2 - 1; ++r) load_w2(r, ring2[r]);

  // ---- phase 0: the 64 x 256 tile of x, split, into LDS (a step = 64 k = 16 pieces of 16 bytes per row) --------
  {
    const int s_piece = tid & 15, s_row0 = tid >> 4;
    const int cc = s_piece >> 2, kh = (s_piece >> 1) & 1, w2 = (s_piece & 1) * 2;
    constexpr int RS = 4 * NW, NR = BM / RS;            // rows per pass, passes
    f32x4 xa[4][NR];
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        long long mr = m0 + s_row0 + RS * r;
        mr = mr < M ? mr : M - 1;
        xa[st][r] = *reinterpret_cast<const f32x4*>(x + mr * kFfnD + st * kStepK + s_piece * 4);
      }
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        uint32_t hi[2], lo[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const float fa = xa[st][r][2 * p], fb = xa[st][r][2 * p + 1];
          const uint32_t ah = __float_as_uint(fa) & 0xffff0000u, bh = __float_as_uint(fb) & 0xffff0000u;
          const uint32_t al = __float_as_uint(fa - __uint_as_float(ah)), bl = __float_as_uint(fb - __uint_as_float(bh));
          hi[p] = (ah >> 16) | bh;
          lo[p] = ((al + 0x8000u) >> 16) | ((bl + 0x8000u) & 0xffff0000u);   // lo rounded to nearest
        }
        *reinterpret_cast<uint2*>(&Xs[0][st * 4 + cc][kh][(s_row0 + RS * r) * 4 + w2]) = make_uint2(hi[0], hi[1]);
        *reinterpret_cast<uint2*>(&Xs[1][st * 4 + cc][kh][(s_row0 + RS * r) * 4 + w2]) = make_uint2(lo[0], lo[1]);
      }
  }
  __syncthreads();

  f32x16 acc2[2][WJ2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < WJ2; ++jn)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc2[i][jn][v] = 0.f;

  const int nhc = F / kFfnHC;
  // ---- phase 1: hT[hidden column, row] for this wave's 32 hidden columns x 64 rows of hidden chunk hc ------------
  auto phase1 = [&](int hc, f32x16 (&acc1)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc1[i][v] = 0.f;
    const int g1 = hc * KC1;
#pragma unroll
    for (int kc = 0; kc < KC1; ++kc) {
      load_w1(g1 + kc + R1 - 1, ring1[(kc + R1 - 1) % R1]);
      const W1F& wf = ring1[kc % R1];
      const bf16x8 wh = __builtin_bit_cast(bf16x8, wf.hi), wl = __builtin_bit_cast(bf16x8, wf.lo);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(&Xs[0][kc][half][(i * 32 + r32) * 4]));
        const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(&Xs[1][kc][half][(i * 32 + r32) * 4]));
        acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ah, acc1[i], 0, 0, 0);   // weights as A: transposed tile
        acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al, acc1[i], 0, 0, 0);
        acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah, acc1[i], 0, 0, 0);
      }
    }
  };
  // bias + ReLU + split -> Hs[hc & 1]; register v of lane l is (hidden column 8 (v / 4) + 4 (l / 32) + v % 4, row l % 32)
  auto hidden_to_lds = [&](int hc, const f32x16 (&acc1)[2]) {
    const int hb = hc & (NBUF - 1);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = hc * kFfnHC + wv * 32 + 8 * g + 4 * half;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (bias1) bv = *reinterpret_cast<const f32x4*>(bias1 + col);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        uint32_t hi[2], lo[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const float fa = fmaxf(acc1[i][4 * g + 2 * p] + bv[2 * p], 0.f), fb = fmaxf(acc1[i][4 * g + 2 * p + 1] + bv[2 * p + 1], 0.f);
          const uint32_t ah = __float_as_uint(fa) & 0xffff0000u, bh = __float_as_uint(fb) & 0xffff0000u;
          const uint32_t al = __float_as_uint(fa - __uint_as_float(ah)), bl = __float_as_uint(fb - __uint_as_float(bh));
          hi[p] = (ah >> 16) | bh;
          lo[p] = ((al + 0x8000u) >> 16) | ((bl + 0x8000u) & 0xffff0000u);
        }
        const int kc = wv * 2 + (g >> 1), word = (i * 32 + r32) * 4 + 2 * half;   // k = 8 (g & 1) + 4 half .. + 3 of the chunk
        *reinterpret_cast<uint2*>(&Hs[hb][0][kc][g & 1][word]) = make_uint2(hi[0], hi[1]);
        *reinterpret_cast<uint2*>(&Hs[hb][1][kc][g & 1][word]) = make_uint2(lo[0], lo[1]);
      }
    }
  };
  // ---- phase 2: out[row, column] += h[row, 128 hidden of chunk hc] W2[column, the same 128 hidden] ---------------
  auto phase2 = [&](int hc) {
    const int hb = hc & (NBUF - 1), g2 = hc * KC2;
#pragma unroll
    for (int kc = 0; kc < KC2; ++kc) {
      load_w2(g2 + kc + R2 - 1, ring2[(kc + R2 - 1) % R2]);
      const W2F& wf = ring2[kc % R2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(&Hs[hb][0][kc][half][(i * 32 + r32) * 4]));
        const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(&Hs[hb][1][kc][half][(i * 32 + r32) * 4]));
#pragma unroll
        for (int jn = 0; jn < WJ2; ++jn) {
          const bf16x8 wh = __builtin_bit_cast(bf16x8, wf.hi[jn]), wl = __builtin_bit_cast(bf16x8, wf.lo[jn]);
          acc2[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl, acc2[i][jn], 0, 0, 0);
          acc2[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh, acc2[i][jn], 0, 0, 0);
          acc2[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, acc2[i][jn], 0, 0, 0);
        }
      }
    }
  };
  // NW = 4: one barrier per chunk -- it publishes Hs[hc & 1]; the other buffer is free again by then (every wave has
  // left the phase 2 that read it).  NW = 8: a single buffer, so a second barrier retires its readers.  Issuing
  // phase2(hc - 1) between phase1(hc) and its epilogue instead (to give the matrix pipe work while the VALU converts)
  // measured 17 % slower: the in-order wave only reaches the VALU code later.
  for (int hc = 0; hc < nhc; ++hc) {
    f32x16 acc1[2];
    phase1(hc, acc1);
    if (NBUF == 1 && hc > 0) __syncthreads();
    hidden_to_lds(hc, acc1);
    __syncthreads();
    phase2(hc);
  }

  const int wn = wv * 32 * WJ2;
  if constexpr (LN) {
    // Xs is dead since the last barrier (phase 2 only reads the hidden tile): it lends its LDS to the row statistics
    layernorm_epilogue<WJ2, NW>(acc2, bias2, ln, reinterpret_cast<float*>(ffn_smem), m0, M, kFfnD, wn, tid, out);
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jn = 0; jn < WJ2; ++jn) {
        const int n = wn + jn * 32 + r32;
        const float bv = bias2 ? bias2[n] : 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const long long m = m0 + i * 32 + 8 * (v / 4) + 4 * half + (v % 4);
          if (m < M) out[m * kFfnD + n] = acc2[i][jn][v] + bv + (ln.residual ? ln.residual[m * kFfnD + n] : 0.f);
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
                       long long rows, int in_features, int out_features, int hm_rows, int act, float* out, void* stream,
                       float* out2 = nullptr, int split_col = 0) {
  if (split_col != 0 && (split_col < 0 || split_col >= out_features || split_col % 128 != 0 || hm_rows != 0 || !out2))
    return dynmask_set_error(LINEAR_ERR_BAD_DIMS, "linear (two outputs): the split column must be a multiple of 128 inside (0, out_features)");
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
  if (mt >= (1ll << 31) || (long long)(out_features + 63) / 64 > 65535 || out_features >= (1 << 24))
    return dynmask_set_error(LINEAR_ERR_BAD_DIMS, "linear: problem too large");
  if (!x || !packed || !out) return dynmask_set_error(LINEAR_ERR_NULL_POINTER, "linear: null pointer argument");
  const int n_pad = linear::n_padded(out_features);
  const uint32_t* pk = static_cast<const uint32_t*>(packed);
  hipStream_t st = (hipStream_t)stream;
  static const int forced_tj = msda::ab_env_int("LINEAR_TJ", 0);   // A/B hook: 1, 2, 4
  // (the packed weights are padded to 128 columns only: a 256-column workgroup needs out_features % 256 == 0)
  const bool wide_ok = out_features % 256 == 0 && mt * (out_features / 256) >= 512 && split_col % 256 == 0;
  if (wide_ok && (forced_tj == 4 || (forced_tj == 0 && out_features == 256))) {
    // 256 columns per workgroup: with out_features == 256 the activation tile is read, split and staged ONCE
    // (-3.5 % at K = 256, -7 % at K = 1024; no gain for wider outputs, profiles/r01_linear_tiles.txt)
    dim3 grid((unsigned)mt, (unsigned)((out_features + 255) / 256));
    if (x2) hipLaunchKernelGGL((linear::linear_packed<4, BM, true, 1>), grid, dim3(linear::kThreads), 0, st, x, x2, pk, bias, row_mask, rows,
                       in_features, out_features, n_pad, hm_rows, act, out, linear::LnArgs{nullptr, nullptr, nullptr, 0.f}, out2, split_col);
    else hipLaunchKernelGGL((linear::linear_packed<4, BM, false, 1>), grid, dim3(linear::kThreads), 0, st, x, x2, pk, bias, row_mask, rows,
                       in_features, out_features, n_pad, hm_rows, act, out, linear::LnArgs{nullptr, nullptr, nullptr, 0.f}, out2, split_col);
  } else
  // 128 columns per workgroup unless that leaves CUs idle
  if (forced_tj == 2 || (forced_tj == 0 && out_features > 64 && mt * ((out_features + 127) / 128) >= 512)) {
    dim3 grid((unsigned)mt, (unsigned)((out_features + 127) / 128));
    if (x2) hipLaunchKernelGGL((linear::linear_packed<2, BM, true, 1>), grid, dim3(linear::kThreads), 0, st, x, x2, pk, bias, row_mask, rows,
                       in_features, out_features, n_pad, hm_rows, act, out, linear::LnArgs{nullptr, nullptr, nullptr, 0.f}, out2, split_col);
    else hipLaunchKernelGGL((linear::linear_packed<2, BM, false, 1>), grid, dim3(linear::kThreads), 0, st, x, x2, pk, bias, row_mask, rows,
                       in_features, out_features, n_pad, hm_rows, act, out, linear::LnArgs{nullptr, nullptr, nullptr, 0.f}, out2, split_col);
  } else {
    dim3 grid((unsigned)mt, (unsigned)((out_features + 63) / 64));
    if (x2) hipLaunchKernelGGL((linear::linear_packed<1, BM, true, 2>), grid, dim3(linear::kThreads), 0, st, x, x2, pk, bias, row_mask, rows,
                       in_features, out_features, n_pad, hm_rows, act, out, linear::LnArgs{nullptr, nullptr, nullptr, 0.f}, out2, split_col);
    else hipLaunchKernelGGL((linear::linear_packed<1, BM, false, 2>), grid, dim3(linear::kThreads), 0, st, x, x2, pk, bias, row_mask, rows,
                       in_features, out_features, n_pad, hm_rows, act, out, linear::LnArgs{nullptr, nullptr, nullptr, 0.f}, out2, split_col);
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
  // the epilogue addresses an image pair with 32-bit element offsets and lets a 64-row tile straddle one image boundary
  if (rows_per_image < 64 || (long long)rows_per_image * out_features >= (1ll << 30))
    return dynmask_set_error(LINEAR_ERR_UNSUPPORTED, "linear (head-major): rows_per_image must be >= 64 and rows_per_image * out_features < 2^30");
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

int linear_hip_packed_split_f32(const float* x, const float* x_add, const void* packed, const float* bias, long long rows,
                                int in_features, int out_features, int split_col, float* out_a, float* out_b, void* stream) {
  if (split_col <= 0) return dynmask_set_error(LINEAR_ERR_BAD_DIMS, "linear (two outputs): split_col must be > 0");
  return linear_impl(x, x_add, packed, bias, nullptr, rows, in_features, out_features, 0, 0, out_a, stream, out_b, split_col);
}

int linear_hip_packed_ffn_f32(const float* x, const void* packed1, const float* bias1, const void* packed2,
                              const float* bias2, const float* residual, const float* gamma, const float* beta, float eps,
                              int layer_norm, long long rows, int d_model, int d_ffn, float* out, void* stream) {
  if (rows < 0 || d_model <= 0 || d_ffn <= 0) return dynmask_set_error(LINEAR_ERR_BAD_DIMS, "ffn: bad dimensions");
  if (d_model != linear::kFfnD || d_ffn % 128 != 0)
    return dynmask_set_error(LINEAR_ERR_UNSUPPORTED, "ffn: d_model must be 256 and d_ffn a multiple of 128");
  if (rows == 0) return 0;
  const long long mt = (rows + linear::kFfnBM - 1) / linear::kFfnBM;
  if (mt >= (1ll << 31)) return dynmask_set_error(LINEAR_ERR_BAD_DIMS, "ffn: problem too large");
  if (!x || !packed1 || !packed2 || !out) return dynmask_set_error(LINEAR_ERR_NULL_POINTER, "ffn: null pointer argument");
  const linear::LnArgs ln{residual, gamma, beta, eps};
  const uint32_t* p1 = static_cast<const uint32_t*>(packed1);
  const uint32_t* p2 = static_cast<const uint32_t*>(packed2);
  const int f_pad = linear::n_padded(d_ffn);
  static const int forced_nw = msda::ab_env_int("LINEAR_FFN_WAVES", 0);   // A/B hook
  const bool eight = d_ffn % 256 == 0 && forced_nw != 4;
  static std::atomic<uint64_t> opted_in[4];
  auto launch = [&](auto kernel, int nw, std::atomic<uint64_t>& done) -> int {
    if (int rc = msda::ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), linear::kFfnLdsBytes, done))
      return dynmask_set_error(rc, "ffn: dynamic LDS opt-in failed");
    hipLaunchKernelGGL(kernel, dim3((unsigned)mt), dim3(64 * nw), linear::kFfnLdsBytes, (hipStream_t)stream, x, p1, bias1,
                       p2, bias2, rows, d_ffn, f_pad, out, ln);
    return 0;
  };
  int rc;
  if (layer_norm) rc = eight ? launch(linear::ffn_packed<true, 8>, 8, opted_in[0]) : launch(linear::ffn_packed<true, 4>, 4, opted_in[1]);
  else rc = eight ? launch(linear::ffn_packed<false, 8>, 8, opted_in[2]) : launch(linear::ffn_packed<false, 4>, 4, opted_in[3]);
  if (rc) return rc;
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : dynmask_set_error((int)e, hipGetErrorString(e));
}

}  // extern "C"
