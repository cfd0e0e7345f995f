// 3x3 convolution (stride 1, zero padding 1) + bias + optional ReLU as an fp32 implicit GEMM on the gfx950 matrix
// cores -- see include/conv3x3_hip.h.
//
//   C[n, m] = sum_k B[n, k] * A[m, k]     m = (b, y, x) output pixel, n = output channel, k = (c, ky, kx)
//   A[m, k] = in[b, c, y + ky - 1, x + kx - 1] (0 outside the image)  -- read in place: for a fixed k the pixels of
//             an image row are contiguous, so a wave loads 64 consecutive floats per k
//   B[n, k] = weight[n, c, ky, kx]                                   -- the Conv2d weight as stored (k contiguous)
//
// Workgroup: 256 threads = 4 waves (2 x 2), tile BM pixels x BN channels x 16 k (128 x 128 or 64 x 64); the A tile
// sits in LDS as [k][pixel] (written and read lane-contiguously: no bank conflicts), the B tile as [channel][k] with
// an 80-byte pitch (one ds_read_b128 = four k of a channel).  Both are double buffered: the global loads of tile
// t + 1 are in flight while tile t is multiplied, one barrier per tile.  The MFMA is issued with the weight fragment
// as its first operand, so the 32 x 32 accumulator tile has channels as rows and pixels as columns and the NCHW store
// of a lane row is 128 contiguous bytes.  Bound: the fp32 matrix pipe (157 TFLOP/s dense).
//
// precision 1 (conv3x3_gemm_bf16x3): every fp32 operand is split into two bf16 halves while it is staged into LDS,
// x = hi + lo with hi = the upper 16 bits of x and lo = bf16(x - hi), and a product is formed as
// hi_a hi_b + hi_a lo_b + lo_a hi_b with v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  Each operand keeps 16 mantissa
// bits, the dropped lo lo term is 2^-16 relative: errors of ~2e-5 of the output scale, inside the path's 1e-4 bound,
// at 3/16 of the matrix-pipe time of the exact kernel (the bf16 MFMA runs 16x faster).  Non-finite inputs give NaN.
#include "../../include/conv3x3_hip.h"

#include <cstdlib>

#include "msda_common.hpp"

namespace conv3x3 {

using msda::f32x4;
typedef float f32x16 __attribute__((__vector_size__(64)));

constexpr int kThreads = 256;
constexpr int BK = 16;
constexpr int kPitchB = 20;   // floats per weight row in LDS: 16 + 4 pad

struct Geom {
  int B, Cin, H, W, Cout, Mtot, K;
};

template <int BM, int BN, bool RELU>
__global__ void __launch_bounds__(kThreads, 2)
conv3x3_gemm(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias, Geom g,
             float* __restrict__ out) {
  constexpr int kAPer = BM * BK / kThreads;            // k values per thread for its pixel: 8 (BM 128) or 4 (BM 64)
  constexpr int kBLoads = BN * BK / 4 / kThreads;      // float4 weight loads per thread: 2 or 1
  constexpr int TI = BM / 64, TJ = BN / 64;            // 32 x 32 MFMA tiles per wave (waves are 2 x 2)
  constexpr int kPitchA = BM + 4;
  __shared__ __attribute__((aligned(16))) float As[2][BK][kPitchA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN][kPitchB];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int HW = g.H * g.W;

  // ---- this thread's slice of the operand tiles -------------------------------------------------------------
  const int a_m = tid % BM, a_kg = tid / BM;           // pixel of the tile, group of kAPer consecutive k
  int py, px;
  const float* a_base;                                 // &in[b, 0, y, x]
  bool a_live;
  {
    const int m = m0 + a_m;
    a_live = m < g.Mtot;
    const int mc = a_live ? m : g.Mtot - 1;
    const int b = mc / HW, sp = mc - b * HW;
    py = sp / g.W;
    px = sp - py * g.W;
    a_base = in + ((int64_t)b * g.Cin * g.H + py) * g.W + px;
  }
  const float* b_ptr[kBLoads];
  int b_row[kBLoads], b_kq[kBLoads];
#pragma unroll
  for (int i = 0; i < kBLoads; ++i) {
    const int idx = tid + i * kThreads;
    b_row[i] = idx / 4;
    b_kq[i] = idx % 4;
    int n = n0 + b_row[i];
    n = n < g.Cout ? n : g.Cout - 1;
    b_ptr[i] = w + (int64_t)n * g.K + b_kq[i] * 4;
  }

  float a_reg[kAPer];
  f32x4 b_reg[kBLoads];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < kAPer; ++i) {
      const int k = kt * BK + a_kg * kAPer + i;        // wave-uniform
      const int c = k / 9, r = k - 9 * c;
      const int dy = r / 3 - 1, dx = r - 3 * (r / 3) - 1;
      const bool ok = a_live && (unsigned)(py + dy) < (unsigned)g.H && (unsigned)(px + dx) < (unsigned)g.W;
      a_reg[i] = ok ? a_base[((int64_t)c * g.H + dy) * g.W + dx] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < kBLoads; ++i) b_reg[i] = *reinterpret_cast<const f32x4*>(b_ptr[i] + kt * BK);
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < kAPer; ++i) As[buf][a_kg * kAPer + i][a_m] = a_reg[i];
#pragma unroll
    for (int i = 0; i < kBLoads; ++i) *reinterpret_cast<f32x4*>(&Bs[buf][b_row[i]][b_kq[i] * 4]) = b_reg[i];
  };

  // ---- main loop --------------------------------------------------------------------------------------------
  const int wm = (wv >> 1) * (BM / 2), wn = (wv & 1) * (BN / 2);   // this wave's corner of the tile
  const int r32 = lane & 31, half = lane >> 5;
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int jn = 0; jn < TJ; ++jn)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][jn][v] = 0.f;

  const int KT = g.K / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) load_tile(kt + 1);
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {                    // 8 k per step: lanes 0-31 take k 0..3, lanes 32-63 k 4..7
      float af[TI][4];
      f32x4 bf[TJ];
#pragma unroll
      for (int jn = 0; jn < TJ; ++jn) bf[jn] = *reinterpret_cast<const f32x4*>(&Bs[buf][wn + jn * 32 + r32][ss * 8 + half * 4]);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) af[i][t] = As[buf][ss * 8 + half * 4 + t][wm + i * 32 + r32];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int jn = 0; jn < TJ; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[jn][t], af[i][t], acc[i][jn], 0, 0, 0);
    }
    if (kt + 1 < KT) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: accumulator register v of lane l is (channel row 8 (v / 4) + 4 (l / 32) + v % 4, pixel column l % 32)
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int m = m0 + wm + i * 32 + r32;
    const int b = m / HW, sp = m - b * HW;
#pragma unroll
    for (int jn = 0; jn < TJ; ++jn)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int n = n0 + wn + jn * 32 + 8 * (v / 4) + 4 * half + (v % 4);
        if (m < g.Mtot && n < g.Cout) {
          float r = acc[i][jn][v] + (bias ? bias[n] : 0.f);
          if (RELU) r = fmaxf(r, 0.f);
          out[((int64_t)b * g.Cout + n) * HW + sp] = r;
        }
      }
  }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4v __attribute__((__vector_size__(16)));

// split 8 floats into 8 bf16 "hi" (truncated upper halves) and 8 bf16 "lo" (upper halves of the remainders)
__device__ __forceinline__ void split8(const float (&v)[8], u32x4v& hi, u32x4v& lo) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const uint32_t a = __float_as_uint(v[2 * p]), b = __float_as_uint(v[2 * p + 1]);
    const uint32_t ah = a & 0xffff0000u, bh = b & 0xffff0000u;
    const uint32_t al = __float_as_uint(v[2 * p] - __uint_as_float(ah));
    const uint32_t bl = __float_as_uint(v[2 * p + 1] - __uint_as_float(bh));
    hi[p] = (ah >> 16) | bh;                       // element 2p in the low half-word
    lo[p] = ((al + 0x8000u) >> 16) | ((bl + 0x8000u) & 0xffff0000u);   // lo rounded to nearest
  }
}

template <int BM, int BN, bool RELU>
__global__ void __launch_bounds__(kThreads, 2)
conv3x3_gemm_bf16x3(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias, Geom g,
                    float* __restrict__ out) {
  static_assert(BM == 128 && BN == 128, "one thread per (row, 8 k) of each operand tile");
  constexpr int TI = BM / 64, TJ = BN / 64;
  // [buffer][hi / lo][row][16 bf16 = 8 words]: a row is 32 bytes, lanes read / write 16-byte halves contiguously
  __shared__ __attribute__((aligned(16))) uint32_t As[2][2][BM][8];
  __shared__ __attribute__((aligned(16))) uint32_t Bs[2][2][BN][8];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int HW = g.H * g.W;
  const int row = tid & 127, kg = tid >> 7;            // this thread stages 8 consecutive k of one row of A and of B
  int py, px;
  const float* a_base;
  bool a_live;
  {
    const int m = m0 + row;
    a_live = m < g.Mtot;
    const int mc = a_live ? m : g.Mtot - 1;
    const int b = mc / HW, sp = mc - b * HW;
    py = sp / g.W;
    px = sp - py * g.W;
    a_base = in + ((int64_t)b * g.Cin * g.H + py) * g.W + px;
  }
  // which of the 9 taps fall inside the image for this pixel: bit (dy + 1) * 3 + (dx + 1)
  uint32_t tap_ok = 0;
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const int dy = r / 3 - 1, dx = r % 3 - 1;
    if (a_live && (unsigned)(py + dy) < (unsigned)g.H && (unsigned)(px + dx) < (unsigned)g.W) tap_ok |= 1u << r;
  }
  const float* b_src;
  {
    int n = n0 + row;
    n = n < g.Cout ? n : g.Cout - 1;
    b_src = w + (int64_t)n * g.K + kg * 8;
  }

  float a_reg[8];
  f32x4 b_reg[2];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = kt * BK + kg * 8 + i;              // wave-uniform
      const int c = k / 9, r = k - 9 * c;
      const int dy = r / 3 - 1, dx = r - 3 * (r / 3) - 1;
      a_reg[i] = ((tap_ok >> r) & 1u) ? a_base[((int64_t)c * g.H + dy) * g.W + dx] : 0.f;
    }
    b_reg[0] = *reinterpret_cast<const f32x4*>(b_src + kt * BK);
    b_reg[1] = *reinterpret_cast<const f32x4*>(b_src + kt * BK + 4);
  };
  auto store_tile = [&](int buf) {
    u32x4v hi, lo;
    split8(a_reg, hi, lo);
    *reinterpret_cast<u32x4v*>(&As[buf][0][row][kg * 4]) = hi;
    *reinterpret_cast<u32x4v*>(&As[buf][1][row][kg * 4]) = lo;
    const float bv[8] = {b_reg[0][0], b_reg[0][1], b_reg[0][2], b_reg[0][3], b_reg[1][0], b_reg[1][1], b_reg[1][2], b_reg[1][3]};
    split8(bv, hi, lo);
    *reinterpret_cast<u32x4v*>(&Bs[buf][0][row][kg * 4]) = hi;
    *reinterpret_cast<u32x4v*>(&Bs[buf][1][row][kg * 4]) = lo;
  };

  const int wm = (wv >> 1) * (BM / 2), wn = (wv & 1) * (BN / 2);
  const int r32 = lane & 31, half = lane >> 5;
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int jn = 0; jn < TJ; ++jn)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][jn][v] = 0.f;

  const int KT = g.K / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) load_tile(kt + 1);
    bf16x8 ah[TI], al[TI], bh[TJ], bl[TJ];           // lane: row r32, k = 8 half .. 8 half + 7
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      ah[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(&As[buf][0][wm + i * 32 + r32][half * 4]));
      al[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(&As[buf][1][wm + i * 32 + r32][half * 4]));
    }
#pragma unroll
    for (int jn = 0; jn < TJ; ++jn) {
      bh[jn] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(&Bs[buf][0][wn + jn * 32 + r32][half * 4]));
      bl[jn] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(&Bs[buf][1][wn + jn * 32 + r32][half * 4]));
    }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int jn = 0; jn < TJ; ++jn) {                // small terms first
        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[jn], ah[i], acc[i][jn], 0, 0, 0);
        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[jn], al[i], acc[i][jn], 0, 0, 0);
        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[jn], ah[i], acc[i][jn], 0, 0, 0);
      }
    if (kt + 1 < KT) store_tile(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int m = m0 + wm + i * 32 + r32;
    const int b = m / HW, sp = m - b * HW;
#pragma unroll
    for (int jn = 0; jn < TJ; ++jn)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int n = n0 + wn + jn * 32 + 8 * (v / 4) + 4 * half + (v % 4);
        if (m < g.Mtot && n < g.Cout) {
          float r = acc[i][jn][v] + (bias ? bias[n] : 0.f);
          if (RELU) r = fmaxf(r, 0.f);
          out[((int64_t)b * g.Cout + n) * HW + sp] = r;
        }
      }
  }
}

static int launch_bf16x3(const float* in, const float* w, const float* bias, const Geom& g, int relu, float* out,
                         hipStream_t stream) {
  dim3 grid((unsigned)((g.Mtot + 127) / 128), (unsigned)((g.Cout + 127) / 128));
  if (relu) hipLaunchKernelGGL((conv3x3_gemm_bf16x3<128, 128, true>), grid, dim3(kThreads), 0, stream, in, w, bias, g, out);
  else hipLaunchKernelGGL((conv3x3_gemm_bf16x3<128, 128, false>), grid, dim3(kThreads), 0, stream, in, w, bias, g, out);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// conv3x3_packed<TJ, RELU>: the split-bf16 convolution restructured around what the generic kernel above showed: with
// the matrix-pipe time cut to 3/16 the per-tile barrier, the nine-fold re-read of the input and the weight staging
// become the bound (465 us vs 659 us exact for the 256 -> 256 layer at 100 x 167).  Here
//   * a workgroup owns an 8 x 16 pixel tile of one image and 64 TJ output channels; per chunk of 16 input channels it
//     stages the 10 x 18 HALO of the tile once (split into bf16 hi / lo, [pixel][16 channels] = one ds_read_b128 per
//     MFMA operand) and serves all nine taps from it with immediate LDS offsets -- one barrier per 108 MFMAs;
//   * the weights are split and re-ordered ONCE by conv3x3_hip_pack_weight_f32 into [chunk][tap][hi / lo][cout][16
//     channels] bf16, so a wave's weight fragment of a tap is 1 KB of contiguous memory that it loads straight into
//     registers (no LDS), two taps ahead of its use (ring of three register sets; 9 taps = 3 turns per chunk).
constexpr int kTH = 8, kTW = 16, kHaloW = kTW + 2, kHaloPx = (kTH + 2) * kHaloW;   // 180
constexpr int kChunk = 16;

// WM: waves along the pixel rows (2: a wave owns 4 rows = 2 sub-tiles and 32 TJ channels; 1: a wave owns all 8 rows =
// 4 sub-tiles and 16 TJ channels, so a weight fragment feeds twice as many MFMAs -- half the weight traffic through L1).
template <int TJ, bool RELU, int WM>
__global__ void __launch_bounds__(kThreads, 2)
conv3x3_packed(const float* __restrict__ in, const uint32_t* __restrict__ packed, const float* __restrict__ bias, Geom g,
               int tiles_x, int tiles_per_image, int cout_pad, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint32_t As[2][2][kHaloPx][8];   // [buffer][hi / lo][halo pixel][16 bf16]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.x / tiles_per_image, t_in = blockIdx.x - b * tiles_per_image;
  const int ty0 = (t_in / tiles_x) * kTH, tx0 = (t_in % tiles_x) * kTW;
  const int n0 = blockIdx.y * (64 * TJ);
  const int HW = g.H * g.W;

  // ---- halo staging: items (halo pixel, channel half); this thread owns items tid and tid + 256 ----------------
  const float* h_ptr[2];
  bool h_ok[2], h_has[2];
  int h_px[2], h_half[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int item = tid + r * kThreads;
    h_has[r] = item < 2 * kHaloPx;
    const int hp = item % kHaloPx;
    h_px[r] = hp;
    h_half[r] = (item / kHaloPx) & 1;
    const int gy = ty0 - 1 + hp / kHaloW, gx = tx0 - 1 + hp % kHaloW;
    h_ok[r] = h_has[r] && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W;
    h_ptr[r] = in + ((int64_t)b * g.Cin + h_half[r] * 8) * HW + (h_ok[r] ? gy * g.W + gx : 0);
  }
  float h_reg[2][8];
  auto load_halo = [&](int chunk) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int e = 0; e < 8; ++e) h_reg[r][e] = h_ok[r] ? h_ptr[r][(int64_t)(chunk * kChunk + e) * HW] : 0.f;
  };
  auto store_halo = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
      if (h_has[r]) {
        u32x4v hi, lo;
        split8(h_reg[r], hi, lo);
        *reinterpret_cast<u32x4v*>(&As[buf][0][h_px[r]][h_half[r] * 4]) = hi;
        *reinterpret_cast<u32x4v*>(&As[buf][1][h_px[r]][h_half[r] * 4]) = lo;
      }
  };

  // ---- MFMA fragments ----------------------------------------------------------------------------------------
  constexpr int WN = 4 / WM, TI = 4 / WM, WJ = 2 * TJ / WN;   // per wave: TI 32-pixel sub-tiles x WJ 32-channel tiles
  static_assert(WJ >= 1, "wave layout");
  const int wm = wv / WN, wn = wv % WN;
  const int r32 = lane & 31, half = lane >> 5;
  // LDS word offset of this lane's A fragment for sub-tile i, tap (0, 0): pixel (4 wm + 2 i + r32 / 16, r32 % 16)
  int a_off[TI];
#pragma unroll
  for (int i = 0; i < TI; ++i) a_off[i] = ((wm * 2 * TI + i * 2 + (r32 >> 4)) * kHaloW + (r32 & 15)) * 8 + half * 4;
  // packed weights: u32 index of (flat tap ft, part, channel n, half) = ((ft * 2 + part) * cout_pad + n) * 8 + half * 4
  const int nb = n0 + wn * 32 * WJ + r32;
  const uint32_t* w_lane = packed + (int64_t)nb * 8 + half * 4;
  const int64_t tap_stride = (int64_t)2 * cout_pad * 8, part_stride = (int64_t)cout_pad * 8;
  const int nchunks = g.Cin / kChunk, ntaps = nchunks * 9;
  struct WFrag { u32x4v hi[WJ], lo[WJ]; };
  auto load_w = [&](int ft, WFrag& f) {
    const int fc = ft < ntaps ? ft : ntaps - 1;        // past the end: re-read the last tap (never used)
    const uint32_t* p = w_lane + fc * tap_stride;
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

  WFrag w0, w1, w2;
  load_halo(0);
  load_w(0, w0);
  load_w(1, w1);
  store_halo(0);
  __syncthreads();

  auto tap_mfma = [&](int buf, int tap, const WFrag& wf) {   // tap compile-time after unrolling
    const int toff = ((tap / 3) * kHaloW + (tap % 3)) * 8;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(&As[buf][0][0][0] + a_off[i] + toff));
      const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(&As[buf][1][0][0] + a_off[i] + toff));
#pragma unroll
      for (int jn = 0; jn < WJ; ++jn) {                // small terms first
        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf.lo[jn]), ah, acc[i][jn], 0, 0, 0);
        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf.hi[jn]), al, acc[i][jn], 0, 0, 0);
        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf.hi[jn]), ah, acc[i][jn], 0, 0, 0);
      }
    }
  };

  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int buf = chunk & 1, ft = chunk * 9;
    if (chunk + 1 < nchunks) load_halo(chunk + 1);
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) {                   // ring of three weight register sets, loads two taps ahead
      load_w(ft + 3 * t3 + 2, w2);
      tap_mfma(buf, 3 * t3, w0);
      load_w(ft + 3 * t3 + 3, w0);
      tap_mfma(buf, 3 * t3 + 1, w1);
      load_w(ft + 3 * t3 + 4, w1);
      tap_mfma(buf, 3 * t3 + 2, w2);
    }
    if (chunk + 1 < nchunks) store_halo(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: accumulator register v of lane l is (channel row 8 (v / 4) + 4 (l / 32) + v % 4, pixel l % 32) ------
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int gy = ty0 + wm * 2 * TI + i * 2 + (r32 >> 4), gx = tx0 + (r32 & 15);
    const bool pix_ok = gy < g.H && gx < g.W;
#pragma unroll
    for (int jn = 0; jn < WJ; ++jn)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int n = n0 + wn * 32 * WJ + jn * 32 + 8 * (v / 4) + 4 * half + (v % 4);
        if (pix_ok && n < g.Cout) {
          float r = acc[i][jn][v] + (bias ? bias[n] : 0.f);
          if (RELU) r = fmaxf(r, 0.f);
          out[((int64_t)b * g.Cout + n) * HW + gy * g.W + gx] = r;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------
// conv3x3_exact<TJ, RELU, WM> (round 4): the EXACT fp32 convolution in the structure of conv3x3_packed.  The generic exact kernel
// at the top of this file (one barrier per 16 k, nine-fold re-read of the input) reaches 38.8 % of the fp32 matrix pipe and loses
// to MIOpen (646 vs 397 us on the 256 -> 256 layer at 100 x 167); the halo structure does not care about the operand width:
//   * the 10 x 18 halo of an 8 x 16 pixel tile is staged ONCE per chunk of 16 input channels, as fp32 [halo pixel][16] (64 bytes
//     per pixel, double buffered: 23 KB) and serves all nine taps with immediate LDS offsets -- one barrier per 288 MFMAs and wave;
//   * v_mfma_f32_32x32x2_f32 takes ONE float per lane and operand: lane (r, h) holds channel 2 s + h of row r in k-step s.  Both
//     operands are therefore stored with the chunk's channels in the order (h, s) -- position 8 h + s = channel 2 s + h -- so that
//     the eight values a lane feeds into the eight k-steps of a tap are 32 contiguous bytes: two ds_read_b128 for the pixels, two
//     16-byte global loads (two taps ahead, ring of three register sets) for the weights, which
//     conv3x3_hip_pack_weight_exact_f32 re-orders once into [chunk][tap][cout][16];
//   * every output element is ONE chain of fp32 fused multiply-adds over (chunk, tap, k-step, h): exact fp32 arithmetic in a fixed
//     order (the f32 MFMA is an exact FMA), so the result is bitwise repeatable and within fp32 round-off of any other order.
// TH: rows of the pixel tile (8 or 4; 16 columns): smaller tiles = more, shorter work units -- at these sizes the kernel runs at
// the matrix pipe's rate and what is left to lose is the balance of units over the 1024 SIMDs (profiles/r04_conv3x3_exact.txt)
// WIDE (round 6): the halo travels as 16-byte loads of FOUR pixels of one channel (raw buffer loads: a row's first / last load reaches
// one pixel past the image and reads zeros or a neighbour that is masked out) instead of one dword per (pixel, channel): a thread
// issues 2 vector-memory instructions per chunk instead of 8, and it is the COUNT that costs -- the timing-only build with two 16-byte
// loads in place of the eight dwords was 5.8 % faster on the 256 -> 256 layer at 100 x 167 (profiles/r06_conv3x3_exact.txt, section 4).
template <int TH, int TJ, bool RELU, int WM, bool WIDE = false>
__global__ void __launch_bounds__(kThreads, 2)
conv3x3_exact(const float* __restrict__ in, const float* __restrict__ packed, const float* __restrict__ bias, Geom g,
              int tiles_x, int tiles_per_image, int cout_pad, float* __restrict__ out) {
  constexpr int kHalo = (TH + 2) * kHaloW;                                   // halo pixels of the tile
  // pixel pitch 80 bytes, not 64: a ds_read_b128 is served 16 lanes at a time = 16 consecutive halo pixels, and 16 x 64 B covers
  // only a quarter of the banks (4-way conflicts: 74 % of the LDS-active cycles, profiles/r04_conv3x3_exact.txt); 5 x 16 B is odd
constexpr int CONV3X3_EXACT_PITCH = 20;
  constexpr int kPitch = CONV3X3_EXACT_PITCH;                               // floats per halo pixel in the LDS
  __shared__ __attribute__((aligned(16))) float As[2][kHalo][kPitch];       // [buffer][halo pixel][(h, s)]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.x / tiles_per_image, t_in = blockIdx.x - b * tiles_per_image;
  const int ty0 = (t_in / tiles_x) * TH, tx0 = (t_in % tiles_x) * kTW;
  const int n0 = blockIdx.y * (64 * TJ);
  const int HW = g.H * g.W;

  // ---- halo staging ---------------------------------------------------------------------------------------------------------
  // narrow form: items (halo pixel, h); this thread owns items tid and tid + 256: eight channels 2 s + h each
  const float* h_ptr[2];
  bool h_ok[2], h_has[2];
  int h_px[2], h_half[2];
  float h_reg[2][8];
  // wide form: items (row of the halo, group of four columns, channel), channel fastest -- the 64 lanes of a wave write 16 channel
  // positions x 4 column groups = 64 different LDS banks per ds_write_b32
  constexpr int kGroups = (kHaloW + 3) / 4;                                  // 5: columns 0-3, .., 12-15, 16-17
  constexpr int kWideItems = kChunk * (TH + 2) * kGroups, kWidePer = (kWideItems + kThreads - 1) / kThreads;
  uint32_t w_off[kWidePer];                                                  // byte offset of the item's first pixel in chunk 0 (out of range: a row outside the image)
  int w_dst[kWidePer];                                                       // float index of the item's first pixel in a halo buffer
  unsigned w_ok[kWidePer];                                                   // bits 0..3: the column is inside the image; bits 4..7: the column exists in the halo
  f32x4 w_reg[kWidePer];
  __amdgpu_buffer_rsrc_t in_rsrc;
  if constexpr (WIDE) {
    in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)((uint32_t)g.B * (uint32_t)g.Cin * (uint32_t)HW * 4u), 0x00020000);
#pragma unroll
    for (int i = 0; i < kWidePer; ++i) {
      const int item = tid + i * kThreads;
      const bool has = item < kWideItems;
      const int c = item & (kChunk - 1), rg = item / kChunk, gq = rg % kGroups, r = rg / kGroups;
      const int gy = ty0 - 1 + r, gx0 = tx0 - 1 + 4 * gq;
      const bool row_ok = has && (unsigned)gy < (unsigned)g.H;
      unsigned ok = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (has && 4 * gq + j < kHaloW) ok |= 16u << j;
        if (row_ok && 4 * gq + j < kHaloW && (unsigned)(gx0 + j) < (unsigned)g.W) ok |= 1u << j;
      }
      // a tile at the image's left edge: the first group starts at column -1 -- one float in front of the row, in front of the TENSOR in
      // the first row of all (an offset that wraps: the whole load would read zeros).  Such an item loads columns 0..3 and stores them one
      // halo column to the right (bit 8; halo column 0 is zero, the row's column 3 is the next group's first)
      const bool shl = gx0 < 0;
      if (shl) ok |= 256u;
      w_ok[i] = ok;
      w_off[i] = row_ok ? ((uint32_t)((b * g.Cin + c) * g.H + gy) * (uint32_t)g.W + (uint32_t)(shl ? 0 : gx0)) * 4u : msda::kOobOffset;
      w_dst[i] = (r * kHaloW + 4 * gq) * kPitch + 8 * (c & 1) + (c >> 1);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int item = tid + r * kThreads;
      h_has[r] = item < 2 * kHalo;
      const int hp = item % kHalo;
      h_px[r] = hp;
      h_half[r] = (item / kHalo) & 1;
      const int gy = ty0 - 1 + hp / kHaloW, gx = tx0 - 1 + hp % kHaloW;
      h_ok[r] = h_has[r] && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W;
      h_ptr[r] = in + ((int64_t)b * g.Cin + h_half[r]) * HW + (h_ok[r] ? gy * g.W + gx : 0);
    }
  }
  auto load_halo = [&](int chunk) {
    if constexpr (WIDE) {
      const uint32_t co = (uint32_t)chunk * (uint32_t)(kChunk * 4) * (uint32_t)HW;
#pragma unroll
      for (int i = 0; i < kWidePer; ++i)
        w_reg[i] = msda::buffer_load_f32x4(in_rsrc, w_off[i] == msda::kOobOffset ? msda::kOobOffset : w_off[i] + co, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) h_reg[r][e] = h_ok[r] ? h_ptr[r][(int64_t)(chunk * kChunk + 2 * e) * HW] : 0.f;
    }
  };
  auto store_halo = [&](int buf) {
    if constexpr (WIDE) {
#pragma unroll
      for (int i = 0; i < kWidePer; ++i) {
        float* dst = &As[buf][0][0] + w_dst[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = (w_ok[i] & 256u) ? (j ? w_reg[i][j ? j - 1 : 0] : 0.f) : w_reg[i][j];
          if (w_ok[i] & (16u << j)) dst[j * kPitch] = (w_ok[i] & (1u << j)) ? v : 0.f;
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 2; ++r)
        if (h_has[r]) {
          float* dst = &As[buf][h_px[r]][h_half[r] * 8];
          *reinterpret_cast<f32x4*>(dst) = f32x4{h_reg[r][0], h_reg[r][1], h_reg[r][2], h_reg[r][3]};
          *reinterpret_cast<f32x4*>(dst + 4) = f32x4{h_reg[r][4], h_reg[r][5], h_reg[r][6], h_reg[r][7]};
        }
    }
  };

  // ---- MFMA fragments (as conv3x3_packed: weights are the first operand, so accumulator rows are channels) ----------------
  constexpr int WN = 4 / WM, TI = (TH / 2) / WM, WJ = 2 * TJ / WN;   // per wave: TI 32-pixel sub-tiles x WJ 32-channel tiles
  static_assert(WJ >= 1 && TI >= 1, "wave layout");
  const int wm = wv / WN, wn = wv % WN;
  const int r32 = lane & 31, half = lane >> 5;
  int a_off[TI];                                         // float offset of this lane's eight pixel values, sub-tile i, tap (0, 0)
#pragma unroll
  for (int i = 0; i < TI; ++i) a_off[i] = ((wm * 2 * TI + i * 2 + (r32 >> 4)) * kHaloW + (r32 & 15)) * kPitch + half * 8;
  const int nb = n0 + wn * 32 * WJ + r32;
  const float* w_lane = packed + (int64_t)nb * kChunk + half * 8;
  const int64_t tap_stride = (int64_t)cout_pad * kChunk;
  const int nchunks = g.Cin / kChunk, ntaps = nchunks * 9;
  struct WFrag { f32x4 a[WJ], b[WJ]; };
  auto load_w = [&](int ft, WFrag& f) {
    const int fc = ft < ntaps ? ft : ntaps - 1;        // past the end: re-read the last tap (never used)
    const float* p = w_lane + fc * tap_stride;
#pragma unroll
    for (int jn = 0; jn < WJ; ++jn) {
      f.a[jn] = *reinterpret_cast<const f32x4*>(p + jn * 32 * kChunk);
      f.b[jn] = *reinterpret_cast<const f32x4*>(p + jn * 32 * kChunk + 4);
    }
  };

  f32x16 acc[TI][WJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int jn = 0; jn < WJ; ++jn)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][jn][v] = 0.f;

  // Everything an MFMA waits for is requested a long way ahead: the NINE taps' weights of chunk c + 1 while chunk c is multiplied
  // (72 registers; two taps ahead -- ~2000 clocks -- was inside the L2's latency under load), the halo of chunk c + 1 likewise, and
  // the pixel fragments of tap t + 1 before the MFMAs of tap t (CONV3X3_EXACT_AHEAD=0: the round-4 first version, for A/B).
  struct XFrag { f32x4 a[TI], b[TI]; };
  auto load_x = [&](int buf, int tap, XFrag& x) {        // tap compile-time after unrolling
    const int toff = ((tap / 3) * kHaloW + (tap % 3)) * kPitch;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      x.a[i] = *reinterpret_cast<const f32x4*>(&As[buf][0][0] + a_off[i] + toff);
      x.b[i] = *reinterpret_cast<const f32x4*>(&As[buf][0][0] + a_off[i] + toff + 4);
    }
  };
  // k-step outermost: consecutive MFMAs go to DIFFERENT accumulators (a chain of MFMAs on one accumulator issues at its
  // dependent latency and loses every slot another instruction takes in between)
  auto tap_mfma = [&](const XFrag& x, const WFrag& wf) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int jn = 0; jn < WJ; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(s < 4 ? wf.a[jn][s & 3] : wf.b[jn][s & 3],
                                                            s < 4 ? x.a[i][s & 3] : x.b[i][s & 3], acc[i][jn], 0, 0, 0);
  };
  WFrag wc[9], wnx[9];
  load_halo(0);
#pragma unroll
  for (int t = 0; t < 9; ++t) load_w(t, wc[t]);
  store_halo(0);
  __syncthreads();
  XFrag x0, x1;
  load_x(0, 0, x0);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int buf = chunk & 1, ft = (chunk + 1) * 9;
    if (chunk + 1 < nchunks) load_halo(chunk + 1);
#pragma unroll
    for (int t = 0; t < 9; ++t) load_w(ft + t, wnx[t]);  // (past the end: re-reads the last tap, never used)
#pragma unroll
    for (int t = 0; t < 9; t += 2) {
      if (t + 1 < 9) load_x(buf, t + 1, x1);
      tap_mfma(x0, wc[t]);
      if (t + 1 < 9) {
        if (t + 2 < 9) load_x(buf, t + 2, x0);
        tap_mfma(x1, wc[t + 1]);
      }
      // Round 6: the next chunk's halo goes into the other buffer in the MIDDLE of the chunk, not in front of its barrier (that
      // buffer's last readers left before the previous barrier; the loads were issued at the chunk's start): the barrier no longer
      // waits for the slowest wave's global loads and LDS writes -- 418 -> 376 us on the 256 -> 256 layer at 100 x 167, 803 -> 743 us
      // for the mask head's module forward (profiles/r06_conv3x3_exact.txt; after tap 2 or tap 6 instead: slower)
      if (t == 4 && chunk + 1 < nchunks) store_halo(buf ^ 1);
    }
    __syncthreads();
    load_x(buf ^ 1, 0, x0);                            // (after the last chunk: stale data, never used)
#pragma unroll
    for (int t = 0; t < 9; ++t) wc[t] = wnx[t];
  }

  // ---- epilogue: accumulator register v of lane l is (channel row 8 (v / 4) + 4 (l / 32) + v % 4, pixel l % 32) ------
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int gy = ty0 + wm * 2 * TI + i * 2 + (r32 >> 4), gx = tx0 + (r32 & 15);
    const bool pix_ok = gy < g.H && gx < g.W;
#pragma unroll
    for (int jn = 0; jn < WJ; ++jn)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int n = n0 + wn * 32 * WJ + jn * 32 + 8 * (v / 4) + 4 * half + (v % 4);
        if (pix_ok && n < g.Cout) {
          float r = acc[i][jn][v] + (bias ? bias[n] : 0.f);
          if (RELU) r = fmaxf(r, 0.f);
          out[((int64_t)b * g.Cout + n) * HW + gy * g.W + gx] = r;
        }
      }
  }
}

// weight [cout, cin, 3, 3] fp32 -> packed [cin / 16][9 taps][cout_pad][16] fp32, position 8 h + s of a chunk = channel 2 s + h
__global__ void pack_weight_exact_kernel(const float* __restrict__ w, int cout, int cin, int cout_pad, float* __restrict__ packed) {
  const int64_t total = (int64_t)(cin / kChunk) * 9 * cout_pad * kChunk;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int pos = (int)(idx % kChunk);
    const int n = (int)((idx / kChunk) % cout_pad);
    const int tap = (int)((idx / kChunk / cout_pad) % 9);
    const int chunk = (int)(idx / kChunk / cout_pad / 9);
    const int cl = 2 * (pos & 7) + (pos >> 3);
    packed[idx] = n < cout ? w[((int64_t)n * cin + chunk * kChunk + cl) * 9 + tap] : 0.f;
  }
}

// weight [cout, cin, 3, 3] fp32 -> packed [cin / 16][9 taps][hi, lo][cout_pad][16 channels] bf16
__global__ void pack_weight_kernel(const float* __restrict__ w, int cout, int cin, int cout_pad, uint16_t* __restrict__ packed) {
  const int64_t total = (int64_t)(cin / kChunk) * 9 * cout_pad * kChunk;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int cl = (int)(idx % kChunk);
    const int n = (int)((idx / kChunk) % cout_pad);
    const int tap = (int)((idx / kChunk / cout_pad) % 9);
    const int chunk = (int)(idx / kChunk / cout_pad / 9);
    const float v = n < cout ? w[((int64_t)n * cin + chunk * kChunk + cl) * 9 + tap] : 0.f;
    const uint32_t bits = __float_as_uint(v), hb = bits & 0xffff0000u;
    const uint32_t lb = __float_as_uint(v - __uint_as_float(hb));
    const int64_t o = (((int64_t)(chunk * 9 + tap) * 2) * cout_pad + n) * kChunk + cl;
    packed[o] = (uint16_t)(hb >> 16);
    packed[o + (int64_t)cout_pad * kChunk] = (uint16_t)((lb + 0x8000u) >> 16);   // lo rounded to nearest
  }
}

static inline int cout_padded(int cout) { return (cout + 127) / 128 * 128; }

template <int BM, int BN>
static int launch_tile(const float* in, const float* w, const float* bias, const Geom& g, int relu, float* out,
                       hipStream_t stream) {
  dim3 grid((unsigned)((g.Mtot + BM - 1) / BM), (unsigned)((g.Cout + BN - 1) / BN));
  if (relu) hipLaunchKernelGGL((conv3x3_gemm<BM, BN, true>), grid, dim3(kThreads), 0, stream, in, w, bias, g, out);
  else hipLaunchKernelGGL((conv3x3_gemm<BM, BN, false>), grid, dim3(kThreads), 0, stream, in, w, bias, g, out);
  return (int)hipGetLastError();
}

// out = skip + nearest(low): one thread per 4 consecutive x of a row (16-byte loads / stores where the row allows it)
__global__ void __launch_bounds__(256)
upsample_add_kernel(const float* __restrict__ skip, const float* __restrict__ low, int planes, int H, int W, int h, int w,
                    float* __restrict__ out) {
  const int wq = (W + 3) / 4;
  const long long total = (long long)planes * H * wq;
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int xq = (int)(idx % wq);
    const long long row = idx / wq;               // plane * H + y
    const int y = (int)(row % H);
    const long long plane = row / H;
    const int ys = min((int)floorf((float)y * sy), h - 1);
    const float* lrow = low + (plane * h + ys) * w;
    const long long base = row * W + 4 * xq;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int x = 4 * xq + e;
      if (x < W) out[base + e] = skip[base + e] + lrow[min((int)floorf((float)x * sx), w - 1)];
    }
  }
}

}  // namespace conv3x3

extern "C" {

int dynmask_set_error(int code, const char* what);   // msda_capi.hip (shared last-error slot)

int conv3x3_hip_f32(const float* in, const float* weight, const float* bias, int batch, int cin, int height, int width,
                    int cout, int relu, int precision, float* out, void* stream) {
  if (precision != 0 && precision != 1) return dynmask_set_error(CONV3X3_ERR_BAD_DIMS, "conv3x3: precision must be 0 or 1");
  if (batch < 0 || cin <= 0 || height <= 0 || width <= 0 || cout <= 0)
    return dynmask_set_error(CONV3X3_ERR_BAD_DIMS, "conv3x3: bad dimensions");
  const long long K = 9ll * cin;
  if (K % conv3x3::BK != 0)
    return dynmask_set_error(CONV3X3_ERR_UNSUPPORTED, "conv3x3: 9 * cin must be a multiple of 16");
  const long long M = (long long)batch * height * width;
  if (M == 0) return 0;
  if (M >= (1ll << 31) || K >= (1ll << 31) || (long long)(cout + 63) / 64 > 65535)
    return dynmask_set_error(CONV3X3_ERR_BAD_DIMS, "conv3x3: problem too large");
  if (!in || !weight || !out) return dynmask_set_error(CONV3X3_ERR_NULL_POINTER, "conv3x3: null pointer argument");
  conv3x3::Geom g;
  g.B = batch; g.Cin = cin; g.H = height; g.W = width; g.Cout = cout; g.Mtot = (int)M; g.K = (int)K;
  if (precision == 1) {
    const int rc = conv3x3::launch_bf16x3(in, weight, bias, g, relu, out, (hipStream_t)stream);
    return rc == 0 ? 0 : dynmask_set_error(rc, hipGetErrorString((hipError_t)rc));
  }
  // 128 x 128 tiles unless they would leave CUs without a workgroup or most of a 128-channel tile empty
  static const int forced = msda::ab_env_int("CONV3X3_TILE", 0);
  const long long tiles128 = ((M + 127) / 128) * ((cout + 127) / 128);
  bool big = cout > 64 && tiles128 >= 256;
  if (forced == 1) big = true;
  if (forced == 2) big = false;
  const int rc = big ? conv3x3::launch_tile<128, 128>(in, weight, bias, g, relu, out, (hipStream_t)stream)
                     : conv3x3::launch_tile<64, 64>(in, weight, bias, g, relu, out, (hipStream_t)stream);
  return rc == 0 ? 0 : dynmask_set_error(rc, hipGetErrorString((hipError_t)rc));
}


size_t conv3x3_hip_packed_weight_bytes(int cout, int cin) {
  if (cout <= 0 || cin <= 0 || cin % conv3x3::kChunk != 0) return 0;
  return (size_t)(cin / conv3x3::kChunk) * 9 * 2 * conv3x3::cout_padded(cout) * conv3x3::kChunk * sizeof(uint16_t);
}

int conv3x3_hip_pack_weight_f32(const float* weight, int cout, int cin, void* packed, void* stream) {
  if (cout <= 0 || cin <= 0) return dynmask_set_error(CONV3X3_ERR_BAD_DIMS, "conv3x3: bad dimensions");
  if (cin % conv3x3::kChunk != 0)
    return dynmask_set_error(CONV3X3_ERR_UNSUPPORTED, "conv3x3: packed weights need cin to be a multiple of 16");
  if (!weight || !packed) return dynmask_set_error(CONV3X3_ERR_NULL_POINTER, "conv3x3: null pointer argument");
  hipLaunchKernelGGL(conv3x3::pack_weight_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, weight, cout, cin,
                     conv3x3::cout_padded(cout), static_cast<uint16_t*>(packed));
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : dynmask_set_error((int)e, hipGetErrorString(e));
}

int conv3x3_hip_packed_f32(const float* in, const void* packed, const float* bias, int batch, int cin, int height,
                           int width, int cout, int relu, float* out, void* stream) {
  if (batch < 0 || cin <= 0 || height <= 0 || width <= 0 || cout <= 0)
    return dynmask_set_error(CONV3X3_ERR_BAD_DIMS, "conv3x3: bad dimensions");
  if (cin % conv3x3::kChunk != 0)
    return dynmask_set_error(CONV3X3_ERR_UNSUPPORTED, "conv3x3: packed weights need cin to be a multiple of 16");
  const long long M = (long long)batch * height * width;
  if (M == 0) return 0;
  const int tiles_x = (width + conv3x3::kTW - 1) / conv3x3::kTW, tiles_y = (height + conv3x3::kTH - 1) / conv3x3::kTH;
  const long long tiles = (long long)batch * tiles_x * tiles_y;
  if (M >= (1ll << 31) || tiles >= (1ll << 31) || (long long)cin * height * width >= (1ll << 31))
    return dynmask_set_error(CONV3X3_ERR_BAD_DIMS, "conv3x3: problem too large");
  if (!in || !packed || !out) return dynmask_set_error(CONV3X3_ERR_NULL_POINTER, "conv3x3: null pointer argument");
  conv3x3::Geom g;
  g.B = batch; g.Cin = cin; g.H = height; g.W = width; g.Cout = cout; g.Mtot = (int)M; g.K = 9 * cin;
  const int cout_pad = conv3x3::cout_padded(cout);
  const uint32_t* pk = static_cast<const uint32_t*>(packed);
  hipStream_t st = (hipStream_t)stream;
  // 128 output channels per workgroup unless that leaves CUs idle (small feature maps): then 64
  static const int forced_tj = msda::ab_env_int("CONV3X3_TJ", 0);
  bool wide = cout > 64 && tiles * ((cout + 127) / 128) >= 512;
  if (forced_tj == 1) wide = false;
  if (forced_tj == 2) wide = cout > 64;
  if (wide) {
    dim3 grid((unsigned)tiles, (unsigned)((cout + 127) / 128));
    if (relu) hipLaunchKernelGGL((conv3x3::conv3x3_packed<2, true, 1>), grid, dim3(conv3x3::kThreads), 0, st, in, pk, bias, g, tiles_x, tiles_x * tiles_y, cout_pad, out);
    else hipLaunchKernelGGL((conv3x3::conv3x3_packed<2, false, 1>), grid, dim3(conv3x3::kThreads), 0, st, in, pk, bias, g, tiles_x, tiles_x * tiles_y, cout_pad, out);
  } else {
    dim3 grid((unsigned)tiles, (unsigned)((cout + 63) / 64));
    if (relu) hipLaunchKernelGGL((conv3x3::conv3x3_packed<1, true, 2>), grid, dim3(conv3x3::kThreads), 0, st, in, pk, bias, g, tiles_x, tiles_x * tiles_y, cout_pad, out);
    else hipLaunchKernelGGL((conv3x3::conv3x3_packed<1, false, 2>), grid, dim3(conv3x3::kThreads), 0, st, in, pk, bias, g, tiles_x, tiles_x * tiles_y, cout_pad, out);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : dynmask_set_error((int)e, hipGetErrorString(e));
}


size_t conv3x3_hip_packed_exact_weight_bytes(int cout, int cin) {
  if (cout <= 0 || cin <= 0 || cin % conv3x3::kChunk != 0) return 0;
  return (size_t)(cin / conv3x3::kChunk) * 9 * conv3x3::cout_padded(cout) * conv3x3::kChunk * sizeof(float);
}

int conv3x3_hip_pack_weight_exact_f32(const float* weight, int cout, int cin, void* packed, void* stream) {
  if (cout <= 0 || cin <= 0) return dynmask_set_error(CONV3X3_ERR_BAD_DIMS, "conv3x3: bad dimensions");
  if (cin % conv3x3::kChunk != 0)
    return dynmask_set_error(CONV3X3_ERR_UNSUPPORTED, "conv3x3: packed weights need cin to be a multiple of 16");
  if (!weight || !packed) return dynmask_set_error(CONV3X3_ERR_NULL_POINTER, "conv3x3: null pointer argument");
  hipLaunchKernelGGL(conv3x3::pack_weight_exact_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, weight, cout, cin,
                     conv3x3::cout_padded(cout), static_cast<float*>(packed));
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : dynmask_set_error((int)e, hipGetErrorString(e));
}

int conv3x3_hip_packed_exact_f32(const float* in, const void* packed, const float* bias, int batch, int cin, int height,
                                 int width, int cout, int relu, float* out, void* stream) {
  if (batch < 0 || cin <= 0 || height <= 0 || width <= 0 || cout <= 0)
    return dynmask_set_error(CONV3X3_ERR_BAD_DIMS, "conv3x3: bad dimensions");
  if (cin % conv3x3::kChunk != 0)
    return dynmask_set_error(CONV3X3_ERR_UNSUPPORTED, "conv3x3: packed weights need cin to be a multiple of 16");
  const long long M = (long long)batch * height * width;
  if (M == 0) return 0;
  const int tiles_x = (width + conv3x3::kTW - 1) / conv3x3::kTW, tiles_y = (height + conv3x3::kTH - 1) / conv3x3::kTH;
  const long long tiles = (long long)batch * tiles_x * tiles_y;
  if (M >= (1ll << 31) || tiles >= (1ll << 31) || (long long)cin * height * width >= (1ll << 31))
    return dynmask_set_error(CONV3X3_ERR_BAD_DIMS, "conv3x3: problem too large");
  if (!in || !packed || !out) return dynmask_set_error(CONV3X3_ERR_NULL_POINTER, "conv3x3: null pointer argument");
  conv3x3::Geom g;
  g.B = batch; g.Cin = cin; g.H = height; g.W = width; g.Cout = cout; g.Mtot = (int)M; g.K = 9 * cin;
  const int cout_pad = conv3x3::cout_padded(cout);
  const float* pk = static_cast<const float*>(packed);
  hipStream_t st = (hipStream_t)stream;
  // Work units: the kernel runs at the matrix pipe's rate, so what decides the time is how evenly (tile, channel group) units
  // spread over the SIMDs.  Largest unit (8 x 16 pixels x 128 channels: best weight-fragment reuse) when there are >= 4 per
  // workgroup slot of the chip, then 8 x 16 x 64, else 4 x 16 x 64 -- CONV3X3_EXACT_UNIT=1|2|3 pins one (A/B).
  static const int forced = msda::ab_env_int("CONV3X3_EXACT_UNIT", 0);
  int unit = 3;
  if (cout > 64 && tiles * ((cout + 127) / 128) >= 2048) unit = 1;
  else if (tiles * ((cout + 63) / 64) >= 2048) unit = 2;
  if (forced >= 1 && forced <= 3) unit = forced;
  if (unit == 1 && cout <= 64) unit = 2;
#define CONV3X3_EXACT_LAUNCH(TH, TJ, WM, WIDE, GX, GY, TX, TPI)                                                                       \
  do {                                                                                                                                  \
    dim3 grid((unsigned)(GX), (unsigned)(GY));                                                                                          \
    if (relu) hipLaunchKernelGGL((conv3x3::conv3x3_exact<TH, TJ, true, WM, WIDE>), grid, dim3(conv3x3::kThreads), 0, st, in, pk, bias, g, TX, TPI, cout_pad, out); \
    else hipLaunchKernelGGL((conv3x3::conv3x3_exact<TH, TJ, false, WM, WIDE>), grid, dim3(conv3x3::kThreads), 0, st, in, pk, bias, g, TX, TPI, cout_pad, out);     \
  } while (0)
  // the halo as 16-byte raw buffer loads (32-bit byte offsets: inputs below 2 GiB); CONV3X3_EXACT_WIDE=0: one dword per load (A/B)
  static const int wide_env = msda::ab_env_int("CONV3X3_EXACT_WIDE", 1);
  const bool wide = wide_env != 0 && (long long)batch * cin * height * width * 4 < (1ll << 31);
  if (unit == 1) {
    if (wide) CONV3X3_EXACT_LAUNCH(8, 2, 1, true, tiles, (cout + 127) / 128, tiles_x, tiles_x * tiles_y);
    else CONV3X3_EXACT_LAUNCH(8, 2, 1, false, tiles, (cout + 127) / 128, tiles_x, tiles_x * tiles_y);
  } else if (unit == 2) {
    if (wide) CONV3X3_EXACT_LAUNCH(8, 1, 2, true, tiles, (cout + 63) / 64, tiles_x, tiles_x * tiles_y);
    else CONV3X3_EXACT_LAUNCH(8, 1, 2, false, tiles, (cout + 63) / 64, tiles_x, tiles_x * tiles_y);
  } else {
    const int tiles_y4 = (height + 3) / 4;
    const long long tiles4 = (long long)batch * tiles_x * tiles_y4;
    if (tiles4 >= (1ll << 31)) return dynmask_set_error(CONV3X3_ERR_BAD_DIMS, "conv3x3: problem too large");
    if (wide) CONV3X3_EXACT_LAUNCH(4, 1, 2, true, tiles4, (cout + 63) / 64, tiles_x, tiles_x * tiles_y4);
    else CONV3X3_EXACT_LAUNCH(4, 1, 2, false, tiles4, (cout + 63) / 64, tiles_x, tiles_x * tiles_y4);
  }
#undef CONV3X3_EXACT_LAUNCH
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : dynmask_set_error((int)e, hipGetErrorString(e));
}


int upsample_add_hip_f32(const float* skip, const float* low, int batch, int channels, int height, int width, int low_h,
                         int low_w, float* out, void* stream) {
  if (batch < 0 || channels <= 0 || height <= 0 || width <= 0 || low_h <= 0 || low_w <= 0)
    return dynmask_set_error(CONV3X3_ERR_BAD_DIMS, "upsample_add: bad dimensions");
  const long long planes = (long long)batch * channels;
  if (planes == 0) return 0;
  if (planes * height >= (1ll << 31)) return dynmask_set_error(CONV3X3_ERR_BAD_DIMS, "upsample_add: problem too large");
  if (!skip || !low || !out) return dynmask_set_error(CONV3X3_ERR_NULL_POINTER, "upsample_add: null pointer argument");
  const long long total = planes * height * ((width + 3) / 4);
  long long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(conv3x3::upsample_add_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, skip, low,
                     (int)planes, height, width, low_h, low_w, out);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : dynmask_set_error((int)e, hipGetErrorString(e));
}

}  // extern "C"
