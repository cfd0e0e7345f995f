// Patch-embedding convolutions (kernel == stride, no padding) as an fp32 implicit GEMM on the gfx950 matrix cores --
// see include/patch_embed_hip.h.
//
//   C[m, n] = sum_k A[m, k] * B[n, k]      m = (b, py, px) patch,  n = output channel,  k = (c, ky, kx)
//   A[m, k] = x[b, c, py*KS + ky, px*KS + kx]   -- read in place: for a fixed (c, ky) the pixels of consecutive
//                                                  patches of an image row are contiguous, so the tile loads are
//                                                  coalesced without any im2col copy
//   B[n, k] = weight[n, c, ky, kx]              -- the Conv2d weight as stored (k contiguous)
//
// Workgroup: 256 threads = 4 waves (2 x 2), tile 128 (m) x 128 (n) x 16 (k); a wave owns 64 x 64 = 2 x 2 MFMA tiles
// of 32 x 32 (64 accumulator VGPRs).  Both operand tiles go through LDS as [row][16 k] with an 80-byte row pitch and
// are double buffered: the global loads of tile t + 1 are in flight while tile t is multiplied, one barrier per tile.
// v_mfma_f32_32x32x2_f32 takes ONE float of A and B per lane (row = lane % 32, k = lane / 32); the k order inside a
// tile is free as long as A and B agree, so each half-wave fetches 4 consecutive k with a single ds_read_b128
// (lanes 0-31: k 0..3, lanes 32-63: k 4..7 of an 8-deep step) and feeds them to 4 MFMAs.  Per 16-deep tile a wave
// issues 8 ds_read_b128 and 32 MFMAs of 64 cycles each: the matrix pipe is the bound (157 TFLOP/s fp32 dense).
// channels_last == 0 swaps the operands of the MFMA, which transposes the accumulator tile so that lanes run along
// the patches and the NCHW store is coalesced too.
#include "../../include/patch_embed_hip.h"

#include <cstdlib>

#include "msda_common.hpp"

namespace patch_embed {

constexpr int kThreads = 256;
constexpr int BK = 16;
constexpr int kPitch = 20;   // floats per LDS row: 16 + 4 pad (rows stay 16-byte aligned)

using msda::f32x4;
typedef float f32x16 __attribute__((__vector_size__(64)));

struct Geom {
  int B, C, H, W, E, Hp, Wp, Mtot, K;
};

template <int KS, bool NHWC, int BM, int BN>
__global__ void __launch_bounds__(kThreads, 2)
patch_embed_gemm(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, Geom g,
                 float* __restrict__ out) {
  constexpr int VW = KS >= 4 ? 4 : 2;                  // floats per A load (kx run)
  constexpr int kALoads = BM * BK / VW / kThreads;     // 128 rows: 2 (float4) or 4 (float2)
  constexpr int kBLoads = BN * BK / 4 / kThreads;
  constexpr int TI = BM / 64, TJ = BN / 64;            // 32 x 32 MFMA tiles per wave (waves are 2 x 2)
  constexpr int kQ = BK / VW;                          // k groups per row
  __shared__ __attribute__((aligned(16))) float As[2][BM][kPitch];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN][kPitch];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int HpWp = g.Hp * g.Wp;

  // ---- this thread's slice of the operand tiles -------------------------------------------------------------
  int64_t a_base[kALoads];   // offset of x[b, 0, py*KS, px*KS]
  int a_row[kALoads], a_kq[kALoads];
#pragma unroll
  for (int i = 0; i < kALoads; ++i) {
    const int idx = tid + i * kThreads;
    a_row[i] = idx / kQ;
    a_kq[i] = idx % kQ;
    int m = m0 + a_row[i];
    m = m < g.Mtot ? m : g.Mtot - 1;                   // rows past the end re-read the last patch; never stored
    const int b = m / HpWp, sp = m - b * HpWp;
    const int py = sp / g.Wp, px = sp - py * g.Wp;
    a_base[i] = ((int64_t)b * g.C * g.H + (int64_t)py * KS) * g.W + (int64_t)px * KS;
  }
  const float* b_ptr[kBLoads];
  int b_row[kBLoads], b_kq[kBLoads];
#pragma unroll
  for (int i = 0; i < kBLoads; ++i) {
    const int idx = tid + i * kThreads;
    b_row[i] = idx / 4;
    b_kq[i] = idx % 4;
    int n = n0 + b_row[i];
    n = n < g.E ? n : g.E - 1;
    b_ptr[i] = w + (int64_t)n * g.K + b_kq[i] * 4;
  }

  f32x4 a_reg[kALoads];
  float a_reg2[kALoads][2];
  f32x4 b_reg[kBLoads];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < kALoads; ++i) {
      const int kk = kt * BK + a_kq[i] * VW;           // = (c * KS + ky) * KS + kx, kx a multiple of VW
      const int c = kk / (KS * KS), r = kk % (KS * KS);
      const int ky = r / KS, kx = r % KS;
      const float* p = x + a_base[i] + ((int64_t)c * g.H + ky) * g.W + kx;
      if constexpr (VW == 4) {
        a_reg[i] = f32x4{p[0], p[1], p[2], p[3]};      // 4-byte aligned only (odd image widths): dword loads merge
      } else {
        a_reg2[i][0] = p[0];
        a_reg2[i][1] = p[1];
      }
    }
#pragma unroll
    for (int i = 0; i < kBLoads; ++i) b_reg[i] = *reinterpret_cast<const f32x4*>(b_ptr[i] + kt * BK);
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < kALoads; ++i) {
      if constexpr (VW == 4) {
        *reinterpret_cast<f32x4*>(&As[buf][a_row[i]][a_kq[i] * 4]) = a_reg[i];
      } else {
        As[buf][a_row[i]][a_kq[i] * 2] = a_reg2[i][0];
        As[buf][a_row[i]][a_kq[i] * 2 + 1] = a_reg2[i][1];
      }
    }
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
    for (int ss = 0; ss < 2; ++ss) {
      f32x4 af[TI], bf[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i) af[i] = *reinterpret_cast<const f32x4*>(&As[buf][wm + i * 32 + r32][ss * 8 + half * 4]);
#pragma unroll
      for (int jn = 0; jn < TJ; ++jn) bf[jn] = *reinterpret_cast<const f32x4*>(&Bs[buf][wn + jn * 32 + r32][ss * 8 + half * 4]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int jn = 0; jn < TJ; ++jn) {
            if constexpr (NHWC) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[jn][t], acc[i][jn], 0, 0, 0);
            else acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[jn][t], af[i][t], acc[i][jn], 0, 0, 0);
          }
    }
    if (kt + 1 < KT) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: accumulator register v of lane l is element (row 8 (v / 4) + 4 (l / 32) + v % 4, column l % 32) ----
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int jn = 0; jn < TJ; ++jn) {
      if constexpr (NHWC) {          // rows = patches, columns = channels: a lane row writes 32 consecutive channels
        const int n = n0 + wn + jn * 32 + r32;
        const float bv = (bias && n < g.E) ? bias[n] : 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int m = m0 + wm + i * 32 + 8 * (v / 4) + 4 * half + (v % 4);
          if (m < g.Mtot && n < g.E) out[(int64_t)m * g.E + n] = acc[i][jn][v] + bv;
        }
      } else {                       // transposed tile: rows = channels, columns = patches
        const int m = m0 + wm + i * 32 + r32;
        const int b = m / HpWp, sp = m - b * HpWp;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int n = n0 + wn + jn * 32 + 8 * (v / 4) + 4 * half + (v % 4);
          if (m < g.Mtot && n < g.E)
            out[((int64_t)b * g.E + n) * HpWp + sp] = acc[i][jn][v] + (bias ? bias[n] : 0.f);
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// patch_embed_packed<KS, NHWC, TJ>: split-bf16 products (see conv3x3.hip / include/patch_embed_hip.h) from weights that
// were split and re-ordered once into [K / 16 chunks][hi, lo][E padded][16 k] bf16.  A tile = 128 patches x 48 k per
// step (three 16-k chunks), staged through double-buffered LDS as [chunk][patch][16 bf16] hi / lo (one ds_read_b128
// per MFMA operand); a wave's weight fragment of a chunk is 1 KB of contiguous memory loaded straight into registers
// two chunks ahead (ring of three register sets).  36 v_mfma_f32_32x32x16_bf16 per wave and barrier (TJ = 2).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4v __attribute__((__vector_size__(16)));
constexpr int kChunk = 16, kStepChunks = 3, kStepK = kChunk * kStepChunks;   // 48 k per barrier

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

// WM: waves along the patches (1 or 2); with WM = 1 a wave spans the tile's 64 patches, so a weight fragment feeds two
// row tiles (a wave with one row tile re-loads weights faster than the 64 B/clk L1 path delivers them).
template <int KS, bool NHWC, int TJ, int BM, int WM>
__global__ void __launch_bounds__(kThreads, 2)
patch_embed_packed(const float* __restrict__ x, const uint32_t* __restrict__ packed, const float* __restrict__ bias, Geom g,
                   int e_pad, float* __restrict__ out) {
  constexpr int VW = KS >= 4 ? 4 : 2;                 // floats per load (a kx run)
  constexpr int PP = kChunk / VW;                     // pieces per (patch, chunk)
  constexpr int kItems = kStepChunks * BM * PP / kThreads;   // staging items per thread and step
  constexpr int WN = 4 / WM, TI = BM / (32 * WM), WJ = 2 * TJ / WN;   // per wave: TI row tiles x WJ column tiles
  static_assert(TI >= 1 && WJ >= 1, "wave layout");
  // [buffer][hi / lo][chunk][patch (+1 pad row: staggers the banks of the staging stores)][16 bf16]
  __shared__ __attribute__((aligned(16))) uint32_t As[2][2][kStepChunks][BM + 1][8];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * (64 * TJ);
  const int HpWp = g.Hp * g.Wp;

  // staging item i = tid + 256 r: piece i % PP (VW consecutive k), patch (i / PP) % BM, chunk i / (PP BM) -- the
  // lanes of a wave run along the pieces of a patch and then along patches, i.e. along contiguous image memory
  int64_t a_base[kItems];
#pragma unroll
  for (int r = 0; r < kItems; ++r) {
    const int i = tid + r * kThreads;
    int m = m0 + (i / PP) % BM;
    m = m < g.Mtot ? m : g.Mtot - 1;
    const int b = m / HpWp, sp = m - b * HpWp;
    const int py = sp / g.Wp, px = sp - py * g.Wp;
    a_base[r] = ((int64_t)b * g.C * g.H + (int64_t)py * KS) * g.W + (int64_t)px * KS;
  }
  float a_reg[kItems][VW];
  auto load_step = [&](int st) {
#pragma unroll
    for (int r = 0; r < kItems; ++r) {
      const int i = tid + r * kThreads;
      const int kk = st * kStepK + (i / (PP * BM)) * kChunk + (i % PP) * VW;   // (c * KS + ky) * KS + kx
      const int c = kk / (KS * KS), rr = kk % (KS * KS);
      const int ky = rr / KS, kx = rr % KS;
      const float* p = x + a_base[r] + ((int64_t)c * g.H + ky) * g.W + kx;
#pragma unroll
      for (int e = 0; e < VW; ++e) a_reg[r][e] = p[e];
    }
  };
  auto store_step = [&](int buf) {
#pragma unroll
    for (int r = 0; r < kItems; ++r) {
      const int i = tid + r * kThreads;
      const int piece = i % PP, row = (i / PP) % BM, cc = i / (PP * BM);
      uint32_t hi[VW / 2], lo[VW / 2];
#pragma unroll
      for (int p = 0; p < VW / 2; ++p) {
        const float fa = a_reg[r][2 * p], fb = a_reg[r][2 * p + 1];
        const uint32_t ah = __float_as_uint(fa) & 0xffff0000u, bh = __float_as_uint(fb) & 0xffff0000u;
        const uint32_t al = __float_as_uint(fa - __uint_as_float(ah)), bl = __float_as_uint(fb - __uint_as_float(bh));
        hi[p] = (ah >> 16) | bh;
        lo[p] = ((al + 0x8000u) >> 16) | ((bl + 0x8000u) & 0xffff0000u);   // lo rounded to nearest
      }
      if constexpr (VW == 4) {
        *reinterpret_cast<uint2*>(&As[buf][0][cc][row][piece * 2]) = make_uint2(hi[0], hi[1]);
        *reinterpret_cast<uint2*>(&As[buf][1][cc][row][piece * 2]) = make_uint2(lo[0], lo[1]);
      } else {
        As[buf][0][cc][row][piece] = hi[0];
        As[buf][1][cc][row][piece] = lo[0];
      }
    }
  };

  const int wm = (wv / WN) * (BM / WM), wn = (wv % WN) * 32 * WJ;
  const int r32 = lane & 31, half = lane >> 5;
  const int nb = n0 + wn + r32;
  const uint32_t* w_lane = packed + (int64_t)nb * 8 + half * 4;
  const int64_t chunk_stride = (int64_t)2 * e_pad * 8, part_stride = (int64_t)e_pad * 8;
  const int nchunks = g.K / kChunk, nsteps = g.K / kStepK;
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
      for (int jn = 0; jn < WJ; ++jn) {
        const bf16x8 wh = __builtin_bit_cast(bf16x8, wf.hi[jn]), wl = __builtin_bit_cast(bf16x8, wf.lo[jn]);
        if constexpr (NHWC) {
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl, acc[i][jn], 0, 0, 0);
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh, acc[i][jn], 0, 0, 0);
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, acc[i][jn], 0, 0, 0);
        } else {
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ah, acc[i][jn], 0, 0, 0);
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al, acc[i][jn], 0, 0, 0);
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah, acc[i][jn], 0, 0, 0);
        }
      }
    }
  };

  WFrag w0, w1, w2;
  load_step(0);
  load_w(0, w0);
  load_w(1, w1);
  store_step(0);
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    const int buf = st & 1, ch = st * kStepChunks;
    if (st + 1 < nsteps) load_step(st + 1);
    load_w(ch + 2, w2);
    chunk_mfma(buf, 0, w0);
    load_w(ch + 3, w0);
    chunk_mfma(buf, 1, w1);
    load_w(ch + 4, w1);
    chunk_mfma(buf, 2, w2);
    if (st + 1 < nsteps) store_step(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int jn = 0; jn < WJ; ++jn) {
      if constexpr (NHWC) {
        const int n = n0 + wn + jn * 32 + r32;
        const float bv = (bias && n < g.E) ? bias[n] : 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int m = m0 + wm + i * 32 + 8 * (v / 4) + 4 * half + (v % 4);
          if (m < g.Mtot && n < g.E) out[(int64_t)m * g.E + n] = acc[i][jn][v] + bv;
        }
      } else {
        const int m = m0 + wm + i * 32 + r32;
        const int b = m / HpWp, sp = m - b * HpWp;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int n = n0 + wn + jn * 32 + 8 * (v / 4) + 4 * half + (v % 4);
          if (m < g.Mtot && n < g.E)
            out[((int64_t)b * g.E + n) * HpWp + sp] = acc[i][jn][v] + (bias ? bias[n] : 0.f);
        }
      }
    }
}

// weight [E, K] fp32 (K = C * patch^2 contiguous) -> packed [K / 16][hi, lo][e_pad][16] bf16
__global__ void pack_weight_kernel(const float* __restrict__ w, int E, int K, int e_pad, uint16_t* __restrict__ packed) {
  const int64_t total = (int64_t)(K / kChunk) * e_pad * kChunk;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int kl = (int)(idx % kChunk);
    const int n = (int)((idx / kChunk) % e_pad);
    const int chunk = (int)(idx / kChunk / e_pad);
    const float v = n < E ? w[(int64_t)n * K + chunk * kChunk + kl] : 0.f;
    const uint32_t bits = __float_as_uint(v), hb = bits & 0xffff0000u;
    const uint32_t lb = __float_as_uint(v - __uint_as_float(hb));
    const int64_t o = ((int64_t)chunk * 2 * e_pad + n) * kChunk + kl;
    packed[o] = (uint16_t)(hb >> 16);
    packed[o + (int64_t)e_pad * kChunk] = (uint16_t)((lb + 0x8000u) >> 16);   // lo rounded to nearest
  }
}

static inline int e_padded(int E) { return (E + 127) / 128 * 128; }

template <int KS, int BM>
static int launch_packed_bm(const float* x, const uint32_t* packed, const float* bias, const Geom& g, int channels_last,
                            float* out, hipStream_t stream) {
  const int e_pad = e_padded(g.E);
  const long long mt = (g.Mtot + BM - 1) / BM;
  const bool wide = g.E > 64 && mt * ((g.E + 127) / 128) >= 512;   // 128 channels per workgroup unless CUs would idle
  if (wide) {
    dim3 grid((unsigned)mt, (unsigned)((g.E + 127) / 128));
    if (channels_last) hipLaunchKernelGGL((patch_embed_packed<KS, true, 2, BM, (BM == 64 ? 1 : 2)>), grid, dim3(kThreads), 0, stream, x, packed, bias, g, e_pad, out);
    else hipLaunchKernelGGL((patch_embed_packed<KS, false, 2, BM, (BM == 64 ? 1 : 2)>), grid, dim3(kThreads), 0, stream, x, packed, bias, g, e_pad, out);
  } else {
    dim3 grid((unsigned)mt, (unsigned)((g.E + 63) / 64));
    if (channels_last) hipLaunchKernelGGL((patch_embed_packed<KS, true, 1, BM, 2>), grid, dim3(kThreads), 0, stream, x, packed, bias, g, e_pad, out);
    else hipLaunchKernelGGL((patch_embed_packed<KS, false, 1, BM, 2>), grid, dim3(kThreads), 0, stream, x, packed, bias, g, e_pad, out);
  }
  return (int)hipGetLastError();
}

template <int KS>
static int launch_packed(const float* x, const uint32_t* packed, const float* bias, const Geom& g, int channels_last,
                         float* out, hipStream_t stream) {
  static const int forced = msda::ab_env_int("PATCH_EMBED_BM", 0);
  if (forced == 128) return launch_packed_bm<KS, 128>(x, packed, bias, g, channels_last, out, stream);
  return launch_packed_bm<KS, 64>(x, packed, bias, g, channels_last, out, stream);
}

template <int KS, int BM, int BN>
static int launch_tile(const float* x, const float* w, const float* bias, const Geom& g, int channels_last, float* out,
                       hipStream_t stream) {
  dim3 grid((unsigned)((g.Mtot + BM - 1) / BM), (unsigned)((g.E + BN - 1) / BN));
  if (channels_last)
    hipLaunchKernelGGL((patch_embed_gemm<KS, true, BM, BN>), grid, dim3(kThreads), 0, stream, x, w, bias, g, out);
  else
    hipLaunchKernelGGL((patch_embed_gemm<KS, false, BM, BN>), grid, dim3(kThreads), 0, stream, x, w, bias, g, out);
  return (int)hipGetLastError();
}

// Tile choice (tools/patch_embed_bench.py, PATCH_EMBED_TILE=1..3 forces one): 128 x 128 unless the launch would not
// even give every CU one workgroup (then 64 x 128), or K is so short that the kernel is bound by writing the output
// (ConvNeXt stem, K = 48: 64 x 64 tiles, 47 us vs 69 us).
template <int KS>
static int launch(const float* x, const float* w, const float* bias, const Geom& g, int channels_last, float* out,
                  hipStream_t stream) {
  auto tiles = [&](int bm, int bn) { return (long long)((g.Mtot + bm - 1) / bm) * ((g.E + bn - 1) / bn); };
  static const int forced = msda::ab_env_int("PATCH_EMBED_TILE", 0);
  int cfg = g.K <= 64 ? 2 : (tiles(128, 128) >= 256 ? 0 : 1);
  if (forced >= 1 && forced <= 3) cfg = forced - 1;
  if (cfg == 0) return launch_tile<KS, 128, 128>(x, w, bias, g, channels_last, out, stream);
  if (cfg == 1) return launch_tile<KS, 64, 128>(x, w, bias, g, channels_last, out, stream);
  return launch_tile<KS, 64, 64>(x, w, bias, g, channels_last, out, stream);
}

}  // namespace patch_embed

extern "C" {

int dynmask_set_error(int code, const char* what);   // msda_capi.hip (shared last-error slot)

int patch_embed_hip_f32(const float* x, const float* weight, const float* bias, int batch, int in_chans, int height,
                        int width, int embed_dim, int patch, int channels_last, float* out, void* stream) {
  if (batch < 0 || in_chans <= 0 || height <= 0 || width <= 0 || embed_dim <= 0 || patch <= 0)
    return dynmask_set_error(PATCH_EMBED_ERR_BAD_DIMS, "patch_embed: bad dimensions");
  if (patch != 2 && patch != 4 && patch != 8 && patch != 16)
    return dynmask_set_error(PATCH_EMBED_ERR_UNSUPPORTED, "patch_embed: patch size must be 2, 4, 8 or 16");
  const long long K = (long long)in_chans * patch * patch;
  if (K % patch_embed::BK != 0)
    return dynmask_set_error(PATCH_EMBED_ERR_UNSUPPORTED, "patch_embed: in_chans * patch^2 must be a multiple of 16");
  patch_embed::Geom g;
  g.B = batch; g.C = in_chans; g.H = height; g.W = width; g.E = embed_dim;
  g.Hp = height / patch; g.Wp = width / patch;
  const long long M = (long long)batch * g.Hp * g.Wp;
  if (M == 0) return 0;
  if (M >= (1ll << 31) || K >= (1ll << 31) || (long long)(embed_dim + 63) / 64 > 65535)
    return dynmask_set_error(PATCH_EMBED_ERR_BAD_DIMS, "patch_embed: problem too large");
  if (!x || !weight || !out) return dynmask_set_error(PATCH_EMBED_ERR_NULL_POINTER, "patch_embed: null pointer argument");
  g.Mtot = (int)M; g.K = (int)K;
  int rc;
  switch (patch) {
    case 2: rc = patch_embed::launch<2>(x, weight, bias, g, channels_last, out, (hipStream_t)stream); break;
    case 4: rc = patch_embed::launch<4>(x, weight, bias, g, channels_last, out, (hipStream_t)stream); break;
    case 8: rc = patch_embed::launch<8>(x, weight, bias, g, channels_last, out, (hipStream_t)stream); break;
    default: rc = patch_embed::launch<16>(x, weight, bias, g, channels_last, out, (hipStream_t)stream); break;
  }
  return rc == 0 ? 0 : dynmask_set_error(rc, hipGetErrorString((hipError_t)rc));
}


size_t patch_embed_hip_packed_weight_bytes(int embed_dim, int in_chans, int patch) {
  const long long K = (long long)in_chans * patch * patch;
  if (embed_dim <= 0 || in_chans <= 0 || (patch != 2 && patch != 4 && patch != 8 && patch != 16) ||
      K % patch_embed::kStepK != 0)
    return 0;
  return (size_t)(K / patch_embed::kChunk) * 2 * patch_embed::e_padded(embed_dim) * patch_embed::kChunk * sizeof(uint16_t);
}

int patch_embed_hip_pack_weight_f32(const float* weight, int embed_dim, int in_chans, int patch, void* packed, void* stream) {
  if (patch_embed_hip_packed_weight_bytes(embed_dim, in_chans, patch) == 0)
    return dynmask_set_error(PATCH_EMBED_ERR_UNSUPPORTED, "patch_embed: packed weights need patch in {2,4,8,16} and C * patch^2 a multiple of 48");
  if (!weight || !packed) return dynmask_set_error(PATCH_EMBED_ERR_NULL_POINTER, "patch_embed: null pointer argument");
  hipLaunchKernelGGL(patch_embed::pack_weight_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, weight, embed_dim,
                     in_chans * patch * patch, patch_embed::e_padded(embed_dim), static_cast<uint16_t*>(packed));
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : dynmask_set_error((int)e, hipGetErrorString(e));
}

int patch_embed_hip_packed_f32(const float* x, const void* packed, const float* bias, int batch, int in_chans, int height,
                               int width, int embed_dim, int patch, int channels_last, float* out, void* stream) {
  if (batch < 0 || in_chans <= 0 || height <= 0 || width <= 0 || embed_dim <= 0 || patch <= 0)
    return dynmask_set_error(PATCH_EMBED_ERR_BAD_DIMS, "patch_embed: bad dimensions");
  if (patch_embed_hip_packed_weight_bytes(embed_dim, in_chans, patch) == 0)
    return dynmask_set_error(PATCH_EMBED_ERR_UNSUPPORTED, "patch_embed: packed weights need patch in {2,4,8,16} and C * patch^2 a multiple of 48");
  patch_embed::Geom g;
  g.B = batch; g.C = in_chans; g.H = height; g.W = width; g.E = embed_dim;
  g.Hp = height / patch; g.Wp = width / patch;
  const long long M = (long long)batch * g.Hp * g.Wp, K = (long long)in_chans * patch * patch;
  if (M == 0) return 0;
  if (M >= (1ll << 31) || K >= (1ll << 31) || (long long)(embed_dim + 63) / 64 > 65535)
    return dynmask_set_error(PATCH_EMBED_ERR_BAD_DIMS, "patch_embed: problem too large");
  if (!x || !packed || !out) return dynmask_set_error(PATCH_EMBED_ERR_NULL_POINTER, "patch_embed: null pointer argument");
  g.Mtot = (int)M; g.K = (int)K;
  const uint32_t* pk = static_cast<const uint32_t*>(packed);
  int rc;
  switch (patch) {
    case 2: rc = patch_embed::launch_packed<2>(x, pk, bias, g, channels_last, out, (hipStream_t)stream); break;
    case 4: rc = patch_embed::launch_packed<4>(x, pk, bias, g, channels_last, out, (hipStream_t)stream); break;
    case 8: rc = patch_embed::launch_packed<8>(x, pk, bias, g, channels_last, out, (hipStream_t)stream); break;
    default: rc = patch_embed::launch_packed<16>(x, pk, bias, g, channels_last, out, (hipStream_t)stream); break;
  }
  return rc == 0 ? 0 : dynmask_set_error(rc, hipGetErrorString((hipError_t)rc));
}

}  // extern "C"
