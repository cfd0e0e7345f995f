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
  static const int forced = std::getenv("PATCH_EMBED_TILE") ? std::atoi(std::getenv("PATCH_EMBED_TILE")) : 0;
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

}  // extern "C"
