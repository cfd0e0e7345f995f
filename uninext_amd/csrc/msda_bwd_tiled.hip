// msda_bwd_tiled -- MSDeformAttn backward for encoder-style calls (Lq == S) with grad_value privatised in LDS.
// fp32, D = 32, P = 4, L <= 4.  gfx950 only.
//
// Why: the value gradient is a scatter of 728 M float adds per encoder call (R50, N = 2).  With global atomics the
// L2 atomic units cost one operation per (instruction, 128-byte line) pair -- measured 11.4 G line-ops/s -- so even
// the best pattern (one full line per instruction, msda_bwd_generic) takes 22.7 M line-ops = 2.0 ms.  Queries that
// are neighbours in the image hit the same value pixels (21 ... 1300 contributions per pixel from level 0 ... 3),
// so the adds are combined on chip first:
//
//   work item = (image b, head m, 8 x 16 tile of level-0 pixels); the tile's queries are the pixels of every
//   level whose centre falls into the tile (same exact partition as msda_fwd_tiled).  For every level a WH x WW
//   window of accumulators (128 B per pixel) lives in LDS, placed where this head's samples fall (running mean
//   offset of the previous tile of the same head).  Measured on MI355X (tools/micro/lds_atomic_bench.cpp):
//   ds_add_f32 costs ~194 cycles per wave instruction, ds_add_u32 4.4 -- so the accumulators are 32-bit FIXED
//   POINT with a per-tile power-of-two scale chosen from a bound that cannot overflow:
//   |sum| <= (#pairs in tile) * max|grad_out| * max_pair(sum_s |attn|) < 2^30 / scale  (bilinear weights <= 1),
//   i.e. a resolution of ~2e-7 of the largest upstream gradient of the tile.  Non-finite inputs switch the
//   tile to the direct path so NaN/Inf propagate exactly as with float atomics.
//   When the tile is done each touched window pixel leaves the CU as ONE full-line global float atomic
//   (32 lanes x 4 B).  That is ~830 line-ops per tile instead of ~10 000.
//   A sample with a live corner outside its window ("far", a few %) is added to global memory directly, but
//   transposed through ds_bpermute so that it also costs one full-line atomic per corner.
//
//   Lane mapping and per-sample gradient math are msda_bwd_lanegroup's: 8 lanes x 4 channels per (query, head)
//   pair, DPP reductions for grad_attn / grad_loc, coalesced 16/8-byte gradient stores.  `value` itself is read
//   through the L1 path (raw buffer loads, out-of-range offset = 0 for dead corners).
//
// One 768-thread workgroup per CU (12 waves = 3 per SIMD), persistent grid of 256, items walked head-minor.
// Sample records are built 8 samples at a time (two passes per query) to keep the record LDS at 34 KB.
#include "msda_common.hpp"

namespace msda {

constexpr int kBT = 768;                          // threads per workgroup: 12 waves = 3 per SIMD (168 VGPRs); 512 threads were
                                                  // 463 -> 418 us after the step diet, 768 give 353 us: the step is a latency chain
constexpr int kBWaves = kBT / 64;
constexpr int kBTH = 8, kBTW = 16;                // tile in level-0 pixels
constexpr int kBMaxL = 4, kBP = 4;
constexpr int kBSlots = 832;                      // 104 KiB of 128-byte accumulator slots (+1 dummy slot)
constexpr int kBRec = 48;                         // one sample record: {lw, lh, a, flags} {4 value offsets} {4 LDS addresses}
constexpr int kBPairRec = 8 * kBRec + 16;         // 8 sample records per pass, padded
constexpr int kBWaveRec = 8 * kBPairRec;
constexpr int kBRecBytes = kBWaves * kBWaveRec;
constexpr int kBMaxTileQ = 256;

struct BwdMeta {
  int H[kBMaxL], W[kBMaxL], start[kBMaxL];
  int WH[kBMaxL], WW[kBMaxL], slot[kBMaxL + 1];
  int ys[kBMaxL], xs[kBMaxL], ny[kBMaxL], nx[kBMaxL];
  int oy[kBMaxL], ox[kBMaxL], base[kBMaxL];
  float gcy[kBMaxL], gcx[kBMaxL];
  float dev[kBMaxL][2];
  float devacc[kBMaxL][4];
  int TY, TX;
  uint32_t gmax_bits, amax_bits;                  // per-tile max |grad_out| and max_pair sum |attn| (float bits)
  int qtab[kBMaxTileQ];
  uint32_t slot_tab[kBSlots];                     // (level << 28) | (row * W_level + col)
};

// Accumulator slots are 33 words apart: with 32-word slots every pair's lane j would hit bank 4j + c whatever
// the pixel (8-way conflicts, measured 64 % of the LDS cycles); with 33 the pixel index rotates the banks.
constexpr int kBSlotBytes = 132;
constexpr int kBWinBytes = ((kBSlots + 8) * kBSlotBytes + 15) / 16 * 16;   // + 8 per-pair sink slots
constexpr int kBLdsBytes = kBWinBytes + kBRecBytes + ((sizeof(BwdMeta) + 15) / 16) * 16;
static_assert(kBLdsBytes <= 160 * 1024, "one workgroup per CU");

__device__ __forceinline__ int bwd_ceil_div_signed(int a, int b) { return a >= 0 ? (a + b - 1) / b : -((-a) / b); }

template <int CTRL>
__device__ __forceinline__ float bdpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float group8_sum(float v) {  // over the 8 lanes of a pair; every lane gets the total
  // v_add_f32 with the DPP modifier on its first source: one instruction per butterfly step (the compiler keeps
  // v_mov_b32_dpp + v_add_f32 apart for float adds -- 9 extra VALU instructions per sample step)
  // (s_nop 1: a DPP read of a VGPR needs two wait states after the VALU write of it; the compiler does not see into
  // the asm and would not insert them)
  asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(v) : "v"(v));
  asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(v) : "v"(v));
  asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(v) : "v"(v));
  return v;
}

typedef int __attribute__((address_space(3)))* lds_int_ptr;
__device__ __forceinline__ void lds_add(uint32_t lds_byte_addr, int v) {   // ds_add_u32, no return value
  __hip_atomic_fetch_add(reinterpret_cast<lds_int_ptr>((uintptr_t)lds_byte_addr), v, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int cvt_rn_i32(float x) {   // floor(x + 0.5): one VALU instruction
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ float abs_or_inf(float x) {  // |x|, +inf for NaN/Inf (so that a max() sees it)
  const float a = fabsf(x);
  return (a <= 3.402823466e+38f) ? a : __builtin_inff();
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// GV = false: grad_sampling_loc and grad_attn_weight only -- no accumulator windows, no far atomics, no flush; grad_value is
// somebody else's (msda_bwd_regions.hip adds it on the destination side).
template <bool GV>
__device__ __forceinline__ void bwd_tiled_body(const float* __restrict__ grad_out, const float* __restrict__ value,
                                               const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                                               const float* __restrict__ loc, const float* __restrict__ attn, const Dims& d,
                                               float* __restrict__ grad_value, float* __restrict__ grad_loc,
                                               float* __restrict__ grad_attn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  BwdMeta& mt = *reinterpret_cast<BwdMeta*>(smem + kBWinBytes + kBRecBytes);
  constexpr int P = kBP;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int L = d.L, LP = L * P, M = d.M;
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // ---- once per launch ------------------------------------------------------------------------------
  if (tid == 0) {
    for (int l = 0; l < L; ++l) {
      mt.H[l] = (int)shapes[2 * l];
      mt.W[l] = (int)shapes[2 * l + 1];
      mt.start[l] = (int)lsi[l];
    }
    const int H0 = mt.H[0], W0 = mt.W[0];
    mt.TY = (H0 + kBTH - 1) / kBTH;
    mt.TX = (W0 + kBTW - 1) / kBTW;
    int total = 0;
    // Window of a level = the tile's footprint on it + a margin.  Round 5: the margin GROWS with the level (-2 / 0 / +2 / +2 around the
    // common value, which is the largest that fits): the samples spread by the same number of pixels on every level, but on a fine level
    // only the queries near the tile's edge can leave the window, on a coarse one (footprint 2 x 1) every query can.  At the R50 shapes:
    // 14x22 / 12x16 / 12x14 / 11x12 instead of 16x24 / 12x16 / 10x12 / 9x10 -- far samples 4.6 -> 3.0 % at a spread of 2 px, 14.7 -> ~11 % at
    // 3 px (tools/win_geometry_search.py --big; each far sample costs four full-line float atomics here).
    for (int margin = 10; margin >= 0; --margin) {
      total = 0;
      for (int l = 0; l < L; ++l) {
        const int ex = (kBTW * mt.W[l] + W0 - 1) / W0, ey = (kBTH * mt.H[l] + H0 - 1) / H0;
        const int ml = max(0, margin + (l == 0 ? -2 : l == 1 ? 0 : 2));
        mt.WW[l] = min(mt.W[l], ex + ml);
        mt.WH[l] = min(mt.H[l], ey + ml);
        total += mt.WW[l] * mt.WH[l];
      }
      if (total <= kBSlots) break;
    }
    while (total > kBSlots) {  // odd pyramids: the largest window is dropped, that level goes straight to global
      int big = 0;
      for (int l = 1; l < L; ++l)
        if (mt.WW[l] * mt.WH[l] > mt.WW[big] * mt.WH[big]) big = l;
      total -= mt.WW[big] * mt.WH[big];
      mt.WW[big] = mt.WH[big] = 0;
    }
    int acc = 0;
    for (int l = 0; l < kBMaxL; ++l) {
      mt.slot[l] = acc;
      if (l < L) acc += mt.WW[l] * mt.WH[l];
      mt.dev[l][0] = mt.dev[l][1] = 0.f;
      mt.devacc[l][0] = mt.devacc[l][1] = mt.devacc[l][2] = 0.f;
    }
    mt.slot[kBMaxL] = acc;
  }
  __syncthreads();
  const int nslots = mt.slot[kBMaxL];
  for (int p = tid; p < nslots; p += kBT) {
    int l = 0;
    for (int ll = 1; ll < L; ++ll)
      if (p >= mt.slot[ll] && mt.WW[ll] > 0) l = ll;
    const int rel = p - mt.slot[l], ww = mt.WW[l];
    const int r = rel / ww, c = rel - r * ww;
    mt.slot_tab[p] = ((uint32_t)l << 28) | (uint32_t)(r * mt.W[l] + c);
  }
  __syncthreads();

  // ---- per-lane constants ----------------------------------------------------------------------------
  const int pw = lane >> 3, j = lane & 7;            // pair of the wave, lane of the pair
  const uint32_t lane_off = (uint32_t)j * 16u;
  const uint32_t sink = smem_base + (kBSlots + pw) * kBSlotBytes;   // this pair's sink slot (never flushed)
  const uint32_t rec_pair = kBWinBytes + wv * kBWaveRec + pw * kBPairRec;
  // per-level launch constants used by the compile-time-unrolled sample steps: force them into SGPRs
  int lvH[kBMaxL], lvW[kBMaxL];
  uint32_t lvRowG[kBMaxL], lvRowL[kBMaxL];
  const uint32_t pix_bytes = (uint32_t)M * 128u;
#pragma unroll
  for (int l = 0; l < kBMaxL; ++l) {
    const int ll = min(l, L - 1);
    lvH[l] = __builtin_amdgcn_readfirstlane(mt.H[ll]);
    lvW[l] = __builtin_amdgcn_readfirstlane(mt.W[ll]);
    lvRowG[l] = (uint32_t)lvW[l] * pix_bytes;
    lvRowL[l] = (uint32_t)__builtin_amdgcn_readfirstlane(mt.WW[ll]) * (uint32_t)kBSlotBytes;
  }
  const int TY = mt.TY, TX = mt.TX;
  const int items = d.N * M * TY * TX;
  const int pairs_per_image = d.Lq * M;

  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int m = item % M;
    const int tile = (item / M) % (TY * TX);
    const int b = item / (M * TY * TX);
    const int ty = tile / TX, tx = tile % TX;

    if (tid < L) {
      const int l = tid;
      const int H0 = mt.H[0], W0 = mt.W[0], Hl = mt.H[l], Wl = mt.W[l];
      int xs = bwd_ceil_div_signed(tx * 2 * kBTW * Wl - W0, 2 * W0);
      int xe = bwd_ceil_div_signed((tx + 1) * 2 * kBTW * Wl - W0, 2 * W0);
      int ys = bwd_ceil_div_signed(ty * 2 * kBTH * Hl - H0, 2 * H0);
      int ye = bwd_ceil_div_signed((ty + 1) * 2 * kBTH * Hl - H0, 2 * H0);
      xs = max(0, min(xs, Wl)); xe = max(xs, min(xe, Wl));
      ys = max(0, min(ys, Hl)); ye = max(ys, min(ye, Hl));
      if (tx == TX - 1) xe = Wl;
      if (ty == TY - 1) ye = Hl;
      mt.xs[l] = xs; mt.ys[l] = ys; mt.nx[l] = xe - xs; mt.ny[l] = ye - ys;
      const float x_lo = (float)(tx * kBTW) / W0, x_hi = (float)min((tx + 1) * kBTW, W0) / W0;
      const float y_lo = (float)(ty * kBTH) / H0, y_hi = (float)min((ty + 1) * kBTH, H0) / H0;
      const float gcx = 0.5f * (x_lo + x_hi) * Wl - 0.5f, gcy = 0.5f * (y_lo + y_hi) * Hl - 0.5f;
      mt.gcx[l] = gcx; mt.gcy[l] = gcy;
      if (mt.devacc[l][2] > 0.f) {
        mt.dev[l][0] = mt.devacc[l][0] / mt.devacc[l][2];
        mt.dev[l][1] = mt.devacc[l][1] / mt.devacc[l][2];
      }
      mt.devacc[l][0] = mt.devacc[l][1] = mt.devacc[l][2] = 0.f;
      const int oy = (int)floorf(gcy + mt.dev[l][0] + 1.0f - 0.5f * (float)mt.WH[l]);
      const int ox = (int)floorf(gcx + mt.dev[l][1] + 1.0f - 0.5f * (float)mt.WW[l]);
      const int oyc = max(0, min(oy, Hl - mt.WH[l])), oxc = max(0, min(ox, Wl - mt.WW[l]));
      mt.oy[l] = oyc; mt.ox[l] = oxc;
      mt.base[l] = mt.start[l] + oyc * Wl + oxc;
    }
    if (tid == 0) { mt.gmax_bits = 0u; mt.amax_bits = 0u; }
    // zero the accumulator windows (and the sink)
    if constexpr (GV)
      for (int o = tid * 16; o < nslots * kBSlotBytes; o += kBT * 16) *reinterpret_cast<f32x4*>(smem + o) = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    int cum[kBMaxL + 1];
    cum[0] = 0;
#pragma unroll
    for (int l = 0; l < kBMaxL; ++l) cum[l + 1] = cum[l] + (l < L ? mt.nx[l] * mt.ny[l] : 0);
    const int nq = cum[kBMaxL];

    const int64_t img_val = (int64_t)b * d.S * M * 32;
    const __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(value) + img_val, 0, (int)((uint32_t)d.S * pix_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t lsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(loc) + (int64_t)b * pairs_per_image * (2 * LP), 0,
        (int)((uint32_t)pairs_per_image * (uint32_t)(LP * 8)), 0x00020000);
    const __amdgpu_buffer_rsrc_t asrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(attn) + (int64_t)b * pairs_per_image * LP, 0,
        (int)((uint32_t)pairs_per_image * (uint32_t)(LP * 4)), 0x00020000);
    const __amdgpu_buffer_rsrc_t gsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(grad_out) + (int64_t)b * pairs_per_image * 32, 0,
        (int)((uint32_t)pairs_per_image * 128u), 0x00020000);
    const uint32_t hoff = (uint32_t)m * 128u;
    char* gv_head = reinterpret_cast<char*>(grad_value + img_val) + hoff;   // + pixel byte offset + channel * 4

    float dsum_y[2] = {0.f, 0.f}, dsum_x[2] = {0.f, 0.f}, dsum_n[2] = {0.f, 0.f};   // per pass (levels differ)

    float scale = 1.f, inv_scale = 1.f;   // fixed-point scale of the LDS accumulators of this tile
    bool use_lds = false;
    for (int qbase = 0; qbase < nq; qbase += kBMaxTileQ) {
      const int nround = min(kBMaxTileQ, nq - qbase);
      if (tid < nround) {
        const int qi = qbase + tid;
        int l = 0;
#pragma unroll
        for (int ll = 1; ll < kBMaxL; ++ll) l += (qi >= cum[ll]) ? 1 : 0;
        const int jj = qi - (l == 0 ? cum[0] : (l == 1 ? cum[1] : (l == 2 ? cum[2] : cum[3])));
        const int nx = mt.nx[l];
        const int yy = (int)(((float)jj + 0.5f) / (float)nx);
        const int xx = jj - yy * nx;
        mt.qtab[tid] = mt.start[l] + (mt.ys[l] + yy) * mt.W[l] + mt.xs[l] + xx;
      }
      __syncthreads();

      // ---- fixed-point scale of this tile (LDS accumulation needs the whole tile in one round) ----------
      const bool one_round = GV && nq <= kBMaxTileQ;
      if (one_round) {
        float gm = 0.f, am = 0.f;
        for (int e = tid; e < nround * 8; e += kBT) {   // (pair, lane of the pair)
          const int fp = mt.qtab[e >> 3] * M + m, jj = e & 7;
          const f32x4 g4 = buffer_load_f32x4(gsrc, (uint32_t)fp * 128u + (uint32_t)jj * 16u, 0);
          gm = fmaxf(gm, fmaxf(fmaxf(abs_or_inf(g4[0]), abs_or_inf(g4[1])), fmaxf(abs_or_inf(g4[2]), abs_or_inf(g4[3]))));
          float as = 0.f;
          if (2 * jj < LP) {
            const uint2 a2 = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(
                asrc, (uint32_t)fp * (uint32_t)(LP * 4) + (uint32_t)jj * 8u, 0, 0));
            as = abs_or_inf(__uint_as_float(a2.x)) + abs_or_inf(__uint_as_float(a2.y));
          }
          am = fmaxf(am, group8_sum(as));
        }
        gm = wave_max(gm); am = wave_max(am);
        if (lane == 0) {   // non-negative floats order like their bit patterns
          atomicMax(&mt.gmax_bits, __float_as_uint(gm));
          atomicMax(&mt.amax_bits, __float_as_uint(am));
        }
      }
      __syncthreads();
      if (one_round) {
        const float bound = (float)nq * __uint_as_float(mt.gmax_bits) * __uint_as_float(mt.amax_bits);
        use_lds = bound < 0x1p120f;                     // false for NaN / Inf, and for bounds the clamped exponent below cannot scale into int32
        if (use_lds && bound > 0.f) {
          int k;
          (void)frexpf(bound, &k);                       // bound < 2^k
          k = max(-90, min(90, 30 - k));
          scale = ldexpf(1.f, k);
          inv_scale = ldexpf(1.f, -k);
        }
      }

      const int niter = (nround + kBWaves * 8 - 1) / (kBWaves * 8);
      for (int it = 0; it < niter; ++it) {
        const int qi = it * (kBWaves * 8) + wv * 8 + pw;
        const int pair = qi < nround ? mt.qtab[qi] * M + m : -1;   // pair index inside image b
        // this lane's 4 channels of the upstream gradient, and the sampling data of samples j and 8 + j
        f32x4 go = {0.f, 0.f, 0.f, 0.f};
        float2 lc[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
        float at[2] = {0.f, 0.f};
        if (pair >= 0) {
          go = buffer_load_f32x4(gsrc, (uint32_t)pair * 128u + lane_off, 0);
#pragma unroll
          for (int ps = 0; ps < 2; ++ps) {
            const int s = 8 * ps + j;
            if (s < LP) {
              const uint2 l2 = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(
                  lsrc, (uint32_t)pair * (uint32_t)(LP * 8) + (uint32_t)s * 8u, 0, 0));
              lc[ps] = make_float2(__uint_as_float(l2.x), __uint_as_float(l2.y));
              at[ps] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                  asrc, (uint32_t)pair * (uint32_t)(LP * 4) + (uint32_t)s * 4u, 0, 0));
            }
          }
        }
        f32x4 gl = {0.f, 0.f, 0.f, 0.f};   // grad_loc of samples 2j, 2j+1
        float ga0 = 0.f, ga1 = 0.f;         // grad_attn of samples 2j, 2j+1

#pragma unroll 1   // unrolling both passes interleaves them and doubles the register pressure (spills)
        for (int ps = 0; ps < 2; ++ps) {
          // -- prepare sample s = 8 ps + j: {lw, lh, a, flags} {g00, l00} -------------------------------
          {
            const int s = 8 * ps + j;
            const int l = (j >> 2) + 2 * ps;          // level of the sample (P = 4)
            const int lq = min(l, L - 1);
            const int Hl = mt.H[lq], Wl = mt.W[lq], St = mt.start[lq];
            const int Oy = mt.oy[lq], Ox = mt.ox[lq], Sl = mt.slot[lq], Wh = mt.WH[lq], Ww = mt.WW[lq];
            const float x = lc[ps].x * (float)Wl - 0.5f, y = lc[ps].y * (float)Hl - 0.5f;
            const bool inr = (pair >= 0) && (s < LP) && (y > -1.f) && (x > -1.f) && (y < (float)Hl) && (x < (float)Wl);
            f32x4 r0 = {0.f, 0.f, 0.f, 0.f};
            uint32_t g00 = 0u, l00 = 0u, flags = 0u;
            if (inr) {
              const float yf = floorf(y), xf = floorf(x);
              const int y0 = (int)yf, x0 = (int)xf;
              const bool t_ok = y0 >= 0, b_ok = y0 + 1 <= Hl - 1, l_ok = x0 >= 0, r_ok = x0 + 1 <= Wl - 1;
              flags = (t_ok && l_ok ? 1u : 0u) | (t_ok && r_ok ? 2u : 0u) | (b_ok && l_ok ? 4u : 0u) | (b_ok && r_ok ? 8u : 0u);
              const int ry = y0 - Oy, cx = x0 - Ox;
              const bool rows_in = (!t_ok || (unsigned)ry < (unsigned)Wh) && (!b_ok || (unsigned)(ry + 1) < (unsigned)Wh);
              const bool cols_in = (!l_ok || (unsigned)cx < (unsigned)Ww) && (!r_ok || (unsigned)(cx + 1) < (unsigned)Ww);
              if (!(rows_in && cols_in) || !use_lds) flags |= 16u;   // far: straight to global memory
              r0[0] = x - xf; r0[1] = y - yf; r0[2] = at[ps];
              g00 = (uint32_t)(St + y0 * Wl + x0) * pix_bytes;
              l00 = smem_base + (uint32_t)(Sl + ry * Ww + cx) * (uint32_t)kBSlotBytes;
              if (Ww > 0) {   // statistics for the next tile's window placement
                const float dx = x - mt.gcx[lq], dy = y - mt.gcy[lq];
                if (fabsf(dx) <= 12.f && fabsf(dy) <= 12.f) { dsum_x[ps] += dx; dsum_y[ps] += dy; dsum_n[ps] += 1.f; }
              }
            }
            r0[3] = __uint_as_float(flags);
            // per-corner byte offsets into `value` (out-of-range for dead corners: the raw buffer load returns 0) and LDS
            // accumulator addresses (the pair's sink slot for dead corners and far samples), selected HERE, once per
            // sample, instead of in every lane of the pair at every step
            const bool near = (flags & 16u) == 0u;
            const uint32_t rowg = (uint32_t)Wl * pix_bytes, rowl = (uint32_t)Ww * (uint32_t)kBSlotBytes;
            u32x4 gk, ak;
            gk[0] = (flags & 1u) ? g00 : kOobOffset;
            gk[1] = (flags & 2u) ? g00 + pix_bytes : kOobOffset;
            gk[2] = (flags & 4u) ? g00 + rowg : kOobOffset;
            gk[3] = (flags & 8u) ? g00 + rowg + pix_bytes : kOobOffset;
            ak[0] = ((flags & 1u) && near) ? l00 : sink;
            ak[1] = ((flags & 2u) && near) ? l00 + (uint32_t)kBSlotBytes : sink;
            ak[2] = ((flags & 4u) && near) ? l00 + rowl : sink;
            ak[3] = ((flags & 8u) && near) ? l00 + rowl + (uint32_t)kBSlotBytes : sink;
            *reinterpret_cast<f32x4*>(smem + rec_pair + j * kBRec) = r0;
            *reinterpret_cast<u32x4*>(smem + rec_pair + j * kBRec + 16) = gk;
            *reinterpret_cast<u32x4*>(smem + rec_pair + j * kBRec + 32) = ak;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

          // -- the 8 samples of this pass, software-pipelined by hand: the record and the four value loads of
          //    sample s+1 are issued before the arithmetic of sample s; sched_barriers keep the compiler from
          //    hoisting further ahead (it would spill: 16 data VGPRs per sample in flight)
          uint64_t far_any = 0;
          struct StepIn { f32x4 r0; u32x4 ak; f32x4 v1, v2, v3, v4; };
          auto fetch = [&](int s8, StepIn& in) {
            in.r0 = *reinterpret_cast<const f32x4*>(smem + rec_pair + s8 * kBRec);
            const u32x4 gk = *reinterpret_cast<const u32x4*>(smem + rec_pair + s8 * kBRec + 16);
            in.ak = *reinterpret_cast<const u32x4*>(smem + rec_pair + s8 * kBRec + 32);
            in.v1 = buffer_load_f32x4(vsrc, gk[0] + lane_off, hoff);
            in.v2 = buffer_load_f32x4(vsrc, gk[1] + lane_off, hoff);
            in.v3 = buffer_load_f32x4(vsrc, gk[2] + lane_off, hoff);
            in.v4 = buffer_load_f32x4(vsrc, gk[3] + lane_off, hoff);
          };
          StepIn cur, nxt;
          fetch(0, cur);
#pragma unroll
          for (int s8 = 0; s8 < 8; ++s8) {
            const int s = 8 * ps + s8;
            const int l = s / P;                       // compile-time
            if (s8 + 1 < 8) fetch(s8 + 1, nxt);
            __builtin_amdgcn_sched_barrier(0);
            const float lw = cur.r0[0], lh = cur.r0[1], a = cur.r0[2];
            const uint32_t flags = __float_as_uint(cur.r0[3]);
            const float hw = 1.f - lw, hh = 1.f - lh;
            // Bilinear value and its two location derivatives per channel, on channel PAIRS (v_pk_* math; the kernel is
            // bound by VALU issue, a wave64 instruction holds its SIMD for 4 clocks whether it does one or two lanes of
            // work):   top = v1 + lw (v2 - v1), bot = v3 + lw (v4 - v3), val = top + lh (bot - top),
            //          d val / d y = bot - top,  d val / d x = hh (v2 - v1) + lh (v4 - v3)          (cuh:113-158, refactored)
            typedef float v2f __attribute__((ext_vector_type(2)));
            // (round 4: the three sums are linear in the two corner rows -- A_r = sum_c g_c (left_c + lw (right_c - left_c)) and
            // D_r = sum_c g_c (right_c - left_c) per row: grad_attn = hh A_top + lh A_bot, d/dx = a (hh D_top + lh D_bot),
            // d/dy = a (A_bot - A_top): 9 packed operations per channel pair instead of 13)
            v2f At2 = {0.f, 0.f}, Ab2 = {0.f, 0.f}, Dt2 = {0.f, 0.f}, Db2 = {0.f, 0.f};
            v2f tgs[2];     // go * a * scale: the fixed-point value-gradient of a unit-weight corner
            const float a_sc = a * scale;
#pragma unroll
            for (int cp = 0; cp < 2; ++cp) {
              const v2f V1 = {cur.v1[2 * cp], cur.v1[2 * cp + 1]}, V2 = {cur.v2[2 * cp], cur.v2[2 * cp + 1]};
              const v2f V3 = {cur.v3[2 * cp], cur.v3[2 * cp + 1]}, V4 = {cur.v4[2 * cp], cur.v4[2 * cp + 1]};
              const v2f G = {go[2 * cp], go[2 * cp + 1]};
              const v2f tt = V2 - V1, tb = V4 - V3;
              const v2f top = __builtin_elementwise_fma(v2f{lw, lw}, tt, V1), bot = __builtin_elementwise_fma(v2f{lw, lw}, tb, V3);
              At2 = __builtin_elementwise_fma(G, top, At2);
              Ab2 = __builtin_elementwise_fma(G, bot, Ab2);
              Dt2 = __builtin_elementwise_fma(G, tt, Dt2);
              Db2 = __builtin_elementwise_fma(G, tb, Db2);
              tgs[cp] = G * a_sc;
            }
            const float at_ = At2.x + At2.y, ab_ = Ab2.x + Ab2.y, dt_ = Dt2.x + Dt2.y, db_ = Db2.x + Db2.y;
            const float pa = fmaf(lh, ab_, hh * at_), pwx = a * fmaf(lh, db_, hh * dt_), phy = a * (ab_ - at_);
            // near samples: accumulate into the LDS window; dead corners and far samples go to the pair's sink slot
            const v2f wh = v2f{hh, lh} * hw, wl = v2f{hh, lh} * lw;      // (w1, w3), (w2, w4)
            const uint32_t a1 = cur.ak[0] + lane_off, a2 = cur.ak[1] + lane_off, a3 = cur.ak[2] + lane_off, a4 = cur.ak[3] + lane_off;
#pragma unroll
            for (int cp = 0; GV && cp < 2; ++cp) {
              const v2f g1 = tgs[cp] * wh.x, g2 = tgs[cp] * wl.x, g3 = tgs[cp] * wh.y, g4 = tgs[cp] * wl.y;
              constexpr uint32_t kC0 = 8u, kC1 = 4u;
              lds_add(a1 + kC0 * cp, cvt_rn_i32(g1.x)); lds_add(a1 + kC0 * cp + kC1, cvt_rn_i32(g1.y));
              lds_add(a2 + kC0 * cp, cvt_rn_i32(g2.x)); lds_add(a2 + kC0 * cp + kC1, cvt_rn_i32(g2.y));
              lds_add(a3 + kC0 * cp, cvt_rn_i32(g3.x)); lds_add(a3 + kC0 * cp + kC1, cvt_rn_i32(g3.y));
              lds_add(a4 + kC0 * cp, cvt_rn_i32(g4.x)); lds_add(a4 + kC0 * cp + kC1, cvt_rn_i32(g4.y));
            }
            if constexpr (GV) far_any |= __ballot((flags & 16u) != 0u);
            const float ra = group8_sum(pa);
            const float rw = group8_sum(pwx) * (float)lvW[l];
            const float rh = group8_sum(phy) * (float)lvH[l];
            const bool mine = (j == (s >> 1));
            if ((s & 1) == 0) {
              gl[0] = mine ? rw : gl[0]; gl[1] = mine ? rh : gl[1]; ga0 = mine ? ra : ga0;
            } else {
              gl[2] = mine ? rw : gl[2]; gl[3] = mine ? rh : gl[3]; ga1 = mine ? ra : ga1;
            }
            // (no scheduling barrier at the end of a step: the next step's arithmetic may overlap this one's LDS atomics)
            cur = nxt;
          }
          // -- far samples of this pass: the value gradient w_k * a * g_c does not depend on the sampled values,
          //    so a half-wave (32 lanes = 32 channels) redoes it from the record and the upstream gradient and
          //    issues ONE full-line atomic per live corner.
          if (far_any) {
            const int hl = lane & 31;
#pragma unroll 1
            for (int s8 = 0; s8 < 8; ++s8) {
              const uint32_t myflags = __float_as_uint(*reinterpret_cast<const float*>(smem + rec_pair + s8 * kBRec + 12));
              const uint64_t fmask = __ballot((myflags & 16u) != 0u);
              if (fmask == 0) continue;                  // wave-uniform
              uint32_t hm = (lane < 32) ? (uint32_t)fmask : (uint32_t)(fmask >> 32);
              while (__ballot(hm != 0u)) {
                const bool act = hm != 0u;
                const int first = act ? (__builtin_ctz(hm) & ~7) : 0;          // first lane of the far pair in this half
                const int fp = ((lane & 32) + first) >> 3;                     // that pair's index in the wave
                const uint32_t rp = kBWinBytes + wv * kBWaveRec + fp * kBPairRec + s8 * kBRec;
                const f32x4 r0 = *reinterpret_cast<const f32x4*>(smem + rp);
                const u32x4 gk = *reinterpret_cast<const u32x4*>(smem + rp + 16);   // kOobOffset marks a dead corner
                const int fqi = it * (kBWaves * 8) + wv * 8 + fp;
                const int fpair = (act && fqi < nround) ? mt.qtab[fqi] * M + m : 0;
                const float g = act ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                                          gsrc, (uint32_t)fpair * 128u + (uint32_t)hl * 4u, 0, 0)) : 0.f;
                const float lw = r0[0], lh = r0[1], tgv = g * r0[2];
                if (act && gk[0] != kOobOffset) atomic_add(reinterpret_cast<float*>(gv_head + gk[0]) + hl, (1.f - lh) * (1.f - lw) * tgv);
                if (act && gk[1] != kOobOffset) atomic_add(reinterpret_cast<float*>(gv_head + gk[1]) + hl, (1.f - lh) * lw * tgv);
                if (act && gk[2] != kOobOffset) atomic_add(reinterpret_cast<float*>(gv_head + gk[2]) + hl, lh * (1.f - lw) * tgv);
                if (act && gk[3] != kOobOffset) atomic_add(reinterpret_cast<float*>(gv_head + gk[3]) + hl, lh * lw * tgv);
                if (act) hm &= ~(0xFFu << first);
              }
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();   // the next pass rewrites the records
        }

        if (pair >= 0 && 2 * j < LP) {
          const int64_t gp = (int64_t)b * pairs_per_image + pair;
          *reinterpret_cast<f32x4*>(grad_loc + gp * (2 * LP) + 4 * j) = gl;
          *reinterpret_cast<float2*>(grad_attn + gp * LP + 2 * j) = make_float2(ga0, ga1);
        }
      }
      __syncthreads();
    }

    // ---- flush: every touched window pixel leaves as one full-line atomic (32 lanes x 4 B) -------------
    if constexpr (GV) {
      const int ch = tid & 31;
      const int b0 = mt.base[0], b1 = mt.base[1], b2 = mt.base[2], b3 = mt.base[3];
      for (int p = tid >> 5; p < nslots; p += kBT / 32) {
        const float v = (float)*reinterpret_cast<const int*>(smem + p * kBSlotBytes + ch * 4) * inv_scale;
        const uint32_t e = mt.slot_tab[p];
        const uint32_t l = e >> 28;
        const int base = l == 0 ? b0 : (l == 1 ? b1 : (l == 2 ? b2 : b3));
        const uint32_t gpix = (uint32_t)base + (e & 0x0fffffffu);
        if (__ballot(v != 0.f) != 0)   // wave = two pixels; skip the instruction when both are untouched
          if (v != 0.f) atomic_add(reinterpret_cast<float*>(gv_head + (size_t)gpix * pix_bytes) + ch, v);
      }
    }
    // sample statistics -> window placement of the next tile of this head
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      // lanes with the same (j >> 2) share a level: reduce over lane bits 0,1 (quad) and 3,4,5 (pairs of the wave)
      float sy = dsum_y[ps], sx = dsum_x[ps], sn = dsum_n[ps];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        if (o == 4) continue;
        sy += __shfl_xor(sy, o, 64); sx += __shfl_xor(sx, o, 64); sn += __shfl_xor(sn, o, 64);
      }
      const int l = min((j >> 2) + 2 * ps, L - 1);
      if ((lane == 0 || lane == 4) && sn > 0.f) {
        atomicAdd(&mt.devacc[l][0], sy);
        atomicAdd(&mt.devacc[l][1], sx);
        atomicAdd(&mt.devacc[l][2], sn);
      }
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(kBT, 3)
msda_bwd_tiled(const float* __restrict__ grad_out, const float* __restrict__ value,
               const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
               const float* __restrict__ loc, const float* __restrict__ attn, Dims d,
               float* __restrict__ grad_value, float* __restrict__ grad_loc, float* __restrict__ grad_attn) {
  bwd_tiled_body<true>(grad_out, value, shapes, lsi, loc, attn, d, grad_value, grad_loc, grad_attn);
}

__global__ void __launch_bounds__(kBT, 3)
msda_bwd_tiled_nogv(const float* __restrict__ grad_out, const float* __restrict__ value,
                    const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                    const float* __restrict__ loc, const float* __restrict__ attn, Dims d,
                    float* __restrict__ grad_loc, float* __restrict__ grad_attn) {
  bwd_tiled_body<false>(grad_out, value, shapes, lsi, loc, attn, d, nullptr, grad_loc, grad_attn);
}

// Host side ---------------------------------------------------------------------------------------
bool tiled_backward_ok(const Dims& d) {
  return d.D == 32 && d.P == kBP && d.L <= kBMaxL && d.Lq == d.S &&
         (int64_t)d.S * d.M * 128 < (int64_t)kOobOffset && d.N <= 65535;
}

int launch_backward_tiled(const float* grad_out, const float* value, const int64_t* shapes, const int64_t* lsi,
                          const float* loc, const float* attn, const Dims& d, float* grad_value, float* grad_loc,
                          float* grad_attn, hipStream_t stream) {
  static std::atomic<uint64_t> lds_opted_in{0};
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(msda_bwd_tiled), kBLdsBytes, lds_opted_in)) return rc;
  // persistent grid: one 1024-thread workgroup per CU; a multiple of 8 so that item % M tracks blockIdx % 8
  hipLaunchKernelGGL(msda_bwd_tiled, dim3(256), dim3(kBT), kBLdsBytes, stream, grad_out, value, shapes, lsi, loc, attn,
                     d, grad_value, grad_loc, grad_attn);
  return (int)hipGetLastError();
}

// grad_sampling_loc and grad_attn_weight only (first step of launch_backward_regions)
int launch_backward_tiled_nogv(const float* grad_out, const float* value, const int64_t* shapes, const int64_t* lsi,
                               const float* loc, const float* attn, const Dims& d, float* grad_loc, float* grad_attn,
                               hipStream_t stream) {
  static std::atomic<uint64_t> lds_opted_in{0};
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(msda_bwd_tiled_nogv), kBLdsBytes, lds_opted_in)) return rc;
  hipLaunchKernelGGL(msda_bwd_tiled_nogv, dim3(256), dim3(kBT), kBLdsBytes, stream, grad_out, value, shapes, lsi, loc, attn,
                     d, grad_loc, grad_attn);
  return (int)hipGetLastError();
}

}  // namespace msda
