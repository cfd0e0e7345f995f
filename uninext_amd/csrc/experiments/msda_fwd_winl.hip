// msda_fwd_winl -- MSDeformAttn forward for encoder-style calls (Lq == S): LDS windows on all four pyramid levels, ONE LANE
// per (query, head) pair.  fp32, D = 32, L = P = 4.  gfx950 only.  Replaces, for these calls, the work of
// ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299.
//
// Why another formulation (round 6).  msda_fwd_win gives a QUAD of lanes to a pair: every sample's
// weights and LDS addresses travel through the quad by DPP, lane k prepares LEVEL k so a level's constants are vector
// registers, and an item issues 3.85 vector instructions per useful packed FMA (21.9 M per launch for 5.7 M FMAs).  Its time
// follows that count (profiles/r03_forward_window_analysis.txt).  Here a lane owns all 32 channels of its pair:
//
//   lane           = one (query, head) pair; its 16 samples are walked level by level, point by point, by every lane at the
//                    same time -- so the level's size, window origin and window limits are SCALARS, nothing is broadcast, and a
//                    sample's preparation (33 vector instructions) is paid once per 64 packed FMAs instead of once per 16.
//   work item      = (image, head, 8 x 16 tile of level-0 pixels + the pixels of levels 1..3 whose centres fall into the
//                    tile's rectangle) -- the partition, the windows (12x20 / 10x14 / 10x12 / 10x10 pixels, 76 KB), their
//                    placement at the mean sample position and the LDS-DMA staging are msda_fwd_win's.  170 pairs at the R50
//                    shapes = a 192-thread workgroup: waves 0 / 1 take the tile's rows 0-3 / 4-7, wave 2 the queries of
//                    levels 1..3; two workgroups per CU.
//   LDS banks      = a ds_read_b128 is served in four groups of 16 lanes.  Lane class (e, t) = (bit 3, bits 0-2 of the lane
//                    id -- 16 different classes in each service group): of the two x-adjacent corner pixels of a bilinear row
//                    the lane reads the one whose window slot has parity e FIRST, and at its j-th read the 16-byte piece
//                    j ^ t.  Bank quad = 8 * parity + piece: the 16 lanes of a group cover all 64 banks exactly once at every
//                    instruction, for ANY sample positions.  Register set j of a lane therefore accumulates piece j ^ t --
//                    a fixed piece per lane, so nothing is permuted until the final store.  The bottom corner row is the
//                    same address + a compile-time offset.  Cost: one v_add_u32 per two ds_read_b128 (4 packed FMAs).
//   far            = an in-range sample with a corner outside its window: flagged in a per-lane bit mask during the pass and
//                    worked off behind it with raw buffer loads (invalid corners at an out-of-range offset).  Correctness never
//                    depends on where the windows are; only speed does.
//
// Per sample and wave: 64 v_pk_fma_f32 + 16 v_add_u32 + 33 of preparation; per item ~6 k vector instructions against 11.3 k.
//
// All geometry comes from the int64 shape tensors on the device; the host only knows S.  The grid is persistent: the two
// workgroups a CU holds walk the items kk, kk + K, ... of their head.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "../msda_common.hpp"

#ifndef WINL_LB
#define WINL_LB 3
#endif
#ifndef WINL_RING
#define WINL_RING 8     // LDS reads in flight per lane during the pass (8 or 16)
#endif

namespace msda {
namespace {

constexpr int kT = 384, kWaves = kT / 64, kPairs = kT / 2;       // two lanes per (query, head) pair
constexpr int kTH = 8, kTW = 16;                                // level-0 pairs of an item: waves 0..3, 32 pairs each
constexpr int kWH[4] = {12, 10, 10, 10};
constexpr int kWW[4] = {20, 14, 12, 10};                        // even: slot parity == column parity in every row
constexpr int kBase[5] = {0, 240, 384, 504, 608};               // first slot of each window, multiples of 8: a 1 KB
                                                                // DMA chunk (8 slots) never straddles two levels
static_assert(kBase[1] >= kWH[0] * kWW[0] && kBase[2] >= kBase[1] + kWH[1] * kWW[1] &&
              kBase[3] >= kBase[2] + kWH[2] * kWW[2] && kBase[4] >= kBase[3] + kWH[3] * kWW[3], "window table");
static_assert(kBase[1] % 8 == 0 && kBase[2] % 8 == 0 && kBase[3] % 8 == 0 && kBase[4] % 8 == 0, "DMA chunks / parity");
constexpr int kSlots = kBase[4];
constexpr int kZeroOff = kSlots * 128;                          // all-zero region: target of dead / far samples
constexpr int kZeroBytes = kWW[0] * 128 + 256;                  // a bottom-row read lands at most one level-0 row further
static_assert(kZeroOff % 256 == 0, "zero region: slot parity by address bit 7");
struct Meta {
  int part[4][4][4];                                            // [level-0 wave] per level: sum x0, sum y0, count, unused
  int lvl[4][8];                                                // per level: H, W, first pixel, window rows, window columns
  int org[4][4];                                                // per level: window origin x, y; last near column / row (later rounds)
  int stat[4];
};
constexpr int kMetaOff = kZeroOff + kZeroBytes;
constexpr int kLdsBytes = kMetaOff + ((sizeof(Meta) + 15) / 16) * 16;
static_assert(kLdsBytes <= 80 * 1024, "two workgroups per CU");

typedef const f32x4 __attribute__((address_space(3)))* lds4;
typedef float v2f __attribute__((ext_vector_type(2)));        // packed fp32 math: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32

template <int CTRL, int ROWS = 0xF>
__device__ __forceinline__ float dppf(float v) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, ROWS, 0xF, true));
}
// sum over the wave, valid in lane 63 (quad_perm x 2, row_shr 4 / 8, row_bcast 15 / 31)
__device__ __forceinline__ float wave_total(float v) {
  v += dppf<0xB1>(v);
  v += dppf<0x4E>(v);
  v += dppf<0x114>(v);
  v += dppf<0x118>(v);
  v += dppf<0x142, 0xA>(v);
  v += dppf<0x143, 0xC>(v);
  return v;
}
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) {   // (a & 0xffffff) * (b & 0xffffff) + c
  uint32_t r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ int cvt_i32(float x) {             // saturating, NaN -> 0 (a C cast is undefined out of range)
  int r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
// (a - b) clamped to [0, 1], NaN -> 0 (kernels run with DX10_CLAMP): the fractional parts, safe for poisoned locations
__device__ __forceinline__ v2f sub_clamp01(v2f a, v2f b) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// m = 2 * m + (bit set in `mask` for this lane)
__device__ __forceinline__ uint32_t shift_in(uint32_t m, bool bit) {
  const unsigned long long mask = __builtin_amdgcn_ballot_w64(bit);
  uint32_t r;
  unsigned long long junk;
  asm("v_addc_co_u32 %0, %1, %2, %2, %3" : "=v"(r), "=s"(junk) : "v"(m), "s"(mask));
  return r;
}

}  // namespace

template <int I> using IC = std::integral_constant<int, I>;

template <int REFD>
__device__ __forceinline__ void winl_body(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                          const int64_t* __restrict__ lsi, const float* __restrict__ loc,
                                          const float* __restrict__ attn, const Dims& d, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Meta& mt = *reinterpret_cast<Meta*>(smem + kMetaOff);
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = d.M;
  const int m = blockIdx.x, kk = blockIdx.y, K = gridDim.y;   // workgroup kk of K on head m

  // ---- launch constants straight from the shape tensors (uniform addresses: scalar loads) ---------------------------------
  int lvH[4], lvW[4], lvS[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    lvH[l] = (int)shapes[2 * l];
    lvW[l] = (int)shapes[2 * l + 1];
    lvS[l] = (int)lsi[l];
  }
  const int TY = (lvH[0] + kTH - 1) / kTH, TX = (lvW[0] + kTW - 1) / kTW;
  const int ntiles = TY * TX, nitems = d.N * ntiles;       // work items of this head: (image, tile), image-major
  if (kk >= nitems) return;
  for (int o = tid * 16; o < kZeroBytes; o += kT * 16) *reinterpret_cast<f32x4*>(smem + kZeroOff + o) = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tid >= 64 && tid < 68) {
    const int l = tid - 64;
    const int4 a = make_int4(l == 0 ? lvH[0] : l == 1 ? lvH[1] : l == 2 ? lvH[2] : lvH[3], l == 0 ? lvW[0] : l == 1 ? lvW[1] : l == 2 ? lvW[2] : lvW[3],
                             l == 0 ? lvS[0] : l == 1 ? lvS[1] : l == 2 ? lvS[2] : lvS[3], l == 0 ? kWH[0] : l == 1 ? kWH[1] : l == 2 ? kWH[2] : kWH[3]);
    const int4 b = make_int4(l == 0 ? kWW[0] : l == 1 ? kWW[1] : l == 2 ? kWW[2] : kWW[3], 0, 0, 0);
    *reinterpret_cast<int4*>(&mt.lvl[l][0]) = a;
    *reinterpret_cast<int4*>(&mt.lvl[l][4]) = b;
  }
  __syncthreads();

  const uint32_t pixB = (uint32_t)M * 128u;                 // bytes from a pixel of head m to the next one
  const uint32_t hoff = (uint32_t)m * 128u;

  // ---- this lane's half of its pair and its bank class: parity read first, piece rotation; the eight piece offsets stay in
  // registers -------------------------------------------------------------------------------------------------------------
  const uint32_t hh = (uint32_t)lane & 1u;                  // this lane samples the points 2 hh, 2 hh + 1 of every level
  uint32_t t16 = 16u * (uint32_t)(lane & 7);               // piece j of this lane's j-th read: byte offset t16 ^ 16 j
  asm volatile("" : "+v"(t16));
  uint32_t E7 = ((uint32_t)(lane >> 3) & 1u) << 7;
  asm volatile("" : "+v"(E7));
  const uint32_t zero_base = smem_base + kZeroOff;
  float fW[4], fH[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) { fW[l] = (float)lvW[l]; fH[l] = (float)lvH[l]; }

#ifdef WINL_STAGGER
  if (blockIdx.y * gridDim.x + blockIdx.x >= 256u) { for (int z = 0; z < WINL_STAGGER; ++z) __builtin_amdgcn_s_sleep(127); }
#endif
  for (int item = kk, it = 0; item < nitems; item += K, ++it) {
    if (it > 0) __syncthreads();                            // everybody left the previous item's windows
    const int b = (int)(((float)item + 0.5f) * __builtin_amdgcn_rcpf((float)ntiles));
    const int64_t pair_img = (int64_t)b * d.Lq * M;         // first (query, head) pair of this item's image
    const float* const loc_img = loc + pair_img * 32;       // uniform bases: per-lane offsets stay 32-bit (S * M * 128 < 2^31)
    const float* const attn_img = attn + pair_img * 16;
    float* const out_img = out + pair_img * 32;
    const __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(value) + (int64_t)b * d.S * M * 32, 0, (int)((uint32_t)d.S * pixB), 0x00020000);

    // ---- tile geometry: lane k (of every quad) works out level k's query rectangle, readlane makes it scalar -------------
    // level-k pixels [f(t), f(t + 1)) with f(t) = ceil(t * T * n / n0 - 1/2) are the ones whose centre falls into tile t
    // (msda_fwd_win's partition: any monotone f with f(0) = 0 is exact as long as every workgroup evaluates the same expression)
    int gxs[4], gys[4], gnx[4], gny[4];
    int vxs, vys;                                           // lane k: level k's first column / row (for the default window origin)
    {
      const int kq = lane & 3;
      const int2 gHW = *reinterpret_cast<const int2*>(&mt.lvl[kq][0]);
      const int gW = gHW.y, gH = gHW.x;
      const float fxs = (float)(kTW * gW) * __builtin_amdgcn_rcpf((float)lvW[0]), fys = (float)(kTH * gH) * __builtin_amdgcn_rcpf((float)lvH[0]);
      const int tile_ = item - b * ntiles;
      const int ty = (int)(((float)tile_ + 0.5f) * __builtin_amdgcn_rcpf((float)TX)), tx = tile_ - ty * TX;
      const int xs = min(max((int)ceilf((float)tx * fxs - 0.5f), 0), gW);
      const int xe = tx == TX - 1 ? gW : min(max((int)ceilf((float)(tx + 1) * fxs - 0.5f), xs), gW);
      const int ys = min(max((int)ceilf((float)ty * fys - 0.5f), 0), gH);
      const int ye = ty == TY - 1 ? gH : min(max((int)ceilf((float)(ty + 1) * fys - 0.5f), ys), gH);
      vxs = xs; vys = ys;
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        gxs[l] = __builtin_amdgcn_readlane(xs, l);
        gys[l] = __builtin_amdgcn_readlane(ys, l);
        gnx[l] = __builtin_amdgcn_readlane(xe - xs, l);
        gny[l] = __builtin_amdgcn_readlane(ye - ys, l);
      }
    }
    const int e1 = gnx[1] * gny[1], e2 = e1 + gnx[2] * gny[2], nrest = e2 + gnx[3] * gny[3];
    // rounds of the workgroup: round 0 = the level-0 tile (waves 0..3, 32 pairs each) + the first 64 queries of levels 1..3
    // (waves 4, 5); odd pyramids with more of them take further rounds of 192 pairs
    const int nrounds = 1 + (nrest > 64 ? (nrest - 64 + kPairs - 1) / kPairs : 0);

    for (int rnd = 0; rnd < nrounds; ++rnd) {
      // ---- this lane's query ---------------------------------------------------------------------------------------------
      bool live;
      uint32_t qidx;
      const bool l0wave = rnd == 0 && wv < 4;               // wave-uniform
      const int ri0 = rnd == 0 ? (wv - 4) * 32 : 64 + (rnd - 1) * kPairs + wv * 32;   // first rest index of this wave in this round
      if (rnd > 0 && ri0 >= nrest) continue;                // (later rounds only: a wave without a query; no barrier follows)
      if (l0wave) {
        const int pi = tid >> 1, row = pi >> 4, col = pi & 15;
        live = col < gnx[0] && row < gny[0];
        qidx = (uint32_t)(lvS[0] + (gys[0] + row) * lvW[0] + gxs[0] + col);
      } else {
        const int ri = ri0 + (lane >> 1);
        live = ri < nrest;
        const bool c1 = ri >= e1, c2 = ri >= e2;
        const int j = ri - (c2 ? e2 : c1 ? e1 : 0);
        const int nx = c2 ? gnx[3] : c1 ? gnx[2] : gnx[1];
        const int xs = c2 ? gxs[3] : c1 ? gxs[2] : gxs[1], ys = c2 ? gys[3] : c1 ? gys[2] : gys[1];
        const int Wq = c2 ? lvW[3] : c1 ? lvW[2] : lvW[1], Sq = c2 ? lvS[3] : c1 ? lvS[2] : lvS[1];
        const int yy = (int)(((float)j + 0.5f) * __builtin_amdgcn_rcpf((float)max(nx, 1)));
        qidx = (uint32_t)(Sq + (ys + yy) * Wq + xs + (j - yy * nx));
      }
      live = live && qidx < (uint32_t)d.Lq;                 // (shapes whose pixel count exceeds num_query: never outside the tensors)
      if (!live) qidx = 0u;
      const uint32_t pair = mad_u24(qidx, (uint32_t)M, (uint32_t)m);

      // ---- this lane's 8 locations and weights: per level 16 + 8 bytes, contiguous with the partner lane's ------------------------
      // a dead lane's locations are far outside every level: all its samples are out of range without a `live &&`
      f32x4 lc[4];
      msda::f32x2 at[4];
#pragma unroll
      for (int l = 0; l < 4; ++l) { lc[l] = f32x4{-4.f, -4.f, -4.f, -4.f}; at[l] = msda::f32x2{0.f, 0.f}; }
#ifdef WINL_NOLOADS
      if (live) {
        const float qx = ((float)(qidx % 167u) + 0.5f) * (1.f / 167.f), qy = ((float)((qidx / 167u) % 100u) + 0.5f) * 0.01f;
        for (int l = 0; l < 4; ++l) { lc[l] = f32x4{qx + 0.004f * l, qy + 0.003f * l, qx - 0.002f * l, qy + 0.001f * l}; at[l] = msda::f32x2{0.0625f, 0.0625f}; }
      }
      if (false) {
#else
      if (live) {
#endif
        const f32x4* lp = reinterpret_cast<const f32x4*>(loc_img + (pair * 32u + 4u * hh));
        const msda::f32x2* ap = reinterpret_cast<const msda::f32x2*>(attn_img + (pair * 16u + 2u * hh));
#pragma unroll
        for (int l = 0; l < 4; ++l) lc[l] = __builtin_nontemporal_load(lp + 2 * l);
#pragma unroll
        for (int l = 0; l < 4; ++l) at[l] = __builtin_nontemporal_load(ap + 2 * l);
      }

      int ogx[4], ogy[4], cxm[4], rym[4];                    // window origin, last near column / row: scalars
      if (rnd == 0) {
        // ---- window placement: mean top-left corner of the in-range samples of the tile's level-0 queries, per level ----------
        if (wv < 4) {
          int tot[4][3];
          auto place = [&](auto ltag) __attribute__((always_inline)) {
            constexpr int LV = decltype(ltag)::value;
            const v2f fWH = {fW[LV], fH[LV]};
            float ax = 0.f, ay = 0.f;
            int an = 0;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              const v2f l2 = p ? v2f{lc[LV][2], lc[LV][3]} : v2f{lc[LV][0], lc[LV][1]};
              const v2f xy = __builtin_elementwise_fma(l2, fWH, v2f{-0.5f, -0.5f});
              const bool inr = (xy.y > -1.f) && (xy.x > -1.f) && (xy.y < fWH.y) && (xy.x < fWH.x);
              ax += inr ? floorf(xy.x) : 0.f;                 // small integers: float sums are exact
              ay += inr ? floorf(xy.y) : 0.f;
              an += __builtin_popcountll(__builtin_amdgcn_ballot_w64(inr));   // scalar
            }
            tot[LV][0] = (int)wave_total(ax);
            tot[LV][1] = (int)wave_total(ay);
            tot[LV][2] = an;
          };
          place(IC<0>{}); place(IC<1>{}); place(IC<2>{}); place(IC<3>{});
          if (lane == 63) {
#pragma unroll
            for (int l = 0; l < 4; ++l) *reinterpret_cast<int4*>(&mt.part[wv][l][0]) = make_int4(tot[l][0], tot[l][1], tot[l][2], 0);
          }
        }
        __syncthreads();
        {
          const int k = lane & 3;
          int4 sm = *reinterpret_cast<const int4*>(&mt.part[0][k][0]);
#pragma unroll
          for (int w = 1; w < 4; ++w) {
            const int4 t = *reinterpret_cast<const int4*>(&mt.part[w][k][0]);
            sm.x += t.x; sm.y += t.y; sm.z += t.z;
          }
          const int4 lv4 = *reinterpret_cast<const int4*>(&mt.lvl[k][0]);
          const int myH = lv4.x, myW = lv4.y, myWH = lv4.w, myWW = mt.lvl[k][4];
          int myOx = vxs - 3, myOy = vys - 3;
          if (sm.z > 0) {   // v_rcp_f32: every lane of the workgroup evaluates the same expression on the same sums
            const float inv = __builtin_amdgcn_rcpf((float)sm.z);
            myOx = (int)floorf((float)sm.x * inv + 0.5f) - (myWW - 2) / 2;
            myOy = (int)floorf((float)sm.y * inv + 0.5f) - (myWH - 2) / 2;
          }
          myOx = max(-1, min(myOx, myW + 1 - myWW));
          myOy = max(-1, min(myOy, myH + 1 - myWH));
          // a level smaller than its window: top-left corners past the last in-range one are not "near"
          const int cxmax = min(myOx + myWW - 2, myW - 1) - myOx, rymax = min(myOy + myWH - 2, myH - 1) - myOy;
          if (nrounds > 1 && tid < 4) *reinterpret_cast<int4*>(&mt.org[k][0]) = make_int4(myOx, myOy, cxmax, rymax);
#pragma unroll
          for (int l = 0; l < 4; ++l) {
            ogx[l] = __builtin_amdgcn_readlane(myOx, l);
            ogy[l] = __builtin_amdgcn_readlane(myOy, l);
            cxm[l] = __builtin_amdgcn_readlane(cxmax, l);
            rym[l] = __builtin_amdgcn_readlane(rymax, l);
          }
        }

        // ---- stage the four windows: LDS-DMA, one instruction = 8 consecutive window slots (1 KB) of ONE level.  The wave's
        // number is a compile-time constant of each copy below, so a chunk's window row / column / wrap position are constants and
        // its offset is (invariant per-lane part) + (scalar base of the chunk's row), plus one select where the chunk wraps into
        // the next window row -------------------------------------------------------------------------------------------------
        {
          // (nothing below may be hoisted out of the item loop: as loop invariants the per-chunk scalars do not fit the scalar
          // registers and come back as v_readlane of spilled SGPRs)
          int lane_ = lane;
          asm volatile("" : "+v"(lane_));
          uint32_t pixB_ = pixB, ldsb = smem_base;
          asm volatile("" : "+s"(pixB_), "+s"(ldsb));
          const uint32_t chunk = (uint32_t)(lane_ & 7) * 16u;
          const uint32_t vsub = (uint32_t)(lane_ >> 3);
          const uint32_t vlane = mad_u24(vsub, pixB_, chunk);
          auto stage_level = [&](auto wtag, auto ltag) __attribute__((always_inline)) {
            constexpr int WV = decltype(wtag)::value, LV = decltype(ltag)::value;
            constexpr int WW = kWW[LV], C0 = kBase[LV] / 8, C1 = kBase[LV + 1] / 8;
            constexpr int I0 = C0 + ((WV - C0) % kWaves + kWaves) % kWaves;   // this wave's first chunk of the level
            const int Hs = lvH[LV], xS = ogx[LV] + lvS[LV], oy = ogy[LV], ox = ogx[LV];
            int Ws = lvW[LV];
            asm volatile("" : "+s"(Ws));
            const uint32_t pixB = pixB_;
            if (Ws + 2 >= WW) {                                  // at most ONE window column outside the image on either side
              const int border = (int)((uint32_t)ox >> 31) | (int)((uint32_t)(Ws - ox - WW) >> 31);
#pragma unroll
              for (int i = I0; i < C1; i += kWaves) {
                const int rel0 = 8 * (i - C0);
                const int r0 = rel0 / WW, c0 = rel0 - r0 * WW;   // constants after unrolling
                const int thr = WW - c0;                          // lanes with sub >= thr sit in window row r0 + 1
                const int yA = oy + r0;
                const bool okA = (unsigned)yA < (unsigned)Hs, okB = (unsigned)(yA + 1) < (unsigned)Hs;
                const int pixA = yA * Ws + xS + c0;               // pixel of slot 0 of the chunk
                const uint32_t baseA = okA ? (uint32_t)pixA * pixB : kOobOffset;
                uint32_t off = vlane + baseA;                     // (kOobOffset + vlane stays out of range, vlane < 2^31)
                if (thr < 8) {
                  const uint32_t baseB = okB ? (uint32_t)(pixA + Ws - WW) * pixB : kOobOffset;
                  off += vsub >= (uint32_t)thr ? baseB - baseA : 0u;   // (mod 2^32: baseA + (baseB - baseA) = baseB)
                }
                if (border != 0) {                                // border tiles only: a real branch
                  asm volatile("; window column outside the image");
                  if (ox < 0) off = vsub == (uint32_t)(c0 == 0 ? 0 : thr) ? kOobOffset : off;
                  if (ox + WW > Ws) off = vsub == (uint32_t)(WW - 1 - c0) ? kOobOffset : off;   // (wrapped lanes never reach column WW - 1: WW >= 8)
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(vsrc, (__attribute__((address_space(3))) void*)(uintptr_t)(ldsb + (uint32_t)i * 1024u), 16,
                                                         off, hoff, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
              }
              return;
            }
            // (levels narrower than their window: per-lane row / column / inside-the-image arithmetic for every DMA instruction)
#pragma unroll
            for (int i = I0; i < C1; i += kWaves) {
              const int rel = 8 * (i - C0) + (int)vsub;           // slot of this lane in the level's window
              const int r = (int)(((float)rel + 0.5f) * (1.f / WW)), c = rel - r * WW;
              const int y = oy + r;
              const bool inside = (unsigned)y < (unsigned)Hs && (unsigned)(ox + c) < (unsigned)Ws;
              const uint32_t pix = mad_u24((uint32_t)y, (uint32_t)Ws, (uint32_t)(xS + c));
              const uint32_t in_off = mad_u24(pix, pixB, chunk);
              const uint32_t off = inside ? in_off : kOobOffset;
              __builtin_amdgcn_raw_ptr_buffer_load_lds(vsrc, (__attribute__((address_space(3))) void*)(uintptr_t)(ldsb + (uint32_t)i * 1024u), 16,
                                                       off, hoff, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
            }
          };
          auto stage_all = [&](auto wtag) __attribute__((always_inline)) {
            stage_level(wtag, IC<0>{}); stage_level(wtag, IC<1>{}); stage_level(wtag, IC<2>{}); stage_level(wtag, IC<3>{});
          };
#ifndef WINL_NODMA
          switch (wv) {
            case 0: stage_all(IC<0>{}); break;
            case 1: stage_all(IC<1>{}); break;
            case 2: stage_all(IC<2>{}); break;
            case 3: stage_all(IC<3>{}); break;
            case 4: stage_all(IC<4>{}); break;
            default: stage_all(IC<5>{}); break;
          }
#endif
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of the windows has landed
        __syncthreads();                                     // ... and everybody else's
      } else {
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          const int4 og = *reinterpret_cast<const int4*>(&mt.org[l][0]);
          ogx[l] = __builtin_amdgcn_readfirstlane(og.x); ogy[l] = __builtin_amdgcn_readfirstlane(og.y);
          cxm[l] = __builtin_amdgcn_readfirstlane(og.z); rym[l] = __builtin_amdgcn_readfirstlane(og.w);
        }
      }

      // ---- the pass: 8 samples x (2 corner rows x 2 pixels x 8 pieces) out of the LDS windows ---------------------------------
      // One prepared sample: the LDS addresses of the pixel read first / second in the top corner row (+ this lane's piece
      // offsets), the four corner weights with the attention weight folded in.  Dead and far samples carry zero weights and
      // point at the zero region.
      f32x4 acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      uint32_t farmask = 0;                                  // bit 7 - (2 * level + point of this lane)
      uint32_t pF, pS;                                       // LDS address of the pixel read first / second in the top corner row, | t16
      v2f wT, wB;                                            // (first, second) pixel x (top, bottom) row
      auto prep = [&](auto ltag, auto ptag) __attribute__((always_inline)) {
        constexpr int LV = decltype(ltag)::value, PT = decltype(ptag)::value;
        const v2f fWH = {fW[LV], fH[LV]};
        const v2f l2 = PT ? v2f{lc[LV][2], lc[LV][3]} : v2f{lc[LV][0], lc[LV][1]};
        const float a = at[LV][PT];
        // sample coordinates: the reference's arithmetic (cuh:282-288 and :38-46)
        const v2f xy = __builtin_elementwise_fma(l2, fWH, v2f{-0.5f, -0.5f});
        const v2f fl = {floorf(xy.x), floorf(xy.y)};
        const v2f fr = sub_clamp01(xy, fl);                   // (fx, fy)
        const bool inr = (xy.y > -1.f) && (xy.x > -1.f) && (xy.y < fWH.y) && (xy.x < fWH.x);
        const int cx = cvt_i32(fl.x) - ogx[LV], ry = cvt_i32(fl.y) - ogy[LV];
        const bool near = inr && (uint32_t)cx <= (uint32_t)cxm[LV] && (uint32_t)ry <= (uint32_t)rym[LV];
        farmask = shift_in(farmask, inr && !near);
        const uint32_t slot = mad_u24((uint32_t)ry, (uint32_t)kWW[LV], (uint32_t)cx);
        uint32_t tl = smem_base + (uint32_t)(kBase[LV] * 128) + (slot << 7);
        tl = near ? tl : zero_base;
        const uint32_t sw7 = (tl ^ E7) & 128u;                // 128: the right-hand pixel has this lane's first parity
        pF = (tl + sw7) | t16;                               // (pixels are 128-byte aligned: + piece offset == | == ^)
        pS = (tl + (sw7 ^ 128u)) | t16;
        const float an = near ? a : 0.f;                      // dead and far samples: all four weights 0
        const float wb = an * fr.y, wt = an - wb;             // bottom / top row x attention weight
        const float omx = 1.f - fr.x;
        const bool sw = sw7 != 0u;
        const v2f gx = {sw ? fr.x : omx, sw ? omx : fr.x};    // x factors of the (first, second) pixel
        wT = gx * wt;
        wB = gx * wb;
      };
      // In-place ring of kRing reads = kRing / 2 pieces of one corner PIXEL COLUMN (top and bottom row of one of the two x-adjacent
      // pixels): the two rows of a piece share their address register (the bottom row is an immediate offset), so an address lives
      // for two instructions; while a part of a column is consumed, the next part -- of the same pixel, of the sample's other
      // pixel, or of the next sample's first -- is requested piece by piece.
      constexpr int kHalf = WINL_RING / 2, kParts = 8 / kHalf;   // pieces per ring turn, turns per pixel column
      f32x4 R[2 * kHalf];
      auto rd = [&](uint32_t a, int off16) __attribute__((always_inline)) {
        return reinterpret_cast<lds4>((uintptr_t)a)[off16];
      };
      auto fma4 = [&](f32x4& c, float w, const f32x4& v) __attribute__((always_inline)) {
        const v2f W2 = {w, w};
        v2f lo = {c[0], c[1]}, hi = {c[2], c[3]};
        lo = __builtin_elementwise_fma(W2, v2f{v[0], v[1]}, lo);
        hi = __builtin_elementwise_fma(W2, v2f{v[2], v[3]}, hi);
        c = f32x4{lo.x, lo.y, hi.x, hi.y};
      };
      auto issue_part = [&](auto ltag, uint32_t px) __attribute__((always_inline)) {   // part 0 of column px
        constexpr int kRow = kWW[decltype(ltag)::value] * 8;   // one window row, in 16-byte units
#pragma unroll
        for (int i = 0; i < kHalf; ++i) {
          const uint32_t a = px ^ (16u * (uint32_t)i);
          R[2 * i] = rd(a, 0);
          R[2 * i + 1] = rd(a, kRow);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      // consume part `part` of the column in the ring (weights wt / wb for its top / bottom row); request part `npart` of column
      // `px` of level NL behind it
      auto step = [&](int part, float wt, float wb, auto nltag, uint32_t px, int npart, bool more) __attribute__((always_inline)) {
        constexpr int kRow = kWW[decltype(nltag)::value] * 8;
#pragma unroll
        for (int i = 0; i < kHalf; ++i) {
          const int j = part * kHalf + i;
          fma4(acc[j], wt, R[2 * i]);
          fma4(acc[j], wb, R[2 * i + 1]);
          asm volatile("" : "+v"(acc[j]));
          if (more) {
            const uint32_t a = px ^ (16u * (uint32_t)(npart * kHalf + i));
            R[2 * i] = rd(a, 0);
            R[2 * i + 1] = rd(a, kRow);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      // one sample: part 0 of its first pixel's column is in the ring; the rest of that column, then its second pixel's column are
      // requested while consuming; the next sample is prepared before the last part, which requests the next sample's first part
      auto sample = [&](auto ltag, auto ntag, auto nptag, bool more) __attribute__((always_inline)) {
        const v2f wTc = wT, wBc = wB;
        const uint32_t pFc = pF, pSc = pS;
#pragma unroll
        for (int q = 0; q < kParts; ++q) {                    // first pixel
          if (q + 1 < kParts) step(q, wTc.x, wBc.x, ltag, pFc, q + 1, true);
          else step(q, wTc.x, wBc.x, ltag, pSc, 0, true);
        }
#pragma unroll
        for (int q = 0; q + 1 < kParts; ++q) step(q, wTc.y, wBc.y, ltag, pSc, q + 1, true);   // second pixel but its last part
        if (more) prep(ntag, nptag);                          // overwrites pF / pS / wT / wB with the next sample's
        __builtin_amdgcn_sched_barrier(0);
        step(kParts - 1, wTc.y, wBc.y, ntag, pF, 0, more);
      };
#ifndef WINL_NOPASS
      prep(IC<0>{}, IC<0>{});
      issue_part(IC<0>{}, pF);
      sample(IC<0>{}, IC<0>{}, IC<1>{}, true); sample(IC<0>{}, IC<1>{}, IC<0>{}, true);
      sample(IC<1>{}, IC<1>{}, IC<1>{}, true); sample(IC<1>{}, IC<2>{}, IC<0>{}, true);
      sample(IC<2>{}, IC<2>{}, IC<1>{}, true); sample(IC<2>{}, IC<3>{}, IC<0>{}, true);
      sample(IC<3>{}, IC<3>{}, IC<1>{}, true); sample(IC<3>{}, IC<3>{}, IC<1>{}, false);
#endif

#ifdef WINL_NOFAR
      farmask = 0;
#endif
      // ---- far samples: raw buffer loads, one far sample per lane and step ----------------------------------------------------
      f32x4 G[16];
      while (__builtin_amdgcn_ballot_w64(farmask != 0u)) {
        const bool has = farmask != 0u;
        const int bit = has ? 31 - __builtin_clz(farmask) : 0;
        farmask &= ~(1u << bit);
        const int s8 = 7 - bit;                                // 2 * level + point of this lane
        const int s = 4 * (s8 >> 1) + 2 * (int)hh + (s8 & 1);  // 4 * level + point
        // the sample's location and weight come back from memory (8 samples of registers are not kept for a few per cent)
        float lx = -4.f, ly = -4.f, a = 0.f;
        int4 lv4 = make_int4(1, 1, 0, 0);
        if (has) {
          const msda::f32x2 l2 = *reinterpret_cast<const msda::f32x2*>(loc_img + (pair * 32u + 2u * (uint32_t)s));
          lx = l2[0]; ly = l2[1];
          a = attn_img[pair * 16u + (uint32_t)s];
          lv4 = *reinterpret_cast<const int4*>(&mt.lvl[s >> 2][0]);
        }
        const int fHi = lv4.x, fWi = lv4.y, fS = lv4.z;
        const float x = __builtin_fmaf(lx, (float)fWi, -0.5f), y = __builtin_fmaf(ly, (float)fHi, -0.5f);   // the pass's own expression
        const float xf = floorf(x), yf = floorf(y);
        const float lw = x - xf, lh = y - yf;
        const int fx0 = (int)xf, fy0 = (int)yf;                // in range by construction of the mask
        const bool t_ok = has && fy0 >= 0, b_ok = has && fy0 + 1 <= fHi - 1, l_ok = fx0 >= 0, r_ok = fx0 + 1 <= fWi - 1;
        const float wt = (1.f - lh) * a, wb = lh * a;
        const float w1 = wt * (1.f - lw), w2 = wt * lw, w3 = wb * (1.f - lw), w4 = wb * lw;
        // 24-bit multiply-adds on the CLAMPED top-left pixel: with fy0 or fx0 = -1 the live corners sit in row / column 0
        const int cy = max(fy0, 0), cx = max(fx0, 0);
        const uint32_t rowG = mad_u24((uint32_t)fWi, pixB, 0u);
        const uint32_t off = mad_u24(mad_u24((uint32_t)cy, (uint32_t)fWi, (uint32_t)(fS + cx)), pixB, 0u);
        const uint32_t dx = fx0 >= 0 ? pixB : 0u, dy = fy0 >= 0 ? rowG : 0u;   // step to the right / bottom neighbour
        const uint32_t o1 = (t_ok && l_ok) ? off : kOobOffset;
        const uint32_t o2 = (t_ok && r_ok) ? off + dx : kOobOffset;
        const uint32_t o3 = (b_ok && l_ok) ? off + dy : kOobOffset;
        const uint32_t o4 = (b_ok && r_ok) ? off + dy + dx : kOobOffset;
#pragma unroll
        for (int j = 0; j < 8; ++j) { G[j] = buffer_load_f32x4(vsrc, o1 + (t16 ^ (16u * (uint32_t)j)), hoff); G[8 + j] = buffer_load_f32x4(vsrc, o2 + (t16 ^ (16u * (uint32_t)j)), hoff); }
#pragma unroll
        for (int j = 0; j < 8; ++j) { fma4(acc[j], w1, G[j]); fma4(acc[j], w2, G[8 + j]); asm volatile("" : "+v"(acc[j])); }
        __builtin_amdgcn_sched_barrier(0);                     // the bottom corner row is requested behind the top row's FMAs: 16 loads in flight
#pragma unroll
        for (int j = 0; j < 8; ++j) { G[j] = buffer_load_f32x4(vsrc, o3 + (t16 ^ (16u * (uint32_t)j)), hoff); G[8 + j] = buffer_load_f32x4(vsrc, o4 + (t16 ^ (16u * (uint32_t)j)), hoff); }
#pragma unroll
        for (int j = 0; j < 8; ++j) { fma4(acc[j], w3, G[j]); fma4(acc[j], w4, G[8 + j]); }
      }

      // ---- the pair's two halves meet: register set j of this lane holds the piece j ^ t, set j ^ 1 of the partner lane (its t
      // differs in bit 0) the same piece.  Each lane finishes and stores the sets 0, 2, 4, 6: pieces p and p ^ 1 side by side,
      // 32 contiguous bytes per pair and instruction ------------------------------------------------------------------------
      f32x4 fin[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
        for (int c = 0; c < 4; ++c) fin[jj][c] = acc[2 * jj][c] + dppf<0xB1>(acc[2 * jj + 1][c]);
      }
      if (live) {
        char* op = reinterpret_cast<char*>(out_img) + (size_t)pair * 128u;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
#if defined(WINL_NOSTORE)
          if (fin[jj][0] == 12345.f) *reinterpret_cast<f32x4*>(op + (t16 ^ (32u * (uint32_t)jj))) = fin[jj];
#else
          __builtin_nontemporal_store(fin[jj], reinterpret_cast<f32x4*>(op + (t16 ^ (32u * (uint32_t)jj))));
#endif
        }
      }
    }
  }
}

__global__ void __launch_bounds__(kT, WINL_LB)
msda_fwd_winl(const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
              const float* __restrict__ loc, const float* __restrict__ attn, Dims d, float* __restrict__ out) {
  winl_body<0>(value, shapes, lsi, loc, attn, d, out);
}

bool winl_forward_ok(const Dims& d) { return win_forward_ok(d); }

int launch_forward_winl(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn,
                        const Dims& d, float* out, hipStream_t stream) {
  static std::atomic<uint64_t> lds_opted_in{0};
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  const void* fn = reinterpret_cast<const void*>(msda_fwd_winl);
  if (int rc = ensure_dynamic_lds(fn, kLdsBytes, lds_opted_in)) return rc;
  int K = d.N * ((d.S + 127) / 128);
  K = std::min(K, std::max(1, (2 * cus) / std::max(d.M, 1)));
  if (const int k = ab_env_int("MSDA_WINL_K", 0)) K = k;   // A/B: workgroups per head
  if (K < 1) K = 1;
  if (K > 65535) K = 65535;
  hipLaunchKernelGGL(msda_fwd_winl, dim3((unsigned)d.M, (unsigned)K), dim3(kT), kLdsBytes, stream, value, shapes, lsi, loc,
                     attn, d, out);
  return (int)hipGetLastError();
}

}  // namespace msda
