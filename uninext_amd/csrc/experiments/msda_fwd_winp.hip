// msda_fwd_winp -- MSDeformAttn forward for encoder-style calls (Lq == S): LDS windows on all four pyramid levels, ONE
// persistent 12-wave workgroup per CU with TWO window sets, the start-up of the next work item running under the gather of the
// current one.  fp32, D = 32, L = P = 4.  gfx950 only.  Replaces, for these calls, the work of
// ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299.
//
// What round 6 measured first (tools/micro/lds_valu_overlap.cpp, profiles/r06_forward_formulations.txt): a SIMD returns one
// ds_read_b128 per 16 clocks and hides 2 v_pk_fma_f32 (8 clocks) under it from two waves on -- the gather proper is LDS-bound at
// 2.9 us per work item (26 us per launch) as long as >= 2 waves per SIMD are in it and the vector work per read stays under
// 16 clocks.  msda_fwd_win loses the rest of its 69 us outside that: vector work per read AT the 16 clocks (DPP broadcasts,
// per-lane level constants), and an item's start-up (locations -> placement -> barrier -> window DMA -> barrier) that the other
// workgroup of the CU only hides when the two happen to be out of phase.  Hence:
//
//   lanes          = a QUAD per (query, head) pair, split by POINTS: lane p samples point p of every level into all 32 channels
//                    (8 register sets of one 16-byte piece each).  Every lane of the wave walks level 0, 1, 2, 3 together: the
//                    level's size, window origin and limits are SCALARS, the bottom corner row is an immediate offset, nothing
//                    is broadcast; the quad's four partial sums meet at the end in 24 DPP adds.  Per sample and lane: 64 packed
//                    FMAs + 16 v_xor (addresses) + ~33 of preparation = 7 clocks of vector work per read.
//   LDS banks      = lane class (e, t) = (bit 3, bits 0-2 of the lane id; 16 different classes in each 16-lane service group of
//                    ds_read_b128): of the two x-adjacent corner pixels the lane reads the one whose window slot has parity e
//                    first, and at its j-th read the 16-byte piece j ^ t -- every instruction covers all 64 banks exactly
//                    once for ANY sample positions.  Register set j accumulates piece j ^ t.
//   work item      = (image, head, 8 x 16 tile of level-0 pixels + the pixels of levels 1..3 whose centres fall into the tile's
//                    rectangle): msda_fwd_win's partition, windows (12x20 / 10x14 / 10x12 / 10x10 pixels, 76 KB) and placement
//                    rule.  Waves 0..7 = the tile's rows (16 pairs each), waves 8..11 = up to 64 queries of levels 1..3.
//   pipeline       = iteration i of the persistent workgroup:   wait for the locations of item i + 1 and the windows of item i
//                    (both requested one iteration ago) -> placement sums of item i + 1 -> THE barrier of the iteration ->
//                    origins of item i + 1, its window DMA into the other set, the location loads of item i + 2 (all
//                    asynchronous) -> gather of item i -> stores.  No wave ever waits for memory it has just asked for.
//   far            = an in-range sample with a corner outside its window: flagged during the gather, worked off behind it with
//                    raw buffer loads, one far sample per quad and step (each lane takes its two final pieces).  Correctness never
//                    depends on where the windows are; only speed does.
//
// All geometry comes from the int64 shape tensors on the device; the host only knows S.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "../msda_common.hpp"

#ifndef WINP_RING
#define WINP_RING 4     // LDS reads in flight per lane during the gather (4, 8 or 16; 8 spills at 168 registers)
#endif

namespace msda {
namespace {

constexpr int kT = 768, kWaves = kT / 64, kQuads = kT / 4;      // 192 pairs per round
constexpr int kTH = 8, kTW = 16;                                // level-0 pairs of an item: waves 0..7, 16 pairs each
constexpr int kWH[4] = {12, 10, 10, 10};
constexpr int kWW[4] = {20, 14, 12, 10};                        // even: slot parity == column parity in every row
constexpr int kBase[5] = {0, 240, 384, 504, 608};               // first slot of each window, multiples of 8: a 1 KB
                                                                // DMA chunk (8 slots) never straddles two levels
static_assert(kBase[1] >= kWH[0] * kWW[0] && kBase[2] >= kBase[1] + kWH[1] * kWW[1] &&
              kBase[3] >= kBase[2] + kWH[2] * kWW[2] && kBase[4] >= kBase[3] + kWH[3] * kWW[3], "window table");
static_assert(kBase[1] % 8 == 0 && kBase[2] % 8 == 0 && kBase[3] % 8 == 0 && kBase[4] % 8 == 0, "DMA chunks / parity");
constexpr int kSlots = kBase[4];
constexpr int kSetBytes = kSlots * 128;                         // one window set
constexpr int kZeroOff = 2 * kSetBytes;                         // all-zero region: target of dead / far samples
constexpr int kZeroBytes = kWW[0] * 128 + 256;                  // a bottom-row read lands at most one level-0 row further
static_assert(kSetBytes % 256 == 0, "slot parity by address bit 7 in both sets and the zero region");
struct Meta {
  int part[2][8][4][4];                                         // [item parity][level-0 wave] per level: sum x0, sum y0, count, -
  int org[2][4][4];                                             // [item parity] per level: window origin x, y; last near column / row
  int lvl[4][8];                                                // per level: H, W, first pixel, window rows, window columns
  int stat[4];
};
constexpr int kMetaOff = kZeroOff + kZeroBytes;
constexpr int kLdsBytes = kMetaOff + ((sizeof(Meta) + 15) / 16) * 16;
static_assert(kLdsBytes <= 160 * 1024, "one workgroup per CU");

typedef const f32x4 __attribute__((address_space(3)))* lds4;
typedef float v2f __attribute__((ext_vector_type(2)));        // packed fp32 math: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32

template <int CTRL, int ROWS = 0xF>
__device__ __forceinline__ float dppf(float v) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, ROWS, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ uint32_t dppu(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }
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

template <int I> using IC = std::integral_constant<int, I>;

// a lane's share of one work item: its query, and the locations / weights of its point on the four levels
struct Lane {
  msda::f32x2 lc[4];
  float at[4];
  uint32_t pair;                                            // (query, head) pair within the image, 0 for a dead lane
  uint32_t live;
};

}  // namespace

__global__ void __launch_bounds__(kT, 3)
msda_fwd_winp(const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
              const float* __restrict__ loc, const float* __restrict__ attn, Dims d, float* __restrict__ out) {
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
  // ---- this lane's point and bank class: parity read first, piece rotation -------------------------------------------------
  const uint32_t pt = (uint32_t)lane & 3u;                  // this lane samples point pt of every level
  uint32_t t16 = 16u * (uint32_t)(lane & 7);                // piece of this lane's j-th read: byte offset t16 ^ 16 j
  asm volatile("" : "+v"(t16));
  uint32_t E7 = ((uint32_t)(lane >> 3) & 1u) << 7;
  asm volatile("" : "+v"(E7));
  const uint32_t zero_base = smem_base + kZeroOff;
  float fW[4], fH[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) { fW[l] = (float)lvW[l]; fH[l] = (float)lvH[l]; }

  // per-image bases of an item (uniform: per-lane offsets stay 32-bit, S * M * 128 < 2^31)
  auto image_of = [&](int item) __attribute__((always_inline)) { return (int)(((float)item + 0.5f) * __builtin_amdgcn_rcpf((float)ntiles)); };

  // ---- tile geometry of an item: lane k (of every quad) works out level k's query rectangle; readlane makes the level-0
  // rectangle and the query counts scalar, the rectangles of levels 1..3 stay in lanes 1..3 (the queries of those levels fetch
  // theirs with ds_bpermute) ----------------------------------------------------------------------------------------------
  struct Geo { int vxs, vys, vnx, vW, vS; int xs0, ys0, nx0, ny0, e1, e2, nrest; };
  auto geometry = [&](int item, int b) __attribute__((always_inline)) {
    Geo g;
    const int kq = lane & 3;
    const int4 lv4 = *reinterpret_cast<const int4*>(&mt.lvl[kq][0]);
    const int gW = lv4.y, gH = lv4.x;
    const float fxs = (float)(kTW * gW) * __builtin_amdgcn_rcpf((float)lvW[0]), fys = (float)(kTH * gH) * __builtin_amdgcn_rcpf((float)lvH[0]);
    const int tile_ = item - b * ntiles;
    const int ty = (int)(((float)tile_ + 0.5f) * __builtin_amdgcn_rcpf((float)TX)), tx = tile_ - ty * TX;
    const int xs = min(max((int)ceilf((float)tx * fxs - 0.5f), 0), gW);
    const int xe = tx == TX - 1 ? gW : min(max((int)ceilf((float)(tx + 1) * fxs - 0.5f), xs), gW);
    const int ys = min(max((int)ceilf((float)ty * fys - 0.5f), 0), gH);
    const int ye = ty == TY - 1 ? gH : min(max((int)ceilf((float)(ty + 1) * fys - 0.5f), ys), gH);
    g.vxs = xs; g.vys = ys; g.vnx = xe - xs; g.vW = gW; g.vS = lv4.z;
    const int cnt = (xe - xs) * (ye - ys);
    g.xs0 = __builtin_amdgcn_readlane(xs, 0); g.ys0 = __builtin_amdgcn_readlane(ys, 0);
    g.nx0 = __builtin_amdgcn_readlane(xe - xs, 0); g.ny0 = __builtin_amdgcn_readlane(ye - ys, 0);
    g.e1 = __builtin_amdgcn_readlane(cnt, 1);
    g.e2 = g.e1 + __builtin_amdgcn_readlane(cnt, 2);
    g.nrest = g.e2 + __builtin_amdgcn_readlane(cnt, 3);
    return g;
  };
  // the lane's query in round `rnd` of an item, and the loads of its locations / weights (asynchronous: nothing waits here)
  auto fetch = [&](const Geo& g, int b, int rnd, Lane& ln) __attribute__((always_inline)) {
    bool live;
    uint32_t qidx;
    if (rnd == 0 && wv < 8) {                                // wave-uniform: the tile's row wv
      const int col = lane >> 2;
      live = col < g.nx0 && wv < g.ny0;
      qidx = (uint32_t)(lvS[0] + (g.ys0 + wv) * lvW[0] + g.xs0 + col);
    } else {
      const int ri = (rnd == 0 ? (wv - 8) * 16 : 64 + (rnd - 1) * kQuads + wv * 16) + (lane >> 2);
      live = ri < g.nrest;
      const bool c1 = ri >= g.e1, c2 = ri >= g.e2;
      const int ql = 1 + (c1 ? 1 : 0) + (c2 ? 1 : 0);
      const int j = ri - (c2 ? g.e2 : c1 ? g.e1 : 0);
      const int src = ((lane & ~3) | ql) << 2;               // lane ql of this quad holds level ql's rectangle
      const int nx = __builtin_amdgcn_ds_bpermute(src, g.vnx), xs = __builtin_amdgcn_ds_bpermute(src, g.vxs);
      const int ys = __builtin_amdgcn_ds_bpermute(src, g.vys), Wq = __builtin_amdgcn_ds_bpermute(src, g.vW);
      const int Sq = __builtin_amdgcn_ds_bpermute(src, g.vS);
      const int yy = (int)(((float)j + 0.5f) * __builtin_amdgcn_rcpf((float)max(nx, 1)));
      qidx = (uint32_t)(Sq + (ys + yy) * Wq + xs + (j - yy * nx));
    }
    live = live && qidx < (uint32_t)d.Lq;                   // (shapes whose pixel count exceeds num_query: never outside the tensors)
    if (!live) qidx = 0u;
    ln.live = live ? 1u : 0u;
    ln.pair = mad_u24(qidx, (uint32_t)M, (uint32_t)m);
    // a dead lane's locations are far outside every level: all its samples are out of range without a `live &&`
#pragma unroll
    for (int l = 0; l < 4; ++l) { ln.lc[l] = msda::f32x2{-4.f, -4.f}; ln.at[l] = 0.f; }
#ifdef WINP_NOLOADS
    if (live) {
      const float qx = ((float)(qidx % 167u) + 0.5f) * (1.f / 167.f), qy = ((float)((qidx / 167u) % 100u) + 0.5f) * 0.01f;
      for (int l = 0; l < 4; ++l) { ln.lc[l] = msda::f32x2{qx + 0.004f * l, qy + 0.003f * l}; ln.at[l] = 0.0625f; }
    }
    if (false) {
#else
    if (live) {
#endif
      const int64_t pair_img = (int64_t)b * d.Lq * M;
      const msda::f32x2* lp = reinterpret_cast<const msda::f32x2*>(loc + pair_img * 32 + (ln.pair * 32u + 2u * pt));
      const float* ap = attn + pair_img * 16 + (ln.pair * 16u + pt);
#pragma unroll
#ifdef WINP_LOADS_NT
      for (int l = 0; l < 4; ++l) ln.lc[l] = __builtin_nontemporal_load(lp + 4 * l);
#pragma unroll
      for (int l = 0; l < 4; ++l) ln.at[l] = __builtin_nontemporal_load(ap + 4 * l);
#else
      for (int l = 0; l < 4; ++l) ln.lc[l] = lp[4 * l];
#pragma unroll
      for (int l = 0; l < 4; ++l) ln.at[l] = ap[4 * l];
#endif
    }
  };
  // placement sums of an item: the in-range top-left corners of the samples of its level-0 queries, per level; one int4 per
  // (wave, level) into mt.part[parity] (waves 0..7 only)
  auto place = [&](const Lane& ln, int parity) __attribute__((always_inline)) {
    int tot[4][3];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const v2f fWH = {fW[l], fH[l]};
      const v2f xy = __builtin_elementwise_fma(v2f{ln.lc[l][0], ln.lc[l][1]}, fWH, v2f{-0.5f, -0.5f});
      const bool inr = (xy.y > -1.f) && (xy.x > -1.f) && (xy.y < fWH.y) && (xy.x < fWH.x);
      tot[l][0] = (int)wave_total(inr ? floorf(xy.x) : 0.f);   // small integers: float sums are exact
      tot[l][1] = (int)wave_total(inr ? floorf(xy.y) : 0.f);
      tot[l][2] = __builtin_popcountll(__builtin_amdgcn_ballot_w64(inr));
    }
    if (lane == 63) {
#pragma unroll
      for (int l = 0; l < 4; ++l) *reinterpret_cast<int4*>(&mt.part[parity][wv][l][0]) = make_int4(tot[l][0], tot[l][1], tot[l][2], 0);
    }
  };
  // window origins of an item from its placement sums (after the barrier): lane k works out level k; returned per lane
  auto origins = [&](const Geo& g, int parity, int& myOx, int& myOy) __attribute__((always_inline)) {
    const int k = lane & 3;
    int4 sm = *reinterpret_cast<const int4*>(&mt.part[parity][0][k][0]);
#pragma unroll
    for (int w = 1; w < 8; ++w) {
      const int4 t = *reinterpret_cast<const int4*>(&mt.part[parity][w][k][0]);
      sm.x += t.x; sm.y += t.y; sm.z += t.z;
    }
    const int4 lv4 = *reinterpret_cast<const int4*>(&mt.lvl[k][0]);
    const int myH = lv4.x, myW = lv4.y, myWH = lv4.w, myWW = mt.lvl[k][4];
    myOx = g.vxs - 3; myOy = g.vys - 3;
    if (sm.z > 0) {   // v_rcp_f32: every lane of the workgroup evaluates the same expression on the same sums
      const float inv = __builtin_amdgcn_rcpf((float)sm.z);
      myOx = (int)floorf((float)sm.x * inv + 0.5f) - (myWW - 2) / 2;
      myOy = (int)floorf((float)sm.y * inv + 0.5f) - (myWH - 2) / 2;
    }
    myOx = max(-1, min(myOx, myW + 1 - myWW));
    myOy = max(-1, min(myOy, myH + 1 - myWH));
    // a level smaller than its window: top-left corners past the last in-range one are not "near"
    const int cxmax = min(myOx + myWW - 2, myW - 1) - myOx, rymax = min(myOy + myWH - 2, myH - 1) - myOy;
    if (tid < 4) *reinterpret_cast<int4*>(&mt.org[parity][k][0]) = make_int4(myOx, myOy, cxmax, rymax);   // read at the item's gather, a barrier later
  };
  // ---- stage an item's four windows into set `parity`: LDS-DMA, one instruction = 8 consecutive window slots (1 KB) of ONE
  // level.  The wave's number is a compile-time constant of each copy, so a chunk's window row / column / wrap position are
  // constants and its offset is (invariant per-lane part) + (scalar base of the chunk's row), plus one select where the chunk
  // wraps into the next window row ---------------------------------------------------------------------------------------
  auto stage = [&](int b, int parity, int myOx, int myOy) __attribute__((always_inline)) {
    int ogx[4], ogy[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) { ogx[l] = __builtin_amdgcn_readlane(myOx, l); ogy[l] = __builtin_amdgcn_readlane(myOy, l); }
    const __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(value) + (int64_t)b * d.S * M * 32, 0, (int)((uint32_t)d.S * pixB), 0x00020000);
    // (nothing below may be hoisted out of the item loop: as loop invariants the per-chunk scalars do not fit the scalar
    // registers and come back as v_readlane of spilled SGPRs)
    int lane_ = lane;
    asm volatile("" : "+v"(lane_));
    uint32_t pixB_ = pixB, ldsb = smem_base + (uint32_t)parity * (uint32_t)kSetBytes;
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
#ifndef WINP_NODMA
    switch (wv) {
      case 0: stage_all(IC<0>{}); break;
      case 1: stage_all(IC<1>{}); break;
      case 2: stage_all(IC<2>{}); break;
      case 3: stage_all(IC<3>{}); break;
      case 4: stage_all(IC<4>{}); break;
      case 5: stage_all(IC<5>{}); break;
      case 6: stage_all(IC<6>{}); break;
      case 7: stage_all(IC<7>{}); break;
      case 8: stage_all(IC<8>{}); break;
      case 9: stage_all(IC<9>{}); break;
      case 10: stage_all(IC<10>{}); break;
      default: stage_all(IC<11>{}); break;
    }
#endif
  };

  // ---- the gather of one round of an item (lane data `ln`) out of window set `parity`, far samples, the quad's sum, the
  // stores -------------------------------------------------------------------------------------------------------------------
  auto gather = [&](const Lane& ln, int b, int parity) __attribute__((always_inline)) {
    int ogx[4], ogy[4], cxm[4], rym[4];                      // window origin, last near column / row: scalars
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int4 og = *reinterpret_cast<const int4*>(&mt.org[parity][l][0]);
      ogx[l] = __builtin_amdgcn_readfirstlane(og.x); ogy[l] = __builtin_amdgcn_readfirstlane(og.y);
      cxm[l] = __builtin_amdgcn_readfirstlane(og.z); rym[l] = __builtin_amdgcn_readfirstlane(og.w);
    }
    const uint32_t set_base = smem_base + (uint32_t)parity * (uint32_t)kSetBytes;
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint32_t farmask = 0;                                    // bit l: this lane's sample on level l is far
    uint32_t pF, pS;                                         // LDS address of the pixel read first / second in the top corner row, | t16
    v2f wT, wB;                                              // (first, second) pixel x (top, bottom) row
    // One prepared sample: dead and far samples carry zero weights and point at the zero region.
    auto prep = [&](auto ltag) __attribute__((always_inline)) {
      constexpr int LV = decltype(ltag)::value;
      const v2f fWH = {fW[LV], fH[LV]};
      const float a = ln.at[LV];
      // sample coordinates: the reference's arithmetic (cuh:282-288 and :38-46)
      const v2f xy = __builtin_elementwise_fma(v2f{ln.lc[LV][0], ln.lc[LV][1]}, fWH, v2f{-0.5f, -0.5f});
      const v2f fl = {floorf(xy.x), floorf(xy.y)};
      const v2f fr = sub_clamp01(xy, fl);                   // (fx, fy)
      const bool inr = (xy.y > -1.f) && (xy.x > -1.f) && (xy.y < fWH.y) && (xy.x < fWH.x);
      const int cx = cvt_i32(fl.x) - ogx[LV], ry = cvt_i32(fl.y) - ogy[LV];
      const bool near = inr && (uint32_t)cx <= (uint32_t)cxm[LV] && (uint32_t)ry <= (uint32_t)rym[LV];
      farmask |= (inr && !near) ? (1u << LV) : 0u;
      const uint32_t slot = mad_u24((uint32_t)ry, (uint32_t)kWW[LV], (uint32_t)cx);
      uint32_t tl = set_base + (uint32_t)(kBase[LV] * 128) + (slot << 7);
      tl = near ? tl : zero_base;
      const uint32_t sw7 = (tl ^ E7) & 128u;                // 128: the right-hand pixel has this lane's first parity
      pF = (tl + sw7) | t16;                                // (pixels are 128-byte aligned: + piece offset == | == ^)
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
    // pixels): the two rows of a piece share their address register (the bottom row is an immediate offset); while a part of a
    // column is consumed, the next part -- of the same pixel, of the sample's other pixel, or of the next sample's first -- is
    // requested piece by piece.
    constexpr int kHalf = WINP_RING / 2, kParts = 8 / kHalf;   // pieces per ring turn, turns per pixel column
    f32x4 R[2 * kHalf];
    auto rd = [&](uint32_t a, int off16) __attribute__((always_inline)) { return reinterpret_cast<lds4>((uintptr_t)a)[off16]; };
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
    auto sample = [&](auto ltag, auto ntag, bool more) __attribute__((always_inline)) {
      const v2f wTc = wT, wBc = wB;
      const uint32_t pFc = pF, pSc = pS;
#pragma unroll
      for (int q = 0; q < kParts; ++q) {                    // first pixel
        if (q + 1 < kParts) step(q, wTc.x, wBc.x, ltag, pFc, q + 1, true);
        else step(q, wTc.x, wBc.x, ltag, pSc, 0, true);
      }
#pragma unroll
      for (int q = 0; q + 1 < kParts; ++q) step(q, wTc.y, wBc.y, ltag, pSc, q + 1, true);   // second pixel but its last part
      if (more) prep(ntag);                                 // overwrites pF / pS / wT / wB with the next sample's
      __builtin_amdgcn_sched_barrier(0);
      step(kParts - 1, wTc.y, wBc.y, ntag, pF, 0, more);
    };
#ifndef WINP_NOPASS
    prep(IC<0>{});
    issue_part(IC<0>{}, pF);
    sample(IC<0>{}, IC<1>{}, true);
    sample(IC<1>{}, IC<2>{}, true);
    sample(IC<2>{}, IC<3>{}, true);
    sample(IC<3>{}, IC<3>{}, false);
#endif

    // ---- the quad's four partial sums meet: register set j of this lane holds piece j ^ t; the lane whose number differs in bit
    // 0 / bit 1 holds the same piece in set j ^ 1 / j ^ 2.  Each lane finishes the sets 0 and 4: the quad stores 2 x 64 contiguous
    // bytes ---------------------------------------------------------------------------------------------------------------
    f32x4 fin[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float s0 = acc[4 * h][c] + dppf<0xB1>(acc[4 * h + 1][c]);         // quad_perm [1,0,3,2]
        const float s2 = acc[4 * h + 2][c] + dppf<0xB1>(acc[4 * h + 3][c]);
        fin[h][c] = s0 + dppf<0x4E>(s2);                                          // quad_perm [2,3,0,1]
      }
    }

#ifdef WINP_NOFAR
    farmask = 0;
#endif
    // ---- far samples: raw buffer loads, one far sample per quad and step; every lane takes its two final pieces -------------
    {
      uint32_t fm = farmask << (4u * pt);                    // the pair's 16 samples: bit 4 * point + level
      fm |= dppu<0xB1>(fm);
      fm |= dppu<0x4E>(fm);
      if (__builtin_amdgcn_ballot_w64(fm != 0u)) {
        const __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(value) + (int64_t)b * d.S * M * 32, 0, (int)((uint32_t)d.S * pixB), 0x00020000);
        do {
          const bool has = fm != 0u;
          const int idx = has ? __builtin_ctz(fm) : 0;
          fm &= fm - 1u;
          const int fl_ = idx & 3, ps = idx >> 2;            // level and point (= owning lane of the quad) of the far sample
          const int src = ((lane & ~3) | ps) << 2;           // byte address of the owning lane for ds_bpermute
          const bool c1 = (fl_ & 1) != 0, c2 = (fl_ & 2) != 0;
          auto sel4 = [&](float a0, float a1, float a2, float a3) __attribute__((always_inline)) {
            const float t0 = c1 ? a1 : a0, t1 = c1 ? a3 : a2;
            return (int)__float_as_uint(c2 ? t1 : t0);
          };
          // every lane selects its own level-fl_ candidate, the quad pulls the owner's; stand-ins of idle quads must be finite
          const uint32_t hm = has ? 0xffffffffu : 0u;
          const float lx = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, sel4(ln.lc[0][0], ln.lc[1][0], ln.lc[2][0], ln.lc[3][0])) & hm);
          const float ly = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, sel4(ln.lc[0][1], ln.lc[1][1], ln.lc[2][1], ln.lc[3][1])) & hm);
          const float fa = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, sel4(ln.at[0], ln.at[1], ln.at[2], ln.at[3])) & hm);
          const int4 lv4 = *reinterpret_cast<const int4*>(&mt.lvl[fl_][0]);
          const int fHi = lv4.x, fWi = lv4.y, fS = lv4.z;
          const float x = __builtin_fmaf(lx, (float)fWi, -0.5f), y = __builtin_fmaf(ly, (float)fHi, -0.5f);   // the gather's own expression
          const float xf = floorf(x), yf = floorf(y);
          const float lw = x - xf, lh = y - yf;
          const int fx0 = (int)xf, fy0 = (int)yf;            // in range by construction of the mask; 0 for the stand-ins
          const bool t_ok = has && fy0 >= 0, b_ok = has && fy0 + 1 <= fHi - 1, l_ok = fx0 >= 0, r_ok = fx0 + 1 <= fWi - 1;
          const float wt = (1.f - lh) * fa, wb = lh * fa;
          const float w1 = wt * (1.f - lw), w2 = wt * lw, w3 = wb * (1.f - lw), w4 = wb * lw;
          // 24-bit multiply-adds on the CLAMPED top-left pixel: with fy0 or fx0 = -1 the live corners sit in row / column 0
          const int cy = max(fy0, 0), cx = max(fx0, 0);
          const uint32_t rowG = mad_u24((uint32_t)fWi, pixB, 0u);
          const uint32_t off = mad_u24(mad_u24((uint32_t)cy, (uint32_t)fWi, (uint32_t)(fS + cx)), pixB, t16);   // this lane's piece of set 0
          const uint32_t dx = fx0 >= 0 ? pixB : 0u, dy = fy0 >= 0 ? rowG : 0u;   // step to the right / bottom neighbour
          const uint32_t o1 = (t_ok && l_ok) ? off : kOobOffset;
          const uint32_t o2 = (t_ok && r_ok) ? off + dx : kOobOffset;
          const uint32_t o3 = (b_ok && l_ok) ? off + dy : kOobOffset;
          const uint32_t o4 = (b_ok && r_ok) ? off + dy + dx : kOobOffset;
          const f32x4 d1a = buffer_load_f32x4(vsrc, o1, hoff), d1b = buffer_load_f32x4(vsrc, o1 ^ 64u, hoff);
          const f32x4 d2a = buffer_load_f32x4(vsrc, o2, hoff), d2b = buffer_load_f32x4(vsrc, o2 ^ 64u, hoff);
          const f32x4 d3a = buffer_load_f32x4(vsrc, o3, hoff), d3b = buffer_load_f32x4(vsrc, o3 ^ 64u, hoff);
          const f32x4 d4a = buffer_load_f32x4(vsrc, o4, hoff), d4b = buffer_load_f32x4(vsrc, o4 ^ 64u, hoff);
          fma4(fin[0], w1, d1a); fma4(fin[1], w1, d1b);
          fma4(fin[0], w2, d2a); fma4(fin[1], w2, d2b);
          fma4(fin[0], w3, d3a); fma4(fin[1], w3, d3b);
          fma4(fin[0], w4, d4a); fma4(fin[1], w4, d4b);
        } while (__builtin_amdgcn_ballot_w64(fm != 0u));
      }
    }

    if (ln.live) {
      char* op = reinterpret_cast<char*>(out + (int64_t)b * d.Lq * M * 32) + (size_t)ln.pair * 128u;
#ifdef WINP_NOSTORE
      if (fin[0][0] == 12345.f) {
#endif
      __builtin_nontemporal_store(fin[0], reinterpret_cast<f32x4*>(op + t16));
      __builtin_nontemporal_store(fin[1], reinterpret_cast<f32x4*>(op + (t16 ^ 64u)));
#ifdef WINP_NOSTORE
      }
#endif
    }
  };

  // =========================================================================================================================
  // the pipeline.  cur / nxt / nx2 = lane data of the items i, i + 1, i + 2 of this workgroup
  const int first = kk;
  Lane cur, nxt, nx2;
  Geo gn, g2;                                               // geometry of items i + 1, i + 2
  int bn, b2;
  // fill: item `first` goes through loads -> placement -> barrier -> origins -> DMA with nothing to hide behind
  {
    const int b0 = image_of(first);
    const Geo g0 = geometry(first, b0);
    fetch(g0, b0, 0, cur);
    if (wv < 8) place(cur, 0);
    __syncthreads();
    int ox, oy;
    origins(g0, 0, ox, oy);
    stage(b0, 0, ox, oy);
    gn = g0; bn = b0;                                       // (placeholders when there is no second item)
    nxt = cur;
    if (first + K < nitems) {
      bn = image_of(first + K);
      gn = geometry(first + K, bn);
      fetch(gn, bn, 0, nxt);
    }
    // the geometry of item `first` is needed once more for its later rounds (odd pyramids): kept in g2 / b2 until the loop rotates
    g2 = g0; b2 = b0;
  }
  Geo gc = g2;                                              // geometry of the CURRENT item (later rounds only)
  int bc = b2;
  for (int item = first, it = 0; item < nitems; item += K, ++it) {
    const int par = it & 1;
    const bool has1 = item + K < nitems, has2 = item + 2 * K < nitems;
    // a. placement sums of item i + 1 (its locations were requested one iteration ago); the windows of item i have landed
    if (has1 && wv < 8) place(nxt, par ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's share of item i's windows (and everything older)
    __syncthreads();                                         // THE barrier of the iteration
    // b. origins + window DMA of item i + 1 into the other set (whose last reader, the gather of item i - 1, is behind the barrier)
    if (has1) {
      int ox, oy;
      origins(gn, par ^ 1, ox, oy);
      stage(bn, par ^ 1, ox, oy);
    }
    // c. the locations of item i + 2 start travelling
    nx2 = nxt; g2 = gn; b2 = bn;
    if (has2) {
      b2 = image_of(item + 2 * K);
      g2 = geometry(item + 2 * K, b2);
      fetch(g2, b2, 0, nx2);
    }
    // e. gather + stores of item i
    const bool busy = wv < 8 || (wv - 8) * 16 < gc.nrest;   // a wave without a query skips the gather
    if (busy) gather(cur, bc, par);
    // (odd pyramids: more than 64 queries of levels 1..3 in a tile -- further rounds on the same windows, not pipelined)
    if (gc.nrest > 64) {
      const int nrounds = 1 + (gc.nrest - 64 + kQuads - 1) / kQuads;
      for (int rnd = 1; rnd < nrounds; ++rnd) {
        if (64 + (rnd - 1) * kQuads + wv * 16 >= gc.nrest) continue;
        Lane ex;
        fetch(gc, bc, rnd, ex);
        gather(ex, bc, par);
      }
    }
    cur = nxt; nxt = nx2; gc = gn; bc = bn; gn = g2; bn = b2;
  }
}

bool winp_forward_ok(const Dims& d) { return win_forward_ok(d); }

int launch_forward_winp(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn,
                        const Dims& d, float* out, hipStream_t stream) {
  static std::atomic<uint64_t> lds_opted_in{0};
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  const void* fn = reinterpret_cast<const void*>(msda_fwd_winp);
  if (int rc = ensure_dynamic_lds(fn, kLdsBytes, lds_opted_in)) return rc;
  int K = d.N * ((d.S + 127) / 128);
  K = std::min(K, std::max(1, cus / std::max(d.M, 1)));      // one workgroup per CU
  if (const int k = ab_env_int("MSDA_WINP_K", 0)) K = k;   // A/B: workgroups per head
  if (K < 1) K = 1;
  if (K > 65535) K = 65535;
  hipLaunchKernelGGL(msda_fwd_winp, dim3((unsigned)d.M, (unsigned)K), dim3(kT), kLdsBytes, stream, value, shapes, lsi, loc,
                     attn, d, out);
  return (int)hipGetLastError();
}

}  // namespace msda
