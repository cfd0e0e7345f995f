// msda_fwd_win -- MSDeformAttn forward for encoder-style calls (Lq == S) with the gather served from LDS windows
// on ALL four pyramid levels.  fp32, D = 32, L = P = 4.  gfx950 only.  Replaces, for these calls, the work of
// ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299.
//
// Why: the gather moves 8 KB per (query, head) pair for 128 B of output; through the vector L1 (msda_fwd_lg3) the
// launch is bound by the texture path at ~2 clocks per 128-byte line (profiles/r01_ablation.txt: 92 us warm with only
// the coarsest level in LDS, 24 us without any gather).  The LDS delivers 256 B/clk/CU -- 4x the L1 -- but only for
// data that is on the CU, so the work is cut into tiles whose samples land in small windows:
//
//   work item  = (image b, head m, 8 x 16 tile of level-0 pixels).  Its queries are the pixels of EVERY level whose
//                centre falls into the tile's normalised rectangle (an exact partition of the S queries in integer
//                arithmetic): 128 + 32 + 8 + 2 = 170 queries at the R50 shapes.
//   windows    = per level a WH x WW block of head m's value rows (128 B per pixel), 12x20 / 10x14 / 10x12 / 10x10
//                pixels = 600 slots (608 with chunk padding) = 76 KB, so that TWO 512-thread workgroups fit a CU (round 5: rounds 2-4 had
//                14x22 / 10x14 / 8x10 / 7x8 -- half of the far samples sat on levels 2 / 3; tools/win_geometry_search.py picks the
//                geometry that minimises them under the LDS budget over sampling spreads of 1-3 px: 2.2 -> 1.4 % at 1 px, 12.7 -> 8.2 %
//                at 2 px; 72 -> 69 us on the bench's inputs, profiles/r05_window_geometry.txt).  A window is placed where
//                the tile's own samples fall: the mean top-left corner of the in-range samples of the first 128
//                queries, reduced over the workgroup with integer LDS atomics.  Pixels outside the image are staged
//                as zeros (out-of-range raw buffer offsets), so the zero padding of border samples needs no masks.
//                Staging is LDS-DMA (`buffer_load_dwordx4 ... lds`): no registers, no ds_write, and it stays in
//                flight while the wave does other work.
//   lane roles = a QUAD of lanes owns one (query, head) pair; lane k of the quad prepares the four points of level
//                k (x first, then y; corner weights with the attention weight folded in) and accumulates the
//                channels of the 16-byte pieces k and k + 4 of a pixel.  Prepared samples never touch memory: the consumer lanes read them straight out of the
//                preparing lane's registers with DPP quad_perm broadcasts (folded into the FMA / address add).
//   LDS banks  = a ds_read_b128 is served in four groups of 16 lanes = 4 quads; a quad reads 64 contiguous
//                bytes = 16 of a pixel's 32 banks.  The four quads of a group take four different (16-byte half,
//                pixel parity) orders over the two x-adjacent corners of a bilinear row, so that in every
//                instruction the group covers all 64 banks exactly once, for any sample position.
//   far        = an in-range sample with a corner outside its window takes raw buffer loads (invalid corners at an
//                out-of-range offset), one far sample per quad and step; in round 0 the loads of the first step are
//                issued ahead of the window DMA and consumed behind it.  Correctness never depends on where the
//                windows are; only speed does (uniform-random locations are all far).
//   statistic  = one workgroup in 16 runs a copy of the body that counts its far samples; the last of them stores the
//                launch's totals as one 8-byte word into host-mapped memory, which the host's automatic kernel choice
//                reads without a sync (win_forward_auto below, include/msda_hip.h).
//
// Instruction-stream notes (profiles/r02_window_forward_experiments.txt): per-level constants come from an LDS table and
// the far sample's level constants by ds_bpermute -- `k == 0 ? a : k == 1 ? b : ...` on scalar registers compiles into
// trees of exec-masked branches; 24-bit multiply-adds are inline asm (v_mad_u32_u24) -- __mul24 comes back as quarter-rate
// v_mul_lo_u32; quotients use v_rcp_f32; (query, head) pair indices are 32-bit below uniform per-image base pointers.
//
// All geometry comes from the int64 shape tensors on the device; the host only knows S.  The grid is persistent: the two
// workgroups a CU holds walk the items kk, kk + K, ... of their head (win_workgroups_per_head below; 4-5 items each at the
// R50 shapes).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>

#include "msda_common.hpp"

namespace msda {
namespace {

constexpr int kT = 512, kWaves = kT / 64, kQuads = kT / 4;
// auto dispatch (win_forward_auto): the window kernel is used while the last reported far fraction is at most this, and the
// statistic is refreshed every kReprobe-th call otherwise.  Measured on model-like patterns of growing spread (round 4,
// tools/crossover_sweep.py, profiles/r04_crossover_sweep.txt; us per launch, window / gather kernel): far 0.02: 67 / 102, 0.10: 84 /
// 111, 0.21: 103 / 114, 0.30: 112 / 117, 0.37: 120 / 118, 0.41: 126 / 118 -- the lines cross near 0.35 (rounds 2-3 had 0.20 from
// three flavours only, the window kernel of the time being slower)
constexpr double kFarFractionMax = 0.33;
constexpr unsigned kReprobe = 64, kReportEvery = 8;
constexpr int kTH = 8, kTW = 16;
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
struct Meta {
  int sum[2][4][4];                                             // [item parity] per level: sum x0, sum y0, count, unused
  int geo[4][4];                                                // per level: first column / row and width of the tile's queries
  int org[4][4];                                                // per level: window origin x, y; last near column / row
  int lvl[4][8];                                                // per level: H, W, first pixel, window rows, window columns, window byte offset
  int stat[4];                                                  // this workgroup's far samples, live pairs, waves done, -
};
constexpr int kMetaOff = kZeroOff + kZeroBytes;
constexpr int kLdsBytes = kMetaOff + ((sizeof(Meta) + 15) / 16) * 16;
static_assert(kLdsBytes <= 80 * 1024, "two workgroups per CU");

typedef const f32x4 __attribute__((address_space(3)))* lds4;

// Phase timestamps (profiling builds only: make -C uninext_amd/csrc prof; tools/win_prof.py).  One lane per workgroup
// writes the 100 MHz real-time counter at each phase boundary of its FIRST tile.
#ifdef MSDA_WIN_PROF
constexpr int kProfBlocks = 8192, kProfSlots = 16;
__device__ unsigned long long g_win_prof[kProfBlocks * kProfSlots];
#define WIN_STAMP(i)                                                                                         \
  do {                                                                                                       \
    if (threadIdx.x == 0 && tile == g) {                                                                     \
      const unsigned blk_ = blockIdx.y * gridDim.x + blockIdx.x;                    \
      if (blk_ < (unsigned)kProfBlocks) g_win_prof[blk_ * kProfSlots + (i)] = __builtin_amdgcn_s_memrealtime(); \
    }                                                                                                        \
  } while (0)
#else
#define WIN_STAMP(i) do { } while (0)
#endif

typedef float v2f __attribute__((ext_vector_type(2)));        // packed fp32 math: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32

template <int SRC>
__device__ __forceinline__ uint32_t qb(uint32_t v) {   // value held by lane SRC of this lane's quad
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, SRC * 0x55, 0xF, 0xF, true);
}
template <int SRC>
__device__ __forceinline__ float qbf(float v) { return __uint_as_float(qb<SRC>(__float_as_uint(v))); }
template <int CTRL>
__device__ __forceinline__ int dppi(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }

// One prepared NEAR sample in the preparing lane's registers: corner weights (first-top, second-top), (first-bottom,
// second-bottom) and the LDS byte addresses of the first / second pixel of the top row -- "first" is the pixel whose
// slot parity this quad reads first.  Dead and far samples carry zero weights and point at the zero region.
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) {   // (a & 0xffffff) * (b & 0xffffff) + c
  uint32_t r;
  asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));   // volatile: stays out of branches
  return r;
}

struct Smp {
  v2f wT, wB;
  uint32_t aF, aS;
};

// Locality statistic: how many in-range samples of a launch missed their tile's windows.  Accumulated in device
// memory, moved by the last workgroup of the launch to a host-mapped report the dispatcher reads once the launch is
// known to be complete.  One word, in device memory while a launch runs and (with the launch's sequence number in
// place of the ticket) in the host-mapped report: far samples (bits 0..25), live (query, head) pairs (26..47),
// sampled workgroups done / sequence number (48..63).  EVERY LAUNCH HAS ITS OWN counter and report word (a ring per
// call-site slot, see locality_state below): launches that overlap on different streams, or a graph replay next to
// eager calls, cannot mix their counts.
struct LocalityArgs { unsigned long long* counter; volatile unsigned long long* report; unsigned seq, pad; };
constexpr int kStatPairShift = 26, kStatTicketShift = 48;
// every 2^shift-th workgroup of a head is sampled: 1 in 16, fewer on launches of more than 65536 workgroups (the
// fields above hold 4096 sampled workgroups of <= 192 pairs)
__device__ __forceinline__ int stat_shift(int workgroups) {
  int sh = 4;
  while ((workgroups >> sh) > 4096) ++sh;
  return sh;
}

}  // namespace

// The kernel is VALU-issue bound (a wave64 VALU instruction occupies its SIMD for 4 clocks; profiles/): every
// per-sample instruction below is counted -- hence packed fp32 math wherever two lanes of work sit side by side --
// and short of registers (128 at 4 waves per SIMD): per-lane level constants are re-derived from the lane id in
// every round, tile geometry and window origins are parked in LDS between rounds.
// STAT: this workgroup contributes to the locality statistic of the launch (1 workgroup in 16 does: the kernel below
// runs one of two copies of this body, so that the other 15 keep the register allocation of the plain kernel -- with
// the counting compiled into every workgroup the extra spills cost 5 us).
// REFD: 0 = the operator (loc = normalised sampling locations, attn = softmaxed weights); 2 / 4 = the module's
// elementwise prologue folded in (ops/modules/ms_deform_attn.py:99-112): loc = raw sampling offsets, attn = raw logits
// (same layouts), ref_points [N, Lq, 4 levels, REFD]; head_major: value is [N, M, S, 32] (linear_hip_packed_hm_f32).
template <int DMA_AUX, bool STAT, int REFD>   // DMA_AUX: cache policy bits of the window DMA (0 = default, 2 = non-temporal)
__device__ __forceinline__ void win_body(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                         const int64_t* __restrict__ lsi, const float* __restrict__ loc,
                                         const float* __restrict__ attn, const Dims& d, float* __restrict__ out,
                                         const float* __restrict__ ref_points, const bool head_major,
                                         const LocalityArgs& la) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Meta& mt = *reinterpret_cast<Meta*>(smem + kMetaOff);
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pq = lane >> 2;                                // quad of the wave
  const int M = d.M;
  const int m = blockIdx.x, kk = blockIdx.y, K = gridDim.y;   // workgroup kk of K on head m (2-D grid: no divisions by the head count)

  // ---- launch constants straight from the shape tensors (uniform addresses: scalar loads, no LDS table) ------------
  int lvH[4], lvW[4], lvS[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    lvH[l] = (int)shapes[2 * l];
    lvW[l] = (int)shapes[2 * l + 1];
    lvS[l] = (int)lsi[l];
  }
  const int TY = (lvH[0] + kTH - 1) / kTH, TX = (lvW[0] + kTW - 1) / kTW;
  const int ntiles = TY * TX, nitems = d.N * ntiles;       // work items of this head: (image, tile), image-major
  if (kk >= nitems) return;                                // over-provisioned part of the grid
  if (STAT && tid == 0) {   // stat[3]: the number of sampled workgroups of the launch
    const int per_head = min(K, nitems), sh = stat_shift(M * per_head);
    mt.stat[3] = M * (((per_head - 1) >> sh) + 1);
  }
  // the all-zero region, the placement sums (two sets: consecutive items alternate)
  for (int o = tid * 16; o < kZeroBytes; o += kT * 16) *reinterpret_cast<f32x4*>(smem + kZeroOff + o) = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tid < 32) (&mt.sum[0][0][0])[tid] = 0;
  // per-level constants as an LDS table: a lane fetches its level's row with two reads -- selecting them by the lane's level
  // from scalar registers (k == 0 ? .. : k == 1 ? ..) compiles into trees of exec-masked branches, in every round
  if (tid >= 64 && tid < 68) {
    const int l = tid - 64;
    const int4 a = make_int4(l == 0 ? lvH[0] : l == 1 ? lvH[1] : l == 2 ? lvH[2] : lvH[3], l == 0 ? lvW[0] : l == 1 ? lvW[1] : l == 2 ? lvW[2] : lvW[3],
                             l == 0 ? lvS[0] : l == 1 ? lvS[1] : l == 2 ? lvS[2] : lvS[3], l == 0 ? kWH[0] : l == 1 ? kWH[1] : l == 2 ? kWH[2] : kWH[3]);
    const int4 b = make_int4(l == 0 ? kWW[0] : l == 1 ? kWW[1] : l == 2 ? kWW[2] : kWW[3],
                             128 * (l == 0 ? kBase[0] : l == 1 ? kBase[1] : l == 2 ? kBase[2] : kBase[3]), 0, 0);
    *reinterpret_cast<int4*>(&mt.lvl[l][0]) = a;
    *reinterpret_cast<int4*>(&mt.lvl[l][4]) = b;
  }
  if (tid >= 32 && tid < 35) mt.stat[tid - 32] = 0;
  __syncthreads();                                         // the sums are zero before any wave adds to them

  const uint32_t pixB = head_major ? 128u : (uint32_t)M * 128u;   // bytes from a pixel of head m to the next one
  const uint32_t hoff = head_major ? 0u : (uint32_t)m * 128u;

  for (int item = kk, it = 0; item < nitems; item += K, ++it) {
#ifdef MSDA_WIN_PROF
    const int tile = it, g = 0;                             // WIN_STAMP records the first item of a workgroup
#endif
    WIN_STAMP(0);
    if (it > 0) __syncthreads();                            // (odd pyramids / persistent grids) everybody left the previous item
    // quotients by v_rcp_f32 (the IEEE division sequence is ~10 VALU instructions each, on the critical path of a work
    // item): x + 0.5 is at least 0.5 / divisor away from an integer, far beyond the 1 ulp of the reciprocal
    const int b = (int)(((float)item + 0.5f) * __builtin_amdgcn_rcpf((float)ntiles));
    const int64_t pair_img = (int64_t)b * d.Lq * M;         // first (query, head) pair of this item's image
    const float* const loc_img = loc + pair_img * 32;       // uniform bases: per-lane offsets stay 32-bit (S * M * 128 < 2^31)
    const float* const attn_img = attn + pair_img * 16;
    float* const out_img = out + pair_img * 32;
    const float* const ref_img = REFD ? ref_points + (int64_t)b * d.Lq * (4 * REFD) : nullptr;
    const __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(value) + (head_major ? ((int64_t)b * M + m) * d.S * 32 : (int64_t)b * d.S * M * 32), 0,
        (int)((uint32_t)d.S * pixB), 0x00020000);
    int (*const sums)[4] = mt.sum[it & 1];
    int nrest;                                              // queries of levels 1..3 of this item, and where they start
    int e1, e2;
    f32x4 lcA, lcB, at;                                     // lane k: the four points of level k -- 32 B + 16 B
    f32x4 rf = {0.f, 0.f, 0.f, 0.f};                        // REFD: the query's reference point / box on level k
    uint32_t pair;                                          // (query, head) pair within the image
    uint32_t qidx = 0;                                      // REFD: the query itself
    bool live;
    auto fetch = [&](int kq) __attribute__((always_inline)) {
      lcA = lcB = at = f32x4{0.f, 0.f, 0.f, 0.f};
      if (live) {
        const f32x4* lp = reinterpret_cast<const f32x4*>(loc_img + (pair * 32u + 8u * (uint32_t)kq));
        lcA = __builtin_nontemporal_load(lp);
        lcB = __builtin_nontemporal_load(lp + 1);
        at = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(attn_img + (pair * 16u + 4u * (uint32_t)kq)));
        if constexpr (REFD == 2) {
          const msda::f32x2 r2 = *reinterpret_cast<const msda::f32x2*>(ref_img + (qidx * 8u + 2u * (uint32_t)kq));
          rf[0] = r2[0]; rf[1] = r2[1];
        } else if constexpr (REFD == 4) {
          rf = *reinterpret_cast<const f32x4*>(ref_img + (qidx * 16u + 4u * (uint32_t)kq));
        }
      }
    };
    {
      // ---- tile geometry: lane k owns level k's query rectangle, the quad shares it by DPP, LDS keeps it ---------
      // level-k pixels [f(t), f(t + 1)) with f(t) = ceil(t * T * n / n0 - 1/2) are the ones whose centre falls into
      // tile t.  Evaluated in float: any monotone f with f(0) = 0 gives an exact partition as long as every workgroup
      // evaluates the same expression, which is all that correctness needs.
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
      const int nx = xe - xs, cnt = nx * (ye - ys);
      if (tid < 4) *reinterpret_cast<int4*>(&mt.geo[kq][0]) = make_int4(xs, ys, nx, 0);   // read after the placement barrier
      e1 = __builtin_amdgcn_readfirstlane((int)qb<1>((uint32_t)cnt));
      e2 = e1 + __builtin_amdgcn_readfirstlane((int)qb<2>((uint32_t)cnt));
      nrest = e2 + __builtin_amdgcn_readfirstlane((int)qb<3>((uint32_t)cnt));
      if (STAT && tid == 0) mt.stat[1] += nrest + (int)qb<0>((uint32_t)cnt);   // live pairs of this workgroup
      // round 0 = the level-0 queries of the tile, wave = tile row, quad = tile column (no division)
      live = pq < (int)qb<0>((uint32_t)nx) && wv < (int)qb<0>((uint32_t)(ye - ys));
      const int q = lvS[0] + ((int)qb<0>((uint32_t)ys) + wv) * lvW[0] + (int)qb<0>((uint32_t)xs) + pq;
      live = live && q < d.Lq;                             // (shapes whose pixel count exceeds num_query: never outside the tensors)
      qidx = (uint32_t)(live ? q : 0);
      pair = mad_u24(qidx, (uint32_t)M, (uint32_t)m);
      fetch(kq);
    }
    WIN_STAMP(1);
    // this wave's rounds: a wave without a live quad has nothing to do after round 0 (no barrier in the later rounds)
    const int nrounds = 1 + (nrest > wv * 16 ? (nrest - wv * 16 + kQuads - 1) / kQuads : 0);

    for (int rnd = 0; rnd < nrounds; ++rnd) {
      // ---- per-lane constants of this lane's level, re-derived every round (see the kernel header) ----------------
      int k = lane & 3;
      asm volatile("" : "+v"(k));                            // opaque: nothing below is hoisted out of the round loop
      const int cls_a = (lane >> 3) & 1, cls_e = (lane >> 4) & 1;   // bank class of the quad: half read first, parity read first
      // this lane's channels: the 16-byte pieces k and k + 4 of a pixel, i.e. a quad reads / writes 64 contiguous bytes
      // per instruction; c0 = the piece read first, c0 ^ 64 the other
      const uint32_t c0 = (uint32_t)(16 * k + 64 * cls_a);
      const int4 lv4 = *reinterpret_cast<const int4*>(&mt.lvl[k][0]);
      const int2 lv2 = *reinterpret_cast<const int2*>(&mt.lvl[k][4]);
      const int myH = lv4.x, myW = lv4.y, myS = lv4.z, myWH = lv4.w, myWW = lv2.x;
      const v2f fWH = {(float)myW, (float)myH};

      // ---- sample coordinates (the reference's arithmetic, cuh:282-288 and :38-46) ----------------------------------
      v2f xy[4], fl[4];
      float sa[4];
      bool inr[4];
      int x0[4], y0[4];
      if constexpr (REFD != 0) {
        // softmax over the pair's 16 logits: four per lane, the quad finishes it with DPP (v_rcp_f32: 1 ulp on the weights)
        float mx = fmaxf(fmaxf(at[0], at[1]), fmaxf(at[2], at[3]));
        mx = fmaxf(mx, __uint_as_float((uint32_t)dppi<0xB1>((int)__float_as_uint(mx))));
        mx = fmaxf(mx, __uint_as_float((uint32_t)dppi<0x4E>((int)__float_as_uint(mx))));
        at[0] = __expf(at[0] - mx); at[1] = __expf(at[1] - mx); at[2] = __expf(at[2] - mx); at[3] = __expf(at[3] - mx);
        float sm_ = (at[0] + at[1]) + (at[2] + at[3]);
        sm_ += __uint_as_float((uint32_t)dppi<0xB1>((int)__float_as_uint(sm_)));
        sm_ += __uint_as_float((uint32_t)dppi<0x4E>((int)__float_as_uint(sm_)));
        const float inv = __builtin_amdgcn_rcpf(sm_);
        at[0] *= inv; at[1] *= inv; at[2] *= inv; at[3] *= inv;
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const v2f l2 = p == 0 ? v2f{lcA[0], lcA[1]} : p == 1 ? v2f{lcA[2], lcA[3]} : p == 2 ? v2f{lcB[0], lcB[1]} : v2f{lcB[2], lcB[3]};
        sa[p] = at[p];
        if constexpr (REFD == 0) {
          xy[p] = __builtin_elementwise_fma(l2, fWH, v2f{-0.5f, -0.5f});
        } else if constexpr (REFD == 2) {   // loc = ref + off / (W, H):  loc * (W, H) - 0.5 = ref * (W, H) + (off - 0.5)
          xy[p] = __builtin_elementwise_fma(v2f{rf[0], rf[1]}, fWH, l2 - v2f{0.5f, 0.5f});
        } else {                            // loc = ref_xy + off / P * ref_wh * 0.5  (P = 4)
          const v2f lc = __builtin_elementwise_fma(l2, v2f{rf[2], rf[3]} * 0.125f, v2f{rf[0], rf[1]});
          xy[p] = __builtin_elementwise_fma(lc, fWH, v2f{-0.5f, -0.5f});
        }
        fl[p] = v2f{floorf(xy[p].x), floorf(xy[p].y)};
        inr[p] = live && (xy[p].y > -1.f) && (xy[p].x > -1.f) && (xy[p].y < fWH.y) && (xy[p].x < fWH.x);
        x0[p] = inr[p] ? (int)fl[p].x : 0;
        y0[p] = inr[p] ? (int)fl[p].y : 0;
      }

      int myOx, myOy;                                          // window origin of level k
      uint32_t cxmax, rymax;                                   // largest window column / row a top-left corner may take
      if (rnd == 0) {
        WIN_STAMP(2);                                          // loc / attn have arrived, sample coordinates done
        // ---- window placement: mean top-left corner of this tile's in-range samples, per level ---------------------
        int ax = 0, ay = 0, an = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          ax += x0[p];                                         // 0 when out of range
          ay += y0[p];
          an += inr[p] ? 1 : 0;
        }
        // over the 16 quads of the wave (lanes with the same k): row_shr 4 / 8 inside a row of 16, then rows
        ax += dppi<0x114>(ax); ay += dppi<0x114>(ay); an += dppi<0x114>(an);
        ax += dppi<0x118>(ax); ay += dppi<0x118>(ay); an += dppi<0x118>(an);   // lanes 12..15 of a row: the row's total
        ax += __shfl_xor(ax, 16, 64); ay += __shfl_xor(ay, 16, 64); an += __shfl_xor(an, 16, 64);
        ax += __shfl_xor(ax, 32, 64); ay += __shfl_xor(ay, 32, 64); an += __shfl_xor(an, 32, 64);
        if (pq == 3 && an > 0) {
          atomicAdd(&sums[k][0], ax);
          atomicAdd(&sums[k][1], ay);
          atomicAdd(&sums[k][2], an);
        }
        __syncthreads();
        WIN_STAMP(3);
        {
          const int4 sm = *reinterpret_cast<const int4*>(&sums[k][0]);
          const int4 ge = *reinterpret_cast<const int4*>(&mt.geo[k][0]);
          if (tid < 16) (&mt.sum[(it & 1) ^ 1][0][0])[tid] = 0;   // the next item's sums: nobody reads or adds to them now
          myOx = ge.x - 3; myOy = ge.y - 3;
          if (sm.z > 0) {   // v_rcp_f32: every lane of the workgroup evaluates the same expression on the same sums
            const float inv = __builtin_amdgcn_rcpf((float)sm.z);
            myOx = (int)floorf((float)sm.x * inv + 0.5f) - (myWW - 2) / 2;
            myOy = (int)floorf((float)sm.y * inv + 0.5f) - (myWH - 2) / 2;
          }
          myOx = max(-1, min(myOx, myW + 1 - myWW));
          myOy = max(-1, min(myOy, myH + 1 - myWH));
          // a level smaller than its window: top-left corners past the last in-range one are not "near"
          cxmax = (uint32_t)(min(myOx + myWW - 2, myW - 1) - myOx);
          rymax = (uint32_t)(min(myOy + myWH - 2, myH - 1) - myOy);
          if (tid < 4) *reinterpret_cast<int4*>(&mt.org[k][0]) = make_int4(myOx, myOy, (int)cxmax, (int)rymax);   // for the later rounds
        }
      } else {
        const int4 og = *reinterpret_cast<const int4*>(&mt.org[k][0]);
        myOx = og.x; myOy = og.y; cxmax = (uint32_t)og.z; rymax = (uint32_t)og.w;
      }

      // ---- stage the four windows: LDS-DMA, one instruction = 8 consecutive slots (1 KB) of ONE level per wave.  Straight-line
      // code with the same number of instructions in every wave (a wave without a chunk left in a level issues an
      // out-of-range one into the all-zero region, which costs no memory access), so that loads issued BEFORE it can be
      // waited for with an exact vmcnt while the windows are still in flight -------------------------------------------
      auto stage_windows = [&]() __attribute__((always_inline)) {
        int ogx[4], ogy[4];
        ogx[0] = __builtin_amdgcn_readfirstlane((int)qb<0>((uint32_t)myOx)); ogy[0] = __builtin_amdgcn_readfirstlane((int)qb<0>((uint32_t)myOy));
        ogx[1] = __builtin_amdgcn_readfirstlane((int)qb<1>((uint32_t)myOx)); ogy[1] = __builtin_amdgcn_readfirstlane((int)qb<1>((uint32_t)myOy));
        ogx[2] = __builtin_amdgcn_readfirstlane((int)qb<2>((uint32_t)myOx)); ogy[2] = __builtin_amdgcn_readfirstlane((int)qb<2>((uint32_t)myOy));
        ogx[3] = __builtin_amdgcn_readfirstlane((int)qb<3>((uint32_t)myOx)); ogy[3] = __builtin_amdgcn_readfirstlane((int)qb<3>((uint32_t)myOy));
        const uint32_t chunk = (uint32_t)(lane & 7) * 16u;
        const int sub = lane >> 3;
        // the wave's number, opaque HERE: what is derived from it (first chunk of a level, LDS destinations, row / column of
        // a chunk) is a handful of scalar instructions per item -- hoisted out of the item loop as loop invariants those
        // values do not fit the scalar registers and come back as v_readlane of spilled SGPRs, a vector instruction each
        int wvs = wv;
        asm volatile("" : "+s"(wvs));
        // Round 4: the byte offset of a lane in a DMA instruction is  (scalar base of the chunk)  +  (slot of the lane within
        // the chunk) * pitch + piece, and a chunk of 8 consecutive window slots spans at most two window rows -- so the 17
        // vector instructions of per-lane (row, column, inside-the-image, pixel, offset) arithmetic per DMA instruction become
        // scalar arithmetic plus ~4 vector instructions: one select between the two rows' bases and one add (and one more
        // select per out-of-image window column, which only border tiles have).
        uint32_t vsub = (uint32_t)sub;
        asm volatile("" : "+v"(vsub));                        // a VGPR: the selects below compare it with scalars
        const uint32_t vlane = mad_u24(vsub, pixB, chunk);
        auto stage_level = [&](auto ltag) __attribute__((always_inline)) {
          constexpr int LV = decltype(ltag)::value;
          constexpr int WW = kWW[LV], C0 = kBase[LV] / 8, C1 = kBase[LV + 1] / 8;
          constexpr int kSteps = (C1 - C0 + kWaves - 1) / kWaves;
          const int Hs = lvH[LV], Ws = lvW[LV], xS = ogx[LV] + lvS[LV], oy = ogy[LV], ox = ogx[LV];
          int i = C0 + ((wvs - C0) & (kWaves - 1));              // this wave's first chunk of the level
          if (Ws + 2 >= WW) {                                    // at most ONE window column outside the image on either side
            // window column 0 is x = -1 / column WW - 1 is x = W  (an int, not a bool: the compiler keeps booleans of uniform
            // compares as lane masks and re-materialises them through a VGPR at every use)
            const int border = (int)((uint32_t)ox >> 31) | (int)((uint32_t)(Ws - ox - WW) >> 31);
#pragma unroll
            for (int t = 0; t < kSteps; ++t, i += kWaves) {
              const bool have = i < C1;                           // everything in this loop is wave-uniform except vsub / vlane / off
              const int rel0 = 8 * (i - C0);
              const int r0 = rel0 / WW, c0 = rel0 - r0 * WW;      // (division by a constant, scalar)
              const int thr = WW - c0;                            // lanes with sub >= thr sit in window row r0 + 1
              const int yA = oy + r0;
              const bool okA = have && (unsigned)yA < (unsigned)Hs, okB = have && (unsigned)(yA + 1) < (unsigned)Hs;
              const int pixA = yA * Ws + xS + c0;                 // pixel of slot 0 of the chunk
              const uint32_t baseA = okA ? (uint32_t)pixA * pixB : kOobOffset;
              const uint32_t baseB = okB ? (uint32_t)(pixA + Ws - WW) * pixB : kOobOffset;
              // (mod 2^32: baseA + (baseB - baseA) = baseB; kOobOffset + vlane stays out of range, vlane < 2^31)
              uint32_t off = (vlane + baseA) + (vsub >= (uint32_t)thr ? baseB - baseA : 0u);
              if (border != 0) {                                  // border tiles only: a real branch (the asm keeps it from being if-converted)
                asm volatile("; window column outside the image");
                if (ox < 0) off = vsub == (uint32_t)(c0 == 0 ? 0 : thr) ? kOobOffset : off;
                if (ox + WW > Ws) off = vsub == (uint32_t)(WW - 1 - c0) ? kOobOffset : off;   // (wrapped lanes never reach column WW - 1: WW >= 8)
              }
              const int dst = have ? i * 1024 : kZeroOff;
              __builtin_amdgcn_raw_ptr_buffer_load_lds(vsrc, (__attribute__((address_space(3))) void*)(smem + dst), 16,
                                                       off, hoff, 0, DMA_AUX);
              __builtin_amdgcn_sched_barrier(0);
            }
            return;
          }
          // (levels narrower than their window: per-lane row / column / inside-the-image arithmetic for every DMA instruction)
          int subv = sub;
          asm volatile("" : "+v"(subv));                      // opaque: the level's start is computed HERE
          const int rel = 8 * (i - C0) + subv;                  // slot of this lane in the level's window
          int r = (int)(((float)rel + 0.5f) * (1.f / WW)), c = rel - r * WW;
#pragma unroll
          for (int t = 0; t < kSteps; ++t, i += kWaves) {
            const bool have = i < C1;                             // wave-uniform
            const int y = oy + r;
            const bool inside = have && (unsigned)y < (unsigned)Hs && (unsigned)(ox + c) < (unsigned)Ws;
            // pixel index < 2^24 and pixel pitch M * 128 < 2^24 by win_forward_ok: two full-rate 24-bit multiply-adds
            // (left to the compiler, __mul24 comes back as quarter-rate v_mul_lo_u32)
            const uint32_t pix = mad_u24((uint32_t)y, (uint32_t)Ws, (uint32_t)(xS + c));
            const uint32_t in_off = mad_u24(pix, pixB, chunk);
            const uint32_t off = inside ? in_off : kOobOffset;
            const int dst = have ? i * 1024 : kZeroOff;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vsrc, (__attribute__((address_space(3))) void*)(smem + dst), 16,
                                                     off, hoff, 0, DMA_AUX);
            c += 64 % WW; r += 64 / WW;
            if (c >= WW) { c -= WW; r += 1; }
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        stage_level(std::integral_constant<int, 0>{});
        stage_level(std::integral_constant<int, 1>{});
        stage_level(std::integral_constant<int, 2>{});
        stage_level(std::integral_constant<int, 3>{});
        WIN_STAMP(4);                                            // window DMA issued
      };

      // ---- prepare this lane's four samples ---------------------------------------------------------------------------
      Smp smp[4];
      uint32_t farmask = 0;
      {
        const uint32_t myWin = smem_base + (uint32_t)lv2.y;
        const uint32_t zero_first = smem_base + kZeroOff + 128u * (uint32_t)cls_e;        // parity cls_e
        const uint32_t zero_second = smem_base + kZeroOff + 128u * (uint32_t)(cls_e ^ 1);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          v2f fr = xy[p] - fl[p];                                // (fx, fy); inf - inf / NaN for poisoned locations ...
          fr.x = fmaxf(fr.x, 0.f); fr.y = fmaxf(fr.y, 0.f);      // ... which must not turn the zero weights of dead samples into NaN
          const v2f om = v2f{1.f, 1.f} - fr;                     // (1 - fx, 1 - fy)
          const int cx = x0[p] - myOx, ry = y0[p] - myOy;
          const bool near = inr[p] && (uint32_t)cx <= cxmax && (uint32_t)ry <= rymax;
          farmask |= (inr[p] && !near) ? (1u << p) : 0u;
          const uint32_t sw = (uint32_t)(cx ^ cls_e) & 1u;       // 1: the right-hand pixel has this quad's first parity
          const float an = near ? sa[p] : 0.f;                   // dead and far samples: all four weights 0
          const v2f gx = sw ? v2f{fr.x, om.x} : v2f{om.x, fr.x}; // x factors of the (first, second) pixel
          const v2f wtb = v2f{om.y, fr.y} * an;                  // (top, bottom) row weight x attention weight
          smp[p].wT = gx * wtb.x;
          smp[p].wB = gx * wtb.y;
          const uint32_t tl = myWin + (uint32_t)(__mul24(ry, myWW) + cx) * 128u;
          smp[p].aF = near ? tl + (sw << 7) : zero_first;
          smp[p].aS = near ? tl + 128u - (sw << 7) : zero_second;
        }
      }

      // locality statistic of the launch (feeds the host's kernel choice, see win_forward_auto)
      if (STAT && farmask) atomicAdd(&mt.stat[0], __builtin_popcount(farmask));   // per lane: no scalar state

      f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};   // channels at c0 and at c0 ^ 64
      if (rnd == 0) WIN_STAMP(5);                              // samples prepared

      // ---- far samples: raw buffer loads, one far sample per quad and step ------------------------------------------------
      {
        uint32_t fm = farmask << (4 * k);                      // the pair's 16 samples: bit 4 * level + point
        fm |= (uint32_t)dppi<0xB1>((int)fm);                   // quad_perm [1,0,3,2]
        fm |= (uint32_t)dppi<0x4E>((int)fm);                   // quad_perm [2,3,0,1]
        struct Far { f32x4 d1a, d1b, d2a, d2b, d3a, d3b, d4a, d4b; float w1, w2, w3, w4; };
        auto far_issue = [&](Far& f) __attribute__((always_inline)) {
          const bool has = fm != 0u;
          const int idx = has ? __builtin_ctz(fm) : 0;
          fm &= fm - 1u;
          const int ps = idx & 3, fl_ = idx >> 2;               // point and level of the far sample
          const int src = ((lane & ~3) | fl_) << 2;            // byte address of the preparing lane for ds_bpermute
          // the preparing lane's point `ps`: every lane selects its own candidate, the quad pulls the right one and
          // redoes the (cheap) sample arithmetic -- far samples are a few per cent, their state is not kept around
          // (selects on scalars: as ?: chains on register PAIRS the compiler builds exec-masked branches here)
          const bool c1 = (ps & 1) != 0, c2 = (ps & 2) != 0;
          auto sel4 = [&](float a0, float a1, float a2, float a3) __attribute__((always_inline)) {
            const float t0 = c1 ? a1 : a0, t1 = c1 ? a3 : a2;   // scalars: v_cndmask_b32
            return (int)__float_as_uint(c2 ? t1 : t0);
          };
          const int cx_ = sel4(xy[0].x, xy[1].x, xy[2].x, xy[3].x), cy_ = sel4(xy[0].y, xy[1].y, xy[2].y, xy[3].y);
          const int ca_ = sel4(sa[0], sa[1], sa[2], sa[3]);
          // quads without a far sample left run along with zero weights: their stand-in coordinates must be finite
          const uint32_t hm = has ? 0xffffffffu : 0u;
          const float fxv = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, cx_) & hm);
          const float fyv = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, cy_) & hm);
          const float fav = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, ca_) & hm);
          // the far sample's level constants come from the preparing lane as well (it is lane fl_ of the quad and holds its
          // own level's): selecting them by fl_ from scalar registers compiles into trees of exec-masked branches
          const int fW = __builtin_amdgcn_ds_bpermute(src, myW), fH = __builtin_amdgcn_ds_bpermute(src, myH);
          const int fS = __builtin_amdgcn_ds_bpermute(src, myS);
          const uint32_t rowG = mad_u24((uint32_t)fW, pixB, 0u);
          const float xf = floorf(fxv), yf = floorf(fyv);
          const float lw = fxv - xf, lh = fyv - yf;
          const int fx0 = (int)xf, fy0 = (int)yf;                // 0 for the stand-ins
          const bool t_ok = has && fy0 >= 0, b_ok = has && fy0 + 1 <= fH - 1, l_ok = fx0 >= 0, r_ok = fx0 + 1 <= fW - 1;
          const float wt = (1.f - lh) * fav, wb = lh * fav;
          f.w1 = wt * (1.f - lw); f.w2 = wt * lw; f.w3 = wb * (1.f - lw); f.w4 = wb * lw;
          // 24-bit multiply-adds (pixel index < 2^24, pitch < 2^24) on the CLAMPED top-left pixel: with fy0 or fx0 = -1 the
          // live corners sit in row / column 0, and a 24-bit product of a negative index is not what a 32-bit one wraps
          // to (it differed by 2^31 for odd head counts: the live corners of such samples were dropped)
          const int cy = max(fy0, 0), cx = max(fx0, 0);
          const uint32_t off = mad_u24(mad_u24((uint32_t)cy, (uint32_t)fW, (uint32_t)(fS + cx)), pixB, c0);
          const uint32_t dx = fx0 >= 0 ? pixB : 0u, dy = fy0 >= 0 ? rowG : 0u;   // step to the right / bottom neighbour
          const uint32_t o1 = (t_ok && l_ok) ? off : kOobOffset;
          const uint32_t o2 = (t_ok && r_ok) ? off + dx : kOobOffset;
          const uint32_t o3 = (b_ok && l_ok) ? off + dy : kOobOffset;
          const uint32_t o4 = (b_ok && r_ok) ? off + dy + dx : kOobOffset;
          f.d1a = buffer_load_f32x4(vsrc, o1, hoff); f.d1b = buffer_load_f32x4(vsrc, o1 ^ 64u, hoff);
          f.d2a = buffer_load_f32x4(vsrc, o2, hoff); f.d2b = buffer_load_f32x4(vsrc, o2 ^ 64u, hoff);
          f.d3a = buffer_load_f32x4(vsrc, o3, hoff); f.d3b = buffer_load_f32x4(vsrc, o3 ^ 64u, hoff);
          f.d4a = buffer_load_f32x4(vsrc, o4, hoff); f.d4b = buffer_load_f32x4(vsrc, o4 ^ 64u, hoff);
        };
        auto far_consume = [&](const Far& f) __attribute__((always_inline)) {
          const v2f W1 = {f.w1, f.w1}, W2 = {f.w2, f.w2}, W3 = {f.w3, f.w3}, W4 = {f.w4, f.w4};
#pragma unroll
          for (int h = 0; h < 2; ++h) {                         // channel pairs: v_pk_fma_f32
            v2f a = {accA[2 * h], accA[2 * h + 1]}, b = {accB[2 * h], accB[2 * h + 1]};
            a = __builtin_elementwise_fma(W1, v2f{f.d1a[2 * h], f.d1a[2 * h + 1]}, a);
            b = __builtin_elementwise_fma(W1, v2f{f.d1b[2 * h], f.d1b[2 * h + 1]}, b);
            a = __builtin_elementwise_fma(W2, v2f{f.d2a[2 * h], f.d2a[2 * h + 1]}, a);
            b = __builtin_elementwise_fma(W2, v2f{f.d2b[2 * h], f.d2b[2 * h + 1]}, b);
            a = __builtin_elementwise_fma(W3, v2f{f.d3a[2 * h], f.d3a[2 * h + 1]}, a);
            b = __builtin_elementwise_fma(W3, v2f{f.d3b[2 * h], f.d3b[2 * h + 1]}, b);
            a = __builtin_elementwise_fma(W4, v2f{f.d4a[2 * h], f.d4a[2 * h + 1]}, a);
            b = __builtin_elementwise_fma(W4, v2f{f.d4b[2 * h], f.d4b[2 * h + 1]}, b);
            accA[2 * h] = a.x; accA[2 * h + 1] = a.y; accB[2 * h] = b.x; accB[2 * h + 1] = b.y;
          }
          asm volatile("" : "+v"(accA), "+v"(accB));
        };
        Far f0;
        // round 0: the loads of the FIRST far step are issued ahead of the window DMA and consumed behind it (they do not
        // queue behind the CU's whole staging burst, and the wait for them is an exact vmcnt: everything younger is the
        // fixed number of DMA instructions); the barrier follows, further far steps run after it
        if (rnd == 0) {
          far_issue(f0);                                       // unconditional: quads without a far sample load nothing
          __builtin_amdgcn_sched_barrier(0);
          stage_windows();
          __builtin_amdgcn_sched_barrier(0);
          far_consume(f0);
        } else if (__ballot(fm != 0u)) {
          far_issue(f0);
          far_consume(f0);
        }
        if (rnd == 0) {
          WIN_STAMP(6);                                        // first far step done
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the windows has landed
          __syncthreads();                                   // ... and everybody else's
          WIN_STAMP(7);                                        // windows complete
        }
        while (__ballot(fm != 0u)) {
          far_issue(f0);
          far_consume(f0);
        }
      }

      // ---- the next round's locations and weights travel while this round reads the LDS ----------------------------------
      const uint32_t cur_pair = pair;
      const bool cur_live = live;
      if (rnd + 1 < nrounds) {                                 // ri-th query of levels 1..3
        const int ri = rnd * kQuads + wv * 16 + pq;
        live = ri < nrest;
        const int ql = 1 + (ri >= e1 ? 1 : 0) + (ri >= e2 ? 1 : 0);
        const int j = ri - (ql == 1 ? 0 : ql == 2 ? e1 : e2);
        const int4 ge = *reinterpret_cast<const int4*>(&mt.geo[ql][0]);     // xs, ys, nx of that level
        const int2 qWS = *reinterpret_cast<const int2*>(&mt.lvl[ql][1]);
        const int Wq = qWS.x, Sq = qWS.y;
        const int yy = (int)(((float)j + 0.5f) * __builtin_amdgcn_rcpf((float)max(ge.z, 1)));
        const uint32_t qn = mad_u24((uint32_t)(ge.y + yy), (uint32_t)Wq, (uint32_t)(Sq + ge.x + j)) - mad_u24((uint32_t)yy, (uint32_t)ge.z, 0u);
        live = live && qn < (uint32_t)d.Lq;
        qidx = live ? qn : 0u;
        pair = mad_u24(qidx, (uint32_t)M, (uint32_t)m);
        fetch(k);
      }

      // ---- near samples: 16 samples x 4 corners x 2 halves from the LDS windows ---------------------------------------
      v2f aA0 = {accA[0], accA[1]}, aA1 = {accA[2], accA[3]}, aB0 = {accB[0], accB[1]}, aB1 = {accB[2], accB[3]};
      // Software pipeline in half-samples (a corner row = 4 reads + 8 packed FMAs): the top row of sample s + 1 is
      // requested before the top row of sample s is consumed, and likewise the bottom rows -- three rows (48 registers)
      // in flight at most, so one wave covers most of the LDS latency on its own (the other workgroup of the CU is
      // usually waiting on memory); a full two-sample pipeline would not fit 128 registers.
      struct Row { f32x4 Fa, Fb, Sa, Sb; };
      struct Adr { lds4 pF, pF2, pS, pS2; };
      auto fetch_top = [&](auto ltag, int p, Row& r, Adr& ad) __attribute__((always_inline)) {
        constexpr int LV = decltype(ltag)::value;
        const uint32_t aF = qb<LV>(smp[p].aF) + c0, aS = qb<LV>(smp[p].aS) + c0;
        ad.pF = reinterpret_cast<lds4>((uintptr_t)aF); ad.pF2 = reinterpret_cast<lds4>((uintptr_t)(aF ^ 64u));
        ad.pS = reinterpret_cast<lds4>((uintptr_t)aS); ad.pS2 = reinterpret_cast<lds4>((uintptr_t)(aS ^ 64u));
        r.Fa = ad.pF[0]; r.Fb = ad.pF2[0]; r.Sa = ad.pS[0]; r.Sb = ad.pS2[0];
        __builtin_amdgcn_sched_barrier(0);
      };
      auto fetch_bot = [&](auto ltag, Row& r, const Adr& ad) __attribute__((always_inline)) {
        constexpr int kRow = kWW[decltype(ltag)::value] * 8;   // one window row, in 16-byte units
        r.Fa = ad.pF[kRow]; r.Fb = ad.pF2[kRow]; r.Sa = ad.pS[kRow]; r.Sb = ad.pS2[kRow];
        __builtin_amdgcn_sched_barrier(0);
      };
      auto consume = [&](auto ltag, v2f wrow, const Row& r) __attribute__((always_inline)) {
        constexpr int LV = decltype(ltag)::value;
        const float wf = qbf<LV>(wrow.x), ws = qbf<LV>(wrow.y);
        const v2f WF = {wf, wf}, WS = {ws, ws};
        aA0 = __builtin_elementwise_fma(WF, v2f{r.Fa[0], r.Fa[1]}, aA0); aA1 = __builtin_elementwise_fma(WF, v2f{r.Fa[2], r.Fa[3]}, aA1);
        aB0 = __builtin_elementwise_fma(WF, v2f{r.Fb[0], r.Fb[1]}, aB0); aB1 = __builtin_elementwise_fma(WF, v2f{r.Fb[2], r.Fb[3]}, aB1);
        aA0 = __builtin_elementwise_fma(WS, v2f{r.Sa[0], r.Sa[1]}, aA0); aA1 = __builtin_elementwise_fma(WS, v2f{r.Sa[2], r.Sa[3]}, aA1);
        aB0 = __builtin_elementwise_fma(WS, v2f{r.Sb[0], r.Sb[1]}, aB0); aB1 = __builtin_elementwise_fma(WS, v2f{r.Sb[2], r.Sb[3]}, aB1);
        asm volatile("" : "+v"(aA0), "+v"(aA1), "+v"(aB0), "+v"(aB1));   // pins the FMAs here (IR-level sinking ignores sched_barrier)
        __builtin_amdgcn_sched_barrier(0);
      };
      {
        Row t0, t1, b0, b1;
        Adr ad;
        using L0 = std::integral_constant<int, 0>; using L1 = std::integral_constant<int, 1>;
        using L2 = std::integral_constant<int, 2>; using L3 = std::integral_constant<int, 3>;
#define WIN_STEP(LC, PC, LN, PN, TC, BC, TN, BN)                                                \
        fetch_top(LN{}, PN, TN, ad); consume(LC{}, smp[PC].wT, TC);                              \
        fetch_bot(LN{}, BN, ad);     consume(LC{}, smp[PC].wB, BC);
        fetch_top(L0{}, 0, t0, ad); fetch_bot(L0{}, b0, ad);
        WIN_STEP(L0, 0, L0, 1, t0, b0, t1, b1) WIN_STEP(L0, 1, L0, 2, t1, b1, t0, b0)
        WIN_STEP(L0, 2, L0, 3, t0, b0, t1, b1) WIN_STEP(L0, 3, L1, 0, t1, b1, t0, b0)
        WIN_STEP(L1, 0, L1, 1, t0, b0, t1, b1) WIN_STEP(L1, 1, L1, 2, t1, b1, t0, b0)
        WIN_STEP(L1, 2, L1, 3, t0, b0, t1, b1) WIN_STEP(L1, 3, L2, 0, t1, b1, t0, b0)
        WIN_STEP(L2, 0, L2, 1, t0, b0, t1, b1) WIN_STEP(L2, 1, L2, 2, t1, b1, t0, b0)
        WIN_STEP(L2, 2, L2, 3, t0, b0, t1, b1) WIN_STEP(L2, 3, L3, 0, t1, b1, t0, b0)
        WIN_STEP(L3, 0, L3, 1, t0, b0, t1, b1) WIN_STEP(L3, 1, L3, 2, t1, b1, t0, b0)
        WIN_STEP(L3, 2, L3, 3, t0, b0, t1, b1)
        consume(L3{}, smp[3].wT, t1); consume(L3{}, smp[3].wB, b1);
#undef WIN_STEP
      }

      if (rnd == 0) WIN_STAMP(8);                              // LDS pass of the first round done
      if (cur_live) {   // a quad writes 2 x 64 contiguous bytes
        float* op = out_img + (cur_pair * 32u + 4u * (uint32_t)k);
        __builtin_nontemporal_store(f32x4{aA0.x, aA0.y, aA1.x, aA1.y}, reinterpret_cast<f32x4*>(op + 16 * cls_a));
        __builtin_nontemporal_store(f32x4{aB0.x, aB0.y, aB1.x, aB1.y}, reinterpret_cast<f32x4*>(op + 16 * (cls_a ^ 1)));
      }
    }
    WIN_STAMP(9);                                              // all rounds done (wave 0)
  }
  // ---- publish the locality count: workgroup (LDS) -> launch (ONE 64-bit L2 atomic per sampled workgroup: same-address
  // atomics from all XCDs cost ~14 ns each, 3000 of them would double the kernel's time) -> host (the last sampled
  // workgroup stores the totals into host-mapped memory) ----------------------------------------------------------------
  if (STAT && lane == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (atomicAdd(&mt.stat[2], 1) == kWaves - 1) {             // the last wave of the workgroup
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const unsigned long long mine = (unsigned long long)(unsigned)atomicAdd(&mt.stat[0], 0) |
                                      ((unsigned long long)(unsigned)mt.stat[1] << kStatPairShift) | (1ull << kStatTicketShift);
      const unsigned long long now = atomicAdd(la.counter, mine) + mine;
      const unsigned expected = (unsigned)mt.stat[3];
      if ((unsigned)(now >> kStatTicketShift) == expected) {      // the last sampled workgroup of the launch
        atomicExch(la.counter, 0ull);                           // this ring entry's next launch starts from zero
        // ONE 8-byte store into host-mapped memory: no system-scope fence (it would write the L2 back), no read over PCIe
        *la.report = (now & ((1ull << kStatTicketShift) - 1ull)) | ((unsigned long long)(la.seq & 0xffffu) << kStatTicketShift);
      }
    }
  }
}

// which copy of the body a workgroup runs: the counting one for every 2^shift-th workgroup of a head
__device__ __forceinline__ bool win_sampled(const int64_t* __restrict__ shapes, const Dims& d) {
  const int M = d.M, kk = blockIdx.y, K = gridDim.y;
  const int TY = ((int)shapes[0] + kTH - 1) / kTH, TX = ((int)shapes[1] + kTW - 1) / kTW;
  const int nitems = d.N * TY * TX;
  return (kk & ((1 << stat_shift(M * min(K, nitems))) - 1)) == 0;
}

template <int DMA_AUX, bool STAT>
__global__ void __launch_bounds__(kT, 4)
msda_fwd_win(const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
             const float* __restrict__ loc, const float* __restrict__ attn, Dims d, float* __restrict__ out, LocalityArgs la) {
  if (STAT && win_sampled(shapes, d)) {
    win_body<DMA_AUX, true, 0>(value, shapes, lsi, loc, attn, d, out, nullptr, false, la);
    return;
  }
  win_body<DMA_AUX, false, 0>(value, shapes, lsi, loc, attn, d, out, nullptr, false, la);
}

// the module's inference path: prologue folded in (REFD = 2 / 4), value in the reference or the head-major layout
template <bool STAT, int REFD>
__global__ void __launch_bounds__(kT, 4)
msda_fwd_win_fused(const float* __restrict__ value, int head_major, const int64_t* __restrict__ shapes,
                   const int64_t* __restrict__ lsi, const float* __restrict__ ref_points, const float* __restrict__ offsets,
                   const float* __restrict__ logits, Dims d, float* __restrict__ out, LocalityArgs la) {
  if (STAT && win_sampled(shapes, d)) {
    win_body<0, true, REFD>(value, shapes, lsi, offsets, logits, d, out, ref_points, head_major != 0, la);
    return;
  }
  win_body<0, false, REFD>(value, shapes, lsi, offsets, logits, d, out, ref_points, head_major != 0, la);
}

#ifdef MSDA_WIN_PROF
extern "C" int msda_debug_read_prof(void* dst, int nblocks) {
  if (nblocks > kProfBlocks) nblocks = kProfBlocks;
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_win_prof), (size_t)nblocks * kProfSlots * 8, 0, hipMemcpyDeviceToHost);
}
#endif

bool win_forward_ok(const Dims& d) {
  // (the last condition keeps the work-item index, and item + 0.5, exact in float: the kernel splits it into (image,
  // tile) with a reciprocal)
  return d.D == 32 && d.P == 4 && d.L == 4 && d.Lq == d.S && d.S >= 1024 &&
         (int64_t)d.S * d.M * 128 < (int64_t)kOobOffset && d.N <= 65535 &&
         (int64_t)d.N * ((d.S + 127) / 128) < ((int64_t)1 << 22);
}

namespace {

// ---- automatic choice between msda_fwd_win and msda_fwd_lg3 (variant 0 on the encoder shape) -----------------------
// The window kernel wins while the samples of a tile stay near it and loses (up to 1.7x) when they do not, and only the
// locations know which.  Every STAT launch of the window kernel reports the far fraction of its own inputs.
//
//   per call site   The state is kept per SLOT (msda_hip_set_call_context: the MSDeformAttn module passes its own, the
//                   bare operator uses slot 0): six encoder layers of a trained checkpoint need not share one locality.
//   per launch      A slot owns a ring of kRing (counter, report) pairs; launch n of the slot uses entry n % kRing and
//                   tags its report with n -- overlapping launches (other streams, graph replays) cannot mix counts.
//   deterministic   The report of launch n is consumed at the slot's SECOND call after it, behind a wait on the event
//                   recorded after that launch (normally long complete, so the wait costs nothing; a host running far
//                   ahead is throttled to two window launches per slot).  The kernel a call takes is therefore a function
//                   of the call sequence alone -- never of whether a report happened to land in time -- and two runs of
//                   the same script give bitwise equal results.
//   pinned          No context, geometry not vouched for (window kernels need sum H_l W_l == spatial_size, include/
//                   msda_hip.h), MSDA_CTX_DETERMINISTIC, MSDA_HIP_FWD_ADAPTIVE=0 or a stream capture in progress (events
//                   cannot be waited for inside one): the gather kernel.
constexpr int kSites = 64, kRing = 4, kLag = 2;
struct Slot {
  std::mutex mu;
  int mode = 0;                          // 0: nothing known yet (window kernel, reporting), 1: window kernel, 2: gather kernel
  unsigned calls = 0;                    // auto-dispatched calls on this slot
  unsigned seq = 0;                      // STAT launches so far; launch n (1-based) uses ring entry n % kRing
  unsigned consumed = 0;                 // reports consumed so far (== seq of the last one)
  unsigned launch_call[kRing] = {};      // the call number of the launch in each ring entry
  unsigned since_probe = 0;
  double far = 0.0;
  hipEvent_t ev[kRing] = {};
};
struct LocalityState {
  volatile unsigned long long* report = nullptr;   // pinned host memory, mapped into the device: [kSites][kRing]
  unsigned long long* report_dev = nullptr;        // the device's address of it
  unsigned long long* counters = nullptr;          // device memory: [kSites][kRing]
  Slot slots[kSites];
  int last_slot = 0;                               // msda_hip_forward_locality reports the slot used last
};
constexpr int kMaxDevices = 64;
LocalityState* g_loc[kMaxDevices];
std::atomic<int> g_loc_ready[kMaxDevices];
std::mutex g_loc_mutex;

bool capturing(hipStream_t stream) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}

LocalityState* locality_state(hipStream_t stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  if (g_loc_ready[dev].load(std::memory_order_acquire) == 1) return g_loc[dev];
  if (capturing(stream)) return nullptr;        // first use allocates and clears: not inside a stream capture
  std::lock_guard<std::mutex> lock(g_loc_mutex);
  if (g_loc_ready[dev].load(std::memory_order_relaxed) == 1) return g_loc[dev];
  if (g_loc_ready[dev].load(std::memory_order_relaxed) == -1) return nullptr;
  LocalityState* st = new LocalityState();
  void* rp = nullptr;
  const size_t bytes = sizeof(unsigned long long) * kSites * kRing;
  bool ok = hipHostMalloc(&rp, bytes, hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer(reinterpret_cast<void**>(&st->report_dev), rp, 0) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&st->counters), bytes) == hipSuccess &&
            hipMemset(st->counters, 0, bytes) == hipSuccess;
  for (int i = 0; ok && i < kSites; ++i)
    for (int r = 0; ok && r < kRing; ++r) ok = hipEventCreateWithFlags(&st->slots[i].ev[r], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    g_loc_ready[dev].store(-1, std::memory_order_relaxed);
    return nullptr;                              // (leaks what was allocated: once per process, on a failing device)
  }
  st->report = static_cast<volatile unsigned long long*>(rp);
  std::memset(rp, 0, bytes);
  g_loc[dev] = st;
  g_loc_ready[dev].store(1, std::memory_order_release);
  return st;
}

struct CallContext { int slot = -1; unsigned flags = 0; bool set = false; };
thread_local CallContext t_ctx;

// what the launchers below need to know about the call that is being dispatched (set by win_forward_auto)
struct Pending { LocalityState* st = nullptr; int slot = -1; bool stat = false; };
thread_local Pending t_pending;

}  // namespace

void set_call_context(int slot, unsigned flags) {
  t_ctx.slot = slot;
  t_ctx.flags = flags;
  t_ctx.set = true;
}

void drop_call_context() {                       // a call that does not dispatch automatically still consumes its context
  t_ctx = CallContext();
  t_pending = Pending();
}

int forward_locality(double* far_fraction) {
  LocalityState* st = locality_state(nullptr);
  if (far_fraction) *far_fraction = 0.0;
  if (!st) return 0;
  Slot& sl = st->slots[st->last_slot];
  std::lock_guard<std::mutex> lock(sl.mu);
  // everything launched so far on the slot (the caller asks after a synchronisation; an event that is not complete yet
  // is waited for -- this is a diagnostic entry point, not the dispatcher)
  while (sl.consumed < sl.seq) {
    const unsigned n = sl.consumed + 1;
    if (hipEventSynchronize(sl.ev[n % kRing]) != hipSuccess) { (void)hipGetLastError(); break; }
    const unsigned long long w = st->report[st->last_slot * kRing + n % kRing];
    if ((unsigned)(w >> kStatTicketShift) == (n & 0xffffu)) {
      const unsigned long long f = w & ((1ull << kStatPairShift) - 1ull);
      const unsigned long long pairs = 16ull * ((w >> kStatPairShift) & ((1ull << (kStatTicketShift - kStatPairShift)) - 1ull));
      sl.far = pairs ? (double)f / (double)pairs : 0.0;
      sl.mode = sl.far <= kFarFractionMax ? 1 : 2;
    }
    sl.consumed = n;
  }
  if (far_fraction) *far_fraction = sl.far;
  return (int)sl.consumed;
}

// include/msda_hip.h: msda_hip_reset_call_site.  The slot forgets what its calls have reported: its next call is a "first call"
// again.  Reports of launches still in flight are dropped (their tags lie behind `consumed`).
void reset_call_site(int slot) {
  LocalityState* st = locality_state(nullptr);
  if (!st) return;
  for (int i = 0; i < kSites; ++i) {
    if (slot >= 0 && i != slot) continue;
    Slot& sl = st->slots[i];
    std::lock_guard<std::mutex> lock(sl.mu);
    sl.mode = 0;
    sl.calls = 0;
    sl.since_probe = 0;
    sl.far = 0.0;
    sl.consumed = sl.seq;
  }
}

// auto dispatch of the encoder shape: window kernel or msda_fwd_lg3?  Consumes the thread's call context.
// Backward of an encoder-shaped call made with a call context: msda_bwd_win (value and gradient windows in LDS) when the
// FORWARD calls of the same site have reported that the samples stay near their tiles, msda_bwd_tiled otherwise.  Nothing
// is launched or waited for here -- the slot's state only changes inside forward calls, so the choice is a function of
// the call sequence like the forward's.  Measured (profiles/r03_backward_window.txt): far 0.02 -> 336 us against 360,
// far 0.41 -> 1128 against 924; the lines cross near 0.06.
constexpr double kFarFractionMaxBwd = 0.05;
// ... and msda_bwd_regions (destination-side sums, no global atomics: 0.7-0.9 ms whatever the locations) from the far fraction
// on at which msda_bwd_tiled's far-corner atomics cost more than that (tiled: 0.02 -> 0.37 ms, 0.41 -> 0.91 ms, 0.93 -> 2.07 ms;
// regions 0.78 / 0.71 / 0.89 ms on the same inputs: the lines cross near 0.25; profiles/r03_backward_regions.txt)
// (round 4 sweep, tiled / regions: far 0.21: 477 / 706 us, 0.30: 655 / 688, 0.37: 801 / 664, 0.41: 922 / 642 -- they cross near 0.32)
constexpr double kFarFractionMinRegions = 0.33;
int backward_site_choice(const Dims& d) {
  static const int mode_env = [] { const char* e = std::getenv("MSDA_HIP_FWD_ADAPTIVE"); return e ? std::atoi(e) : 1; }();
  const CallContext ctx = t_ctx;
  t_ctx = CallContext();
  t_pending = Pending();
  if (mode_env == 0 || !ctx.set || ctx.slot < 0 || ctx.slot >= kSites || !(ctx.flags & 1u) || (ctx.flags & 2u) ||
      !win_backward_ok(d))
    return 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices ||
      g_loc_ready[dev].load(std::memory_order_acquire) != 1)
    return 0;                                     // no forward call has dispatched automatically on this device yet
  Slot& sl = g_loc[dev]->slots[ctx.slot];
  std::lock_guard<std::mutex> lock(sl.mu);
  if (sl.mode == 1 && sl.far <= kFarFractionMaxBwd) return 1;
  if (sl.mode == 2 && sl.far >= kFarFractionMinRegions && regions_backward_ok(d)) return 2;
  return 0;
}

bool win_forward_auto(const Dims& d, hipStream_t stream) {
  static const int mode_env = [] { const char* e = std::getenv("MSDA_HIP_FWD_ADAPTIVE"); return e ? std::atoi(e) : 1; }();
  const CallContext ctx = t_ctx;
  t_ctx = CallContext();
  t_pending = Pending();
  if (mode_env == 0 || !ctx.set || ctx.slot < 0 || ctx.slot >= kSites || !(ctx.flags & 1u) || (ctx.flags & 2u) ||
      !win_forward_ok(d) || capturing(stream))
    return false;
  LocalityState* st = locality_state(stream);
  if (!st) return false;
  Slot& sl = st->slots[ctx.slot];
  std::lock_guard<std::mutex> lock(sl.mu);
  st->last_slot = ctx.slot;
  sl.calls += 1;
  // consume the reports that are due: launched at least kLag calls ago on this slot
  while (sl.consumed < sl.seq && sl.calls - sl.launch_call[(sl.consumed + 1) % kRing] >= (unsigned)kLag) {
    const unsigned n = sl.consumed + 1;
    if (hipEventSynchronize(sl.ev[n % kRing]) != hipSuccess) { (void)hipGetLastError(); break; }
    const unsigned long long w = st->report[ctx.slot * kRing + n % kRing];
    if ((unsigned)(w >> kStatTicketShift) == (n & 0xffffu)) {     // (a launch that failed leaves its word stale: ignored)
      const unsigned long long f = w & ((1ull << kStatPairShift) - 1ull);
      const unsigned long long pairs = 16ull * ((w >> kStatPairShift) & ((1ull << (kStatTicketShift - kStatPairShift)) - 1ull));
      sl.far = pairs ? (double)f / (double)pairs : 0.0;
      const int mode = sl.far <= kFarFractionMax ? 1 : 2;
      if (mode != sl.mode) sl.since_probe = 0;
      sl.mode = mode;
    }
    sl.consumed = n;
  }
  bool window, report;
  if (sl.mode == 2) {                            // gather kernel, with every kReprobe-th call through the window kernel
    sl.since_probe += 1;
    window = report = sl.since_probe >= kReprobe;
    if (window) sl.since_probe = 0;
  } else {                                       // window kernel; while nothing is known every launch reports, afterwards
    window = true;                               // every kReportEvery-th (the counting copy + the event cost ~2 % of a launch)
    report = sl.mode == 0 || sl.since_probe == 0;
    sl.since_probe = sl.mode == 0 ? 0 : (sl.since_probe + 1) % kReportEvery;
  }
  // (a ring entry still waiting to be consumed is never reused: cannot happen with kLag < kRing)
  const bool stat = report && sl.seq - sl.consumed < (unsigned)kRing - 1u;
  t_pending.st = st;
  t_pending.slot = ctx.slot;
  t_pending.stat = stat;
  return window;
}

namespace {
// the (counter, report, sequence number) of the STAT launch that is about to be made for the pending auto-dispatched
// call, and the event to record behind it; a pinned window launch (variant 9) reports nothing
bool begin_stat_launch(LocalityArgs* la, hipEvent_t* ev) {
  *la = LocalityArgs{nullptr, nullptr, 0u, 0u};
  const Pending p = t_pending;
  t_pending = Pending();
  if (!p.st || !p.stat) return false;
  Slot& sl = p.st->slots[p.slot];
  std::lock_guard<std::mutex> lock(sl.mu);
  const unsigned n = ++sl.seq;
  sl.launch_call[n % kRing] = sl.calls;
  la->counter = p.st->counters + (p.slot * kRing + n % kRing);
  la->report = p.st->report_dev + (p.slot * kRing + n % kRing);
  la->seq = n;
  *ev = sl.ev[n % kRing];
  return true;
}
}  // namespace

// Workgroups per head (grid y; head m = blockIdx.x of the 2-D grid, i.e. -- by the observed round-robin placement of the linear
// workgroup id -- XCD m % 8 only ever touches head m's slice of `value`).  Persistent: the two workgroups a CU can hold (79 KB
// of LDS each) walk the items kk, kk + K, ... of their head -- 74.4 against 77.5 us per launch in the bench step, 70-72 against
// 73 us back to back, three A/B rounds each on one box (profiles/r03_forward_window_analysis.txt section 9; what the one-shot
// grid pays is a kernel prologue per item -- NOT dispatch time, a freed slot is refilled within 0.3 us, and NOT the partial
// last turn of the items, which neither a largest-first order nor splitting its items shortened).  The one-shot count -- a
// call with fewer items than that, or MSDA_WIN_PERSIST=0 -- is ceil(S / 128) per image, all the host knows: at least the tile
// count of any pyramid whose level 0 holds <= ~3/4 of the pixels; surplus workgroups exit at once.
static int win_workgroups_per_head(const Dims& d) {
  static const bool oneshot = ab_env_int("MSDA_WIN_PERSIST", 1) == 0;   // A/B switch
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  int K = d.N * ((d.S + 127) / 128);
  if (!oneshot) K = std::min(K, std::max(1, (2 * cus) / std::max(d.M, 1)));
  if (K < 1) K = 1;
  if (K > 65535) K = 65535;                                 // (grid y extent; a workgroup walks units kk, kk + K, ...)
  return K;
}

int launch_forward_win(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn,
                       const Dims& d, float* out, hipStream_t stream) {
  static const bool nt = ab_env_int("MSDA_WIN_NT", 0) == 1;   // A/B switch
  static std::atomic<uint64_t> lds_opted_in[3] = {{0}, {0}, {0}};
  LocalityArgs la;
  hipEvent_t ev = nullptr;
  const bool stat = begin_stat_launch(&la, &ev);
  const auto kern = stat ? (nt ? msda_fwd_win<2, true> : msda_fwd_win<0, true>) : msda_fwd_win<0, false>;
  const void* fn = reinterpret_cast<const void*>(kern);
  if (int rc = ensure_dynamic_lds(fn, kLdsBytes, lds_opted_in[stat ? (nt ? 1 : 0) : 2])) return rc;
  const int K = win_workgroups_per_head(d);
  const dim3 grid((unsigned)d.M, (unsigned)K);
  hipLaunchKernelGGL(kern, grid, dim3(kT), kLdsBytes, stream, value, shapes, lsi, loc, attn, d, out, la);
  const int rc = (int)hipGetLastError();
  if (stat && rc == 0) (void)hipEventRecord(ev, stream);
  return rc;
}

int launch_forward_win_fused(const float* value, int head_major, const int64_t* shapes, const int64_t* lsi,
                             const float* ref_points, int ref_dim, const float* offsets, const float* logits, const Dims& d,
                             float* out, hipStream_t stream) {
  static std::atomic<uint64_t> lds_opted_in[4] = {{0}, {0}, {0}, {0}};
  LocalityArgs la;
  hipEvent_t ev = nullptr;
  const bool stat = begin_stat_launch(&la, &ev);
  const auto kern = ref_dim == 2 ? (stat ? msda_fwd_win_fused<true, 2> : msda_fwd_win_fused<false, 2>)
                                 : (stat ? msda_fwd_win_fused<true, 4> : msda_fwd_win_fused<false, 4>);
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), kLdsBytes, lds_opted_in[(ref_dim == 2 ? 0 : 2) + (stat ? 1 : 0)]))
    return rc;
  const int K = win_workgroups_per_head(d);
  hipLaunchKernelGGL(kern, dim3((unsigned)d.M, (unsigned)K), dim3(kT), kLdsBytes, stream, value, head_major, shapes, lsi,
                     ref_points, offsets, logits, d, out, la);
  const int rc = (int)hipGetLastError();
  if (stat && rc == 0) (void)hipEventRecord(ev, stream);
  return rc;
}

}  // namespace msda
