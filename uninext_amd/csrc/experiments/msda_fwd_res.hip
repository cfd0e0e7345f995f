// msda_fwd_res (round 5): encoder-sized forward for ANY sampling pattern, the two coarsest pyramid levels RESIDENT in LDS.
//
// Replaces ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299 (ms_deformable_im2col_gpu_kernel) on the calls whose samples do not
// stay near their queries -- the calls the LDS-window kernel (msda_fwd_win.hip) hands back.  The gather kernel that took them so
// far (msda_fwd_lg3, msda_fwd.hip) is bound by the vector L1: 12 of a pair's 16 samples x 4 corners = 48 cache lines per
// (query, head) pair, 17 M lines per launch, with a 35 KB copy of level 3 re-made by each of 2784 one-shot workgroups.
//
// Here ONE persistent workgroup per CU owns the CU's whole LDS (160 KB) for a single (image, head): level 3 entirely and
// as many rows of level 2 as fit behind it (13 x 21 + 23 of 25 rows x 42 pixels at the 800 x 1333 pyramid; both levels whole for
// anything smaller), copied ONCE per workgroup.  Half of every pair's samples are then LDS reads; the L1 carries levels 0 and 1
// only: 32 lines per pair.  No barrier after the copy: a wave walks its own chunks of 16 queries.
//
//   lanes        8 lanes per (query, head) pair, 16 bytes of a pixel's 128 each: one b128 instruction of a wave touches 8 whole
//                cache lines.  (First version, a QUAD per pair reading a pixel as two 64-byte halves: 16 half lines per
//                instruction, 64 L1 look-ups per pair against msda_fwd_lg3's 48 -- 120 / 134 / 170 us on model / wide / uniform
//                against 106 / 122 / 118: the L1 path is bound by line look-ups, not by bytes.)  Lane k of each QUAD prepares
//                the pair's four samples of LEVEL k (both quads of the group do, identically); a prepared sample reaches the
//                quad by DPP quad_perm broadcasts -- no sample records in LDS.
//   levels 0, 1  raw buffer loads (dead corners: an offset past num_records reads zeros), four samples (16 loads) in flight
//                per lane; levels 2, 3: ds_read_b128 from the resident copy (dead corners: an all-zero slot), issued
//                between the global samples -- the two pipes run side by side.
//   the rows of level 2 that did not fit (the image's top and bottom strips): corners there are left out of the LDS pass
//                and gathered through the L1 afterwards, one such sample per pair and step (the far-sample idiom of
//                msda_fwd_win); every corner is added exactly once.
//   arithmetic   make_sample (msda_common.hpp) = the reference's per-sample arithmetic; the sum of a pair runs over its
//                samples point by point, levels in the order 2, 0, 3, 1 -- fp32 summation order is the only
//                difference from the other forward kernels.
#include <algorithm>
#include <type_traits>

#include "../msda_common.hpp"

namespace msda {
namespace {

#ifndef RES_THREADS
#define RES_THREADS 512
#endif
constexpr int kResThreads = RES_THREADS;                             // 8 waves, 2 per SIMD (A/B: 12 waves gain nothing and cost the register room)
constexpr int kResWaves = kResThreads / 64;
constexpr int kResZeroOff = 128;                               // 128 bytes of zeros: the dead corners of LDS samples
constexpr int kResPixOff = 256;                                // resident pixels from here on, 128 bytes each
constexpr int kResLdsBytes = 160 * 1024;
constexpr int kResSlots = (kResLdsBytes - kResPixOff) / 128;   // 1278

typedef float v2f __attribute__((ext_vector_type(2)));        // v_pk_fma_f32
typedef const f32x4 __attribute__((address_space(3)))* lds4;

template <int SRC>
__device__ __forceinline__ uint32_t qb(uint32_t v) {   // value held by lane SRC of this lane's quad
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, SRC * 0x55, 0xF, 0xF, true);
}
template <int SRC>
__device__ __forceinline__ float qbf(float v) { return __uint_as_float(qb<SRC>(__float_as_uint(v))); }
template <int CTRL>
__device__ __forceinline__ int dppi(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) {   // (a & 0xffffff) * (b & 0xffffff) + c
  uint32_t r;
  asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

struct Corners { f32x4 c1, c2, c3, c4; };   // the lane's 16 bytes of a sample's four corner pixels
struct Smp { float w[4]; uint32_t o[4]; }; // corner weights x attention weight; byte offsets / LDS addresses of the corners

__global__ void __launch_bounds__(kResThreads)
msda_fwd_res(const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
             const float* __restrict__ loc, const float* __restrict__ attn, Dims d, int K, int seq, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(128))) char smem[];
  int* tab = reinterpret_cast<int*>(smem);   // H[4], W[4], start[4]
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x;
  if (tid < 4) {
    tab[tid] = (int)shapes[2 * tid];
    tab[4 + tid] = (int)shapes[2 * tid + 1];
    tab[8 + tid] = (int)lsi[tid];
  }
  if (tid >= 64 && tid < 96) reinterpret_cast<int*>(smem + kResZeroOff)[tid - 64] = 0;
  __syncthreads();
  const int H2 = __builtin_amdgcn_readfirstlane(tab[2]), W2 = __builtin_amdgcn_readfirstlane(tab[6]);
  const int S2 = __builtin_amdgcn_readfirstlane(tab[10]), S3 = __builtin_amdgcn_readfirstlane(tab[11]);
  const int n3 = __builtin_amdgcn_readfirstlane(tab[3]) * __builtin_amdgcn_readfirstlane(tab[7]);
  const bool fits3 = n3 <= kResSlots;
  const int n3r = fits3 ? n3 : 0;                                        // resident pixels of level 3: all or none
  const int R2 = fits3 ? min(H2, (kResSlots - n3r) / max(W2, 1)) : 0;    // resident rows of level 2 ...
  const int r0 = (H2 - R2) >> 1;                                         // ... the central band [r0, r0 + R2)
  const int nfill = n3r + R2 * W2;

  // seq != 0: the grid is M x K workgroups and every workgroup takes the images one after the other -- with one head per XCD
  // (workgroup id mod 8 = head when M = 8) an XCD's L2 then serves ONE (image, head) slice of `value` at a time (2.8 MB at the
  // bench shape, of 4 MB); with the images side by side it is 5.7 MB and the L2 thrashes
  const int m = blockIdx.x % d.M, kk = blockIdx.x / d.M;
  const uint32_t pixB = (uint32_t)d.M * 128u;
  const uint32_t hoff = (uint32_t)m * 128u;

  const int lane = tid & 63, wv = tid >> 6, grp = lane >> 3, k = lane & 3;
  const uint32_t c0 = (uint32_t)(lane & 7) * 16u;
  const int nchunks = (d.Lq + 7) >> 3;                   // chunks of 8 queries: one per 8-lane group of a wave
  const int cstride = K * kResWaves;
  // ---- constants of the lane's level -------------------------------------------------------------------------------------
  const int Hk = tab[k], Wk = tab[4 + k], Sk = tab[8 + k];
  const bool glob = k < 2;                                             // levels 0, 1: through the L1
  const int rlo = k == 2 ? r0 : 0;                                     // rows [rlo, rlo + rn) of the level can be read
  const uint32_t rn = (uint32_t)(k == 2 ? R2 : (k == 3 && !fits3) ? 0 : Hk);
  const uint32_t stride = glob ? pixB : 128u;
  const uint32_t rowstride = mad_u24((uint32_t)Wk, stride, 0u);
  // address of the level's pixel (0, 0): a byte offset into the image's `value`, or an LDS address (rows above the band
  // are never dereferenced; the arithmetic wraps consistently)
  const uint32_t A0 = glob ? mad_u24((uint32_t)Sk, pixB, 0u)
                           : smem_base + kResPixOff + (uint32_t)((k == 2 ? n3r - r0 * W2 : 0) * 128);
  const uint32_t dead = glob ? kOobOffset : smem_base + kResZeroOff;

  for (int b = seq ? 0 : (int)blockIdx.y, b_end = seq ? d.N : b + 1; b < b_end; ++b) {
  __syncthreads();                                                     // (every wave is done with the previous image's copy)
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(value) + (int64_t)b * d.S * d.M * 32, 0, (int)((uint32_t)d.S * pixB), 0x00020000);
  const float* loc_b = loc + (int64_t)b * d.Lq * d.M * 32;
  const float* attn_b = attn + (int64_t)b * d.Lq * d.M * 16;
  float* out_b = out + (int64_t)b * d.Lq * d.M * 32;

  f32x4 A, B, Wt;           // this lane's level: (x, y) of points 0, 1 / 2, 3; attention weights
  uint32_t pair;
  bool live;
  auto fetch = [&](int ch, f32x4& fa, f32x4& fb, f32x4& fw, uint32_t& pr, bool& lv) __attribute__((always_inline)) {
    const int q = ch * 8 + grp;
    lv = ch < nchunks && q < d.Lq;
    pr = (uint32_t)((lv ? q : 0) * d.M + m);
    fa = f32x4{0.f, 0.f, 0.f, 0.f}; fb = fa; fw = fa;
    if (lv) {
      const f32x4* lp = reinterpret_cast<const f32x4*>(loc_b + (size_t)pr * 32 + k * 8);
      fa = __builtin_nontemporal_load(lp);
      fb = __builtin_nontemporal_load(lp + 1);
      fw = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(attn_b + (size_t)pr * 16 + k * 4));
    }
  };
  int chunk = kk * kResWaves + wv;
  fetch(chunk, A, B, Wt, pair, live);      // travels under the copy

  // ---- the resident copy: level 3, then rows [r0, r0 + R2) of level 2 ----------------------------------------------------
  {
    constexpr int kIt = (kResSlots * 8 + kResThreads - 1) / kResThreads;
    f32x4 v[kIt];
#pragma unroll
    for (int t = 0; t < kIt; ++t) {
      const int i = (tid >> 3) + t * (kResThreads / 8);
      const int pix = i < n3r ? S3 + i : S2 + r0 * W2 + (i - n3r);
      const uint32_t off = i < nfill ? (uint32_t)pix * pixB + (uint32_t)(tid & 7) * 16u : kOobOffset;
      v[t] = buffer_load_f32x4(rsrc, off, hoff);
    }
#pragma unroll
    for (int t = 0; t < kIt; ++t) {
      const int i = (tid >> 3) + t * (kResThreads / 8);
      if (i < nfill) *reinterpret_cast<f32x4*>(smem + kResPixOff + i * 128 + (tid & 7) * 16) = v[t];
    }
  }

  __syncthreads();                                                     // the copy is complete

  // ---- this lane's sample of point p of a chunk: corner weights x attention weight, corner addresses -----------------------
  auto prepare = [&](int p, const f32x4& rA, const f32x4& rB, const f32x4& rW, bool lv, uint32_t& fixm) __attribute__((always_inline)) {
    Smp r;
    const float lx = p < 2 ? rA[2 * p] : rB[2 * p - 4], ly = p < 2 ? rA[2 * p + 1] : rB[2 * p - 3], a = rW[p];
    const Sample<float> sm = make_sample<float>(lx, ly, Hk, Wk);
    const bool t_in = lv && sm.in_range && sm.h_low >= 0, b_in = lv && sm.in_range && sm.h_low + 1 <= Hk - 1;
    const bool lf = sm.w_low >= 0, rt = sm.w_low + 1 <= Wk - 1;
    const bool t_res = (uint32_t)(sm.h_low - rlo) < rn, b_res = (uint32_t)(sm.h_low + 1 - rlo) < rn;
    const bool t = t_in && t_res, bt = b_in && b_res;
    fixm |= ((t_in && !t_res) || (b_in && !b_res)) ? (1u << p) : 0u;   // (never on levels 0, 1)
    const float wa = sm.hh * a, wb = sm.lh * a;
    r.w[0] = (t && lf) ? wa * sm.hw : 0.f;
    r.w[1] = (t && rt) ? wa * sm.lw : 0.f;
    r.w[2] = (bt && lf) ? wb * sm.hw : 0.f;
    r.w[3] = (bt && rt) ? wb * sm.lw : 0.f;
    const int cy = max(sm.h_low, 0), cx = max(sm.w_low, 0);
    const uint32_t off = mad_u24(mad_u24((uint32_t)cy, (uint32_t)Wk, (uint32_t)cx), stride, A0);
    const uint32_t dx = sm.w_low >= 0 ? stride : 0u, dy = sm.h_low >= 0 ? rowstride : 0u;
    r.o[0] = (t && lf) ? off : dead;
    r.o[1] = (t && rt) ? off + dx : dead;
    r.o[2] = (bt && lf) ? off + dy : dead;
    r.o[3] = (bt && rt) ? off + dy + dx : dead;
    __builtin_amdgcn_sched_barrier(0);
    return r;
  };
  v2f a0, a1;                                            // the lane's four channels of the chunk in progress
  auto g_issue = [&](auto ltag, const Smp& sp, Corners& c) __attribute__((always_inline)) {
    constexpr int LV = decltype(ltag)::value;
    const uint32_t o1 = qb<LV>(sp.o[0]) + c0, o2 = qb<LV>(sp.o[1]) + c0;
    const uint32_t o3 = qb<LV>(sp.o[2]) + c0, o4 = qb<LV>(sp.o[3]) + c0;
#if defined(RES_NOG)          // timing only (wrong results): no global loads at all
    (void)o1; (void)o2; (void)o3; (void)o4;
    c.c1 = c.c2 = c.c3 = c.c4 = f32x4{0.f, 0.f, 0.f, 0.f};
#else
    c.c1 = buffer_load_f32x4(rsrc, o1, hoff); c.c2 = buffer_load_f32x4(rsrc, o2, hoff);
    c.c3 = buffer_load_f32x4(rsrc, o3, hoff); c.c4 = buffer_load_f32x4(rsrc, o4, hoff);
#endif
    __builtin_amdgcn_sched_barrier(0);
  };
  auto add_px = [&](float w, const f32x4& px) __attribute__((always_inline)) {
    const v2f W = {w, w};
    a0 = __builtin_elementwise_fma(W, v2f{px[0], px[1]}, a0); a1 = __builtin_elementwise_fma(W, v2f{px[2], px[3]}, a1);
  };
  auto consume = [&](auto ltag, const Smp& sp, const Corners& c) __attribute__((always_inline)) {
    constexpr int LV = decltype(ltag)::value;
    add_px(qbf<LV>(sp.w[0]), c.c1); add_px(qbf<LV>(sp.w[1]), c.c2);
    add_px(qbf<LV>(sp.w[2]), c.c3); add_px(qbf<LV>(sp.w[3]), c.c4);
    asm volatile("" : "+v"(a0), "+v"(a1));                           // pins the FMAs here
    __builtin_amdgcn_sched_barrier(0);
  };
  auto l_smp = [&](auto ltag, const Smp& sp) __attribute__((always_inline)) {   // an LDS sample: 4 reads, 8 packed FMAs
    constexpr int LV = decltype(ltag)::value;
    const uint32_t p1 = qb<LV>(sp.o[0]) + c0, p2 = qb<LV>(sp.o[1]) + c0, p3 = qb<LV>(sp.o[2]) + c0, p4 = qb<LV>(sp.o[3]) + c0;
    Corners c;
#ifdef RES_NOL                // timing only: no LDS reads
    (void)p1; (void)p2; (void)p3; (void)p4;
    c.c1 = c.c2 = c.c3 = c.c4 = f32x4{0.f, 0.f, 0.f, 0.f};
#else
    c.c1 = *reinterpret_cast<lds4>((uintptr_t)p1); c.c2 = *reinterpret_cast<lds4>((uintptr_t)p2);
    c.c3 = *reinterpret_cast<lds4>((uintptr_t)p3); c.c4 = *reinterpret_cast<lds4>((uintptr_t)p4);
#endif
    __builtin_amdgcn_sched_barrier(0);
    add_px(qbf<LV>(sp.w[0]), c.c1); add_px(qbf<LV>(sp.w[1]), c.c2); add_px(qbf<LV>(sp.w[2]), c.c3); add_px(qbf<LV>(sp.w[3]), c.c4);
    asm volatile("" : "+v"(a0), "+v"(a1));
    __builtin_amdgcn_sched_barrier(0);
  };
  using L0 = std::integral_constant<int, 0>; using L1 = std::integral_constant<int, 1>;
  using L2 = std::integral_constant<int, 2>; using L3 = std::integral_constant<int, 3>;

  // ---- corners of level-2 (level-3) samples in rows that are not resident: through the L1, one sample per pair and step ------
  auto fix_pass = [&](uint32_t fixm) __attribute__((always_inline)) {
    {
      uint32_t fm = k >= 2 ? fixm << (4 * (k - 2)) : 0u;            // the pair's: bit 4 * (level - 2) + point
      fm |= (uint32_t)dppi<0xB1>((int)fm);                             // quad_perm [1,0,3,2]
      fm |= (uint32_t)dppi<0x4E>((int)fm);                             // quad_perm [2,3,0,1]
#ifdef RES_NOFIX
      fm = 0;
#endif
      while (__ballot(fm != 0u)) {
        const bool has = fm != 0u;
        const int idx = has ? __builtin_ctz(fm) : 0;
        fm &= fm - 1u;
        const int ps = idx & 3;
#ifdef RES_PIPELINED
        const int fl = 2 + (idx >> 2);                                 // the sample's level
        // its location and weight come from memory again (L2 hits, a cold path): keeping the chunk's 12 raw registers alive
        // for this loop would not fit beside the two load buffers
        float fx = 0.f, fy = 0.f, fa = 0.f;                            // pairs with nothing left run along on finite stand-ins
        if (has) {
          const float2 xy = *reinterpret_cast<const float2*>(loc_b + (size_t)pair * 32 + fl * 8 + ps * 2);
          fx = xy.x; fy = xy.y;
          fa = attn_b[(size_t)pair * 16 + fl * 4 + ps];
        }
        const int fH = tab[fl], fW = tab[4 + fl], fS = tab[8 + fl];
        const int flo = fl == 2 ? r0 : 0;
        const uint32_t fn = (uint32_t)(fl == 2 ? R2 : fits3 ? fH : 0);
#else
        // the sample's location, weight and level constants come from the lane that prepared it
        const int src = ((lane & ~3) | (2 + (idx >> 2))) << 2;         // for ds_bpermute
        const bool c1 = (ps & 1) != 0, c2 = (ps & 2) != 0;
        auto sel4 = [&](float v0, float v1, float v2, float v3) __attribute__((always_inline)) {
          const float t0 = c1 ? v1 : v0, t1 = c1 ? v3 : v2;
          return (int)__float_as_uint(c2 ? t1 : t0);
        };
        const uint32_t hm = has ? 0xffffffffu : 0u;                    // pairs with nothing left run along on finite stand-ins
        const float fx = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, sel4(A[0], A[2], B[0], B[2])) & hm);
        const float fy = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, sel4(A[1], A[3], B[1], B[3])) & hm);
        const float fa = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, sel4(Wt[0], Wt[1], Wt[2], Wt[3])) & hm);
        const int fW = __builtin_amdgcn_ds_bpermute(src, Wk), fH = __builtin_amdgcn_ds_bpermute(src, Hk);
        const int fS = __builtin_amdgcn_ds_bpermute(src, Sk), flo = __builtin_amdgcn_ds_bpermute(src, rlo);
        const uint32_t fn = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)rn);
#endif
        const Sample<float> sm = make_sample<float>(fx, fy, fH, fW);
        const bool t_in = has && sm.in_range && sm.h_low >= 0, b_in = has && sm.in_range && sm.h_low + 1 <= fH - 1;
        const bool lf = sm.w_low >= 0, rt = sm.w_low + 1 <= fW - 1;
        const bool t = t_in && !((uint32_t)(sm.h_low - flo) < fn), bt = b_in && !((uint32_t)(sm.h_low + 1 - flo) < fn);
        const float wa = sm.hh * fa, wb = sm.lh * fa;
        const float w1 = (t && lf) ? wa * sm.hw : 0.f, w2 = (t && rt) ? wa * sm.lw : 0.f;
        const float w3 = (bt && lf) ? wb * sm.hw : 0.f, w4 = (bt && rt) ? wb * sm.lw : 0.f;
        const int cy = max(sm.h_low, 0), cx = max(sm.w_low, 0);
        const uint32_t off = mad_u24(mad_u24((uint32_t)cy, (uint32_t)fW, (uint32_t)(fS + cx)), pixB, c0);
        const uint32_t dx = sm.w_low >= 0 ? pixB : 0u, dy = sm.h_low >= 0 ? mad_u24((uint32_t)fW, pixB, 0u) : 0u;
        const uint32_t o1 = (t && lf) ? off : kOobOffset, o2 = (t && rt) ? off + dx : kOobOffset;
        const uint32_t o3 = (bt && lf) ? off + dy : kOobOffset, o4 = (bt && rt) ? off + dy + dx : kOobOffset;
        Corners c;
        c.c1 = buffer_load_f32x4(rsrc, o1, hoff); c.c2 = buffer_load_f32x4(rsrc, o2, hoff);   // (pairs with nothing left: out of range)
        c.c3 = buffer_load_f32x4(rsrc, o3, hoff); c.c4 = buffer_load_f32x4(rsrc, o4, hoff);
        add_px(w1, c.c1); add_px(w2, c.c2); add_px(w3, c.c3); add_px(w4, c.c4);
      }
    }

  };
#ifdef RES_PIPELINED
  // Software pipeline over HALF chunks (points 0-1 / points 2-3: 16 global loads each, buffers X / Y): the loads of the next
  // half are requested before the current half is consumed, ACROSS chunk boundaries -- a wave always has 16 to 32 loads in
  // flight while it computes.  Without it the waves of a CU fall into step (everybody requests, everybody waits for the L1
  // queue to drain, everybody computes): the launch took (vector work) + (L1 time), 46 + 43 us, with nothing overlapped.
  f32x4 nA, nB, nW;                                      // the chunk after this one (travels a whole iteration ahead)
  uint32_t npair, fixmask = 0, nfixmask = 0;
  bool nlive;
  fetch(chunk + cstride, nA, nB, nW, npair, nlive);
  Corners x0, x1, x2, x3, y0, y1, y2, y3;
  Smp s0 = prepare(0, A, B, Wt, live, fixmask), s1 = prepare(1, A, B, Wt, live, fixmask);
  g_issue(L0{}, s0, x0); g_issue(L1{}, s0, x1); g_issue(L0{}, s1, x2); g_issue(L1{}, s1, x3);
  while (chunk < nchunks) {                                            // wave-uniform
    const Smp s2 = prepare(2, A, B, Wt, live, fixmask), s3 = prepare(3, A, B, Wt, live, fixmask);
    g_issue(L0{}, s2, y0); g_issue(L1{}, s2, y1); g_issue(L0{}, s3, y2); g_issue(L1{}, s3, y3);
    a0 = v2f{0.f, 0.f}; a1 = v2f{0.f, 0.f};
    l_smp(L2{}, s0); l_smp(L3{}, s0); l_smp(L2{}, s1); l_smp(L3{}, s1);
    consume(L0{}, s0, x0); consume(L1{}, s0, x1); consume(L0{}, s1, x2); consume(L1{}, s1, x3);
    // the next chunk's first half (a chunk past the end: nlive is false everywhere, every offset dead)
    s0 = prepare(0, nA, nB, nW, nlive, nfixmask); s1 = prepare(1, nA, nB, nW, nlive, nfixmask);
    g_issue(L0{}, s0, x0); g_issue(L1{}, s0, x1); g_issue(L0{}, s1, x2); g_issue(L1{}, s1, x3);
    l_smp(L2{}, s2); l_smp(L3{}, s2); l_smp(L2{}, s3); l_smp(L3{}, s3);
    consume(L0{}, s2, y0); consume(L1{}, s2, y1); consume(L0{}, s3, y2); consume(L1{}, s3, y3);

    fix_pass(fixmask);
    if (live)     // the group writes the pair's 128 contiguous bytes
      __builtin_nontemporal_store(f32x4{a0.x, a0.y, a1.x, a1.y}, reinterpret_cast<f32x4*>(out_b + ((size_t)pair * 32u + (c0 >> 2))));
    A = nA; B = nB; Wt = nW; pair = npair; live = nlive; fixmask = nfixmask; nfixmask = 0;
    chunk += cstride;
    fetch(chunk + cstride, nA, nB, nW, npair, nlive);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the loads requested for the chunk past the end)
#else
  // One chunk at a time, point-major: the global samples of point p + 2 are requested when point p's have been consumed
  // (8 to 16 loads in flight per lane), the LDS samples in between.  (RES_PIPELINED -- the same half-chunk buffers carried ACROSS
  // chunk boundaries, 16 to 32 loads in flight at all times -- is 10-15 % slower: 111 / 124 / 119 us against 97 / - / 108 without
  // the fix-up pass.)
  while (chunk < nchunks) {                                            // wave-uniform
    f32x4 nA, nB, nW;
    uint32_t npair, fixmask = 0;
    bool nlive;
    fetch(chunk + cstride, nA, nB, nW, npair, nlive);
    Corners g0, g1, g2, g3;
    a0 = v2f{0.f, 0.f}; a1 = v2f{0.f, 0.f};
    const Smp s0 = prepare(0, A, B, Wt, live, fixmask);
    g_issue(L0{}, s0, g0); g_issue(L1{}, s0, g1);
    const Smp s1 = prepare(1, A, B, Wt, live, fixmask);
    g_issue(L0{}, s1, g2); g_issue(L1{}, s1, g3);
    const Smp s2 = prepare(2, A, B, Wt, live, fixmask);
    l_smp(L2{}, s0); consume(L0{}, s0, g0); g_issue(L0{}, s2, g0);
    l_smp(L3{}, s0); consume(L1{}, s0, g1); g_issue(L1{}, s2, g1);
    const Smp s3 = prepare(3, A, B, Wt, live, fixmask);
    l_smp(L2{}, s1); consume(L0{}, s1, g2); g_issue(L0{}, s3, g2);
    l_smp(L3{}, s1); consume(L1{}, s1, g3); g_issue(L1{}, s3, g3);
    l_smp(L2{}, s2); consume(L0{}, s2, g0);
    l_smp(L3{}, s2); consume(L1{}, s2, g1);
    l_smp(L2{}, s3); consume(L0{}, s3, g2);
    l_smp(L3{}, s3); consume(L1{}, s3, g3);
    fix_pass(fixmask);
    if (live)     // the group writes the pair's 128 contiguous bytes
      __builtin_nontemporal_store(f32x4{a0.x, a0.y, a1.x, a1.y}, reinterpret_cast<f32x4*>(out_b + ((size_t)pair * 32u + (c0 >> 2))));
    A = nA; B = nB; Wt = nW; pair = npair; live = nlive;
    chunk += cstride;
  }
#endif
  }   // images
}

int cu_count() {
  static std::atomic<int> cached{0};
  int n = cached.load(std::memory_order_relaxed);
  if (n > 0) return n;
  int dev = 0;
  n = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
  if (n <= 0) n = 256;
  cached.store(n, std::memory_order_relaxed);
  return n;
}

}  // namespace

// The host knows spatial_size only: levels 2 + 3 of the usual stride-8..64 pyramid are 1/17 of it (1323 of 22223 pixels
// at 800 x 1333, the largest shape whose two coarse levels -- all but two rows -- fit the 1278 slots).  Beyond that the kernel
// would still be correct (whatever is not resident goes through the L1), but msda_fwd_lg3 with its two workgroups per CU is
// the better gather kernel then.
bool res_forward_ok(const Dims& d) {
  return d.D == 32 && d.P == 4 && d.L == 4 && d.Lq >= 4096 && d.S <= 23000 &&
         (int64_t)d.S * d.M * 128 < (int64_t)kOobOffset && d.N <= 65535;
}

int launch_forward_res(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn,
                       const Dims& d, float* out, hipStream_t stream) {
  static std::atomic<uint64_t> lds_opted_in{0};
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(msda_fwd_res), kResLdsBytes, lds_opted_in)) return rc;
  const int nchunks = (d.Lq + 7) / 8, cus = cu_count();
#ifdef RES_NOSEQ
  const int seq = 0;
#else
  const int seq = d.N > 1 && d.M < cus;                  // images one after the other (see the kernel)
#endif
  int K = cus / (seq ? d.M : d.N * d.M);                 // workgroups per head (per (image, head)): one workgroup per CU in total
  K = std::max(1, std::min(K, (nchunks + kResWaves - 1) / kResWaves));
  dim3 grid((unsigned)(d.M * K), (unsigned)(seq ? 1 : d.N));
  hipLaunchKernelGGL(msda_fwd_res, grid, dim3(kResThreads), kResLdsBytes, stream, value, shapes, lsi, loc, attn, d, K, seq, out);
  return (int)hipGetLastError();
}

}  // namespace msda
