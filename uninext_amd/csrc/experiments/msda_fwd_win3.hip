// msda_fwd_win3 -- MSDeformAttn forward for encoder-style calls (Lq == S), third generation of the LDS-window kernel
// (msda_fwd_win.hip, msda_fwd_win2.hip).  fp32, D = 32, L = P = 4.  gfx950 only.  Replaces, for these calls, the work of
// ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299.
//
// What the timelines of the first two generations said (profiles/r03_forward_window_analysis.txt): a work item is a chain
// of waits -- locations 1.7 us, the slowest wave at two barriers 2.2 us, window DMA 1.0 us -- around ~5 us of issue-bound
// work, two such chains fit a CU, and the vector ALUs end up 55 % busy.  This kernel takes the waits out of the chain:
//
//   persistent   ONE 768-thread workgroup per CU walks items kk, kk + K, ...; waves 0..7 take the 8 rows of the level-0
//                tile, waves 8..11 the tile's queries of levels 1..3 (64 quads: one pass at the R50 shapes).
//   two window   the LDS holds two sets of value windows (2 x 74 KB).  While the gather of item i reads set A, item i + 1
//   sets         goes through its whole start-up on the side: its locations are requested before the first half of the
//                gather (levels 0-1), have arrived by the middle, where their placement sums are added, ONE barrier later
//                the window origins are known and the window DMA into set B is issued, which lands under the second half
//                (levels 2-3), the far samples and the stores.  Nobody waits for memory; two barriers per item.
//   far samples  after the gather (their loads queue behind the DMA that was issued long before).
//
// Unchanged from msda_fwd_win2: the exact tile partition of the S queries, window sizes and placement by the mean top-left
// corner of the tile's own in-range samples, LDS-DMA staging with out-of-image slots as zeros, lane roles (a quad per
// (query, head) pair, lane k prepares point k of every level and accumulates the 16-byte pieces k and k + 4), the
// bank-conflict-free read classes, DPP quad broadcasts, the far path (an in-range sample with a corner outside its window
// takes raw buffer loads: correctness never depends on where the windows are), and the reference's sample arithmetic.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "../msda_common.hpp"

namespace msda {
namespace {

constexpr int kL0Waves = 8, kRestWaves = 4, kWaves = kL0Waves + kRestWaves, kT = kWaves * 64;
constexpr int kRestQuads = kRestWaves * 16;
constexpr int kTH = 8, kTW = 16;
static_assert(kTH == kL0Waves, "wave = row of the level-0 tile");
constexpr int kWH[4] = {14, 10, 8, 7};
constexpr int kWW[4] = {22, 14, 10, 8};                         // even: slot parity == column parity in every row
constexpr int kBase[5] = {0, 312, 456, 536, 592};               // first slot of each window, multiples of 8: a 1 KB
                                                                // DMA chunk (8 slots) never straddles two levels
static_assert(kBase[1] >= kWH[0] * kWW[0] && kBase[2] >= kBase[1] + kWH[1] * kWW[1] &&
              kBase[3] >= kBase[2] + kWH[2] * kWW[2] && kBase[4] >= kBase[3] + kWH[3] * kWW[3], "window table");
static_assert(kBase[1] % 8 == 0 && kBase[2] % 8 == 0 && kBase[3] % 8 == 0 && kBase[4] % 8 == 0, "DMA chunks / parity");
constexpr int kSlots = kBase[4];
constexpr int kBufBytes = kSlots * 128;                         // one set of windows
constexpr int kZeroOff = 2 * kBufBytes;                         // all-zero region: target of dead / far samples
constexpr int kZeroBytes = kWW[0] * 128 + 256;                  // a bottom-row read lands at most one level-0 row further
static_assert(kBufBytes % 256 == 0 && kZeroOff % 256 == 0, "slot parity by address bit 7 in both sets and the zero region");
struct Meta {
  int sum[2][4][4];                                             // per window set and level: sum dx, sum dy, count, - (placement)
  int lvl[4][4];                                                // per level: H, W, first pixel, - (far path: level picked per quad)
  int qtab[2][kRestQuads];                                      // per window set: the item's first kRestQuads queries of levels 1..3 (-1: none)
  int nrest[2][4];                                              // ... and how many there are
};
constexpr int kMetaOff = kZeroOff + kZeroBytes;
constexpr int kLdsBytes = kMetaOff + ((sizeof(Meta) + 15) / 16) * 16;
static_assert(kLdsBytes <= 160 * 1024, "one workgroup per CU");

typedef const f32x4 __attribute__((address_space(3)))* lds4;
typedef float v2f __attribute__((ext_vector_type(2)));        // packed fp32 math: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32

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
  asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));   // volatile: stays out of branches
  return r;
}
// the same with a wave-uniform multiplier in a scalar register (msda_fwd_win2.hip)
__device__ __forceinline__ uint32_t mad_u24_s(uint32_t a, uint32_t b_uniform, uint32_t c) {
  uint32_t r;
  asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
  return r;
}
__device__ __forceinline__ uint32_t mul_u24_s(uint32_t a, uint32_t b_uniform) {
  uint32_t r;
  asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "s"(b_uniform), "v"(a));
  return r;
}
__device__ __forceinline__ int cvt_i32(float f) {   // saturating, NaN -> 0 (a C++ cast of a huge float is undefined)
  int r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
  return r;
}
// A wave-uniform value computed on the vector ALU into a SCALAR register; as inline asm with the wait states the hazard
// recogniser cannot see inside it (msda_fwd_win2.hip)
__device__ __forceinline__ int to_sgpr(int v) {
  int r;
  asm volatile("s_nop 1\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 4" : "=s"(r) : "v"(v));
  return r;
}
template <typename T>
__device__ __forceinline__ T sel4(bool b0, bool b1, T a0, T a1, T a2, T a3) {
  const T t = b0 ? a1 : a0, u = b0 ? a3 : a2;
  return b1 ? u : t;
}

// Phase timestamps (profiling builds only: -DMSDA_WIN3_PROF; tools/win3_prof.py): lane 0 of EVERY wave writes the 100 MHz
// real-time counter at each phase boundary of the workgroup's THIRD item (steady state).
#ifdef MSDA_WIN3_PROF
constexpr int kProfBlocks = 512, kProfSlots = 16;
__device__ unsigned long long g_win3_prof[kProfBlocks * kWaves * kProfSlots];
#define W3_STAMP(i)                                                                                          \
  do {                                                                                                       \
    const unsigned blk_ = blockIdx.y * gridDim.x + blockIdx.x;                                               \
    if ((threadIdx.x & 63) == 0 && body && item == kk + 2 * K && blk_ < (unsigned)kProfBlocks)               \
      g_win3_prof[(blk_ * kWaves + (threadIdx.x >> 6)) * kProfSlots + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define W3_STAMP(i) do { } while (0)
#endif

// Workgroup barrier for LDS traffic only (__syncthreads() waits for vmcnt(0) too: the locations and the window DMA are
// meant to stay in flight across it).  LDS operations of a CU complete in order: lgkmcnt(0) is enough.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct Smp {      // one prepared NEAR sample (dead and far samples: zero weights, addresses in the zero region)
  v2f wT, wB;     // corner weights (first-top, second-top), (first-bottom, second-bottom); "first" = the pixel whose slot parity this quad reads first
  uint32_t aF, aS;   // LDS byte addresses of the first / second pixel of the top row
};

}  // namespace

__global__ void __launch_bounds__(kT, 3)
msda_fwd_win3(const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
              const float* __restrict__ loc, const float* __restrict__ attn, Dims d, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Meta& mt = *reinterpret_cast<Meta*>(smem + kMetaOff);
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x;
  int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = d.M;
  const int m = blockIdx.x, kk = blockIdx.y, K = gridDim.y;   // workgroup kk of K on head m

  // ---- launch constants straight from the shape tensors (uniform addresses: scalar loads) --------------------------
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

  // the all-zero region, the placement sums, the level table of the far path (visible after the first barrier)
  if (tid < kZeroBytes / 16) *reinterpret_cast<f32x4*>(smem + kZeroOff + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  static_assert(kZeroBytes / 16 <= 256, "zero fill by the first four waves");
  if (tid >= 256 && tid < 288) (&mt.sum[0][0][0])[tid - 256] = 0;
  if (tid >= 320 && tid < 324) {
    const bool t0 = (tid & 1) != 0, t1 = (tid & 2) != 0;
    *reinterpret_cast<int4*>(&mt.lvl[tid & 3][0]) = make_int4(sel4(t0, t1, lvH[0], lvH[1], lvH[2], lvH[3]), sel4(t0, t1, lvW[0], lvW[1], lvW[2], lvW[3]),
                                                              sel4(t0, t1, lvS[0], lvS[1], lvS[2], lvS[3]), 0);
  }

  const uint32_t pixB = (uint32_t)M * 128u;                // bytes from a pixel of head m to the next one
  const uint32_t hoff = (uint32_t)m * 128u;
  bool first_table = true;                                 // the first item's table of level 1..3 queries: in the first iteration, before barrier B

  // ---- the pipeline: an iteration gathers item `item` (body) and starts up item `nxt` (more); the first iteration only
  // starts up, the last only gathers ----------------------------------------------------------------------------------
  bool body = false;
  int item = 0, nxt = kk;
  int cb = 1;                                              // the window set of `item` (the first item's windows go to set 0)
  bool live = false;                                       // `item`: this quad's (query, head) pair ...
  uint32_t pair = 0;
  v2f lc[4];                                               // ... locations and weights of point k on the four levels
  float sa[4];
  int ogx[4] = {0, 0, 0, 0}, ogy[4] = {0, 0, 0, 0};         // window origins of `item`
  int npass = 1;
#pragma unroll
  for (int l = 0; l < 4; ++l) { lc[l] = v2f{0.f, 0.f}; sa[l] = 0.f; }

  for (;;) {
    // whatever the optimiser can prove invariant in this loop it hoists in front of it and spills: per-lane constants are
    // re-derived per iteration and the level constants pass through an empty asm, in place (msda_fwd_win2.hip)
    int ntiles_ = ntiles, TX_ = TX;
    asm volatile("" : "+s"(wv), "+s"(ntiles_), "+s"(TX_));
#pragma unroll
    for (int l = 0; l < 4; ++l) asm volatile("" : "+s"(lvH[l]), "+s"(lvW[l]), "+s"(lvS[l]));
    int ln;                                                  // lane of the wave
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const int pq = ln >> 2, k = ln & 3;                      // quad of the wave; this lane's point / 16-byte piece
    const bool k0 = (k & 1) != 0, k1 = (k & 2) != 0;
    const int cls_a = (ln >> 3) & 1, cls_e = (ln >> 4) & 1;  // bank class of the quad: half read first, parity read first
    // this lane's channels: the 16-byte pieces k and k + 4 of a pixel, i.e. a quad reads / writes 64 contiguous bytes
    // per instruction; c0 = the piece read first, c0 ^ 64 the other
    const uint32_t c0 = (uint32_t)(16 * k + 64 * cls_a);
    const bool l0 = wv < kL0Waves;                           // a wave of the level-0 rows?
#ifdef W3_RESTPRIO
    if (!l0) __builtin_amdgcn_s_setprio(W3_RESTPRIO);        // the youngest waves of the workgroup lose every issue slot otherwise
#endif
    const bool more = nxt < nitems;
    const int nbuf = cb ^ 1;                                 // the window set of `nxt`

    // (image, tile row, tile column) of a work item; quotients by v_rcp_f32: x + 0.5 is at least 0.5 / divisor away from an
    // integer, far beyond the 1 ulp of the reciprocal
    auto geometry = [&](int it, int& b_, int& tx_, int& ty_) __attribute__((always_inline)) {
      b_ = to_sgpr((int)(((float)it + 0.5f) * __builtin_amdgcn_rcpf((float)ntiles_)));
      const int tile_ = it - b_ * ntiles;
      ty_ = to_sgpr((int)(((float)tile_ + 0.5f) * __builtin_amdgcn_rcpf((float)TX_)));
      tx_ = tile_ - ty_ * TX;
    };
    // the ri-th query of levels 1..3 of tile (tx_, ty_): lane l of a quad evaluates level l's rectangle, the quad's lanes
    // pull the one of the query's level
    auto rest_query = [&](int tx_, int ty_, int ri, int& nrest_, bool& valid_) __attribute__((always_inline)) {
      const int gW = sel4(k0, k1, lvW[0], lvW[1], lvW[2], lvW[3]), gH = sel4(k0, k1, lvH[0], lvH[1], lvH[2], lvH[3]);
      const float fxs = (float)(kTW * gW) * __builtin_amdgcn_rcpf((float)lvW[0]), fys = (float)(kTH * gH) * __builtin_amdgcn_rcpf((float)lvH[0]);
      const int gxs = min(max((int)ceilf((float)tx_ * fxs - 0.5f), 0), gW);
      const int xe = tx_ == TX - 1 ? gW : min(max((int)ceilf((float)(tx_ + 1) * fxs - 0.5f), gxs), gW);
      const int gys = min(max((int)ceilf((float)ty_ * fys - 0.5f), 0), gH);
      const int ye = ty_ == TY - 1 ? gH : min(max((int)ceilf((float)(ty_ + 1) * fys - 0.5f), gys), gH);
      const int gnx = xe - gxs, cnt = gnx * (ye - gys);
      const int e1 = __builtin_amdgcn_readlane(cnt, 1), e2 = e1 + __builtin_amdgcn_readlane(cnt, 2);
      nrest_ = e2 + __builtin_amdgcn_readlane(cnt, 3);
      valid_ = ri < nrest_;
      const int ql = 1 + (ri >= e1 ? 1 : 0) + (ri >= e2 ? 1 : 0);
      const int j = ri - (ri >= e2 ? e2 : ri >= e1 ? e1 : 0);
      const int src = ((ln & ~3) | ql) << 2;                 // lane ql of the quad holds level ql's rectangle
      const int qxs = __builtin_amdgcn_ds_bpermute(src, gxs), qys = __builtin_amdgcn_ds_bpermute(src, gys);
      const int qnx = __builtin_amdgcn_ds_bpermute(src, gnx);
      const int Wq = __builtin_amdgcn_ds_bpermute(src, gW);
      const int Sq = __builtin_amdgcn_ds_bpermute(src, sel4(k0, k1, lvS[0], lvS[1], lvS[2], lvS[3]));
      const int yy = (int)(((float)j + 0.5f) * __builtin_amdgcn_rcpf((float)max(qnx, 1)));
      return mad_u24((uint32_t)(qys + yy), (uint32_t)Wq, (uint32_t)(Sq + qxs + j)) - mad_u24((uint32_t)yy, (uint32_t)qnx, 0u);
    };
    // ONE wave writes the first kRestQuads of them (lane = query) into the table of a window set: the four waves of levels
    // 1..3 then read their query instead of each quad deriving it (2.4 us of their iteration, and they are the critical path)
    auto build_rest_table = [&](int tx_, int ty_, int slot) __attribute__((always_inline)) {
      int nr;
      bool ok;
      const uint32_t q = rest_query(tx_, ty_, ln, nr, ok);
      mt.qtab[slot][ln] = ok ? (int)q : -1;
      if (ln == 0) mt.nrest[slot][0] = nr;
    };
    static_assert(kRestQuads == 64, "one lane per table entry");
    // ---- this quad's query in pass `ps` of an item, its locations and weights requested ---------------------------------
    // Tile geometry: level-k pixels [f(t), f(t + 1)) with f(t) = ceil(t * T * n / n0 - 1/2) are the ones whose centre falls
    // into tile t -- an exact partition as long as every workgroup evaluates the same expression (the one of msda_fwd_win).
    auto fetch_query = [&](int b_, int tx_, int ty_, int ps, int slot, bool& live_, uint32_t& pair_, v2f (&lc_)[4], float (&sa_)[4], int& npass_)
                           __attribute__((always_inline)) {
      uint32_t qidx = 0;
      bool lv_ = false;
      if (l0) {                                              // wave = tile row, quad = tile column
        const int xs0 = kTW * tx_, ys0 = kTH * ty_;
        lv_ = (ps == 0) & (pq < min(kTW, lvW[0] - xs0)) & (wv < min(kTH, lvH[0] - ys0));
        qidx = (uint32_t)(lvS[0] + (ys0 + wv) * lvW[0] + xs0 + pq);
      } else if (ps == 0) {                                  // from the table
        const int q = mt.qtab[slot][(wv - kL0Waves) * 16 + pq];
        const int nr = __builtin_amdgcn_readfirstlane(mt.nrest[slot][0]);
        // these waves walk their queries kRestQuads at a time (one pass at the R50 shapes; pyramids whose upper levels
        // are large relative to level 0 take more)
        npass_ = max(1, (nr + kRestQuads - 1) / kRestQuads);
        lv_ = q >= 0;
        qidx = (uint32_t)q;
      } else {
        int nr;
        qidx = rest_query(tx_, ty_, ps * kRestQuads + (wv - kL0Waves) * 16 + pq, nr, lv_);
      }
      lv_ = lv_ & (qidx < (uint32_t)d.Lq);                   // (shapes whose pixel count exceeds num_query: never outside the tensors)
      live_ = lv_;
      pair_ = mul_u24_s(lv_ ? qidx : 0u, (uint32_t)M);       // (query, head 0) pair within the image; the head sits in the base pointers
#pragma unroll
      for (int l = 0; l < 4; ++l) { lc_[l] = v2f{0.f, 0.f}; sa_[l] = 0.f; }
      if (lv_) {
        const int64_t pimg = (int64_t)b_ * d.Lq * M + m;     // pair (query 0, head m) of the item's image: uniform bases, 32-bit per-lane offsets
        const v2f* lp = reinterpret_cast<const v2f*>(loc + pimg * 32 + (pair_ * 32u + 2u * (uint32_t)k));
        const float* ap = attn + pimg * 16 + (pair_ * 16u + (uint32_t)k);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          lc_[l] = __builtin_nontemporal_load(lp + 4 * l);
          sa_[l] = __builtin_nontemporal_load(ap + 4 * l);
        }
      }
    };
    // (x, y) of a sample on level l in pixels, and whether it is in range (the reference's arithmetic, cuh:282-288, :38-46)
    auto coord = [&](int l, bool live_, const v2f (&lc_)[4], bool& in) __attribute__((always_inline)) {
      const v2f fWH = {(float)lvW[l], (float)lvH[l]};
      const v2f p = __builtin_elementwise_fma(lc_[l], fWH, v2f{-0.5f, -0.5f});
      in = live_ & (p.y > -1.f) & (p.x > -1.f) & (p.y < fWH.y) & (p.x < fWH.x);
      return p;
    };
    auto value_rsrc = [&](int b_) __attribute__((always_inline)) {
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(value) + (int64_t)b_ * d.S * M * 32, 0,
                                               (int)((uint32_t)d.S * pixB), 0x00020000);
    };

    int b = 0, tx = 0, ty = 0;
    if (body) geometry(item, b, tx, ty);
    if (first_table) {
      first_table = false;
      if (wv == kWaves - 1) {
        int b1, tx1, ty1;
        geometry(nxt, b1, tx1, ty1);
        build_rest_table(tx1, ty1, nbuf);
      }
    }
    W3_STAMP(0);
    lds_barrier();                                           // B: every wave's share of the windows of `item` has landed (each
                                                             // waited for its own before its last stores); everybody left item - K
    W3_STAMP(1);
    if (tid < 16) (&mt.sum[cb][0][0])[tid] = 0;               // read before this barrier; added to again after the next one

    // ---- the next item: its queries, its locations requested -----------------------------------------------------------
    bool live_n = false;
    uint32_t pair_n = 0;
    v2f lc_n[4];
    float sa_n[4];
    int b_n = 0, tx_n = 0, ty_n = 0, npass_n = 1;
#pragma unroll
    for (int l = 0; l < 4; ++l) { lc_n[l] = v2f{0.f, 0.f}; sa_n[l] = 0.f; }
    if (more) {
      geometry(nxt, b_n, tx_n, ty_n);
      fetch_query(b_n, tx_n, ty_n, 0, nbuf, live_n, pair_n, lc_n, sa_n, npass_n);
      // the table of the item after next, into the set of `item` (read before barrier B, next read after the next B)
      if (wv == kWaves - 1 && nxt + K < nitems) {
        int b2, tx2, ty2;
        geometry(nxt + K, b2, tx2, ty2);
        build_rest_table(tx2, ty2, cb);
      }
    }
    W3_STAMP(2);

    // ---- the gather of `item` ------------------------------------------------------------------------------------------
    uint32_t nb = 0, fm = 0;                                 // near bits (bit l); the pair's far samples (bit 4 * point + level)
    // near = all four corners inside the level's window, or outside the image
    auto classify = [&]() __attribute__((always_inline)) {
      uint32_t farmask = 0;
      nb = 0;
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        bool in;
        const v2f p = coord(l, live, lc, in);
        const int cx = cvt_i32(floorf(p.x)), cy = cvt_i32(floorf(p.y));
        // a level smaller than its window: top-left corners past the last in-range one are not "near"
        const int cxm = min(ogx[l] + kWW[l] - 2, lvW[l] - 1) - ogx[l], rym = min(ogy[l] + kWH[l] - 2, lvH[l] - 1) - ogy[l];
        const bool near = in & ((uint32_t)(cx - ogx[l]) <= (uint32_t)cxm) & ((uint32_t)(cy - ogy[l]) <= (uint32_t)rym);
        nb |= near ? (1u << l) : 0u;
        farmask |= (in & !near) ? (1u << l) : 0u;
      }
      fm = farmask << (4 * k);
      fm |= (uint32_t)dppi<0xB1>((int)fm);                   // quad_perm [1,0,3,2]
      fm |= (uint32_t)dppi<0x4E>((int)fm);                   // quad_perm [2,3,0,1]
    };
    v2f aA0 = {0.f, 0.f}, aA1 = {0.f, 0.f}, aB0 = {0.f, 0.f}, aB1 = {0.f, 0.f};   // channels at c0 (A) and at c0 ^ 64 (B)
    const uint32_t win_base = smem_base + (uint32_t)(cb * kBufBytes);
    const uint32_t zero_first = smem_base + kZeroOff + 128u * (uint32_t)cls_e;   // parity cls_e; the other parity: ^ 128
    // this lane's point on level LV (the reference's bilinear weights with the attention weight folded in)
    auto prepare = [&](auto ltag) __attribute__((always_inline)) {
      constexpr int LV = decltype(ltag)::value;
      Smp s;
      bool in_;
      const v2f xy = coord(LV, live, lc, in_);
      const v2f fl = {floorf(xy.x), floorf(xy.y)};
      v2f fr = xy - fl;                                      // (fx, fy); inf - inf / NaN for poisoned locations ...
      fr.x = fmaxf(fr.x, 0.f); fr.y = fmaxf(fr.y, 0.f);      // ... which must not turn the zero weights of dead samples into NaN
      const v2f om = v2f{1.f, 1.f} - fr;                     // (1 - fx, 1 - fy)
      const int cx = cvt_i32(fl.x) - ogx[LV], ry = cvt_i32(fl.y) - ogy[LV];
      const bool near = ((nb >> LV) & 1u) != 0u;
      const uint32_t sw = (uint32_t)(cx ^ cls_e) & 1u;       // 1: the right-hand pixel has this quad's first parity
      const float an = near ? sa[LV] : 0.f;                  // dead and far samples: all four weights 0
      const v2f gx = sw ? v2f{fr.x, om.x} : v2f{om.x, fr.x}; // x factors of the (first, second) pixel
      const v2f wtb = v2f{om.y, fr.y} * an;                  // (top, bottom) row weight x attention weight
      s.wT = gx * wtb.x;
      s.wB = gx * wtb.y;
      const uint32_t tl = win_base + (uint32_t)(kBase[LV] * 128) + (uint32_t)(__mul24(ry, kWW[LV]) + cx) * 128u;
      s.aF = near ? tl + (sw << 7) : zero_first;
      s.aS = near ? tl + 128u - (sw << 7) : (zero_first ^ 128u);
      return s;
    };
    // Half rows (one pixel of a corner row = this lane's two 16-byte pieces, 8 registers) through a ring of three register sets
    struct Half { f32x4 a, b; };
    struct Adr { lds4 pF, pF2, pS, pS2; };
    // half J of the sample (level LV, point PT): 0 = top row / first pixel, 1 = top / second, 2 = bottom / first, 3 = bottom / second
    auto fetch_half = [&](auto ltag, auto ptag, auto jtag, const Smp& s, Half& h, Adr& ad) __attribute__((always_inline)) {
      constexpr int LV = decltype(ltag)::value, PT = decltype(ptag)::value, J = decltype(jtag)::value;
      constexpr int kRow = kWW[LV] * 8;                      // one window row, in 16-byte units
      if constexpr (J == 0) {
        const uint32_t aF = qb<PT>(s.aF) + c0;
        ad.pF = reinterpret_cast<lds4>((uintptr_t)aF); ad.pF2 = reinterpret_cast<lds4>((uintptr_t)(aF ^ 64u));
        h.a = ad.pF[0]; h.b = ad.pF2[0];
      } else if constexpr (J == 1) {
        const uint32_t aS = qb<PT>(s.aS) + c0;
        ad.pS = reinterpret_cast<lds4>((uintptr_t)aS); ad.pS2 = reinterpret_cast<lds4>((uintptr_t)(aS ^ 64u));
        h.a = ad.pS[0]; h.b = ad.pS2[0];
      } else if constexpr (J == 2) {
        h.a = ad.pF[kRow]; h.b = ad.pF2[kRow];
      } else {
        h.a = ad.pS[kRow]; h.b = ad.pS2[kRow];
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto consume_half = [&](auto ptag, auto jtag, const Smp& s, const Half& h) __attribute__((always_inline)) {
      constexpr int PT = decltype(ptag)::value, J = decltype(jtag)::value;
      const float w = qbf<PT>(J == 0 ? s.wT.x : J == 1 ? s.wT.y : J == 2 ? s.wB.x : s.wB.y);
      const v2f W2 = {w, w};
      aA0 = __builtin_elementwise_fma(W2, v2f{h.a[0], h.a[1]}, aA0); aA1 = __builtin_elementwise_fma(W2, v2f{h.a[2], h.a[3]}, aA1);
      aB0 = __builtin_elementwise_fma(W2, v2f{h.b[0], h.b[1]}, aB0); aB1 = __builtin_elementwise_fma(W2, v2f{h.b[2], h.b[3]}, aB1);
      asm volatile("" : "+v"(aA0), "+v"(aA1), "+v"(aB0), "+v"(aB1));   // pins the FMAs here (IR-level sinking ignores sched_barrier)
      __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    Half h0, h1, h2;
    Adr ad;
    Smp s0, s1;
    // generated sequence (msda_fwd_win2.hip): consume half g, request half g + 3 (g = 16 * level + 4 * point + half); the
    // next level's sample is prepared just before its first half is requested.  Levels 0-1 / levels 2-3.
    auto gather_first = [&]() __attribute__((always_inline)) {
      s0 = prepare(I0{});
      fetch_half(I0{}, I0{}, I0{}, s0, h0, ad); fetch_half(I0{}, I0{}, I1{}, s0, h1, ad); fetch_half(I0{}, I0{}, I2{}, s0, h2, ad);
      consume_half(I0{}, I0{}, s0, h0); fetch_half(I0{}, I0{}, I3{}, s0, h0, ad);
      consume_half(I0{}, I1{}, s0, h1); fetch_half(I0{}, I1{}, I0{}, s0, h1, ad);
      consume_half(I0{}, I2{}, s0, h2); fetch_half(I0{}, I1{}, I1{}, s0, h2, ad);
      consume_half(I0{}, I3{}, s0, h0); fetch_half(I0{}, I1{}, I2{}, s0, h0, ad);
      consume_half(I1{}, I0{}, s0, h1); fetch_half(I0{}, I1{}, I3{}, s0, h1, ad);
      consume_half(I1{}, I1{}, s0, h2); fetch_half(I0{}, I2{}, I0{}, s0, h2, ad);
      consume_half(I1{}, I2{}, s0, h0); fetch_half(I0{}, I2{}, I1{}, s0, h0, ad);
      consume_half(I1{}, I3{}, s0, h1); fetch_half(I0{}, I2{}, I2{}, s0, h1, ad);
      consume_half(I2{}, I0{}, s0, h2); fetch_half(I0{}, I2{}, I3{}, s0, h2, ad);
      consume_half(I2{}, I1{}, s0, h0); fetch_half(I0{}, I3{}, I0{}, s0, h0, ad);
      consume_half(I2{}, I2{}, s0, h1); fetch_half(I0{}, I3{}, I1{}, s0, h1, ad);
      consume_half(I2{}, I3{}, s0, h2); fetch_half(I0{}, I3{}, I2{}, s0, h2, ad);
      consume_half(I3{}, I0{}, s0, h0); fetch_half(I0{}, I3{}, I3{}, s0, h0, ad);
      s1 = prepare(I1{});
      consume_half(I3{}, I1{}, s0, h1); fetch_half(I1{}, I0{}, I0{}, s1, h1, ad);
      consume_half(I3{}, I2{}, s0, h2); fetch_half(I1{}, I0{}, I1{}, s1, h2, ad);
      consume_half(I3{}, I3{}, s0, h0); fetch_half(I1{}, I0{}, I2{}, s1, h0, ad);
      consume_half(I0{}, I0{}, s1, h1); fetch_half(I1{}, I0{}, I3{}, s1, h1, ad);
      consume_half(I0{}, I1{}, s1, h2); fetch_half(I1{}, I1{}, I0{}, s1, h2, ad);
      consume_half(I0{}, I2{}, s1, h0); fetch_half(I1{}, I1{}, I1{}, s1, h0, ad);
      consume_half(I0{}, I3{}, s1, h1); fetch_half(I1{}, I1{}, I2{}, s1, h1, ad);
      consume_half(I1{}, I0{}, s1, h2); fetch_half(I1{}, I1{}, I3{}, s1, h2, ad);
      consume_half(I1{}, I1{}, s1, h0); fetch_half(I1{}, I2{}, I0{}, s1, h0, ad);
      consume_half(I1{}, I2{}, s1, h1); fetch_half(I1{}, I2{}, I1{}, s1, h1, ad);
      consume_half(I1{}, I3{}, s1, h2); fetch_half(I1{}, I2{}, I2{}, s1, h2, ad);
      consume_half(I2{}, I0{}, s1, h0); fetch_half(I1{}, I2{}, I3{}, s1, h0, ad);
      consume_half(I2{}, I1{}, s1, h1); fetch_half(I1{}, I3{}, I0{}, s1, h1, ad);
      consume_half(I2{}, I2{}, s1, h2); fetch_half(I1{}, I3{}, I1{}, s1, h2, ad);
      consume_half(I2{}, I3{}, s1, h0); fetch_half(I1{}, I3{}, I2{}, s1, h0, ad);
      consume_half(I3{}, I0{}, s1, h1); fetch_half(I1{}, I3{}, I3{}, s1, h1, ad);
    };
    auto gather_second = [&]() __attribute__((always_inline)) {
      s0 = prepare(I2{});
      consume_half(I3{}, I1{}, s1, h2); fetch_half(I2{}, I0{}, I0{}, s0, h2, ad);
      consume_half(I3{}, I2{}, s1, h0); fetch_half(I2{}, I0{}, I1{}, s0, h0, ad);
      consume_half(I3{}, I3{}, s1, h1); fetch_half(I2{}, I0{}, I2{}, s0, h1, ad);
      consume_half(I0{}, I0{}, s0, h2); fetch_half(I2{}, I0{}, I3{}, s0, h2, ad);
      consume_half(I0{}, I1{}, s0, h0); fetch_half(I2{}, I1{}, I0{}, s0, h0, ad);
      consume_half(I0{}, I2{}, s0, h1); fetch_half(I2{}, I1{}, I1{}, s0, h1, ad);
      consume_half(I0{}, I3{}, s0, h2); fetch_half(I2{}, I1{}, I2{}, s0, h2, ad);
      consume_half(I1{}, I0{}, s0, h0); fetch_half(I2{}, I1{}, I3{}, s0, h0, ad);
      consume_half(I1{}, I1{}, s0, h1); fetch_half(I2{}, I2{}, I0{}, s0, h1, ad);
      consume_half(I1{}, I2{}, s0, h2); fetch_half(I2{}, I2{}, I1{}, s0, h2, ad);
      consume_half(I1{}, I3{}, s0, h0); fetch_half(I2{}, I2{}, I2{}, s0, h0, ad);
      consume_half(I2{}, I0{}, s0, h1); fetch_half(I2{}, I2{}, I3{}, s0, h1, ad);
      consume_half(I2{}, I1{}, s0, h2); fetch_half(I2{}, I3{}, I0{}, s0, h2, ad);
      consume_half(I2{}, I2{}, s0, h0); fetch_half(I2{}, I3{}, I1{}, s0, h0, ad);
      consume_half(I2{}, I3{}, s0, h1); fetch_half(I2{}, I3{}, I2{}, s0, h1, ad);
      consume_half(I3{}, I0{}, s0, h2); fetch_half(I2{}, I3{}, I3{}, s0, h2, ad);
      s1 = prepare(I3{});
      consume_half(I3{}, I1{}, s0, h0); fetch_half(I3{}, I0{}, I0{}, s1, h0, ad);
      consume_half(I3{}, I2{}, s0, h1); fetch_half(I3{}, I0{}, I1{}, s1, h1, ad);
      consume_half(I3{}, I3{}, s0, h2); fetch_half(I3{}, I0{}, I2{}, s1, h2, ad);
      consume_half(I0{}, I0{}, s1, h0); fetch_half(I3{}, I0{}, I3{}, s1, h0, ad);
      consume_half(I0{}, I1{}, s1, h1); fetch_half(I3{}, I1{}, I0{}, s1, h1, ad);
      consume_half(I0{}, I2{}, s1, h2); fetch_half(I3{}, I1{}, I1{}, s1, h2, ad);
      consume_half(I0{}, I3{}, s1, h0); fetch_half(I3{}, I1{}, I2{}, s1, h0, ad);
      consume_half(I1{}, I0{}, s1, h1); fetch_half(I3{}, I1{}, I3{}, s1, h1, ad);
      consume_half(I1{}, I1{}, s1, h2); fetch_half(I3{}, I2{}, I0{}, s1, h2, ad);
      consume_half(I1{}, I2{}, s1, h0); fetch_half(I3{}, I2{}, I1{}, s1, h0, ad);
      consume_half(I1{}, I3{}, s1, h1); fetch_half(I3{}, I2{}, I2{}, s1, h1, ad);
      consume_half(I2{}, I0{}, s1, h2); fetch_half(I3{}, I2{}, I3{}, s1, h2, ad);
      consume_half(I2{}, I1{}, s1, h0); fetch_half(I3{}, I3{}, I0{}, s1, h0, ad);
      consume_half(I2{}, I2{}, s1, h1); fetch_half(I3{}, I3{}, I1{}, s1, h1, ad);
      consume_half(I2{}, I3{}, s1, h2); fetch_half(I3{}, I3{}, I2{}, s1, h2, ad);
      consume_half(I3{}, I0{}, s1, h0); fetch_half(I3{}, I3{}, I3{}, s1, h0, ad);
      consume_half(I3{}, I1{}, s1, h1);
      consume_half(I3{}, I2{}, s1, h2);
      consume_half(I3{}, I3{}, s1, h0);
    };
    // ---- far samples: raw buffer loads, one far sample per quad and step; a step's loads can stay in flight ----------------
    struct FarSt { f32x4 tLa, tLb, tRa, tRb, bLa, bLb, bRa, bRb; float wt, wb, lw; };   // a far step in flight: 8 loads, 3 weights
    auto far_issue = [&](__amdgpu_buffer_rsrc_t vsrc, FarSt& f) __attribute__((always_inline)) {
      const bool has = fm != 0u;
      const int idx = has ? __builtin_ctz(fm) : 0;
      fm &= fm - 1u;
      const int fl_ = idx & 3, ps = idx >> 2;                // level and point (= preparing lane) of the far sample
      const int src = ((ln & ~3) | ps) << 2;                 // byte address of the preparing lane for ds_bpermute
      // every lane evaluates its own candidate on level fl_, the quad pulls the preparing lane's and redoes the (cheap)
      // sample arithmetic -- far samples are a few per cent, their state is not kept around
      const bool c1 = (fl_ & 1) != 0, c2 = (fl_ & 2) != 0;
      const int4 lv = *reinterpret_cast<const int4*>(&mt.lvl[fl_][0]);   // the far sample's level: H, W, first pixel
      const int fH_ = lv.x, fW_ = lv.y, fS_ = lv.z;
      const v2f lsel = {sel4(c1, c2, lc[0].x, lc[1].x, lc[2].x, lc[3].x), sel4(c1, c2, lc[0].y, lc[1].y, lc[2].y, lc[3].y)};
      const v2f pc = __builtin_elementwise_fma(lsel, v2f{(float)fW_, (float)fH_}, v2f{-0.5f, -0.5f});   // == coord(fl_)
      const int ca_ = (int)__float_as_uint(sel4(c1, c2, sa[0], sa[1], sa[2], sa[3]));
      // quads without a far sample left run along with zero weights: their stand-in coordinates must be finite
      const uint32_t hm = has ? 0xffffffffu : 0u;
      const float fxv = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)__float_as_uint(pc.x)) & hm);
      const float fyv = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)__float_as_uint(pc.y)) & hm);
      const float fav = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, ca_) & hm);
      const uint32_t rowG = mul_u24_s((uint32_t)fW_, pixB);
      const float xf = floorf(fxv), yf = floorf(fyv);
      const float lw = fxv - xf, lh = fyv - yf;
      const int fx0 = (int)xf, fy0 = (int)yf;                // in range or 0 for the stand-ins
      const bool t_ok = has & (fy0 >= 0), b_ok = has & (fy0 + 1 <= fH_ - 1), l_ok = fx0 >= 0, r_ok = fx0 + 1 <= fW_ - 1;
      const float wt = (1.f - lh) * fav, wb = lh * fav;
      // 24-bit multiply-adds (pixel index < 2^24, pitch < 2^24) on the CLAMPED top-left pixel: with fy0 or fx0 = -1 the
      // live corners sit in row / column 0, and a 24-bit product of a negative index is not what a 32-bit one wraps to
      const int cy = max(fy0, 0), cx = max(fx0, 0);
      const uint32_t off = mad_u24_s(mad_u24((uint32_t)cy, (uint32_t)fW_, (uint32_t)(fS_ + cx)), pixB, c0);
      const uint32_t dx = fx0 >= 0 ? pixB : 0u, dy = fy0 >= 0 ? rowG : 0u;   // step to the right / bottom neighbour
      const uint32_t o1 = (t_ok & l_ok) ? off : kOobOffset;
      const uint32_t o2 = (t_ok & r_ok) ? off + dx : kOobOffset;
      const uint32_t o3 = (b_ok & l_ok) ? off + dy : kOobOffset;
      const uint32_t o4 = (b_ok & r_ok) ? off + dy + dx : kOobOffset;
      f.wt = wt; f.wb = wb; f.lw = lw;
      f.tLa = buffer_load_f32x4(vsrc, o1, hoff); f.tLb = buffer_load_f32x4(vsrc, o1 ^ 64u, hoff);
      f.tRa = buffer_load_f32x4(vsrc, o2, hoff); f.tRb = buffer_load_f32x4(vsrc, o2 ^ 64u, hoff);
      f.bLa = buffer_load_f32x4(vsrc, o3, hoff); f.bLb = buffer_load_f32x4(vsrc, o3 ^ 64u, hoff);
      f.bRa = buffer_load_f32x4(vsrc, o4, hoff); f.bRb = buffer_load_f32x4(vsrc, o4 ^ 64u, hoff);
    };
    auto far_consume = [&](const FarSt& f) __attribute__((always_inline)) {
      auto row = [&](const f32x4& La, const f32x4& Lb, const f32x4& Ra, const f32x4& Rb, float wrow) __attribute__((always_inline)) {
        const float wl = wrow * (1.f - f.lw), wr = wrow * f.lw;
        const v2f WL = {wl, wl}, WR = {wr, wr};
        aA0 = __builtin_elementwise_fma(WL, v2f{La[0], La[1]}, aA0); aA1 = __builtin_elementwise_fma(WL, v2f{La[2], La[3]}, aA1);
        aB0 = __builtin_elementwise_fma(WL, v2f{Lb[0], Lb[1]}, aB0); aB1 = __builtin_elementwise_fma(WL, v2f{Lb[2], Lb[3]}, aB1);
        aA0 = __builtin_elementwise_fma(WR, v2f{Ra[0], Ra[1]}, aA0); aA1 = __builtin_elementwise_fma(WR, v2f{Ra[2], Ra[3]}, aA1);
        aB0 = __builtin_elementwise_fma(WR, v2f{Rb[0], Rb[1]}, aB0); aB1 = __builtin_elementwise_fma(WR, v2f{Rb[2], Rb[3]}, aB1);
        asm volatile("" : "+v"(aA0), "+v"(aA1), "+v"(aB0), "+v"(aB1));
      };
      row(f.tLa, f.tLb, f.tRa, f.tRb, f.wt);
      row(f.bLa, f.bLb, f.bRa, f.bRb, f.wb);
    };
    // a quad writes 2 x 64 contiguous bytes; dead quads store past the end of the buffer (no branch: the number of
    // memory instructions of an iteration does not depend on the data)
    auto store_out = [&](int b_) __attribute__((always_inline)) {
      const __amdgpu_buffer_rsrc_t osrc = __builtin_amdgcn_make_buffer_rsrc(
          out + (int64_t)b_ * d.Lq * M * 32, 0, (int)((uint32_t)d.Lq * pixB), 0x00020000);
      // (the head's offset goes into the per-lane offset, the scalar offset is the immediate 0: with an SGPR there the
      // compiler's hazard recogniser assumes that a VALU instruction may overwrite the data registers of a
      // buffer_store_dwordx4 in the very next slot -- it did, and on gfx950 lanes 12..15 of every row then stored the new
      // value: one wrong channel per lane in the younger waves of a busy CU, run-to-run different.  Found the hard way.)
      const uint32_t o = live ? (pair * 32u + 4u * (uint32_t)k) * 4u + hoff : kOobOffset;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{aA0.x, aA0.y, aA1.x, aA1.y}), osrc, o + 64u * (uint32_t)cls_a, 0, 2);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{aB0.x, aB0.y, aB1.x, aB1.y}), osrc, o + 64u * (uint32_t)(cls_a ^ 1), 0, 2);
    };

    FarSt f0;                                                // the first far step of the wave: requested before the gather, used after it
    bool far0 = false;
    if (body) {
      classify();
      far0 = __ballot(fm != 0u) != 0;
      if (far0) far_issue(value_rsrc(b), f0);
      gather_first();
    }
    W3_STAMP(3);

    // ---- the next item: placement sums (mean top-left corner of the in-range samples of the tile's level-0 queries, per
    // level: reduce-scatter over the quad -- lane l ends up with level l --, then over the 4 quads of a DPP row) -------------
    if (more & l0) {
      auto quad_scatter = [&](int v0, int v1, int v2, int v3) __attribute__((always_inline)) {
        const int A = (k0 ? v1 : v0) + dppi<0xB1>(k0 ? v0 : v1), B = (k0 ? v3 : v2) + dppi<0xB1>(k0 ? v2 : v3);   // quad_perm [1,0,3,2]
        int R = (k1 ? B : A) + dppi<0x4E>(k1 ? A : B);                                                            // quad_perm [2,3,0,1]
        R += dppi<0x114>(R);                                 // row_shr 4
        R += dppi<0x118>(R);                                 // row_shr 8: lanes 12..15 of a row hold the row's totals of levels 0..3
        return R;
      };
      int px[4], py[4], pn[4];
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        bool in;
        const v2f p = coord(l, live_n, lc_n, in);
        const int cx = cvt_i32(floorf(p.x)), cy = cvt_i32(floorf(p.y));   // (saturated garbage for poisoned locations: masked)
        px[l] = in ? cx : 0; py[l] = in ? cy : 0; pn[l] = in ? 1 : 0;
      }
      const int ax = quad_scatter(px[0], px[1], px[2], px[3]);
      const int ay = quad_scatter(py[0], py[1], py[2], py[3]);
      const int an = quad_scatter(pn[0], pn[1], pn[2], pn[3]);
      if ((ln & 12) == 12 && an != 0) {
        atomicAdd(&mt.sum[nbuf][k][0], ax);
        atomicAdd(&mt.sum[nbuf][k][1], ay);
        atomicAdd(&mt.sum[nbuf][k][2], an);
      }
    }
    W3_STAMP(4);
    lds_barrier();                                           // A: the sums of `nxt` are complete; everybody is past the first
                                                             // half of `item`, i.e. long past item - K, whose window set `nxt` takes
    W3_STAMP(5);
    int ogx_n[4] = {0, 0, 0, 0}, ogy_n[4] = {0, 0, 0, 0};
    if (more) {
      int myOx, myOy;
      {
        const int4 sm = *reinterpret_cast<const int4*>(&mt.sum[nbuf][k][0]);
        const int myWW = sel4(k0, k1, kWW[0], kWW[1], kWW[2], kWW[3]), myWH = sel4(k0, k1, kWH[0], kWH[1], kWH[2], kWH[3]);
        const int myW = sel4(k0, k1, lvW[0], lvW[1], lvW[2], lvW[3]), myH = sel4(k0, k1, lvH[0], lvH[1], lvH[2], lvH[3]);
        // v_rcp_f32: every lane of the workgroup evaluates the same expression on the same sums.  A level without an
        // in-range sample in this tile gets its window at the origin: nothing will be looked up in it
        const float inv = __builtin_amdgcn_rcpf((float)max(sm.z, 1));
        myOx = (int)floorf((float)sm.x * inv + 0.5f) - (myWW - 2) / 2;
        myOy = (int)floorf((float)sm.y * inv + 0.5f) - (myWH - 2) / 2;
        myOx = max(-1, min(myOx, myW + 1 - myWW));
        myOy = max(-1, min(myOy, myH + 1 - myWH));
      }
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        ogx_n[l] = __builtin_amdgcn_readlane(myOx, l);
        ogy_n[l] = __builtin_amdgcn_readlane(myOy, l);
      }
      // ---- stage the four windows of `nxt`: LDS-DMA, one instruction = 8 consecutive slots (1 KB) of ONE level per level-0 wave.
      // Straight-line code with the same number of instructions in every wave (a wave without a chunk left in a level
      // issues an out-of-range one into the all-zero region, which costs no memory access) --------------------------------
    }
    if (more & l0) {                                         // (the four waves of levels 1..3 are the critical path: no DMA there)
      const __amdgpu_buffer_rsrc_t vsrc_n = value_rsrc(b_n);
      const uint32_t chunk = (uint32_t)(ln & 7) * 16u;
      const int sub = ln >> 3;
      auto stage_level = [&](auto ltag) __attribute__((always_inline)) {
        constexpr int LV = decltype(ltag)::value;
        constexpr int WW = kWW[LV], C0 = kBase[LV] / 8, C1 = kBase[LV + 1] / 8;
        constexpr int kSteps = (C1 - C0 + kL0Waves - 1) / kL0Waves;
        constexpr int kDR = (8 * kL0Waves) / WW, kDC = (8 * kL0Waves) % WW;
        const int Hs = lvH[LV], Ws = lvW[LV], xS = ogx_n[LV] + lvS[LV], oy = ogy_n[LV], ox = ogx_n[LV];
        int i = C0 + wv;                                     // this wave's first chunk of the level
        int subv = sub;
        asm volatile("" : "+v"(subv));                       // opaque: the level's start is computed HERE
        const int rel = 8 * wv + subv;                       // slot of this lane in the level's window
        int r = (int)(((float)rel + 0.5f) * (1.f / WW)), c = rel - r * WW;
#pragma unroll
        for (int t = 0; t < kSteps; ++t, i += kL0Waves) {
          const bool have = i < C1;                          // wave-uniform
          const int y = oy + r;
          const bool inside = have & ((unsigned)y < (unsigned)Hs) & ((unsigned)(ox + c) < (unsigned)Ws);
          // pixel index < 2^24 and pixel pitch M * 128 < 2^24 by win3_forward_ok: two full-rate 24-bit multiply-adds
          const uint32_t pix = mad_u24_s((uint32_t)y, (uint32_t)Ws, (uint32_t)(xS + c));
          const uint32_t in_off = mad_u24_s(pix, pixB, chunk);
          const uint32_t off = inside ? in_off : kOobOffset;
          const int dst = have ? nbuf * kBufBytes + i * 1024 : kZeroOff;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(vsrc_n, (__attribute__((address_space(3))) void*)(smem + dst), 16,
                                                   off, hoff, 0, 0);
          if (t + 1 < kSteps) {
            c += kDC; r += kDR;
            if (kDC != 0 && c >= WW) { c -= WW; r += 1; }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      stage_level(I0{});
      stage_level(I1{});
      stage_level(I2{});
      stage_level(I3{});
    }
    W3_STAMP(6);

    if (body) {
      gather_second();
      W3_STAMP(7);
      const __amdgpu_buffer_rsrc_t vsrc = value_rsrc(b);
      if (far0) far_consume(f0);
      while (__ballot(fm != 0u)) {
        FarSt f;
        far_issue(vsrc, f);
        far_consume(f);
      }
      W3_STAMP(8);
      // this wave's share of the NEXT item's windows has landed (issued half a gather ago), and with it everything older;
      // only the stores below stay in flight into the next iteration
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      store_out(b);
      W3_STAMP(9);
      // ---- further passes of the waves of levels 1..3 (pyramids with more than 64 such queries per tile): start to end ----
      for (int ps = 1; ps < npass; ++ps) {
        int np_;
        fetch_query(b, tx, ty, ps, cb, live, pair, lc, sa, np_);
        classify();
        aA0 = v2f{0.f, 0.f}; aA1 = v2f{0.f, 0.f}; aB0 = v2f{0.f, 0.f}; aB1 = v2f{0.f, 0.f};
        gather_first();
        gather_second();
        while (__ballot(fm != 0u)) {
          FarSt f;
          far_issue(vsrc, f);
          far_consume(f);
        }
        store_out(b);
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the first item's windows: nothing to do meanwhile)
    }
    if (!more) break;
    // ---- rotate --------------------------------------------------------------------------------------------------------
    item = nxt; nxt += K; body = true; cb = nbuf;
    live = live_n; pair = pair_n; npass = npass_n;
#pragma unroll
    for (int l = 0; l < 4; ++l) { lc[l] = lc_n[l]; sa[l] = sa_n[l]; ogx[l] = ogx_n[l]; ogy[l] = ogy_n[l]; }
  }
}

#ifdef MSDA_WIN3_PROF
extern "C" int msda_debug_read_prof3(void* dst, int nblocks) {
  if (nblocks > kProfBlocks) nblocks = kProfBlocks;
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_win3_prof), (size_t)nblocks * kWaves * kProfSlots * 8, 0, hipMemcpyDeviceToHost);
}
#endif

bool win3_forward_ok(const Dims& d) {
  // (the last condition keeps the work-item index, and item + 0.5, exact in float: the kernel splits it into (image,
  // tile) with a reciprocal)
  return d.D == 32 && d.P == 4 && d.L == 4 && d.Lq == d.S && d.S >= 1024 && d.M <= 65535 &&
         (int64_t)d.S * d.M * 128 < (int64_t)kOobOffset && d.N <= 65535 &&
         (int64_t)d.N * ((d.S + 127) / 128) < ((int64_t)1 << 22);
}

int launch_forward_win3(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn,
                        const Dims& d, float* out, hipStream_t stream) {
  static std::atomic<uint64_t> lds_opted_in{0};
  const void* fn = reinterpret_cast<const void*>(msda_fwd_win3);
  if (int rc = ensure_dynamic_lds(fn, kLdsBytes, lds_opted_in)) return rc;
  // persistent grid: one resident workgroup per CU (both window sets: 152 KB of LDS), spread over the heads; head m =
  // blockIdx.x, so that (by the observed round-robin placement of the linear workgroup id) XCD m % 8 only touches head m's
  // slice of `value` when M is a multiple of 8.  MSDA_WIN3_WGS=n: n workgroups per head instead (A/B switch).
  static const int wgs_env = std::getenv("MSDA_WIN3_WGS") ? std::atoi(std::getenv("MSDA_WIN3_WGS")) : 0;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  int K = wgs_env > 0 ? wgs_env : (cus + d.M - 1) / d.M;
  const int items = d.N * ((d.S + 127) / 128);             // at least the tile count of any pyramid whose level 0 holds <= ~3/4 of the pixels
  if (K > items) K = items;
  if (K < 1) K = 1;
  if (K > 65535) K = 65535;
  hipLaunchKernelGGL(msda_fwd_win3, dim3((unsigned)d.M, (unsigned)K), dim3(kT), kLdsBytes, stream, value, shapes, lsi, loc,
                     attn, d, out);
  return (int)hipGetLastError();
}

}  // namespace msda
