// msda_fwd_win2 -- MSDeformAttn forward for encoder-style calls (Lq == S), second generation of the LDS-window kernel
// (msda_fwd_win.hip).  fp32, D = 32, L = P = 4.  gfx950 only.  Replaces, for these calls, the work of
// ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299.
//
// What round 2's counters said about msda_fwd_win (profiles/r03_win_valu_budget.txt): a work item (170 queries) is a
// serial chain of ~17.7 us (scalar loads, locations, placement barrier, window DMA, LDS pass, SECOND ROUND for the 42
// queries of levels 1-3) of which only ~7 us keep the vector ALU busy, and at 128 VGPRs only two such chains fit a CU:
// launch time = items per CU / 2 x chain.  And it issues ~1050 VALU instructions per wave and round for 384 that are
// the gather itself.  Hence:
//
//   one pass     704-thread workgroups: waves 0..7 take the 8 rows of the level-0 tile, waves 8..10 the tile's queries
//                of levels 1..3 IN THE SAME PASS (170 queries at the R50 shapes = 10.6 waves); no second round.
//   <= 80 VGPRs  6 waves per SIMD, so that two such workgroups (22 waves) share a CU.  The registers come from the
//                lane roles below and from a two-row (not three-row) read pipeline -- with 5-6 waves per SIMD the LDS
//                latency is covered by the other waves.
//   lane roles   a QUAD of lanes owns one (query, head) pair as before and lane k still accumulates the 16-byte pieces
//                k and k + 4 of a pixel (bank-conflict-free ds_read_b128, see msda_fwd_win.hip), but lane k now
//                prepares POINT k of every level, and the pass walks the levels in order: a level's constants (H, W,
//                window origin, window geometry) are wave-uniform scalars / compile-time constants instead of per-lane
//                selections, and only the current level's prepared sample (6 registers, 12 while the next level's is
//                built) is live instead of all four (24).
//   preamble     2-D grid (head, item): no integer divisions by the head count; level-0 geometry in the wave that
//                needs it; 8 instead of 11 window-DMA instructions per wave.
//
// Unchanged: the exact tile partition of the S queries, window sizes and placement by the mean top-left corner of the
// tile's own in-range samples, LDS-DMA staging with out-of-image slots as zeros, DPP quad broadcasts of the prepared
// samples, the far path (an in-range sample with a corner outside its window takes raw buffer loads; correctness
// never depends on where the windows are).
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "../msda_common.hpp"

namespace msda {
namespace {

constexpr int kL0Waves = 8, kRestWaves = 3, kWaves = kL0Waves + kRestWaves, kT = kWaves * 64;
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
constexpr int kZeroOff = kSlots * 128;                          // all-zero region: target of dead / far samples
constexpr int kZeroBytes = kWW[0] * 128 + 256;                  // a bottom-row read lands at most one level-0 row further
struct Meta {
  int sum[4][4];                                                // per level: sum dx, sum dy, count, - (placement)
  int lvl[4][4];                                                // per level: H, W, first pixel, - (far path: level picked per quad)
  int next[4];                                                  // persistent grid: the workgroup's next item
};
// Persistent grid (launch_forward_win2): K workgroups per head walk the head's items; after its first item (= its index) a
// workgroup draws the next one from a per-head ticket counter, so that nobody idles while a neighbour still has a fifth
// item.  One counter set per launch (ring of kTicketSets, the host passes the index): launches on different streams do not
// share counters; the last workgroup of a head to finish puts the head's two words back to zero.
constexpr int kTicketSets = 32, kTicketHeads = 64, kPersistDefault = 1 << 30;
__device__ unsigned g_win2_tickets[kTicketSets][kTicketHeads][2];   // [set][head]: tickets drawn, workgroups done
constexpr int kMetaOff = kZeroOff + kZeroBytes;
constexpr int kLdsBytes = kMetaOff + ((sizeof(Meta) + 15) / 16) * 16;
static_assert(kLdsBytes <= 80 * 1024, "two workgroups per CU");

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
// the same with a wave-uniform multiplier in a scalar register (one scalar operand per VALU instruction on gfx9): with
// "v" for everything the compiler keeps VGPR copies of M * 128, W_l, ... alive through the whole kernel
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
// A wave-uniform value computed on the vector ALU (float quotients) into a SCALAR register.  As inline asm: the builtin
// readfirstlane of a provably uniform value is folded away, the value then stays in a VGPR, the 64-bit address arithmetic
// that depends on it follows it into the VALU and the buffer descriptors built from it need waterfall loops.
__device__ __forceinline__ int to_sgpr(int v) {
  int r;
  // hazards the compiler's recogniser cannot see inside the asm (gfx940+): a VALU write of the VGPR needs a wait state
  // before v_readfirstlane reads it (without the leading s_nop most waves read a STALE register -- found the hard way);
  // the SGPR it writes needs 5 before a VMEM instruction uses it
  asm volatile("s_nop 1\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 4" : "=s"(r) : "v"(v));
  return r;
}
// 4-way select by a per-lane index (two bit tests, three v_cndmask_b32): `l == 0 ? a : l == 1 ? b : ...` on scalar
// registers compiles into trees of exec-masked branches
template <typename T>
__device__ __forceinline__ T sel4(bool b0, bool b1, T a0, T a1, T a2, T a3) {
  const T t = b0 ? a1 : a0, u = b0 ? a3 : a2;
  return b1 ? u : t;
}

// Phase timestamps (profiling builds only: -DMSDA_WIN2_PROF; tools/win2_prof.py): lane 0 of EVERY wave writes the 100 MHz
// real-time counter at each phase boundary of the workgroup's first item.
#ifdef MSDA_WIN2_PROF
constexpr int kProfBlocks = 4096, kProfSlots = 16;
__device__ unsigned long long g_win2_prof[kProfBlocks * kWaves * kProfSlots];
#define W2_STAMP(i)                                                                                          \
  do {                                                                                                       \
    const unsigned blk_ = blockIdx.y * gridDim.x + blockIdx.x;                                               \
    if ((threadIdx.x & 63) == 0 && item == kk && blk_ < (unsigned)kProfBlocks)                               \
      g_win2_prof[(blk_ * kWaves + (threadIdx.x >> 6)) * kProfSlots + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define W2_STAMP(i) do { } while (0)
#endif

// Wave priorities (s_setprio outranks age in the issue arbitration).  A workgroup's start-up (scalar loads, geometry,
// locations, placement, window DMA) is a latency chain of a few hundred instructions; next to it on the CU the other
// workgroup runs its LDS pass, ~1500 back-to-back VALU instructions per wave, and being OLDER it wins every issue slot:
// in-kernel timestamps put the start-up at 6-7 us of a 17 us workgroup.  So: high priority until the windows are staged,
// low priority for the throughput part.
#ifndef MSDA_WIN2_PRIO_START
#define MSDA_WIN2_PRIO_START 3
#endif
#ifndef MSDA_WIN2_PRIO_PASS
#define MSDA_WIN2_PRIO_PASS 0
#endif

// Workgroup barrier for LDS traffic only: __syncthreads() is a fence + barrier and waits for vmcnt(0) as well, i.e. for the
// locations that are meant to stay in flight across it.  LDS operations of a CU complete in order: lgkmcnt(0) is enough.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct Smp {      // one prepared NEAR sample (dead and far samples: zero weights, addresses in the zero region)
  v2f wT, wB;     // corner weights (first-top, second-top), (first-bottom, second-bottom); "first" = the pixel whose slot parity this quad reads first
  uint32_t aF, aS;   // LDS byte addresses of the first / second pixel of the top row
};

}  // namespace

#ifndef MSDA_WIN2_WAVES_PER_EU
#define MSDA_WIN2_WAVES_PER_EU 6
#endif
__global__ void __launch_bounds__(kT, MSDA_WIN2_WAVES_PER_EU)
msda_fwd_win2(const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
              const float* __restrict__ loc, const float* __restrict__ attn, Dims d, float* __restrict__ out, int ticket_set) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef MSDA_WIN2_PROF
  {
    const unsigned blk_ = blockIdx.y * gridDim.x + blockIdx.x;
    if ((threadIdx.x & 63) == 0 && blk_ < (unsigned)kProfBlocks)
      g_win2_prof[(blk_ * kWaves + (threadIdx.x >> 6)) * kProfSlots + 14] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  __builtin_amdgcn_s_setprio(MSDA_WIN2_PRIO_START);
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
  unsigned* const tickets = ticket_set >= 0 ? &g_win2_tickets[ticket_set][m][0] : nullptr;   // dynamic item distribution?
  auto retire = [&]() __attribute__((always_inline)) {     // one call per workgroup: the head's last one resets the counters
    if (tickets && tid == 0 && atomicAdd(&tickets[1], 1u) == (unsigned)K - 1u) {
      tickets[0] = 0u;
      tickets[1] = 0u;
    }
  };
  if (kk >= nitems) {                                      // over-provisioned part of the grid
    retire();
    return;
  }

  // the all-zero region, the placement sums, the level table of the far path (visible after the first barrier)
  if (tid < kZeroBytes / 16) *reinterpret_cast<f32x4*>(smem + kZeroOff + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  static_assert(kZeroBytes / 16 <= 256, "zero fill by the first four waves");
  if (tid >= 256 && tid < 272) (&mt.sum[0][0])[tid - 256] = 0;
  if (tid >= 320 && tid < 324) {
    const bool t0 = (tid & 1) != 0, t1 = (tid & 2) != 0;
    *reinterpret_cast<int4*>(&mt.lvl[tid & 3][0]) = make_int4(sel4(t0, t1, lvH[0], lvH[1], lvH[2], lvH[3]), sel4(t0, t1, lvW[0], lvW[1], lvW[2], lvW[3]),
                                                              sel4(t0, t1, lvS[0], lvS[1], lvS[2], lvS[3]), 0);
  }

  const uint32_t pixB = (uint32_t)M * 128u;                // bytes from a pixel of head m to the next one
  const uint32_t hoff = (uint32_t)m * 128u;
  for (int item = kk; item < nitems;) {
    // One item per workgroup at every shape this kernel is launched on in practice, so loop-invariant code motion has
    // nothing to gain here -- but it hoists dozens of per-level / per-wave values out of the loop and spills them at once
    // (the kernel lives at the 80-register limit of 6 waves per SIMD).  Everything the body derives values from passes
    // through an empty asm (in place: no second copy stays alive) at the top of the loop.
    int ntiles_ = ntiles, TX_ = TX;
    asm volatile("" : "+s"(wv), "+s"(ntiles_), "+s"(TX_));
#pragma unroll
    for (int l = 0; l < 4; ++l) asm volatile("" : "+s"(lvH[l]), "+s"(lvW[l]), "+s"(lvS[l]));
    W2_STAMP(0);
    unsigned drawn = 0;
    if (tickets && tid == 0) drawn = atomicAdd(&tickets[0], 1u);   // in flight until it is published below
    // quotients by v_rcp_f32: x + 0.5 is at least 0.5 / divisor away from an integer, far beyond the 1 ulp of the reciprocal
    const int b = to_sgpr((int)(((float)item + 0.5f) * __builtin_amdgcn_rcpf((float)ntiles_)));
    const int64_t pair_img = (int64_t)b * d.Lq * M + m;     // pair (query 0, head m) of this item's image: uniform bases, 32-bit per-lane offsets
    const __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(value) + (int64_t)b * d.S * M * 32, 0, (int)((uint32_t)d.S * pixB), 0x00020000);
    // ---- tile geometry.  Level-k pixels [f(t), f(t + 1)) with f(t) = ceil(t * T * n / n0 - 1/2) are the ones whose
    // centre falls into tile t -- an exact partition as long as every workgroup evaluates the same expression, which is
    // all that correctness needs (the expression of msda_fwd_win).  On level 0 it is f(t) = T * t: scalar arithmetic;
    // the waves of levels 1..3 evaluate it in float, lane k for level k (below).
    const int tile_ = item - b * ntiles;
    int ty = to_sgpr((int)(((float)tile_ + 0.5f) * __builtin_amdgcn_rcpf((float)TX_)));
    int tx = tile_ - ty * TX;
    const bool l0 = wv < kL0Waves;                           // a wave of the level-0 rows?
    int ogx[4], ogy[4];                                      // window origins
    int npass = 1;

    for (int pass = 0; pass < npass; ++pass) {
      // per-lane constants are re-derived in every pass and the level constants pass through an empty asm again (in
      // place): whatever the optimiser can prove invariant in THIS loop it hoists in front of it and spills
      asm volatile("" : "+s"(wv), "+s"(tx), "+s"(ty));
#pragma unroll
      for (int l = 0; l < 4; ++l) asm volatile("" : "+s"(lvH[l]), "+s"(lvW[l]), "+s"(lvS[l]));
      int ln;                                                // lane of the wave
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      const int pq = ln >> 2, k = ln & 3;                    // quad of the wave; this lane's point / 16-byte piece
      const bool k0 = (k & 1) != 0, k1 = (k & 2) != 0;
      const int cls_a = (ln >> 3) & 1, cls_e = (ln >> 4) & 1;   // bank class of the quad: half read first, parity read first
      // this lane's channels: the 16-byte pieces k and k + 4 of a pixel, i.e. a quad reads / writes 64 contiguous bytes
      // per instruction; c0 = the piece read first, c0 ^ 64 the other
      const uint32_t c0 = (uint32_t)(16 * k + 64 * cls_a);
      // ---- this quad's query --------------------------------------------------------------------------------------
      bool live;
      uint32_t qidx;
      if (l0) {                                              // wave = tile row, quad = tile column
        const int xs0 = kTW * tx, ys0 = kTH * ty;
        live = (pq < min(kTW, lvW[0] - xs0)) && (wv < min(kTH, lvH[0] - ys0));
        qidx = (uint32_t)(lvS[0] + (ys0 + wv) * lvW[0] + xs0 + pq);
      }
      W2_STAMP(1);
      if (pass == 0) lds_barrier();                          // #1: LDS set-up visible / everybody left the previous item
      W2_STAMP(2);
      if (!l0) {                                             // ri-th query of levels 1..3 (after the barrier: nobody waits for this)
        const int gW = sel4(k0, k1, lvW[0], lvW[1], lvW[2], lvW[3]), gH = sel4(k0, k1, lvH[0], lvH[1], lvH[2], lvH[3]);
        const float fxs = (float)(kTW * gW) * __builtin_amdgcn_rcpf((float)lvW[0]), fys = (float)(kTH * gH) * __builtin_amdgcn_rcpf((float)lvH[0]);
        const int gxs = min(max((int)ceilf((float)tx * fxs - 0.5f), 0), gW);
        const int xe = tx == TX - 1 ? gW : min(max((int)ceilf((float)(tx + 1) * fxs - 0.5f), gxs), gW);
        const int gys = min(max((int)ceilf((float)ty * fys - 0.5f), 0), gH);
        const int ye = ty == TY - 1 ? gH : min(max((int)ceilf((float)(ty + 1) * fys - 0.5f), gys), gH);
        const int gnx = xe - gxs, cnt = gnx * (ye - gys);
        const int e1 = __builtin_amdgcn_readlane(cnt, 1), e2 = e1 + __builtin_amdgcn_readlane(cnt, 2);
        const int nrest = e2 + __builtin_amdgcn_readlane(cnt, 3);
        // these waves walk their queries kRestQuads at a time (one pass at the R50 shapes; pyramids whose upper levels
        // are large relative to level 0 take more)
        npass = max(1, (nrest + kRestQuads - 1) / kRestQuads);
        const int ri = pass * kRestQuads + (wv - kL0Waves) * 16 + pq;
        live = ri < nrest;
        const int ql = 1 + (ri >= e1 ? 1 : 0) + (ri >= e2 ? 1 : 0);
        const int j = ri - (ri >= e2 ? e2 : ri >= e1 ? e1 : 0);
        const int src = ((ln & ~3) | ql) << 2;               // lane ql of the quad holds level ql's rectangle
        const int qxs = __builtin_amdgcn_ds_bpermute(src, gxs), qys = __builtin_amdgcn_ds_bpermute(src, gys);
        const int qnx = __builtin_amdgcn_ds_bpermute(src, gnx);
        const int Wq = __builtin_amdgcn_ds_bpermute(src, gW);
        const int Sq = __builtin_amdgcn_ds_bpermute(src, sel4(k0, k1, lvS[0], lvS[1], lvS[2], lvS[3]));
        const int yy = (int)(((float)j + 0.5f) * __builtin_amdgcn_rcpf((float)max(qnx, 1)));
        qidx = mad_u24((uint32_t)(qys + yy), (uint32_t)Wq, (uint32_t)(Sq + qxs + j)) - mad_u24((uint32_t)yy, (uint32_t)qnx, 0u);
      }
      // ---- locations and weights of point k on the four levels (the quad reads 32 + 16 contiguous bytes per level) --
      live = live && qidx < (uint32_t)d.Lq;                  // (shapes whose pixel count exceeds num_query: never outside the tensors)
      const uint32_t pair = mul_u24_s(live ? qidx : 0u, (uint32_t)M);   // (query, head 0) pair within the image; the head sits in the base pointers
      v2f lc[4];
      float sa[4];
#pragma unroll
      for (int l = 0; l < 4; ++l) { lc[l] = v2f{0.f, 0.f}; sa[l] = 0.f; }
      if (live) {
        const v2f* lp = reinterpret_cast<const v2f*>(loc + pair_img * 32 + (pair * 32u + 2u * (uint32_t)k));
        const float* ap = attn + pair_img * 16 + (pair * 16u + (uint32_t)k);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          lc[l] = __builtin_nontemporal_load(lp + 4 * l);
          sa[l] = __builtin_nontemporal_load(ap + 4 * l);
        }
      }

      // sample coordinates (the reference's arithmetic, cuh:282-288 and :38-46); waits for the locations
      // (x, y) of the sample on level l in pixels, and whether it is in range
      auto coord = [&](int l, bool& in) __attribute__((always_inline)) {
        const v2f fWH = {(float)lvW[l], (float)lvH[l]};
        const v2f p = __builtin_elementwise_fma(lc[l], fWH, v2f{-0.5f, -0.5f});
        in = live & (p.y > -1.f) & (p.x > -1.f) & (p.y < fWH.y) & (p.x < fWH.x);
        return p;
      };

      if (pass == 0) {
        if (l0) {
          // ---- window placement: mean top-left corner of the in-range samples of the tile's level-0 queries, per level:
          // reduce-scatter over the quad (lane l ends up with level l), then over the 4 quads of a DPP row.  (The
          // coordinates are computed again after the barrier: kept, they are spilled.) -------------------------------------
          auto quad_scatter = [&](int v0, int v1, int v2, int v3) __attribute__((always_inline)) {
            const int A = (k0 ? v1 : v0) + dppi<0xB1>(k0 ? v0 : v1), B = (k0 ? v3 : v2) + dppi<0xB1>(k0 ? v2 : v3);   // quad_perm [1,0,3,2]
            int R = (k1 ? B : A) + dppi<0x4E>(k1 ? A : B);                                                            // quad_perm [2,3,0,1]
            R += dppi<0x114>(R);                             // row_shr 4
            R += dppi<0x118>(R);                             // row_shr 8: lanes 12..15 of a row hold the row's totals of levels 0..3
            return R;
          };
          int px[4], py[4], pn[4];
#pragma unroll
          for (int l = 0; l < 4; ++l) {
            bool in;
            const v2f p = coord(l, in);
            const int cx = cvt_i32(floorf(p.x)), cy = cvt_i32(floorf(p.y));   // (saturated garbage for poisoned locations: masked)
            px[l] = in ? cx : 0; py[l] = in ? cy : 0; pn[l] = in ? 1 : 0;
          }
          const int ax = quad_scatter(px[0], px[1], px[2], px[3]);
          const int ay = quad_scatter(py[0], py[1], py[2], py[3]);
          const int an = quad_scatter(pn[0], pn[1], pn[2], pn[3]);
          if ((ln & 12) == 12 && an != 0) {
            atomicAdd(&mt.sum[k][0], ax);
            atomicAdd(&mt.sum[k][1], ay);
            atomicAdd(&mt.sum[k][2], an);
          }
        }
        W2_STAMP(4);                                         // (level-0 waves) coordinates + placement sums done
        lds_barrier();                                       // #2 (the waves of levels 1..3 arrive with their loads still in flight)
        W2_STAMP(5);
        int myOx, myOy;
        {
          const int4 sm = *reinterpret_cast<const int4*>(&mt.sum[k][0]);
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
          ogx[l] = __builtin_amdgcn_readlane(myOx, l);
          ogy[l] = __builtin_amdgcn_readlane(myOy, l);
        }
        W2_STAMP(6);                                         // origins known
      }

      // ---- near or far?  (near = all four corners inside the level's window, or outside the image) ------------------
      v2f xy[4];
      uint32_t nb = 0, fm = 0;                               // near bits (bit l); the pair's far samples (bit 4 * point + level)
      auto classify = [&]() __attribute__((always_inline)) {
        uint32_t farmask = 0;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          bool in;
          xy[l] = coord(l, in);
          const int cx = cvt_i32(floorf(xy[l].x)), cy = cvt_i32(floorf(xy[l].y));
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

      f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};   // channels at c0 and at c0 ^ 64

      // ---- far samples: raw buffer loads, one far sample per quad and step ----------------------------------------------
      {
        // one far step: the top row's four loads, then the bottom row's (16 registers in flight, not 32: the kernel lives at 80)
        auto far_step = [&]() __attribute__((always_inline)) {
          const bool has = fm != 0u;
          const int idx = has ? __builtin_ctz(fm) : 0;
          fm &= fm - 1u;
          const int fl_ = idx & 3, ps = idx >> 2;               // level and point (= preparing lane) of the far sample
          const int src = ((ln & ~3) | ps) << 2;             // byte address of the preparing lane for ds_bpermute
          // every lane selects its own candidate on level fl_, the quad pulls the preparing lane's and redoes the
          // (cheap) sample arithmetic -- far samples are a few per cent, their state is not kept around
          const bool c1 = (fl_ & 1) != 0, c2 = (fl_ & 2) != 0;
          const int cx_ = (int)__float_as_uint(sel4(c1, c2, xy[0].x, xy[1].x, xy[2].x, xy[3].x));
          const int cy_ = (int)__float_as_uint(sel4(c1, c2, xy[0].y, xy[1].y, xy[2].y, xy[3].y));
          const int ca_ = (int)__float_as_uint(sel4(c1, c2, sa[0], sa[1], sa[2], sa[3]));
          // quads without a far sample left run along with zero weights: their stand-in coordinates must be finite
          const uint32_t hm = has ? 0xffffffffu : 0u;
          const float fxv = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, cx_) & hm);
          const float fyv = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, cy_) & hm);
          const float fav = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, ca_) & hm);
          const int4 lv = *reinterpret_cast<const int4*>(&mt.lvl[fl_][0]);   // the far sample's level: H, W, first pixel
          const int fH_ = lv.x, fW_ = lv.y, fS_ = lv.z;
          const uint32_t rowG = mul_u24_s((uint32_t)fW_, pixB);
          const float xf = floorf(fxv), yf = floorf(fyv);
          const float lw = fxv - xf, lh = fyv - yf;
          const int fx0 = (int)xf, fy0 = (int)yf;                // in range or 0 for the stand-ins
          const bool t_ok = has && fy0 >= 0, b_ok = has && fy0 + 1 <= fH_ - 1, l_ok = fx0 >= 0, r_ok = fx0 + 1 <= fW_ - 1;
          const float wt = (1.f - lh) * fav, wb = lh * fav;
          // 24-bit multiply-adds (pixel index < 2^24, pitch < 2^24) on the CLAMPED top-left pixel: with fy0 or fx0 = -1 the
          // live corners sit in row / column 0, and a 24-bit product of a negative index is not what a 32-bit one wraps to
          const int cy = max(fy0, 0), cx = max(fx0, 0);
          const uint32_t off = mad_u24_s(mad_u24((uint32_t)cy, (uint32_t)fW_, (uint32_t)(fS_ + cx)), pixB, c0);
          const uint32_t dx = fx0 >= 0 ? pixB : 0u, dy = fy0 >= 0 ? rowG : 0u;   // step to the right / bottom neighbour
          const uint32_t o1 = (t_ok && l_ok) ? off : kOobOffset;
          const uint32_t o2 = (t_ok && r_ok) ? off + dx : kOobOffset;
          const uint32_t o3 = (b_ok && l_ok) ? off + dy : kOobOffset;
          const uint32_t o4 = (b_ok && r_ok) ? off + dy + dx : kOobOffset;
          auto row = [&](uint32_t oL, uint32_t oR, float wrow) __attribute__((always_inline)) {
            const f32x4 La = buffer_load_f32x4(vsrc, oL, hoff), Lb = buffer_load_f32x4(vsrc, oL ^ 64u, hoff);
            const f32x4 Ra = buffer_load_f32x4(vsrc, oR, hoff), Rb = buffer_load_f32x4(vsrc, oR ^ 64u, hoff);
            const float wl = wrow * (1.f - lw), wr = wrow * lw;
            const v2f WL = {wl, wl}, WR = {wr, wr};
#pragma unroll
            for (int h = 0; h < 2; ++h) {                       // channel pairs: v_pk_fma_f32
              v2f a = {accA[2 * h], accA[2 * h + 1]}, bb = {accB[2 * h], accB[2 * h + 1]};
              a = __builtin_elementwise_fma(WL, v2f{La[2 * h], La[2 * h + 1]}, a);
              bb = __builtin_elementwise_fma(WL, v2f{Lb[2 * h], Lb[2 * h + 1]}, bb);
              a = __builtin_elementwise_fma(WR, v2f{Ra[2 * h], Ra[2 * h + 1]}, a);
              bb = __builtin_elementwise_fma(WR, v2f{Rb[2 * h], Rb[2 * h + 1]}, bb);
              accA[2 * h] = a.x; accA[2 * h + 1] = a.y; accB[2 * h] = bb.x; accB[2 * h + 1] = bb.y;
            }
            asm volatile("" : "+v"(accA), "+v"(accB));
          };
          row(o1, o2, wt);
          row(o3, o4, wb);
        };
        // ---- stage the four windows: LDS-DMA, one instruction = 8 consecutive slots (1 KB) of ONE level per wave.
        // Straight-line code with the same number of instructions in every wave (a wave without a chunk left in a
        // level issues an out-of-range one into the all-zero region, which costs no memory access) ----------------------
        auto stage_windows = [&]() __attribute__((always_inline)) {
          const uint32_t chunk = (uint32_t)(ln & 7) * 16u;
          const int sub = ln >> 3;
            auto stage_level = [&](auto ltag) __attribute__((always_inline)) {
            constexpr int LV = decltype(ltag)::value;
            constexpr int WW = kWW[LV], C0 = kBase[LV] / 8, C1 = kBase[LV + 1] / 8;
            constexpr int kSteps = (C1 - C0 + kWaves - 1) / kWaves;
            constexpr int kDR = (8 * kWaves) / WW, kDC = (8 * kWaves) % WW;
            const int Hs = lvH[LV], Ws = lvW[LV], xS = ogx[LV] + lvS[LV], oy = ogy[LV], ox = ogx[LV];
            int i = C0 + wv;                                  // this wave's first chunk of the level
            int subv = sub;
            asm volatile("" : "+v"(subv));                    // opaque: the level's start is computed HERE
            const int rel = 8 * wv + subv;                    // slot of this lane in the level's window
            int r = (int)(((float)rel + 0.5f) * (1.f / WW)), c = rel - r * WW;
#pragma unroll
            for (int t = 0; t < kSteps; ++t, i += kWaves) {
              const bool have = i < C1;                         // wave-uniform
              const int y = oy + r;
              const bool inside = have && (unsigned)y < (unsigned)Hs && (unsigned)(ox + c) < (unsigned)Ws;
              // pixel index < 2^24 and pixel pitch M * 128 < 2^24 by win2_forward_ok: two full-rate 24-bit multiply-adds
              const uint32_t pix = mad_u24_s((uint32_t)y, (uint32_t)Ws, (uint32_t)(xS + c));
              const uint32_t in_off = mad_u24_s(pix, pixB, chunk);
              const uint32_t off = inside ? in_off : kOobOffset;
              const int dst = have ? i * 1024 : kZeroOff;
              __builtin_amdgcn_raw_ptr_buffer_load_lds(vsrc, (__attribute__((address_space(3))) void*)(smem + dst), 16,
                                                       off, hoff, 0, 0);
              if (t + 1 < kSteps) {
                c += kDC; r += kDR;
                if (kDC != 0 && c >= WW) { c -= WW; r += 1; }
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          };
          stage_level(std::integral_constant<int, 0>{});
          stage_level(std::integral_constant<int, 1>{});
          stage_level(std::integral_constant<int, 2>{});
          stage_level(std::integral_constant<int, 3>{});
        };
        if (pass == 0) {
          // the DMA depends on the origins only: the waves of levels 1..3 issue it with their own locations still in flight
          stage_windows();
          W2_STAMP(8);
        }
        classify();                                          // sample coordinates (the reference's arithmetic, cuh:282-288 and :38-46): waits for the locations
        W2_STAMP(9);
        // far steps while the windows travel (the first wait for far loads covers the wave's own DMA instructions, which
        // are older in the same queue)
        while (__ballot(fm != 0u)) far_step();
        if (pass == 0) {
          if (tickets && tid == 0) mt.next[0] = K + (int)drawn;
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of the windows has landed
          W2_STAMP(10);
          lds_barrier();                                       // #3 ... and everybody else's
          W2_STAMP(11);
          if (ln < 16 && wv == 0) (&mt.sum[0][0])[ln] = 0;      // the next item's sums (everybody has read this item's)
        }
      }
      W2_STAMP(12);                                          // further far steps done
      __builtin_amdgcn_s_setprio(MSDA_WIN2_PRIO_PASS);
      // ---- near samples: 4 levels x 4 points x 4 corners x 2 halves from the LDS windows -------------------------------
      v2f aA0 = {accA[0], accA[1]}, aA1 = {accA[2], accA[3]}, aB0 = {accB[0], accB[1]}, aB1 = {accB[2], accB[3]};
      const uint32_t zero_first = smem_base + kZeroOff + 128u * (uint32_t)cls_e;        // parity cls_e; the other parity: ^ 128
      static_assert(kZeroOff % 256 == 0, "zero region: slot parity by address bit 7");
      // this lane's point on level LV (the reference's bilinear weights with the attention weight folded in)
      auto prepare = [&](auto ltag) __attribute__((always_inline)) {
        constexpr int LV = decltype(ltag)::value;
        Smp s;
        const v2f fl = {floorf(xy[LV].x), floorf(xy[LV].y)};
        v2f fr = xy[LV] - fl;                                  // (fx, fy); inf - inf / NaN for poisoned locations ...
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
        const uint32_t tl = smem_base + (uint32_t)(kBase[LV] * 128) + (uint32_t)(__mul24(ry, kWW[LV]) + cx) * 128u;
        s.aF = near ? tl + (sw << 7) : zero_first;
        s.aS = near ? tl + 128u - (sw << 7) : (zero_first ^ 128u);
        return s;
      };
      // Half rows (one pixel of a corner row = this lane's two 16-byte pieces, 8 registers) through a ring of three register
      // sets: half g + 3 is requested into the set that consuming half g has just freed -- 24 registers in flight instead
      // of the 48 of msda_fwd_win; with 5-6 waves per SIMD the other waves cover the rest of the LDS latency.
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
      {
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        Half h0, h1, h2;
        Adr ad;
        // generated sequence: consume half g, request half g + 3 (g = 16 * level + 4 * point + half); the next level's sample is
        // prepared just before its first half is requested
        Smp s0 = prepare(I0{}), s1;
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
      }
      W2_STAMP(13);                                          // LDS pass done
      __builtin_amdgcn_s_setprio(MSDA_WIN2_PRIO_START);
      if (live) {   // a quad writes 2 x 64 contiguous bytes
        float* op = out + pair_img * 32 + (pair * 32u + 4u * (uint32_t)k);
        __builtin_nontemporal_store(f32x4{aA0.x, aA0.y, aA1.x, aA1.y}, reinterpret_cast<f32x4*>(op + 16 * cls_a));
        __builtin_nontemporal_store(f32x4{aB0.x, aB0.y, aB1.x, aB1.y}, reinterpret_cast<f32x4*>(op + 16 * (cls_a ^ 1)));
      }
#ifdef MSDA_WIN2_PROF
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      W2_STAMP(15);                                          // output stores acknowledged
#endif
    }
    // the next item: the ticket published before barrier #3 (nobody overwrites it before everybody is past the next
    // item's barrier #1), or the static stride
    item = tickets ? __builtin_amdgcn_readfirstlane(*(volatile int*)&mt.next[0]) : item + K;
  }
  retire();
}

#ifdef MSDA_WIN2_PROF
extern "C" int msda_debug_read_prof2(void* dst, int nblocks) {
  if (nblocks > kProfBlocks) nblocks = kProfBlocks;
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_win2_prof), (size_t)nblocks * kWaves * kProfSlots * 8, 0, hipMemcpyDeviceToHost);
}
#endif

bool win2_forward_ok(const Dims& d) {
  // (the last condition keeps the work-item index, and item + 0.5, exact in float: the kernel splits it into (image,
  // tile) with a reciprocal; the grid's y extent is the number of workgroups per head)
  return d.D == 32 && d.P == 4 && d.L == 4 && d.Lq == d.S && d.S >= 1024 && d.M <= 65535 &&
         (int64_t)d.S * d.M * 128 < (int64_t)kOobOffset && d.N <= 65535 &&
         (int64_t)d.N * ((d.S + 127) / 128) < ((int64_t)1 << 22);
}

int launch_forward_win2(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn,
                        const Dims& d, float* out, hipStream_t stream) {
  static std::atomic<uint64_t> lds_opted_in{0};
  const void* fn = reinterpret_cast<const void*>(msda_fwd_win2);
  if (int rc = ensure_dynamic_lds(fn, kLdsBytes, lds_opted_in)) return rc;
  // Workgroups per head: the host only knows S, so ceil(S / 128) per image -- at least the tile count of any pyramid
  // whose level 0 holds <= ~3/4 of the pixels; the surplus exits at once, a workgroup of a pyramid with more tiles walks
  // items kk, kk + K, ...  Head m = blockIdx.x, i.e. (by the observed round-robin placement of the linear workgroup id)
  // XCD m % 8 only ever touches head m's slice of `value` when M is a multiple of 8.
  int K = d.N * ((d.S + 127) / 128);
  // Default: a persistent grid of two resident workgroups per CU spread over the heads, static stride over the items
  // (83.7-86.0 us against 90.4-93.1 for one workgroup per item: 4.4 us pass between a workgroup's last store and its successor's
  // first instruction; per-head ticket counters LOSE to the static stride, 91.5 us -- profiles/r03_forward_window_analysis.txt).
  // MSDA_WIN2_PERSIST=n (A/B switch): 0 = one workgroup per item, -n = n workgroups per head with the static stride,
  // +n = n workgroups per head drawing tickets.
  static const int persist_env = std::getenv("MSDA_WIN2_PERSIST") ? std::atoi(std::getenv("MSDA_WIN2_PERSIST")) : kPersistDefault;
  int persist = persist_env;
  if (persist == kPersistDefault) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    persist = -((2 * cus + d.M - 1) / d.M);
  }
  static std::atomic<unsigned> launch_seq{0};
  int ticket_set = -1;
  if (persist != 0) {
    const int want = persist > 0 ? persist : -persist;
    if (want < K) {                                        // fewer items than that: one workgroup per item is the same thing
      K = want;
      if (persist > 0 && d.M <= kTicketHeads) ticket_set = (int)(launch_seq.fetch_add(1, std::memory_order_relaxed) % kTicketSets);
    }
  }
  if (K < 1) K = 1;
  if (K > 65535) K = 65535;
  hipLaunchKernelGGL(msda_fwd_win2, dim3((unsigned)d.M, (unsigned)K), dim3(kT), kLdsBytes, stream, value, shapes, lsi, loc,
                     attn, d, out, ticket_set);
  return (int)hipGetLastError();
}

}  // namespace msda
