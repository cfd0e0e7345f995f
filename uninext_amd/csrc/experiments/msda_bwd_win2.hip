// msda_bwd_win2 -- MSDeformAttn backward for encoder-style calls (Lq == S), two-phase: ONE set of LDS windows per workgroup,
// first holding `value`, then the gradient accumulators, so that TWO 512-thread workgroups share a CU.  fp32, D = 32,
// L = P = 4.  gfx950 only.  Replaces, for these calls, the work of ops/src/cuda/ms_deform_im2col_cuda.cuh:301-403 (+ :87-159).
//
// msda_bwd_win (round 3) keeps value windows AND accumulator windows (156 KB): one workgroup per CU, whose per-item start-up,
// far samples, flush and barrier skew (23 us of a 37 us item; the LDS-bound pass is 14 us) have nothing to overlap with
// (profiles/r03_backward_window.txt).  Here an item is
//
//   gather   value windows staged by LDS-DMA (74 KB, as in the forward window kernel); every (query, head) pair of the item --
//            round 0: the 8 x 16 level-0 tile, wave = tile row, quad = pixel; later rounds: the tile's queries of levels 1..3 --
//            gathers its 16 samples and forms grad_attn_weight / grad_sampling_loc (cuh:113-158).  Far samples (a corner outside
//            its window): half a wave per sample, coalesced corner loads, full-line float atomics for grad_value.
//   barrier, the SAME 74 KB are zeroed, barrier
//   scatter  every pair comes by again (its locations, weights and upstream gradient re-read: they are L2-resident, 208 bytes
//            per pair) and adds weight x attention x grad_out into int32 fixed-point accumulators (ds_add_u32, conflict-free
//            channel rotation, per-item power-of-two scale -- msda_bwd_win.hip / include/msda_hip.h)
//   barrier, flush: every touched pixel inside the image leaves as one full-line float atomic; the next item's DMA overwrites
//            the windows, so nothing is cleared.
//
// What it pays: the sample coordinates twice, 208 re-read bytes per pair, two more barriers per item.  What it buys: while one
// workgroup sits in a barrier, a DMA wait or its flush, the other one of the CU has the LDS and the vector ALUs.
//
// Round 4 result (profiles/r04_backward_two_phase.txt): parity-green on the 22 cases of tools/bwin_check.py, 292 us against 277 us for
// msda_bwd_win on the same box -- the same 92 M vector instructions, 60 % instead of 44 % of the wave cycles waiting.  An experiment
// (`make experiments`, backward variant 7), not in the product library.
#include <cstdlib>
#include <type_traits>

#include "../msda_common.hpp"

namespace msda {
namespace {

constexpr int kWaves = 8, kT = kWaves * 64, kQuads = kT / 4;
// Tile height (A/B: -DMSDA_BWIN2_TH=5|6|8).  The waves below kTH take the tile's rows, the others its queries of levels 1..3 IN THE
// SAME ROUND: at 6 x 16 a tile of the R50 pyramid has 96 + 28..40 queries -- one round of 128 quads, all eight waves equally long
// (at 8 x 16: 128 + ~42, and three of the eight waves went round twice while five waited at the next barrier for 23 of 65 us).
#ifndef MSDA_BWIN2_TH
#define MSDA_BWIN2_TH 6
#endif
constexpr int kTH = MSDA_BWIN2_TH, kTW = 16;
constexpr int kRest0 = (kWaves - kTH) * 16;                     // rest queries that ride in round 0
static_assert(kTH >= 1 && kTH <= kWaves, "tile rows are waves");
constexpr int kWH[4] = {14, 10, 8, 7};
constexpr int kWW[4] = {22, 14, 10, 8};                         // even: slot parity == column parity in every row
constexpr int kBase[5] = {0, 312, 456, 536, 592};               // first slot of each window (multiples of 8: DMA chunks)
constexpr int kSlots = kBase[4];
constexpr int kZeroOff = kSlots * 128;                          // all-zero region: read target of dead / far samples
constexpr int kZeroBytes = kWW[0] * 128 + 256;
struct Meta {
  int sum[2][4][4];                                             // [item parity] per level: sum x0, sum y0, count, - (placement)
  unsigned gmax_bits, amax_bits, pad0, pad1;                    // per item: max |grad_out|, max_pair sum |attn| (float bits)
  int geo[2][4][4];                                             // [item parity] per level: first column / row, width of the item's queries, W | first pixel
  unsigned off_tab[kSlots];                                     // per item: byte offset of the slot's pixel in grad_value (head 0), ~0: outside
};
constexpr int kMetaOff = kZeroOff + kZeroBytes;
constexpr int kLdsBytes = kMetaOff + ((sizeof(Meta) + 15) / 16) * 16;
static_assert(kLdsBytes <= 80 * 1024, "two workgroups per CU");

// Phase timestamps (profiling builds only: -DMSDA_BWIN2_PROF; tools/bwin2_prof.py): lane 0 of every wave, second item only.
#ifdef MSDA_BWIN2_PROF
constexpr int kProfBlocks = 512, kProfSlots = 16;
__device__ unsigned long long g_bwin2_prof[kProfBlocks * kWaves * kProfSlots];
#define BW_STAMP(i)                                                                                          \
  do {                                                                                                       \
    const unsigned blk_ = blockIdx.y * gridDim.x + blockIdx.x;                                               \
    if ((threadIdx.x & 63) == 0 && it == 1 && blk_ < (unsigned)kProfBlocks)                                  \
      g_bwin2_prof[(blk_ * kWaves + (threadIdx.x >> 6)) * kProfSlots + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define BW_STAMP(i) do { } while (0)
#endif

typedef const f32x4 __attribute__((address_space(3)))* lds4;
typedef int __attribute__((address_space(3)))* lds_int_ptr;
typedef float v2f __attribute__((ext_vector_type(2)));

template <int SRC>
__device__ __forceinline__ uint32_t qb(uint32_t v) {   // value held by lane SRC of this lane's quad
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, SRC * 0x55, 0xF, 0xF, true);
}
template <int SRC>
__device__ __forceinline__ float qbf(float v) { return __uint_as_float(qb<SRC>(__float_as_uint(v))); }
template <int CTRL>
__device__ __forceinline__ int dppi(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
__device__ __forceinline__ float dppf(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true)); }
__device__ __forceinline__ float quad_sum(float v) {   // over the 4 lanes of a quad; every lane gets the total
  v += dppf<0xB1>(v);                                  // quad_perm [1,0,3,2]
  v += dppf<0x4E>(v);                                  // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) {   // (a & 0xffffff) * (b & 0xffffff) + c
  uint32_t r;
  asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
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
__device__ __forceinline__ int cvt_i32(float f) {   // saturating, NaN -> 0
  int r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
  return r;
}
__device__ __forceinline__ int cvt_rn_i32(float x) {   // floor(x + 0.5): one VALU instruction
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ int to_sgpr(int v) {   // a wave-uniform value computed on the vector ALU into a SCALAR register
  int r;
  asm volatile("s_nop 1\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 4" : "=s"(r) : "v"(v));
  return r;
}
template <typename T>
__device__ __forceinline__ T sel4(bool b0, bool b1, T a0, T a1, T a2, T a3) {
  const T t = b0 ? a1 : a0, u = b0 ? a3 : a2;
  return b1 ? u : t;
}
__device__ __forceinline__ void lds_add(uint32_t lds_byte_addr, int v) {   // ds_add_u32, no return value
  __hip_atomic_fetch_add(reinterpret_cast<lds_int_ptr>((uintptr_t)lds_byte_addr), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ float abs_or_inf(float x) {  // |x|, +inf for NaN / Inf (so that a max() sees it)
  const float a = fabsf(x);
  return (a <= 3.402823466e+38f) ? a : __builtin_inff();
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace

__global__ void __launch_bounds__(kT, 4)
msda_bwd_win2(const float* __restrict__ grad_out, const float* __restrict__ value, const int64_t* __restrict__ shapes,
              const int64_t* __restrict__ lsi, const float* __restrict__ loc, const float* __restrict__ attn, Dims d,
              float* __restrict__ grad_value, float* __restrict__ grad_loc, float* __restrict__ grad_attn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Meta& mt = *reinterpret_cast<Meta*>(smem + kMetaOff);
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x;
  int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = d.M;
  const int m = blockIdx.x, kk = blockIdx.y, K = gridDim.y;   // workgroup kk of K on head m

  int lvH[4], lvW[4], lvS[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    lvH[l] = (int)shapes[2 * l];
    lvW[l] = (int)shapes[2 * l + 1];
    lvS[l] = (int)lsi[l];
  }
  const int TY = (lvH[0] + kTH - 1) / kTH, TX = (lvW[0] + kTW - 1) / kTW;
  const int ntiles = TY * TX, nitems = d.N * ntiles;
  if (kk >= nitems) return;

  // ---- once per workgroup: zero region, placement sums, scale words ---------------------------------------------------------
  for (int o = tid * 16; o < kZeroBytes; o += kT * 16) *reinterpret_cast<f32x4*>(smem + kZeroOff + o) = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tid < 32) (&mt.sum[0][0][0])[tid] = 0;
  if (tid == 32) { mt.gmax_bits = 0u; mt.amax_bits = 0u; }
  __syncthreads();

  const uint32_t pixB = (uint32_t)M * 128u;                // bytes from a pixel of head m to the next one
  const uint32_t hoff = (uint32_t)m * 128u;

  // the fetched query of this quad: the gather side reads (lc, sa, gA, gB), the scatter side (lc, sa, gi)
  bool live = false;
  uint32_t pair = 0;
  v2f lc[4];                                               // locations and weights of point k on the four levels
  float sa[4];
  f32x4 gA = {0.f, 0.f, 0.f, 0.f}, gB = {0.f, 0.f, 0.f, 0.f};   // upstream gradient: channels of the pieces at c0 / c0 ^ 64 (gather order)
  float gi[8];                                                  // channels k + 4 (t ^ cls8) (accumulation order)
#pragma unroll
  for (int l = 0; l < 4; ++l) { lc[l] = v2f{0.f, 0.f}; sa[l] = 0.f; }
#pragma unroll
  for (int t = 0; t < 8; ++t) gi[t] = 0.f;

  for (int item = kk, it = 0; item < nitems; item += K, ++it) {
    // per-lane constants are re-derived per item and the level constants pass through an empty asm (in place): whatever
    // the optimiser can prove invariant in this loop it hoists in front of it and spills (msda_fwd_win.hip)
    asm volatile("" : "+s"(wv));
#pragma unroll
    for (int l = 0; l < 4; ++l) asm volatile("" : "+s"(lvH[l]), "+s"(lvW[l]), "+s"(lvS[l]));
    int ln;                                                  // lane of the wave
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const int pq = ln >> 2, k = ln & 3;                      // quad of the wave; this lane's point / 16-byte piece
    const bool k0 = (k & 1) != 0, k1 = (k & 2) != 0;
    const int cls_a = (ln >> 3) & 1, cls_e = (ln >> 4) & 1;  // read classes of the quad: half read first, parity read first
    const int cls8 = pq & 7;                                 // accumulation class: rotation of the channel order
    const uint32_t c0 = (uint32_t)(16 * k + 64 * cls_a);     // the 16-byte piece read first; c0 ^ 64 the other
    const uint32_t rot = 16u * (uint32_t)cls8;
    const int b = to_sgpr((int)(((float)item + 0.5f) * __builtin_amdgcn_rcpf((float)ntiles)));
    const int64_t pair_img = (int64_t)b * d.Lq * M + m;     // pair (query 0, head m) of this item's image
    const int64_t img_val = (int64_t)b * d.S * M * 32;
    const __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(value) + img_val, 0, (int)((uint32_t)d.S * pixB), 0x00020000);
    char* const gv_head = reinterpret_cast<char*>(grad_value + img_val) + hoff;   // + pixel byte offset + channel * 4

    auto load_common = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int l = 0; l < 4; ++l) { lc[l] = v2f{0.f, 0.f}; sa[l] = 0.f; }
      if (live) {
        const v2f* lp = reinterpret_cast<const v2f*>(loc + pair_img * 32 + (pair * 32u + 2u * (uint32_t)k));
        const float* ap = attn + pair_img * 16 + (pair * 16u + (uint32_t)k);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          lc[l] = lp[4 * l];
          sa[l] = ap[4 * l];
        }
      }
    };
    auto load_gather = [&]() __attribute__((always_inline)) {   // the gather side's view of the query (live, pair)
      load_common();
      gA = f32x4{0.f, 0.f, 0.f, 0.f}; gB = f32x4{0.f, 0.f, 0.f, 0.f};
      if (live) {
        const float* gp = grad_out + pair_img * 32 + pair * 32u;
        gA = *reinterpret_cast<const f32x4*>(gp + (c0 >> 2));
        gB = *reinterpret_cast<const f32x4*>(gp + ((c0 ^ 64u) >> 2));
      }
    };
    auto load_scatter = [&]() __attribute__((always_inline)) {  // the scatter side's
      load_common();
#pragma unroll
      for (int t = 0; t < 8; ++t) gi[t] = 0.f;
      if (live) {
        const float* gp = grad_out + pair_img * 32 + pair * 32u;
#pragma unroll
        for (int t = 0; t < 8; ++t) gi[t] = gp[k + 4 * (t ^ cls8)];
      }
    };
    // a query of levels 1..3 from its index among the item's: level, position in the level's rectangle -> (live, pair)
    auto rest_query = [&](int ri, int e1_, int e2_, int nrest_, int qxs, int qys, int qnx, int ql) __attribute__((always_inline)) {
      const bool q0 = (ql & 1) != 0, q1 = (ql & 2) != 0;
      const int Wq = sel4(q0, q1, lvW[0], lvW[1], lvW[2], lvW[3]), Sq = sel4(q0, q1, lvS[0], lvS[1], lvS[2], lvS[3]);
      const int j = ri - (ri >= e2_ ? e2_ : ri >= e1_ ? e1_ : 0);
      const int yy = (int)(((float)j + 0.5f) * __builtin_amdgcn_rcpf((float)max(qnx, 1)));
      const uint32_t qidx = mad_u24((uint32_t)(qys + yy), (uint32_t)Wq, (uint32_t)(Sq + qxs + j)) - mad_u24((uint32_t)yy, (uint32_t)qnx, 0u);
      live = ri < nrest_ && qidx < (uint32_t)d.Lq;
      pair = mul_u24_s(live ? qidx : 0u, (uint32_t)M);       // (query, head 0) pair within the image
    };
    // ---- the item's queries: the level-0 tile and the rectangles of levels 1..3 whose centres fall into it (lane k: level k) --
    const int tile = item - b * ntiles;
    const int ty = to_sgpr((int)(((float)tile + 0.5f) * __builtin_amdgcn_rcpf((float)TX)));
    const int tx = tile - ty * TX;
    int e1, e2, nrest;
    int (*const geo)[4] = mt.geo[it & 1];                    // (written here, read behind barrier A; the other parity: the item before)
    {
      const int gW = sel4(k0, k1, lvW[0], lvW[1], lvW[2], lvW[3]);
      const int gH = sel4(k0, k1, lvH[0], lvH[1], lvH[2], lvH[3]);
      const float fxs = (float)(kTW * gW) * __builtin_amdgcn_rcpf((float)lvW[0]), fys = (float)(kTH * gH) * __builtin_amdgcn_rcpf((float)lvH[0]);
      const int gxs = min(max((int)ceilf((float)tx * fxs - 0.5f), 0), gW);
      const int xe = tx == TX - 1 ? gW : min(max((int)ceilf((float)(tx + 1) * fxs - 0.5f), gxs), gW);
      const int gys = min(max((int)ceilf((float)ty * fys - 0.5f), 0), gH);
      const int ye = ty == TY - 1 ? gH : min(max((int)ceilf((float)(ty + 1) * fys - 0.5f), gys), gH);
      const int gnx = xe - gxs;
      const int cnt = gnx * (ye - gys);
      if (tid < 4) *reinterpret_cast<int4*>(&geo[k][0]) = make_int4(gxs, gys, gnx, 0);
      e1 = __builtin_amdgcn_readlane(cnt, 1);
      e2 = e1 + __builtin_amdgcn_readlane(cnt, 2);
      nrest = e2 + __builtin_amdgcn_readlane(cnt, 3);
      if (kTH < kWaves && wv >= kTH) {
        // a wave whose round 0 is rest queries: its loads go out here (the rectangles are in the quad's registers, the LDS copy is
        // only readable behind barrier A); the waves of the tile's rows issued theirs behind the scatter phase of the item before
        const int ri = (wv - kTH) * 16 + pq;
        const int ql = 1 + (ri >= e1 ? 1 : 0) + (ri >= e2 ? 1 : 0);
        const int src = ((ln & ~3) | ql) << 2;               // lane ql of the quad holds level ql's rectangle
        const int qxs = __builtin_amdgcn_ds_bpermute(src, gxs), qys = __builtin_amdgcn_ds_bpermute(src, gys);
        const int qnx = __builtin_amdgcn_ds_bpermute(src, gnx);
        rest_query(ri, e1, e2, nrest, qxs, qys, qnx, ql);
        load_gather();
      }
    }
    // this wave's rounds: round 0 = its row of the tile (waves < kTH) or rest queries (wv - kTH) * 16 + quad; round r > 0 = rest
    // queries kRest0 + (r - 1) * 128 + wv * 16 + quad
    const int nrounds = 1 + (nrest > kRest0 + wv * 16 ? (nrest - kRest0 - wv * 16 + kQuads - 1) / kQuads : 0);
    auto query_of = [&](int rnd) __attribute__((always_inline)) {   // -> live, pair  (rest queries: behind barrier A)
      if (rnd == 0 && wv < kTH) {
        const int xs0 = kTW * tx, ys0 = kTH * ty;
        live = (pq < min(kTW, lvW[0] - xs0)) && (wv < min(kTH, lvH[0] - ys0));
        const uint32_t qidx = (uint32_t)(lvS[0] + (ys0 + wv) * lvW[0] + xs0 + pq);
        live = live && qidx < (uint32_t)d.Lq;
        pair = mul_u24_s(live ? qidx : 0u, (uint32_t)M);
      } else {
        const int ri = rnd == 0 ? (wv - kTH) * 16 + pq : kRest0 + (rnd - 1) * kQuads + wv * 16 + pq;
        const int ql = 1 + (ri >= e1 ? 1 : 0) + (ri >= e2 ? 1 : 0);
        const int4 ge = *reinterpret_cast<const int4*>(&geo[ql][0]);
        rest_query(ri, e1, e2, nrest, ge.x, ge.y, ge.z, ql);
      }
    };
    auto fetch_gather = [&](int rnd) __attribute__((always_inline)) { query_of(rnd); load_gather(); };
    auto fetch_scatter = [&](int rnd) __attribute__((always_inline)) { query_of(rnd); load_scatter(); };
    auto coord = [&](int l, bool& in) __attribute__((always_inline)) {
      const v2f fWH = {(float)lvW[l], (float)lvH[l]};
      const v2f p = __builtin_elementwise_fma(lc[l], fWH, v2f{-0.5f, -0.5f});
      in = live & (p.y > -1.f) & (p.x > -1.f) & (p.y < fWH.y) & (p.x < fWH.x);
      return p;
    };
    // max |grad_out| and max over pairs of sum |attn| of the fetched (gather-side) queries -> the item's scale words
    auto add_stats = [&]() __attribute__((always_inline)) {
      float gm = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) gm = fmaxf(gm, fmaxf(abs_or_inf(gA[c]), abs_or_inf(gB[c])));
      float as = (abs_or_inf(sa[0]) + abs_or_inf(sa[1])) + (abs_or_inf(sa[2]) + abs_or_inf(sa[3]));
      as = quad_sum(as);
      gm = wave_max(gm);
      as = wave_max(as);
      if (ln == 0) {   // non-negative floats order like their bit patterns
        atomicMax(&mt.gmax_bits, __float_as_uint(gm));
        atomicMax(&mt.amax_bits, __float_as_uint(as));
      }
    };

    BW_STAMP(0);
    if (it == 0 && wv < kTH) fetch_gather(0);                // (later items: issued behind the scatter phase of the item before)
    int (*const sums)[4] = mt.sum[it & 1];
    if (wv < kTH) {                                          // (the waves of levels 1..3: their loads are still on the way)
      // ---- window placement: mean top-left corner of the in-range samples of the tile's level-0 queries, per level ----
      auto quad_scatter = [&](int v0, int v1, int v2, int v3) __attribute__((always_inline)) {
        const int A = (k0 ? v1 : v0) + dppi<0xB1>(k0 ? v0 : v1), B = (k0 ? v3 : v2) + dppi<0xB1>(k0 ? v2 : v3);
        int R = (k1 ? B : A) + dppi<0x4E>(k1 ? A : B);
        R += dppi<0x114>(R);                             // row_shr 4
        R += dppi<0x118>(R);                             // row_shr 8: lanes 12..15 of a row hold the row's totals
        return R;
      };
      int px[4], py[4], pn[4];
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        bool in;
        const v2f p = coord(l, in);
        const int cx = cvt_i32(floorf(p.x)), cy = cvt_i32(floorf(p.y));
        px[l] = in ? cx : 0; py[l] = in ? cy : 0; pn[l] = in ? 1 : 0;
      }
      const int ax = quad_scatter(px[0], px[1], px[2], px[3]);
      const int ay = quad_scatter(py[0], py[1], py[2], py[3]);
      const int an = quad_scatter(pn[0], pn[1], pn[2], pn[3]);
      if ((ln & 12) == 12 && an != 0) {
        atomicAdd(&sums[k][0], ax);
        atomicAdd(&sums[k][1], ay);
        atomicAdd(&sums[k][2], an);
      }
    }
    // barrier A: the placement sums are complete -- and every wave is through the flush of the item before (it comes in front
    // of this point in every wave), so the windows may be overwritten
    BW_STAMP(1);                                             // loads arrived, maxima + placement sums added
    lds_barrier();
    BW_STAMP(2);
    int ogx[4], ogy[4];                                      // window origins of the item
    {
      const int4 sm = *reinterpret_cast<const int4*>(&sums[k][0]);
      if (tid < 16) (&mt.sum[(it & 1) ^ 1][0][0])[tid] = 0;  // the next item's sums: nobody reads or adds to them now
      const int myWW = sel4(k0, k1, kWW[0], kWW[1], kWW[2], kWW[3]), myWH = sel4(k0, k1, kWH[0], kWH[1], kWH[2], kWH[3]);
      const int myW = sel4(k0, k1, lvW[0], lvW[1], lvW[2], lvW[3]), myH = sel4(k0, k1, lvH[0], lvH[1], lvH[2], lvH[3]);
      const float inv = __builtin_amdgcn_rcpf((float)max(sm.z, 1));
      int myOx = (int)floorf((float)sm.x * inv + 0.5f) - (myWW - 2) / 2;
      int myOy = (int)floorf((float)sm.y * inv + 0.5f) - (myWH - 2) / 2;
      myOx = max(-1, min(myOx, myW + 1 - myWW));
      myOy = max(-1, min(myOy, myH + 1 - myWH));
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        ogx[l] = __builtin_amdgcn_readlane(myOx, l);
        ogy[l] = __builtin_amdgcn_readlane(myOy, l);
      }
    }
    // ---- stage the four value windows: LDS-DMA, one instruction = 8 consecutive slots (1 KB) of ONE level per wave ----
    {
      const uint32_t chunk = (uint32_t)(ln & 7) * 16u;
      const int sub = ln >> 3;
      auto stage_level = [&](auto ltag) __attribute__((always_inline)) {
        constexpr int LV = decltype(ltag)::value;
        constexpr int WW = kWW[LV], C0 = kBase[LV] / 8, C1 = kBase[LV + 1] / 8;
        constexpr int kSteps = (C1 - C0 + kWaves - 1) / kWaves;
        constexpr int kDR = (8 * kWaves) / WW, kDC = (8 * kWaves) % WW;
        const int Hs = lvH[LV], Ws = lvW[LV], xS = ogx[LV] + lvS[LV], oy = ogy[LV], ox = ogx[LV];
        int i = C0 + wv;
        const int rel = 8 * wv + sub;
        int r = (int)(((float)rel + 0.5f) * (1.f / WW)), c = rel - r * WW;
#pragma unroll
        for (int t = 0; t < kSteps; ++t, i += kWaves) {
          const bool have = i < C1;
          const int y = oy + r;
          const bool inside = have && (unsigned)y < (unsigned)Hs && (unsigned)(ox + c) < (unsigned)Ws;
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
        }
      };
      stage_level(std::integral_constant<int, 0>{});
      stage_level(std::integral_constant<int, 1>{});
      stage_level(std::integral_constant<int, 2>{});
      stage_level(std::integral_constant<int, 3>{});
    }
    // where the flush will send each accumulator slot (the windows travel meanwhile): thread p computes slots p and p + 512
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int p = tid + h * kT;
      if (p < kSlots) {
        const int l = (p >= kBase[1] ? 1 : 0) + (p >= kBase[2] ? 1 : 0) + (p >= kBase[3] ? 1 : 0);
        const bool e0 = (l & 1) != 0, e1b = (l & 2) != 0;
        const int rel = p - sel4(e0, e1b, kBase[0], kBase[1], kBase[2], kBase[3]);
        const int ww = sel4(e0, e1b, kWW[0], kWW[1], kWW[2], kWW[3]);
        const int r = (int)(((float)rel + 0.5f) * __builtin_amdgcn_rcpf((float)ww)), c = rel - r * ww;
        const int y = sel4(e0, e1b, ogy[0], ogy[1], ogy[2], ogy[3]) + r, x = sel4(e0, e1b, ogx[0], ogx[1], ogx[2], ogx[3]) + c;
        const int Hl = sel4(e0, e1b, lvH[0], lvH[1], lvH[2], lvH[3]), Wl = sel4(e0, e1b, lvW[0], lvW[1], lvW[2], lvW[3]);
        const int Sl = sel4(e0, e1b, lvS[0], lvS[1], lvS[2], lvS[3]);
        const bool inside = ((unsigned)y < (unsigned)Hl) & ((unsigned)x < (unsigned)Wl);
        mt.off_tab[p] = inside ? (uint32_t)(Sl + y * Wl + x) * pixB : 0xffffffffu;
      }
    }
    BW_STAMP(3);                                             // origins, DMA issued, flush table
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's share of the value windows has landed
    BW_STAMP(4);
    lds_barrier();                                           // barrier B ... and everybody else's
    BW_STAMP(5);

    // near = all four corners inside the level's window or outside the image; far = in range and not near
    auto classify = [&](v2f (&xy)[4], uint32_t& inb, uint32_t& nb) __attribute__((always_inline)) {
      inb = 0; nb = 0;
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        bool in;
        xy[l] = coord(l, in);
        const int cx = cvt_i32(floorf(xy[l].x)), cy = cvt_i32(floorf(xy[l].y));
        const int cxm = min(ogx[l] + kWW[l] - 2, lvW[l] - 1) - ogx[l], rym = min(ogy[l] + kWH[l] - 2, lvH[l] - 1) - ogy[l];
        const bool near = in & ((uint32_t)(cx - ogx[l]) <= (uint32_t)cxm) & ((uint32_t)(cy - ogy[l]) <= (uint32_t)rym);
        inb |= in ? (1u << l) : 0u;
        nb |= near ? (1u << l) : 0u;
      }
    };

    // ======== gather phase: grad_attn_weight, grad_sampling_loc; grad_value of the far samples ================================
    for (int rnd = 0; rnd < nrounds; ++rnd) {
      if (rnd > 0) fetch_gather(rnd);
      add_stats();                                           // (read between barriers C and D)
      v2f xy[4];
      uint32_t inb, nb;
      classify(xy, inb, nb);
      float ga[4] = {0.f, 0.f, 0.f, 0.f}, glx[4] = {0.f, 0.f, 0.f, 0.f}, gly[4] = {0.f, 0.f, 0.f, 0.f};

      struct Smp {
        uint32_t aF, aS;          // LDS byte addresses of the first / second pixel of the top row (read order of this quad)
        float u, lh;              // bilinear weight of the second pixel; of the bottom row
      };
      const uint32_t zero_first = smem_base + kZeroOff + 128u * (uint32_t)cls_e;
      auto prepare = [&](auto ltag, float& sgn_a_w, float& a_h) __attribute__((always_inline)) {
        constexpr int LV = decltype(ltag)::value;
        Smp s;
        const v2f fl = {floorf(xy[LV].x), floorf(xy[LV].y)};
        v2f fr = xy[LV] - fl;                                  // (lw, lh)
        fr.x = fmaxf(fr.x, 0.f); fr.y = fmaxf(fr.y, 0.f);      // (NaN of poisoned, dead samples must not reach the weights)
        const int cx = cvt_i32(fl.x) - ogx[LV], ry = cvt_i32(fl.y) - ogy[LV];
        const bool near = ((nb >> LV) & 1u) != 0u;
        const uint32_t sw = (uint32_t)(cx ^ cls_e) & 1u;       // 1: the right-hand pixel has this quad's first parity
        const uint32_t tl = smem_base + (uint32_t)(kBase[LV] * 128) + (uint32_t)(__mul24(ry, kWW[LV]) + cx) * 128u;
        s.aF = near ? tl + (sw << 7) : zero_first;
        s.aS = near ? tl + 128u - (sw << 7) : (zero_first ^ 128u);
        s.u = sw ? 1.f - fr.x : fr.x;                          // weight of the SECOND pixel
        s.lh = fr.y;
        // what turns the quad's reduced d/dx, d/dy sums into this sample's grad_sampling_loc (cuh:157-158: x W, x H)
        sgn_a_w = (sw ? -sa[LV] : sa[LV]) * (float)lvW[LV];
        a_h = sa[LV] * (float)lvH[LV];
        return s;
      };
      // The sample's three sums are linear in its two corner rows: with F / S the first / second pixel of a row in this quad's read
      // order, u the weight of S, A_r = sum_c g_c (F_c + u (S_c - F_c)) and D_r = sum_c g_c (S_c - F_c) per row r,
      //   grad_attn = hh A_top + lh A_bot,   d val / d x = +-(hh D_top + lh D_bot),   d val / d y = A_bot - A_top
      // so a row is consumed on its own (4 packed operations per channel pair) and, as in the forward window kernel, the top row
      // of sample s + 1 is requested before the top row of sample s is consumed, likewise the bottom rows: three rows in flight.
      struct Row { f32x4 Fa, Fb, Sa, Sb; };
      struct Adr { lds4 pF, pF2, pS, pS2; };
      auto fetch_top = [&](auto ltag, auto ptag, const Smp& s, Row& r, Adr& ad) __attribute__((always_inline)) {
        constexpr int PT = decltype(ptag)::value;
        const uint32_t aF = qb<PT>(s.aF) + c0, aS = qb<PT>(s.aS) + c0;
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
      v2f At, Dt, Ab, Db;                                      // the current sample's row sums (per lane: its 8 channels)
      auto consume = [&](auto ptag, const Smp& s, const Row& r, v2f& A, v2f& D) __attribute__((always_inline)) {
        constexpr int PT = decltype(ptag)::value;
        const float u = qbf<PT>(s.u);
        const v2f U = {u, u};
        v2f a = {0.f, 0.f}, dsum = {0.f, 0.f};
        auto chan_pair = [&](v2f F, v2f S, v2f G) __attribute__((always_inline)) {
          const v2f t = S - F;
          const v2f v = __builtin_elementwise_fma(U, t, F);
          a = __builtin_elementwise_fma(G, v, a);
          dsum = __builtin_elementwise_fma(G, t, dsum);
        };
        chan_pair(v2f{r.Fa[0], r.Fa[1]}, v2f{r.Sa[0], r.Sa[1]}, v2f{gA[0], gA[1]});
        chan_pair(v2f{r.Fa[2], r.Fa[3]}, v2f{r.Sa[2], r.Sa[3]}, v2f{gA[2], gA[3]});
        chan_pair(v2f{r.Fb[0], r.Fb[1]}, v2f{r.Sb[0], r.Sb[1]}, v2f{gB[0], gB[1]});
        chan_pair(v2f{r.Fb[2], r.Fb[3]}, v2f{r.Sb[2], r.Sb[3]}, v2f{gB[2], gB[3]});
        A = a; D = dsum;
        asm volatile("" : "+v"(A), "+v"(D));                   // pins the sums here (IR-level sinking ignores sched_barrier)
        __builtin_amdgcn_sched_barrier(0);
      };
      auto finish = [&](auto ltag, auto ptag, const Smp& s, float sgn_a_w, float a_h) __attribute__((always_inline)) {
        constexpr int LV = decltype(ltag)::value, PT = decltype(ptag)::value;
        const float lh = qbf<PT>(s.lh), hh = 1.f - lh;
        const float at = At.x + At.y, ab = Ab.x + Ab.y, dt = Dt.x + Dt.y, db = Db.x + Db.y;
        const float ra = quad_sum(fmaf(lh, ab, hh * at)), rw = quad_sum(fmaf(lh, db, hh * dt)), rh = quad_sum(ab - at);
        const bool near_mine = ((nb >> LV) & 1u) != 0u;
        const bool mine = (k == PT) & near_mine;               // this lane's own sample (far ones: below; dead ones stay 0)
        ga[LV] = mine ? ra : ga[LV];
        glx[LV] = mine ? rw * sgn_a_w : glx[LV];
        gly[LV] = mine ? rh * a_h : gly[LV];
        asm volatile("" : "+v"(ga[LV]), "+v"(glx[LV]), "+v"(gly[LV]));
        __builtin_amdgcn_sched_barrier(0);
      };
      if (__ballot(nb != 0u) != 0ull) {   // (a wave without a near sample: nothing to gather)
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        Row t0, t1, b0, b1;
        Adr ad;
        float sw0, ah0, sw1, ah1;
        // one step: sample (LC, PC) with rows (TC, BC) is consumed while the rows of the next sample (LN, PN) are requested
#define BW_STEP(LC, PC, SC, SWC, AHC, LN, PN, SN, TC, BC, TN, BN)                                        \
        fetch_top(LN{}, PN{}, SN, TN, ad); consume(PC{}, SC, TC, At, Dt);                                 \
        fetch_bot(LN{}, BN, ad);           consume(PC{}, SC, BC, Ab, Db);                                 \
        finish(LC{}, PC{}, SC, SWC, AHC);
#define BW_LEVEL(LC, SC, SWC, AHC, LN, SN, SWN, AHN)                                                     \
        BW_STEP(LC, I0, SC, SWC, AHC, LC, I1, SC, t0, b0, t1, b1)                                        \
        BW_STEP(LC, I1, SC, SWC, AHC, LC, I2, SC, t1, b1, t0, b0)                                        \
        BW_STEP(LC, I2, SC, SWC, AHC, LC, I3, SC, t0, b0, t1, b1)                                        \
        SN = prepare(LN{}, SWN, AHN);                                                                    \
        BW_STEP(LC, I3, SC, SWC, AHC, LN, I0, SN, t1, b1, t0, b0)
        Smp s0 = prepare(I0{}, sw0, ah0), s1 = s0;
        fetch_top(I0{}, I0{}, s0, t0, ad); fetch_bot(I0{}, b0, ad);
        BW_LEVEL(I0, s0, sw0, ah0, I1, s1, sw1, ah1)
        BW_LEVEL(I1, s1, sw1, ah1, I2, s0, sw0, ah0)
        BW_LEVEL(I2, s0, sw0, ah0, I3, s1, sw1, ah1)
        BW_STEP(I3, I0, s1, sw1, ah1, I3, I1, s1, t0, b0, t1, b1)
        BW_STEP(I3, I1, s1, sw1, ah1, I3, I2, s1, t1, b1, t0, b0)
        BW_STEP(I3, I2, s1, sw1, ah1, I3, I3, s1, t0, b0, t1, b1)
        consume(I3{}, s1, t1, At, Dt); consume(I3{}, s1, b1, Ab, Db); finish(I3{}, I3{}, s1, sw1, ah1);
#undef BW_LEVEL
#undef BW_STEP
      }

      if (rnd == 0) BW_STAMP(6);                             // near samples of round 0 gathered
      // ---- far samples: half a wave per sample (lane = channel), two samples per iteration; the corner loads of an iteration
      // are all in flight before anything waits for them (msda_bwd_win.hip) -------------------------------------------------
      {
        const uint32_t farbits = inb & ~nb;
        const int ch = ln & 31;
        auto half_sum = [](float v) __attribute__((always_inline)) {
          v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
          v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
          v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
          v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
          v += __shfl_xor(v, 16, 64);
          return v;
        };
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          uint64_t fm = __ballot(((farbits >> l) & 1u) != 0u);
          const int Hl = lvH[l], Wl = lvW[l], Sl = lvS[l];
          while (fm) {
            const int sA = __builtin_ctzll(fm);
            fm &= fm - 1;
            const bool hasB = fm != 0;
            const int sB = hasB ? __builtin_ctzll(fm) : sA;
            if (hasB) fm &= fm - 1;
            const bool act = (ln < 32) | hasB;
            const int src = (ln < 32 ? sA : sB) << 2;
            const float fx = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(xy[l].x)));
            const float fy = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(xy[l].y)));
            const float fa = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(sa[l])));
            const uint32_t fpair = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)pair);
            const float xf = floorf(fx), yf = floorf(fy);
            const float lw = fx - xf, lh = fy - yf, hw = 1.f - lw, hh = 1.f - lh;
            const int x0 = (int)xf, y0 = (int)yf;              // in range: -1 <= x0 < W, -1 <= y0 < H
            const bool t_ok = act & (y0 >= 0), b_ok = act & (y0 + 1 <= Hl - 1), l_ok = x0 >= 0, r_ok = x0 + 1 <= Wl - 1;
            const uint32_t p00 = (uint32_t)(Sl + y0 * Wl + x0) * pixB + (uint32_t)ch * 4u;   // (garbage where the corner is dead: masked)
            const uint32_t rowG = (uint32_t)Wl * pixB;
            const uint32_t o1 = (t_ok & l_ok) ? p00 : kOobOffset, o2 = (t_ok & r_ok) ? p00 + pixB : kOobOffset;
            const uint32_t o3 = (b_ok & l_ok) ? p00 + rowG : kOobOffset, o4 = (b_ok & r_ok) ? p00 + rowG + pixB : kOobOffset;
            const float v1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vsrc, o1, hoff, 0));
            const float v2 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vsrc, o2, hoff, 0));
            const float v3 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vsrc, o3, hoff, 0));
            const float v4 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vsrc, o4, hoff, 0));
            const float g = act ? grad_out[pair_img * 32 + fpair * 32u + (uint32_t)ch] : 0.f;
            const float tt = v2 - v1, tb_ = v4 - v3;
            const float top = fmaf(lw, tt, v1), bot = fmaf(lw, tb_, v3);
            const float dd = bot - top;
            const float val = fmaf(lh, dd, top), dx = fmaf(lh, tb_, hh * tt);
            const float ra = half_sum(g * val), rw = half_sum(g * dx) * fa * (float)Wl, rh = half_sum(g * dd) * fa * (float)Hl;
            const float raA = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ra), 0)), raB = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ra), 32));
            const float rwA = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rw), 0)), rwB = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rw), 32));
            const float rhA = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rh), 0)), rhB = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rh), 32));
            if (ln == sA) { ga[l] = raA; glx[l] = rwA; gly[l] = rhA; }
            if (hasB && ln == sB) { ga[l] = raB; glx[l] = rwB; gly[l] = rhB; }
            // grad_value: w_corner * a * g_c, one full-line float atomic per live corner
            const float tg = g * fa;
            if (o1 != kOobOffset) atomic_add(reinterpret_cast<float*>(gv_head + o1), hh * hw * tg);
            if (o2 != kOobOffset) atomic_add(reinterpret_cast<float*>(gv_head + o2), hh * lw * tg);
            if (o3 != kOobOffset) atomic_add(reinterpret_cast<float*>(gv_head + o3), lh * hw * tg);
            if (o4 != kOobOffset) atomic_add(reinterpret_cast<float*>(gv_head + o4), lh * lw * tg);
          }
        }
      }

      // ---- this lane's point on the four levels: grad_attn_weight, grad_sampling_loc --------------------------------------
      if (live) {
        float* gap = grad_attn + pair_img * 16 + (pair * 16u + (uint32_t)k);
        v2f* glp = reinterpret_cast<v2f*>(grad_loc + pair_img * 32 + (pair * 32u + 2u * (uint32_t)k));
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          gap[4 * l] = ga[l];
          glp[4 * l] = v2f{glx[l], gly[l]};
        }
      }
    }

    // ======== the windows change hands ============================================================================================
    BW_STAMP(7);                                             // gather phase done (far samples, stores, later rounds)
    fetch_scatter(0);                                        // (travels under the barrier and the zeroing)
    lds_barrier();                                           // barrier C: every wave has read its last value row (and added its stats)
    BW_STAMP(8);
    for (int o = tid * 16; o < kSlots * 128; o += kT * 16) *reinterpret_cast<f32x4*>(smem + o) = f32x4{0.f, 0.f, 0.f, 0.f};
    // fixed-point scale: every slot receives at most (#pairs of the item) x max |grad_out| x max_pair sum |attn|
    float scale = 1.f, inv_scale = 1.f;
    bool use_lds;
    {
      const int npairs = kTH * kTW + nrest;
      const float bound = (float)npairs * __uint_as_float(mt.gmax_bits) * __uint_as_float(mt.amax_bits);
      // false for NaN / Inf, and for items of more than 256 pairs (pyramids with a finer level after the first): every near
      // sample then takes float atomics (thousands of roundings per pixel at a coarse scale add up past 1e-4)
      use_lds = bound < 0x1p120f && npairs <= 256;
      if (use_lds && bound > 0.f) {
        int e;
        (void)frexpf(bound, &e);                             // bound < 2^e
        e = max(-90, min(90, 30 - e));
        scale = ldexpf(1.f, e);
        inv_scale = ldexpf(1.f, -e);
      }
    }
    BW_STAMP(9);
    lds_barrier();                                           // barrier D: the accumulators are zero; everybody has read the scale words
    BW_STAMP(10);
    if (tid == 0) { mt.gmax_bits = 0u; mt.amax_bits = 0u; }  // the next item's (its first add comes behind its barrier A ... of this wave's
                                                             // own stats: in front of it -- see the note at add_stats' call)

    // ======== scatter phase: grad_value of the near samples ========================================================================
    for (int rnd = 0; rnd < nrounds; ++rnd) {
      if (rnd > 0) fetch_scatter(rnd);
      v2f xy[4];
      uint32_t inb, nb;
      classify(xy, inb, nb);
      if (__ballot(nb != 0u) == 0ull) continue;
      struct SmpS {
        uint32_t aL;              // LDS byte address of the LEFT top pixel's accumulator slot
        v2f wT, wB;               // (left, right) corner weights of the top / bottom row x attention weight x scale
      };
      auto prepare_s = [&](auto ltag) __attribute__((always_inline)) {
        constexpr int LV = decltype(ltag)::value;
        SmpS s;
        const v2f fl = {floorf(xy[LV].x), floorf(xy[LV].y)};
        v2f fr = xy[LV] - fl;
        fr.x = fmaxf(fr.x, 0.f); fr.y = fmaxf(fr.y, 0.f);
        const int cx = cvt_i32(fl.x) - ogx[LV], ry = cvt_i32(fl.y) - ogy[LV];
        const bool near = ((nb >> LV) & 1u) != 0u;
        s.aL = smem_base + (uint32_t)(kBase[LV] * 128) + (uint32_t)(__mul24(ry, kWW[LV]) + cx) * 128u;   // (only used for near samples)
        const float an = near ? sa[LV] * scale : 0.f;
        const v2f wrow = v2f{1.f - fr.y, fr.y} * an;           // (top, bottom) x attention weight x scale
        s.wT = v2f{1.f - fr.x, fr.x} * wrow.x;
        s.wB = v2f{1.f - fr.x, fr.x} * wrow.y;
        return s;
      };
      auto scatter = [&](auto ltag, auto ptag, const SmpS& s) __attribute__((always_inline)) {
        constexpr int LV = decltype(ltag)::value, PT = decltype(ptag)::value;
        if (((qb<PT>(nb) >> LV) & 1u) != 0u) {               // this quad's sample near?  (quad-uniform; far and dead samples add nothing)
          constexpr uint32_t kRowB = (uint32_t)kWW[LV] * 128u;
          const uint32_t aL0 = qb<PT>(s.aL);
          const float wTL = qbf<PT>(s.wT.x), wTR = qbf<PT>(s.wT.y), wBL = qbf<PT>(s.wB.x), wBR = qbf<PT>(s.wB.y);
          if (use_lds) {
            const uint32_t aLr = (aL0 + 4u * (uint32_t)k) ^ rot;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              const uint32_t a = aLr ^ (16u * (uint32_t)t);
              const float g = gi[t];
              lds_add(a, cvt_rn_i32(wTL * g));
              lds_add(a + 128u, cvt_rn_i32(wTR * g));
              lds_add(a + kRowB, cvt_rn_i32(wBL * g));
              lds_add(a + kRowB + 128u, cvt_rn_i32(wBR * g));
            }
          } else {
            // float atomics straight to grad_value (scale == 1): the slot's pixel comes from the flush table, ~0 = outside the image
            const uint32_t sl = (aL0 - smem_base) >> 7;
            const uint32_t o00 = mt.off_tab[sl], o01 = mt.off_tab[sl + 1], o10 = mt.off_tab[sl + kWW[LV]], o11 = mt.off_tab[sl + kWW[LV] + 1];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              const uint32_t cb = 4u * (uint32_t)(k + 4 * (t ^ cls8));
              const float g = gi[t];
              if (o00 != 0xffffffffu) atomic_add(reinterpret_cast<float*>(gv_head + (size_t)o00 + cb), wTL * g);
              if (o01 != 0xffffffffu) atomic_add(reinterpret_cast<float*>(gv_head + (size_t)o01 + cb), wTR * g);
              if (o10 != 0xffffffffu) atomic_add(reinterpret_cast<float*>(gv_head + (size_t)o10 + cb), wBL * g);
              if (o11 != 0xffffffffu) atomic_add(reinterpret_cast<float*>(gv_head + (size_t)o11 + cb), wBR * g);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
#define BW_SLEVEL(LC)                                                                                   \
      { const SmpS s = prepare_s(LC{}); scatter(LC{}, I0{}, s); scatter(LC{}, I1{}, s); scatter(LC{}, I2{}, s); scatter(LC{}, I3{}, s); }
      BW_SLEVEL(I0) BW_SLEVEL(I1) BW_SLEVEL(I2) BW_SLEVEL(I3)
#undef BW_SLEVEL
    }

    BW_STAMP(11);                                            // scatter phase done
    // ---- the next item's first queries travel under the flush ------------------------------------------------------------------
    const bool more = item + K < nitems;
    if (more && wv < kTH) {
      // (fetch_gather uses this item's tx / ty / pair_img: the next item's are derived here)
      const int nitem = item + K;
      const int b2 = to_sgpr((int)(((float)nitem + 0.5f) * __builtin_amdgcn_rcpf((float)ntiles)));
      const int64_t pair_img2 = (int64_t)b2 * d.Lq * M + m;
      const int tile2 = nitem - b2 * ntiles;
      const int ty2 = to_sgpr((int)(((float)tile2 + 0.5f) * __builtin_amdgcn_rcpf((float)TX)));
      const int tx2 = tile2 - ty2 * TX;
      const int xs0 = kTW * tx2, ys0 = kTH * ty2;
      live = (pq < min(kTW, lvW[0] - xs0)) && (wv < min(kTH, lvH[0] - ys0));
      const uint32_t qidx = (uint32_t)(lvS[0] + (ys0 + wv) * lvW[0] + xs0 + pq);
      live = live && qidx < (uint32_t)d.Lq;
      pair = mul_u24_s(live ? qidx : 0u, (uint32_t)M);
#pragma unroll
      for (int l = 0; l < 4; ++l) { lc[l] = v2f{0.f, 0.f}; sa[l] = 0.f; }
      gA = f32x4{0.f, 0.f, 0.f, 0.f}; gB = f32x4{0.f, 0.f, 0.f, 0.f};
      if (live) {
        const v2f* lp = reinterpret_cast<const v2f*>(loc + pair_img2 * 32 + (pair * 32u + 2u * (uint32_t)k));
        const float* ap = attn + pair_img2 * 16 + (pair * 16u + (uint32_t)k);
        const float* gp = grad_out + pair_img2 * 32 + pair * 32u;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          lc[l] = lp[4 * l];
          sa[l] = ap[4 * l];
        }
        gA = *reinterpret_cast<const f32x4*>(gp + (c0 >> 2));
        gB = *reinterpret_cast<const f32x4*>(gp + ((c0 ^ 64u) >> 2));
      }
    }

    BW_STAMP(12);
    lds_barrier();                                           // barrier E: every wave's adds are in
    BW_STAMP(13);
    // ---- flush: every touched accumulator pixel inside the image leaves as one full-line float atomic (32 lanes x 4 B) ----
    {
      const int ch = tid & 31;
#pragma unroll 3
      for (int p = tid >> 5; p < kSlots; p += kT / 32) {
        const int raw = *reinterpret_cast<const int*>(smem + p * 128 + ch * 4);
        const uint32_t off = mt.off_tab[p];
        if (off != 0xffffffffu && raw != 0)
          atomic_add(reinterpret_cast<float*>(gv_head + (size_t)off) + ch, (float)raw * inv_scale);
      }
    }
    BW_STAMP(14);
  }
}

#ifdef MSDA_BWIN2_PROF
extern "C" int msda_debug_read_prof_bwin2(void* dst, int nblocks) {
  if (nblocks > kProfBlocks) nblocks = kProfBlocks;
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_bwin2_prof), (size_t)nblocks * kWaves * kProfSlots * 8, 0, hipMemcpyDeviceToHost);
}
#endif

bool win2_backward_ok(const Dims& d) { return win_backward_ok(d); }

int launch_backward_win2(const float* grad_out, const float* value, const int64_t* shapes, const int64_t* lsi,
                         const float* loc, const float* attn, const Dims& d, float* grad_value, float* grad_loc,
                         float* grad_attn, hipStream_t stream) {
  static std::atomic<uint64_t> lds_opted_in{0};
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(msda_bwd_win2), kLdsBytes, lds_opted_in)) return rc;
  // persistent grid: two resident workgroups per CU, spread over the heads; head m = blockIdx.x, so that (by the observed
  // round-robin placement of the linear workgroup id) XCD m % 8 only touches head m's slice of `value` and `grad_value`
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  int K = (2 * cus + d.M - 1) / d.M;
  const int items = d.N * ((d.S + 127) / 128);
  if (K > items) K = items;
  if (K < 1) K = 1;
  if (K > 65535) K = 65535;
  hipLaunchKernelGGL(msda_bwd_win2, dim3((unsigned)d.M, (unsigned)K), dim3(kT), kLdsBytes, stream, grad_out, value, shapes, lsi,
                     loc, attn, d, grad_value, grad_loc, grad_attn);
  return (int)hipGetLastError();
}

}  // namespace msda
