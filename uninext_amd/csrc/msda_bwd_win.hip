// msda_bwd_win -- MSDeformAttn backward for encoder-style calls (Lq == S) with BOTH the sampled values and the value
// gradient in LDS windows.  fp32, D = 32, L = P = 4.  gfx950 only.  Replaces, for these calls, the work of
// ops/src/cuda/ms_deform_im2col_cuda.cuh:301-403 (+ :87-159).
//
// msda_bwd_tiled (round 1 / 2) privatises grad_value in LDS but gathers `value` through the vector L1: 22.7 M 128-byte
// lines per call through a texture path that moves one line per ~2 clocks per CU (>= 84 us, ~150 in practice), at three
// waves per SIMD next to 110 M VALU instructions and LDS atomics with 62 % bank conflicts: 350-370 us per call.  This
// kernel is the forward window kernel (msda_fwd_win2.hip) run backwards:
//
//   work item    (image, head, 8 x 16 tile of level-0 pixels) + the tile's queries of levels 1..3 (the same exact
//                partition), 704-thread workgroups: waves 0..7 = the tile's rows, waves 8..10 = levels 1..3, one pass.
//   LDS          per level a window of `value` pixels (LDS-DMA, as in the forward kernel: 12x20 / 10x14 / 10x12 / 10x10 pixels, 76 KB;
//                round 5 -- rounds 3-4 had 14x22 / 10x14 / 8x10 / 7x8: 273 -> 263 us, see msda_fwd_win.hip) AND a window of the same
//                geometry of 32-bit FIXED-POINT accumulators for grad_value (76 KB; ds_add_f32 costs ~190 clocks per wave
//                instruction on gfx950, ds_add_u32 4.4 -- tools/micro/lds_atomic_bench.cpp; the per-tile power-of-two scale
//                comes from a bound that cannot overflow, msda_bwd_tiled.hip / include/msda_hip.h).  One workgroup per CU.
//   gather       a quad of lanes per (query, head) pair; lane k owns the 16-byte pieces k and k + 4 of a pixel; corner
//                rows are read with bank-conflict-free ds_read_b128 (class rotation over (half, slot parity)).
//   scatter      for the accumulation lane k owns the channels k, k + 4, ..., k + 28 and walks them in an order rotated by
//                the quad's class (pq & 7): the eight quads of a 32-lane service group hit eight different 4-bank groups
//                whatever the parities of their slots -- ds_add_u32 without bank conflicts (62 % of the LDS cycles in
//                msda_bwd_tiled).  The four corners of a sample are immediate offsets from its left-top slot.
//   far          an in-range sample with a corner outside its window is processed by the whole wave (a lane per channel):
//                four coalesced corner loads, the sample's three gradients by a wave reduction, one full-line float atomic per
//                corner.  A tile with non-finite inputs takes this path for every sample (NaN / Inf propagate as with float
//                atomics).
//   flush        when the tile is done every touched accumulator pixel inside the image leaves the CU as ONE full-line
//                float atomic.
//
// Per-sample arithmetic (cuh:113-158, refactored as in msda_bwd_tiled): with F / S the first / second pixel of a corner
// row in this quad's read order and u the bilinear weight of S:
//   top = F_top + u (S_top - F_top), bot = F_bot + u (S_bot - F_bot), val = top + lh (bot - top)
//   grad_attn = sum_c g_c val_c;  d val / d y = bot - top;  d val / d x = +-[hh (S_top - F_top) + lh (S_bot - F_bot)]
// Round 5: u, lh do not depend on the channel, so a sample needs only the four sums  sum_c g_c F_c  and  sum_c g_c S_c  of its two
// corner rows (4 packed FMAs per channel pair; everything else on the reduced sums), and the scatter multiplies two channels per
// v_pk_mul_f32: 3965 -> 3389 vector instructions in the kernel, 264.5 -> 260.7 us (profiles/r05_backward_packed.txt).
#include <cstdlib>
#include <type_traits>

#include "msda_common.hpp"

namespace msda {
namespace {

constexpr int kL0Waves = 8, kRestWaves = 3, kWaves = kL0Waves + kRestWaves, kT = kWaves * 64;
constexpr int kRestQuads = kRestWaves * 16;
constexpr int kTH = 8, kTW = 16;
constexpr int kWH[4] = {12, 10, 10, 10};
constexpr int kWW[4] = {20, 14, 12, 10};                        // even: slot parity == column parity in every row
constexpr int kBase[5] = {0, 240, 384, 504, 608};               // first slot of each window (multiples of 8: DMA chunks)
constexpr int kSlots = kBase[4];
constexpr int kZeroOff = kSlots * 128;                          // all-zero region: read target of dead / far samples
constexpr int kZeroBytes = kWW[0] * 128 + 256;
constexpr int kAccOff = kZeroOff + kZeroBytes;                  // accumulator windows: same slot numbering as the value windows
static_assert(kAccOff % 256 == 0, "slot parity by address bit 7 in both regions");
struct Meta {
  int sum[4][4];                                                // per level: sum x0, sum y0, count, - (placement)
  int lvl[4][4];                                                // per level: H, W, first pixel, -
  int org[4][4];                                                // per level: window origin x, y (flush)
  unsigned gmax_bits, amax_bits, pad0, pad1;                    // per tile: max |grad_out|, max_pair sum |attn| (float bits)
  unsigned slot_tab[kSlots];                                    // (level << 28) | (row << 14) | column of a window slot
  unsigned off_tab[kSlots];                                     // per item: byte offset of the slot's pixel in grad_value (head 0), ~0: outside
};
constexpr int kMetaOff = kAccOff + kSlots * 128;
constexpr int kLdsBytes = kMetaOff + ((sizeof(Meta) + 15) / 16) * 16;
static_assert(kLdsBytes <= 160 * 1024, "one workgroup per CU");

// Phase timestamps (profiling builds only: -DMSDA_BWIN_PROF; tools/bwin_prof.py): lane 0 of every wave, first item only.
#ifdef MSDA_BWIN_PROF
constexpr int kProfBlocks = 512, kProfSlots = 16;
__device__ unsigned long long g_bwin_prof[kProfBlocks * kWaves * kProfSlots];
#define BW_STAMP(i)                                                                                          \
  do {                                                                                                       \
    const unsigned blk_ = blockIdx.y * gridDim.x + blockIdx.x;                                               \
    if ((threadIdx.x & 63) == 0 && item == kk + K && blk_ < (unsigned)kProfBlocks)                           \
      g_bwin_prof[(blk_ * kWaves + (threadIdx.x >> 6)) * kProfSlots + (i)] = __builtin_amdgcn_s_memrealtime(); \
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
// a wave-uniform value computed on the vector ALU into a SCALAR register (see msda_fwd_win2.hip: hazards, folding)
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
__device__ __forceinline__ void lds_add(uint32_t lds_byte_addr, int v) {   // ds_add_u32, no return value
  __hip_atomic_fetch_add(reinterpret_cast<lds_int_ptr>((uintptr_t)lds_byte_addr), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ float abs_or_inf(float x) {  // |x|, +inf for NaN / Inf (so that a max() sees it)
  const float a = fabsf(x);
  return (a <= 3.402823466e+38f) ? a : __builtin_inff();
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace

__global__ void __launch_bounds__(kT, 3)
msda_bwd_win(const float* __restrict__ grad_out, const float* __restrict__ value, const int64_t* __restrict__ shapes,
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

  // ---- once per workgroup: zero region, level table, slot table ---------------------------------------------------------
  for (int o = tid * 16; o < kZeroBytes; o += kT * 16) *reinterpret_cast<f32x4*>(smem + kZeroOff + o) = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tid < 4) {
    const bool t0 = (tid & 1) != 0, t1 = (tid & 2) != 0;
    *reinterpret_cast<int4*>(&mt.lvl[tid][0]) = make_int4(sel4(t0, t1, lvH[0], lvH[1], lvH[2], lvH[3]), sel4(t0, t1, lvW[0], lvW[1], lvW[2], lvW[3]),
                                                          sel4(t0, t1, lvS[0], lvS[1], lvS[2], lvS[3]), 0);
  }
  if (tid < kSlots) {
    const int l = (tid >= kBase[1] ? 1 : 0) + (tid >= kBase[2] ? 1 : 0) + (tid >= kBase[3] ? 1 : 0);
    const int rel = tid - (l == 0 ? kBase[0] : l == 1 ? kBase[1] : l == 2 ? kBase[2] : kBase[3]);
    const int ww = l == 0 ? kWW[0] : l == 1 ? kWW[1] : l == 2 ? kWW[2] : kWW[3];
    const int r = rel / ww, c = rel - r * ww;
    mt.slot_tab[tid] = ((unsigned)l << 28) | ((unsigned)r << 14) | (unsigned)c;
  }
  static_assert(kSlots <= kT, "one slot-table entry per thread");
  for (int o = tid * 16; o < kSlots * 128; o += kT * 16) *reinterpret_cast<f32x4*>(smem + kAccOff + o) = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tid >= 640 && tid < 656) (&mt.sum[0][0])[tid - 640] = 0;
  if (tid == 656) { mt.gmax_bits = 0u; mt.amax_bits = 0u; }   // (pad0: written by the fetch of every item's first pass before it is read)

  const uint32_t pixB = (uint32_t)M * 128u;                // bytes from a pixel of head m to the next one
  const uint32_t hoff = (uint32_t)m * 128u;

  // ---- steps: one (item, pass) each.  The loads of a step are issued at the end of the step before it -- for the first
  // pass of an item that is between barrier #4 and the flush of the item before, so that they travel under the flush -- and
  // the step in front of the first item only fetches.
  int item = kk - K, pass = 0, npass = 1;
  bool body = false;
  int ogx[4] = {0, 0, 0, 0}, ogy[4] = {0, 0, 0, 0};         // window origins of the item
  float scale = 1.f, inv_scale = 1.f;                      // fixed-point scale of the item's accumulators
  bool use_lds = false;
  bool live = false;                                       // the fetched step: this quad's (query, head) pair ...
  uint32_t pair = 0;
  v2f lc[4];                                               // ... locations and weights of point k on the four levels
  float sa[4];
  f32x4 gA = {0.f, 0.f, 0.f, 0.f}, gB = {0.f, 0.f, 0.f, 0.f};   // upstream gradient: channels of the pieces at c0 / c0 ^ 64 (gather order)
  float gi[8];                                                  // channels k + 4 (t ^ cls8) (accumulation order)
#pragma unroll
  for (int l = 0; l < 4; ++l) { lc[l] = v2f{0.f, 0.f}; sa[l] = 0.f; }
#pragma unroll
  for (int t = 0; t < 8; ++t) gi[t] = 0.f;

  for (;;) {
    // per-lane constants are re-derived per item and the level constants pass through an empty asm (in place): whatever
    // the optimiser can prove invariant in this loop it hoists in front of it and spills (msda_fwd_win2.hip)
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
    // this lane's eight accumulation channels are k + 4 (t ^ cls8), t = 0..7: byte offset 4 k + 16 (t ^ cls8) inside a slot, i.e.
    // (slot address + 4 k) ^ rot ^ 16 t with rot = 16 cls8 (the slot address is 128-byte aligned, no carries)
    const uint32_t rot = 16u * (uint32_t)cls8;
    const int itemc = max(item, 0);
    const int b = to_sgpr((int)(((float)itemc + 0.5f) * __builtin_amdgcn_rcpf((float)ntiles)));
    const int64_t pair_img = (int64_t)b * d.Lq * M + m;     // pair (query 0, head m) of this item's image
    const int64_t img_val = (int64_t)b * d.S * M * 32;
    const __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(value) + img_val, 0, (int)((uint32_t)d.S * pixB), 0x00020000);
    char* const gv_head = reinterpret_cast<char*>(grad_value + img_val) + hoff;   // + pixel byte offset + channel * 4
    const bool l0 = wv < kL0Waves;


    if (body) {
      if (pass == 0) {
        // ---- barrier #1: everybody has left the previous item (its flush left the accumulator windows all zero) ----------
        BW_STAMP(0);
        lds_barrier();
        BW_STAMP(1);
      }

      auto coord = [&](int l, bool& in) __attribute__((always_inline)) {
        const v2f fWH = {(float)lvW[l], (float)lvH[l]};
        const v2f p = __builtin_elementwise_fma(lc[l], fWH, v2f{-0.5f, -0.5f});
        in = live & (p.y > -1.f) & (p.x > -1.f) & (p.y < fWH.y) & (p.x < fWH.x);
        return p;
      };

      BW_STAMP(3);                                           // query decoded, loads issued
      if (pass == 0) {
        // ---- fixed-point scale: every slot receives at most (#pairs of the tile) x max |grad_out| x max_pair sum |attn| ----
        {
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
        }
        if (l0) {
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
            atomicAdd(&mt.sum[k][0], ax);
            atomicAdd(&mt.sum[k][1], ay);
            atomicAdd(&mt.sum[k][2], an);
          }
        }
        BW_STAMP(4);                                         // loads arrived; maxima and placement sums added
        lds_barrier();                                       // #2: sums and scale words complete
        BW_STAMP(5);
        {
          const float bound = (float)(kL0Waves * 16 + (int)mt.pad0) * __uint_as_float(mt.gmax_bits) * __uint_as_float(mt.amax_bits);
          // false for NaN / Inf, and for tiles of more than 256 pairs (pyramids with a finer level after the first): every
          // sample then takes the float-atomic path (thousands of roundings per pixel at a coarse scale add up past 1e-4)
          use_lds = bound < 0x1p120f && kL0Waves * 16 + (int)mt.pad0 <= 256;   // (>= 2^120: the clamped exponent below could not scale it into int32)
          if (use_lds && bound > 0.f) {
            int e;
            (void)frexpf(bound, &e);                         // bound < 2^e
            e = max(-90, min(90, 30 - e));
            scale = ldexpf(1.f, e);
            inv_scale = ldexpf(1.f, -e);
          }
        }
        int myOx, myOy;
        {
          const int4 sm = *reinterpret_cast<const int4*>(&mt.sum[k][0]);
          const int myWW = sel4(k0, k1, kWW[0], kWW[1], kWW[2], kWW[3]), myWH = sel4(k0, k1, kWH[0], kWH[1], kWH[2], kWH[3]);
          const int myW = sel4(k0, k1, lvW[0], lvW[1], lvW[2], lvW[3]), myH = sel4(k0, k1, lvH[0], lvH[1], lvH[2], lvH[3]);
          const float inv = __builtin_amdgcn_rcpf((float)max(sm.z, 1));
          myOx = (int)floorf((float)sm.x * inv + 0.5f) - (myWW - 2) / 2;
          myOy = (int)floorf((float)sm.y * inv + 0.5f) - (myWH - 2) / 2;
          myOx = max(-1, min(myOx, myW + 1 - myWW));
          myOy = max(-1, min(myOy, myH + 1 - myWH));
          if (tid < 4) *reinterpret_cast<int2*>(&mt.org[k][0]) = make_int2(myOx, myOy);   // for the flush
        }
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          ogx[l] = __builtin_amdgcn_readlane(myOx, l);
          ogy[l] = __builtin_amdgcn_readlane(myOy, l);
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
      }

      // where the flush will send each accumulator slot: thread p computes slot p (the windows travel meanwhile)
      if (tid < kSlots) {
        const unsigned e = mt.slot_tab[tid];
        const int l = (int)(e >> 28), r = (int)((e >> 14) & 0x3fffu), c = (int)(e & 0x3fffu);
        const bool e0 = (l & 1) != 0, e1 = (l & 2) != 0;
        const int y = sel4(e0, e1, ogy[0], ogy[1], ogy[2], ogy[3]) + r, x = sel4(e0, e1, ogx[0], ogx[1], ogx[2], ogx[3]) + c;
        const int Hl = sel4(e0, e1, lvH[0], lvH[1], lvH[2], lvH[3]), Wl = sel4(e0, e1, lvW[0], lvW[1], lvW[2], lvW[3]);
        const int Sl = sel4(e0, e1, lvS[0], lvS[1], lvS[2], lvS[3]);
        const bool inside = ((unsigned)y < (unsigned)Hl) & ((unsigned)x < (unsigned)Wl);
        mt.off_tab[tid] = inside ? (uint32_t)(Sl + y * Wl + x) * pixB : 0xffffffffu;
      }
      BW_STAMP(6);                                           // origins, window DMA issued
      // ---- sample coordinates; near (all four corners inside the level's window or outside the image) or far? -----------
      v2f xy[4];
      uint32_t inb = 0, nb = 0;
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        bool in;
        xy[l] = coord(l, in);
        const int cx = cvt_i32(floorf(xy[l].x)), cy = cvt_i32(floorf(xy[l].y));
        const int cxm = min(ogx[l] + kWW[l] - 2, lvW[l] - 1) - ogx[l], rym = min(ogy[l] + kWH[l] - 2, lvH[l] - 1) - ogy[l];
        const bool near = use_lds & in & ((uint32_t)(cx - ogx[l]) <= (uint32_t)cxm) & ((uint32_t)(cy - ogy[l]) <= (uint32_t)rym);
        inb |= in ? (1u << l) : 0u;
        nb |= near ? (1u << l) : 0u;
      }
      float ga[4] = {0.f, 0.f, 0.f, 0.f}, glx[4] = {0.f, 0.f, 0.f, 0.f}, gly[4] = {0.f, 0.f, 0.f, 0.f};

      BW_STAMP(7);                                           // classified, far samples done
      if (pass == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of the value windows has landed
        BW_STAMP(8);
        lds_barrier();                                         // #3 ... and everybody else's
        BW_STAMP(9);
        // the next item's placement sums and scale words (everybody has read this item's; the next adds come after barrier #1)
        if (tid < 16) (&mt.sum[0][0])[tid] = 0;
        if (tid == 16) { mt.gmax_bits = 0u; mt.amax_bits = 0u; }   // (pad0 is rewritten for every item, before barrier #4 of the one before)
      }

      if (!l0) __builtin_amdgcn_s_setprio(2);                  // the three youngest waves of the workgroup would finish the pass last
      // ---- near samples: gather from the value windows, three gradients per sample, scatter into the accumulators ------
      struct Smp {
        uint32_t aF, aS;          // LDS byte addresses of the first / second pixel of the top row (read order of this quad)
        uint32_t aL;              // LDS byte address of the LEFT top pixel's accumulator slot
        float u, lh;              // bilinear weight of the second pixel; of the bottom row
        v2f wT, wB;               // (left, right) corner weights of the top / bottom row x attention weight x scale
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
        s.aL = tl + (uint32_t)kAccOff;                         // (only used for near samples)
        s.u = sw ? 1.f - fr.x : fr.x;                          // weight of the SECOND pixel
        s.lh = fr.y;
        const float an = near ? sa[LV] * scale : 0.f;
        const v2f wrow = v2f{1.f - fr.y, fr.y} * an;           // (top, bottom) x attention weight x scale
        s.wT = v2f{1.f - fr.x, fr.x} * wrow.x;
        s.wB = v2f{1.f - fr.x, fr.x} * wrow.y;
        // what turns the quad's reduced d/dx, d/dy sums into this sample's grad_sampling_loc (cuh:157-158: x W, x H)
        sgn_a_w = (sw ? -sa[LV] : sa[LV]) * (float)lvW[LV];
        a_h = sa[LV] * (float)lvH[LV];
        return s;
      };
      struct Row { f32x4 Fa, Fb, Sa, Sb; };
      auto fetch_rows = [&](auto ltag, auto ptag, const Smp& s, Row& top, Row& bot) __attribute__((always_inline)) {
        constexpr int LV = decltype(ltag)::value, PT = decltype(ptag)::value;
        constexpr int kRow = kWW[LV] * 8;                      // one window row, in 16-byte units
        const uint32_t aF = qb<PT>(s.aF) + c0, aS = qb<PT>(s.aS) + c0;
        lds4 pF = reinterpret_cast<lds4>((uintptr_t)aF), pF2 = reinterpret_cast<lds4>((uintptr_t)(aF ^ 64u));
        lds4 pS = reinterpret_cast<lds4>((uintptr_t)aS), pS2 = reinterpret_cast<lds4>((uintptr_t)(aS ^ 64u));
        top.Fa = pF[0]; top.Fb = pF2[0]; top.Sa = pS[0]; top.Sb = pS2[0];
        bot.Fa = pF[kRow]; bot.Fb = pF2[kRow]; bot.Sa = pS[kRow]; bot.Sb = pS2[kRow];
        __builtin_amdgcn_sched_barrier(0);
      };
      // one sample = the three gradients from the rows in registers, then -- with the NEXT sample's rows requested, its reads
      // travel under the 32 atomics -- the scatter
      auto grads = [&](auto ltag, auto ptag, const Smp& s, const Row& top, const Row& bot, float sgn_a_w, float a_h)
                       __attribute__((always_inline)) {
        constexpr int LV = decltype(ltag)::value, PT = decltype(ptag)::value;
        const float u = qbf<PT>(s.u), lh = qbf<PT>(s.lh);
        const float hh = 1.f - lh;
        // the three sums are linear in the two corner rows (round 4): with A_r = sum_c g_c (F_c + u (S_c - F_c)) and
        // D_r = sum_c g_c (S_c - F_c) per row r:  grad_attn = hh A_top + lh A_bot,  d val / d x = hh D_top + lh D_bot,
        // d val / d y = A_bot - A_top.  Round 5: u does not depend on the channel either, so a row needs only sum_c g_c F_c and
        // sum_c g_c S_c -- 4 packed FMAs per channel pair (was 8 + two subtractions that compiled to four v_sub_f32), then
        // D_r = sum gS - sum gF and A_r = sum gF + u D_r on the reduced sums
        v2f SFt = {0.f, 0.f}, SSt = {0.f, 0.f}, SFb = {0.f, 0.f}, SSb = {0.f, 0.f};
        auto chan_pair = [&](v2f Ft, v2f St, v2f Fb, v2f Sb, v2f G) __attribute__((always_inline)) {
          SFt = __builtin_elementwise_fma(G, Ft, SFt);
          SSt = __builtin_elementwise_fma(G, St, SSt);
          SFb = __builtin_elementwise_fma(G, Fb, SFb);
          SSb = __builtin_elementwise_fma(G, Sb, SSb);
        };
        chan_pair(v2f{top.Fa[0], top.Fa[1]}, v2f{top.Sa[0], top.Sa[1]}, v2f{bot.Fa[0], bot.Fa[1]}, v2f{bot.Sa[0], bot.Sa[1]}, v2f{gA[0], gA[1]});
        chan_pair(v2f{top.Fa[2], top.Fa[3]}, v2f{top.Sa[2], top.Sa[3]}, v2f{bot.Fa[2], bot.Fa[3]}, v2f{bot.Sa[2], bot.Sa[3]}, v2f{gA[2], gA[3]});
        chan_pair(v2f{top.Fb[0], top.Fb[1]}, v2f{top.Sb[0], top.Sb[1]}, v2f{bot.Fb[0], bot.Fb[1]}, v2f{bot.Sb[0], bot.Sb[1]}, v2f{gB[0], gB[1]});
        chan_pair(v2f{top.Fb[2], top.Fb[3]}, v2f{top.Sb[2], top.Sb[3]}, v2f{bot.Fb[2], bot.Fb[3]}, v2f{bot.Sb[2], bot.Sb[3]}, v2f{gB[2], gB[3]});
        const float sft = SFt.x + SFt.y, sst = SSt.x + SSt.y, sfb = SFb.x + SFb.y, ssb = SSb.x + SSb.y;
        const float dt = sst - sft, db = ssb - sfb;
        const float at = fmaf(u, dt, sft), ab = fmaf(u, db, sfb);
        const float ra = quad_sum(fmaf(lh, ab, hh * at)), rw = quad_sum(fmaf(lh, db, hh * dt)), rh = quad_sum(ab - at);
        const bool near_mine = ((nb >> LV) & 1u) != 0u;
        const bool mine = (k == PT) & near_mine;               // this lane's own sample (far ones were done above, dead ones stay 0)
        ga[LV] = mine ? ra : ga[LV];
        glx[LV] = mine ? rw * sgn_a_w : glx[LV];
        gly[LV] = mine ? rh * a_h : gly[LV];
        asm volatile("" : "+v"(ga[LV]), "+v"(glx[LV]), "+v"(gly[LV]));   // settled HERE (the selects are otherwise sunk to the end of
        __builtin_amdgcn_sched_barrier(0);                                // the pass and 48 reduced sums stay alive)
      };
      auto scatter = [&](auto ltag, auto ptag, const Smp& s) __attribute__((always_inline)) {
        constexpr int LV = decltype(ltag)::value, PT = decltype(ptag)::value;
        // this quad's sample near?  (quad-uniform; far and dead samples add nothing)
        if (((qb<PT>(nb) >> LV) & 1u) != 0u) {
          constexpr uint32_t kRowB = (uint32_t)kWW[LV] * 128u;
          const uint32_t aLr = (qb<PT>(s.aL) + 4u * (uint32_t)k) ^ rot;
          const float wTL = qbf<PT>(s.wT.x), wTR = qbf<PT>(s.wT.y), wBL = qbf<PT>(s.wB.x), wBR = qbf<PT>(s.wB.y);
          const v2f WTL = {wTL, wTL}, WTR = {wTR, wTR}, WBL = {wBL, wBL}, WBR = {wBR, wBR};
#pragma unroll
          for (int t = 0; t < 8; t += 2) {                     // two channels per v_pk_mul_f32
            const uint32_t a = aLr ^ (16u * (uint32_t)t), b = aLr ^ (16u * (uint32_t)(t + 1));
            const v2f g = {gi[t], gi[t + 1]};
            const v2f pTL = WTL * g, pTR = WTR * g, pBL = WBL * g, pBR = WBR * g;
            lds_add(a, cvt_rn_i32(pTL.x));
            lds_add(a + 128u, cvt_rn_i32(pTR.x));
            lds_add(a + kRowB, cvt_rn_i32(pBL.x));
            lds_add(a + kRowB + 128u, cvt_rn_i32(pBR.x));
            lds_add(b, cvt_rn_i32(pTL.y));
            lds_add(b + 128u, cvt_rn_i32(pTR.y));
            lds_add(b + kRowB, cvt_rn_i32(pBL.y));
            lds_add(b + kRowB + 128u, cvt_rn_i32(pBR.y));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      {
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        Row rt, rb;
        float sw0, ah0, sw1, ah1;
#define BW_LEVEL(LC, SC, SWC, AHC, LN, SN, SWN, AHN, HAVE_NEXT)                                         \
        grads(LC{}, I0{}, SC, rt, rb, SWC, AHC); fetch_rows(LC{}, I1{}, SC, rt, rb); scatter(LC{}, I0{}, SC);   \
        grads(LC{}, I1{}, SC, rt, rb, SWC, AHC); fetch_rows(LC{}, I2{}, SC, rt, rb); scatter(LC{}, I1{}, SC);   \
        grads(LC{}, I2{}, SC, rt, rb, SWC, AHC); fetch_rows(LC{}, I3{}, SC, rt, rb); scatter(LC{}, I2{}, SC);   \
        grads(LC{}, I3{}, SC, rt, rb, SWC, AHC);                                                          \
        if (HAVE_NEXT) { SN = prepare(LN{}, SWN, AHN); fetch_rows(LN{}, I0{}, SN, rt, rb); }              \
        scatter(LC{}, I3{}, SC);
        Smp s0 = prepare(I0{}, sw0, ah0), s1 = s0;
        fetch_rows(I0{}, I0{}, s0, rt, rb);
        BW_LEVEL(I0, s0, sw0, ah0, I1, s1, sw1, ah1, true)
        BW_LEVEL(I1, s1, sw1, ah1, I2, s0, sw0, ah0, true)
        BW_LEVEL(I2, s0, sw0, ah0, I3, s1, sw1, ah1, true)
        BW_LEVEL(I3, s1, sw1, ah1, I3, s1, sw1, ah1, false)
#undef BW_LEVEL
      }

      __builtin_amdgcn_s_setprio(0);
      BW_STAMP(10);                                          // pass done
      // ---- far samples, BEHIND the pass: they need no window, and a wave that is through its pass early does them while the others
      // are still in theirs (in front of barrier #3 their imbalance was waited for by everybody: 281 -> ... us) -- a HALF of the wave
      // per sample (lane = channel), two samples per iteration; the corner loads of an
      // iteration are all in flight before anything waits for them ------------------------------------------------------------
      {
        const uint32_t farbits = inb & ~nb;
        const int ch = ln & 31;
        // over the 32 lanes of a half; every lane gets the total: four DPP butterfly steps inside a row of 16 lanes and ONE
        // cross-row exchange (as a shuffle loop it was a chain of five dependent ds_bpermute round trips per value).
        // (Tried and dropped: software-pipelining this loop -- the next iteration's loads issued before this one's are
        // waited for -- changed nothing, 342.6 vs 342 us and 1145 vs 1139 us on the wide flavour: the loop is bound by its
        // four full-line atomics per sample, not by the round trips of its loads.)
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
            const float tt = v2 - v1, tb = v4 - v3;
            const float top = fmaf(lw, tt, v1), bot = fmaf(lw, tb, v3);
            const float dd = bot - top;
            const float val = fmaf(lh, dd, top), dx = fmaf(lh, tb, hh * tt);
            const float ra = half_sum(g * val), rw = half_sum(g * dx) * fa * (float)Wl, rh = half_sum(g * dd) * fa * (float)Hl;
            // the owners keep their sample's three gradients (every lane of a half holds that half's totals)
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

    const bool last = !body || pass + 1 >= npass;
    BW_STAMP(11);
    // ---- the next step: its query, and its loads issued ---------------------------------------------------------------------
    const int nitem = last ? item + K : item, np = last ? 0 : pass + 1;
    const bool more = nitem < nitems;
    // the three youngest waves derive their queries in float + five ds_bpermute and are the last to have their loads out:
    // they go ahead of the other waves' flush (312.8 -> 308.6 us, A/B on one box)
    if (!l0) __builtin_amdgcn_s_setprio(2);
    if (more) {
      const int b2 = to_sgpr((int)(((float)nitem + 0.5f) * __builtin_amdgcn_rcpf((float)ntiles)));
      const int64_t pair_img2 = (int64_t)b2 * d.Lq * M + m;
      const int tile2 = nitem - b2 * ntiles;
      const int ty2 = to_sgpr((int)(((float)tile2 + 0.5f) * __builtin_amdgcn_rcpf((float)TX)));
      const int tx2 = tile2 - ty2 * TX;
        // ---- this quad's query --------------------------------------------------------------------------------------
        uint32_t qidx;
        if (l0) {
          const int xs0 = kTW * tx2, ys0 = kTH * ty2;
          live = (pq < min(kTW, lvW[0] - xs0)) && (wv < min(kTH, lvH[0] - ys0));
          qidx = (uint32_t)(lvS[0] + (ys0 + wv) * lvW[0] + xs0 + pq);
        } else {
          const int gW = sel4(k0, k1, lvW[0], lvW[1], lvW[2], lvW[3]), gH = sel4(k0, k1, lvH[0], lvH[1], lvH[2], lvH[3]);
          const float fxs = (float)(kTW * gW) * __builtin_amdgcn_rcpf((float)lvW[0]), fys = (float)(kTH * gH) * __builtin_amdgcn_rcpf((float)lvH[0]);
          const int gxs = min(max((int)ceilf((float)tx2 * fxs - 0.5f), 0), gW);
          const int xe = tx2 == TX - 1 ? gW : min(max((int)ceilf((float)(tx2 + 1) * fxs - 0.5f), gxs), gW);
          const int gys = min(max((int)ceilf((float)ty2 * fys - 0.5f), 0), gH);
          const int ye = ty2 == TY - 1 ? gH : min(max((int)ceilf((float)(ty2 + 1) * fys - 0.5f), gys), gH);
          const int gnx = xe - gxs, cnt = gnx * (ye - gys);
          const int e1 = __builtin_amdgcn_readlane(cnt, 1), e2 = e1 + __builtin_amdgcn_readlane(cnt, 2);
          const int nrest = e2 + __builtin_amdgcn_readlane(cnt, 3);
          npass = max(1, (nrest + kRestQuads - 1) / kRestQuads);
          if (np == 0 && tid == kL0Waves * 64) mt.pad0 = (unsigned)nrest;   // for the overflow bound of the accumulators
          const int ri = np * kRestQuads + (wv - kL0Waves) * 16 + pq;
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
        live = live && qidx < (uint32_t)d.Lq;
        pair = mul_u24_s(live ? qidx : 0u, (uint32_t)M);   // (query, head 0) pair within the image

        // ---- locations and weights of point k on the four levels; the upstream gradient in the two channel orders ------
        gA = f32x4{0.f, 0.f, 0.f, 0.f}; gB = f32x4{0.f, 0.f, 0.f, 0.f};
  #pragma unroll
        for (int l = 0; l < 4; ++l) { lc[l] = v2f{0.f, 0.f}; sa[l] = 0.f; }
  #pragma unroll
        for (int t = 0; t < 8; ++t) gi[t] = 0.f;
        if (live) {
          const v2f* lp = reinterpret_cast<const v2f*>(loc + pair_img2 * 32 + (pair * 32u + 2u * (uint32_t)k));
          const float* ap = attn + pair_img2 * 16 + (pair * 16u + (uint32_t)k);
          const float* gp = grad_out + pair_img2 * 32 + pair * 32u;
  #pragma unroll
          for (int l = 0; l < 4; ++l) {
            lc[l] = __builtin_nontemporal_load(lp + 4 * l);
            sa[l] = __builtin_nontemporal_load(ap + 4 * l);
          }
          gA = *reinterpret_cast<const f32x4*>(gp + (c0 >> 2));
          gB = *reinterpret_cast<const f32x4*>(gp + ((c0 ^ 64u) >> 2));
  #pragma unroll
          for (int t = 0; t < 8; ++t) gi[t] = gp[k + 4 * (t ^ cls8)];
        }

    }
    __builtin_amdgcn_s_setprio(0);
    BW_STAMP(14);
    if (body && last) {
      // #4: every wave's atomics are in.  The next item's queries and loads above do not depend on it: the waves of levels 1..3,
      // which finish the pass first, derive theirs while the level-0 waves are still in the pass instead of behind the barrier.
      lds_barrier();
      BW_STAMP(12);
      // ---- flush: every touched accumulator pixel inside the image leaves as one full-line float atomic (32 lanes x 4 B) ----
      {
        const int ch = tid & 31;
  #pragma unroll 3
        for (int p = tid >> 5; p < kSlots; p += kT / 32) {
          // read and clear in one LDS operation: the next item finds the windows zeroed
          const int raw = __hip_atomic_exchange(reinterpret_cast<lds_int_ptr>((uintptr_t)(smem_base + kAccOff + p * 128 + ch * 4)), 0,
                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          const uint32_t off = mt.off_tab[p];
          if (off != 0xffffffffu && raw != 0)
            atomic_add(reinterpret_cast<float*>(gv_head + (size_t)off) + ch, (float)raw * inv_scale);
        }
      }
      BW_STAMP(13);
    }
    if (!more) break;
    item = nitem; pass = np; body = true;
  }
}

#ifdef MSDA_BWIN_PROF
extern "C" int msda_debug_read_prof_bwin(void* dst, int nblocks) {
  if (nblocks > kProfBlocks) nblocks = kProfBlocks;
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_bwin_prof), (size_t)nblocks * kWaves * kProfSlots * 8, 0, hipMemcpyDeviceToHost);
}
#endif

bool win_backward_ok(const Dims& d) {
  return d.D == 32 && d.P == 4 && d.L == 4 && d.Lq == d.S && d.S >= 1024 && d.M <= 65535 &&
         (int64_t)d.S * d.M * 128 < (int64_t)kOobOffset && d.N <= 65535 &&
         (int64_t)d.N * ((d.S + 127) / 128) < ((int64_t)1 << 22);
}

int launch_backward_win(const float* grad_out, const float* value, const int64_t* shapes, const int64_t* lsi,
                        const float* loc, const float* attn, const Dims& d, float* grad_value, float* grad_loc,
                        float* grad_attn, hipStream_t stream) {
  static std::atomic<uint64_t> lds_opted_in{0};
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(msda_bwd_win), kLdsBytes, lds_opted_in)) return rc;
  // persistent grid: one resident workgroup per CU (155 KB of LDS), spread over the heads; head m = blockIdx.x, so that (by
  // the observed round-robin placement of the linear workgroup id) XCD m % 8 only touches head m's slice of `value`
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  int K = (cus + d.M - 1) / d.M;
  const int items = d.N * ((d.S + 127) / 128);
  if (K > items) K = items;
  if (K < 1) K = 1;
  if (K > 65535) K = 65535;
  hipLaunchKernelGGL(msda_bwd_win, dim3((unsigned)d.M, (unsigned)K), dim3(kT), kLdsBytes, stream, grad_out, value, shapes, lsi,
                     loc, attn, d, grad_value, grad_loc, grad_attn);
  return (int)hipGetLastError();
}

}  // namespace msda
