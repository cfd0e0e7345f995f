// msda_fwd_win4 -- MSDeformAttn forward for encoder-style calls (Lq == S), fourth generation of the LDS-window kernel.
// fp32, D = 32, L = P = 4.  gfx950 only.  Replaces, for these calls, the work of ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299.
//
// What the first three generations are bound by (profiles/r03_forward_window_analysis.txt, section 6): not a pipe, not a wait
// -- the NUMBER OF INSTRUCTIONS an item takes.  A quad of lanes per (query, head) pair issues ~1900 instructions per wave
// and item of which 256 are the FMAs; everything else (sample classification and preparation, DPP broadcasts, addresses,
// placement sums, DMA addresses, far steps, stores) is per lane, and a quad spends it four times per pair.  Hence:
//
//   pair of lanes   TWO lanes own one (query, head) pair: lane h accumulates the 16-byte pieces h, h + 2, h + 4, h + 6 of a
//                   pixel (16 channels) and prepares points 2h, 2h + 1 of every level.  A wave holds 32 pairs, a work
//                   item (8 x 16 level-0 tile + its queries of levels 1..3) is a 384-thread workgroup: waves 0..3 take
//                   two tile rows each, waves 4..5 the up to 64 queries of levels 1..3 in one pass.  Per pair that is half
//                   the per-lane work of the quad layout; the FMAs per pair are what they were.
//   bank classes    ds_read_b128 serves 16 lanes = 8 pairs at a time: pair class (quarter rotation cq = pair & 3, slot parity
//                   read first ce = (pair >> 2) & 1) -- at read t a pair reads the 32 bytes at quarter t ^ cq of the pixel
//                   whose slot parity is its own: eight pairs, eight different (parity, quarter) = all 64 banks once.
//
// Unchanged from msda_fwd_win2: the exact tile partition of the S queries, window sizes and placement by the mean top-left
// corner of the tile's own in-range samples, LDS-DMA staging with out-of-image slots as zeros, the persistent grid with a
// static stride (two resident workgroups per CU), the far path (an in-range sample with a corner outside its window takes
// raw buffer loads: correctness never depends on where the windows are), and the reference's sample arithmetic.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "../msda_common.hpp"

namespace msda {
namespace {

constexpr int kL0Waves = 4, kRestWaves = 2, kWaves = kL0Waves + kRestWaves, kT = kWaves * 64;
constexpr int kRestPairs = kRestWaves * 32;
constexpr int kTH = 8, kTW = 16;
static_assert(kTH == 2 * kL0Waves && kTW == 16, "a level-0 wave = two tile rows of 16 pairs");
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
static_assert(kZeroOff % 256 == 0, "zero region: slot parity by address bit 7");
struct Meta {
  int sum[4][4];                                                // per level: sum dx, sum dy, count, - (placement)
  int lvl[4][4];                                                // per level: H, W, first pixel, - (far path: level picked per pair)
};
constexpr int kMetaOff = kZeroOff + kZeroBytes;
constexpr int kLdsBytes = kMetaOff + ((sizeof(Meta) + 15) / 16) * 16;
static_assert(kLdsBytes <= 80 * 1024, "two workgroups per CU");

typedef const f32x4 __attribute__((address_space(3)))* lds4;
typedef float v2f __attribute__((ext_vector_type(2)));        // packed fp32 math: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32

// value held by lane HO (0 / 1) of this lane's PAIR: quad_perm [HO, HO, 2 + HO, 2 + HO]
template <int HO>
__device__ __forceinline__ uint32_t pb(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, HO ? 0xF5 : 0xA0, 0xF, 0xF, true);
}
template <int HO>
__device__ __forceinline__ float pbf(float v) { return __uint_as_float(pb<HO>(__float_as_uint(v))); }
template <int CTRL>
__device__ __forceinline__ int dppi(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }

__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) {   // (a & 0xffffff) * (b & 0xffffff) + c
  uint32_t r;
  asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));   // volatile: stays out of branches
  return r;
}
__device__ __forceinline__ uint32_t mad_u24_s(uint32_t a, uint32_t b_uniform, uint32_t c) {   // wave-uniform multiplier in an SGPR
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
// a wave-uniform value computed on the vector ALU into a SCALAR register, with the wait states the compiler's hazard
// recogniser cannot see inside an asm (msda_fwd_win2.hip)
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

#ifdef MSDA_WIN4_PROF
constexpr int kProfBlocks = 4096, kProfSlots = 16;
__device__ unsigned long long g_win4_prof[kProfBlocks * kWaves * kProfSlots];
#define W4_STAMP(i)                                                                                          \
  do {                                                                                                       \
    const unsigned blk_ = blockIdx.y * gridDim.x + blockIdx.x;                                               \
    if ((threadIdx.x & 63) == 0 && item == kk + K && blk_ < (unsigned)kProfBlocks)                           \
      g_win4_prof[(blk_ * kWaves + (threadIdx.x >> 6)) * kProfSlots + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define W4_STAMP(i) do { } while (0)
#endif

// workgroup barrier for LDS traffic only (__syncthreads() waits for vmcnt(0) too)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct Smp {      // one prepared NEAR sample (dead and far samples: zero weights, addresses in the zero region)
  v2f wT, wB;     // corner weights (first-top, second-top), (first-bottom, second-bottom); "first" = the pixel whose slot parity this pair reads first
  uint32_t aF, aS;   // LDS byte addresses of the first / second pixel of the top row
};

}  // namespace

// (128 VGPRs: with 4 wave slots per SIMD a second 6-wave workgroup fits a CU wherever its waves land; at 135 registers = 3 slots
// it only fits when the dispatcher happens to start it on the right SIMD, and the launch ran as one workgroup per CU)
__global__ void __launch_bounds__(kT, 4)
msda_fwd_win4(const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
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
  if (tid >= 256 && tid < 272) (&mt.sum[0][0])[tid - 256] = 0;
  if (tid >= 320 && tid < 324) {
    const bool t0 = (tid & 1) != 0, t1 = (tid & 2) != 0;
    *reinterpret_cast<int4*>(&mt.lvl[tid & 3][0]) = make_int4(sel4(t0, t1, lvH[0], lvH[1], lvH[2], lvH[3]), sel4(t0, t1, lvW[0], lvW[1], lvW[2], lvW[3]),
                                                              sel4(t0, t1, lvS[0], lvS[1], lvS[2], lvS[3]), 0);
  }

  const uint32_t pixB = (uint32_t)M * 128u;                // bytes from a pixel of head m to the next one
  const uint32_t hoff = (uint32_t)m * 128u;
  for (int item = kk; item < nitems; item += K) {
    // whatever the optimiser can prove invariant in this loop it hoists in front of it and spills: everything the body
    // derives values from passes through an empty asm, in place (msda_fwd_win2.hip)
    int ntiles_ = ntiles, TX_ = TX;
    asm volatile("" : "+s"(wv), "+s"(ntiles_), "+s"(TX_));
#pragma unroll
    for (int l = 0; l < 4; ++l) asm volatile("" : "+s"(lvH[l]), "+s"(lvW[l]), "+s"(lvS[l]));
    W4_STAMP(0);
    // quotients by v_rcp_f32: x + 0.5 is at least 0.5 / divisor away from an integer, far beyond the 1 ulp of the reciprocal
    const int b = to_sgpr((int)(((float)item + 0.5f) * __builtin_amdgcn_rcpf((float)ntiles_)));
    const int64_t pair_img = (int64_t)b * d.Lq * M + m;     // pair (query 0, head m) of this item's image: uniform bases, 32-bit per-lane offsets
    const __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(value) + (int64_t)b * d.S * M * 32, 0, (int)((uint32_t)d.S * pixB), 0x00020000);
    // ---- tile geometry.  Level-k pixels [f(t), f(t + 1)) with f(t) = ceil(t * T * n / n0 - 1/2) are the ones whose
    // centre falls into tile t -- an exact partition as long as every workgroup evaluates the same expression (the one of
    // msda_fwd_win); on level 0 it is f(t) = T * t
    const int tile_ = item - b * ntiles;
    int ty = to_sgpr((int)(((float)tile_ + 0.5f) * __builtin_amdgcn_rcpf((float)TX_)));
    int tx = tile_ - ty * TX;
    const bool l0 = wv < kL0Waves;                           // a wave of the level-0 rows?
    int ogx[4], ogy[4];                                      // window origins
    int npass = 1;

    for (int pass = 0; pass < npass; ++pass) {
      asm volatile("" : "+s"(wv), "+s"(tx), "+s"(ty));
#pragma unroll
      for (int l = 0; l < 4; ++l) asm volatile("" : "+s"(lvH[l]), "+s"(lvW[l]), "+s"(lvS[l]));
      int ln;                                                // lane of the wave
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      const int p = ln >> 1, h = ln & 1;                     // pair of the wave; this lane's half of it
      const int k = ln & 3;                                  // lane of the quad (levels are spread over a quad in the reductions)
      const bool k0 = (k & 1) != 0, k1 = (k & 2) != 0;
      const int cq = p & 3, ce = (p >> 2) & 1;               // bank class of the pair: quarter rotation, slot parity read first
      // this lane's channels: the 16-byte pieces h + 2 q of a pixel; read / accumulator t holds quarter q = t ^ cq
      // (byte offset inside a pixel: 16 h + 32 (t ^ cq) = off0 ^ 32 t -- one register, and pixel addresses are 128-byte aligned)
      const uint32_t off0 = (uint32_t)(16 * h + 32 * cq);
      // ---- this pair's query --------------------------------------------------------------------------------------
      bool live;
      uint32_t qidx;
      if (l0) {                                              // wave = two tile rows, pair = tile column
        const int xs0 = kTW * tx, ys0 = kTH * ty;
        const int row = 2 * wv + (p >> 4), col = p & 15;
        live = (pass == 0) & (col < min(kTW, lvW[0] - xs0)) & (row < min(kTH, lvH[0] - ys0));
        qidx = (uint32_t)(lvS[0] + (ys0 + row) * lvW[0] + xs0 + col);
      }
      W4_STAMP(1);
      if (pass == 0) lds_barrier();                          // #1: LDS set-up visible / everybody left the previous item
      W4_STAMP(2);
      if (!l0) {                                             // ri-th query of levels 1..3: lane l of a quad evaluates level l's rectangle
        const int gW = sel4(k0, k1, lvW[0], lvW[1], lvW[2], lvW[3]), gH = sel4(k0, k1, lvH[0], lvH[1], lvH[2], lvH[3]);
        const float fxs = (float)(kTW * gW) * __builtin_amdgcn_rcpf((float)lvW[0]), fys = (float)(kTH * gH) * __builtin_amdgcn_rcpf((float)lvH[0]);
        const int gxs = min(max((int)ceilf((float)tx * fxs - 0.5f), 0), gW);
        const int xe = tx == TX - 1 ? gW : min(max((int)ceilf((float)(tx + 1) * fxs - 0.5f), gxs), gW);
        const int gys = min(max((int)ceilf((float)ty * fys - 0.5f), 0), gH);
        const int ye = ty == TY - 1 ? gH : min(max((int)ceilf((float)(ty + 1) * fys - 0.5f), gys), gH);
        const int gnx = xe - gxs, cnt = gnx * (ye - gys);
        const int e1 = __builtin_amdgcn_readlane(cnt, 1), e2 = e1 + __builtin_amdgcn_readlane(cnt, 2);
        const int nrest = e2 + __builtin_amdgcn_readlane(cnt, 3);
        // these waves walk their queries kRestPairs at a time (one pass at the R50 shapes)
        npass = max(1, (nrest + kRestPairs - 1) / kRestPairs);
        const int ri = pass * kRestPairs + (wv - kL0Waves) * 32 + p;
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
      // ---- locations and weights of points 2h, 2h + 1 on the four levels (a pair reads 32 + 16 contiguous bytes per level) --
      live = live & (qidx < (uint32_t)d.Lq);                 // (shapes whose pixel count exceeds num_query: never outside the tensors)
      const uint32_t pair = mul_u24_s(live ? qidx : 0u, (uint32_t)M);   // (query, head 0) pair within the image; the head sits in the base pointers
      f32x4 lc[4];                                           // (x, y) of point 2h, (x, y) of point 2h + 1
      v2f sa[4];
#pragma unroll
      for (int l = 0; l < 4; ++l) { lc[l] = f32x4{0.f, 0.f, 0.f, 0.f}; sa[l] = v2f{0.f, 0.f}; }
      if (live) {
        const f32x4* lp = reinterpret_cast<const f32x4*>(loc + pair_img * 32 + (pair * 32u + 4u * (uint32_t)h));
        const v2f* ap = reinterpret_cast<const v2f*>(attn + pair_img * 16 + (pair * 16u + 2u * (uint32_t)h));
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          lc[l] = __builtin_nontemporal_load(lp + 2 * l);
          sa[l] = __builtin_nontemporal_load(ap + 2 * l);
        }
      }
      // (x, y) of local point LP on level l in pixels, and whether it is in range (the reference's arithmetic, cuh:282-288, :38-46)
      auto coord = [&](int l, int LP, bool& in) __attribute__((always_inline)) {
        const v2f fWH = {(float)lvW[l], (float)lvH[l]};
        const v2f q = LP ? v2f{lc[l][2], lc[l][3]} : v2f{lc[l][0], lc[l][1]};
        const v2f pc = __builtin_elementwise_fma(q, fWH, v2f{-0.5f, -0.5f});
        in = live & (pc.y > -1.f) & (pc.x > -1.f) & (pc.y < fWH.y) & (pc.x < fWH.x);
        return pc;
      };

      if (pass == 0) {
        if (l0) {
          // ---- window placement: mean top-left corner of the in-range samples of the tile's level-0 queries, per level:
          // reduce-scatter over the quad (lane l ends up with level l), then over the 4 quads of a DPP row --------------------
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
            px[l] = 0; py[l] = 0; pn[l] = 0;
#pragma unroll
            for (int LP = 0; LP < 2; ++LP) {
              bool in;
              const v2f pc = coord(l, LP, in);
              const int cx = cvt_i32(floorf(pc.x)), cy = cvt_i32(floorf(pc.y));   // (saturated garbage for poisoned locations: masked)
              px[l] += in ? cx : 0; py[l] += in ? cy : 0; pn[l] += in ? 1 : 0;
            }
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
        W4_STAMP(4);
        lds_barrier();                                       // #2 (the waves of levels 1..3 arrive with their loads still in flight)
        W4_STAMP(5);
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
        W4_STAMP(6);
      }

      // ---- near or far?  (near = all four corners inside the level's window, or outside the image) ------------------
      uint32_t nb = 0, fm = 0;                               // near bits (bit 2 l + LP); the pair's far samples (bit 4 * point + level)
      {
        uint32_t farmask = 0;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
#pragma unroll
          for (int LP = 0; LP < 2; ++LP) {
            bool in;
            const v2f pc = coord(l, LP, in);
            const int cx = cvt_i32(floorf(pc.x)), cy = cvt_i32(floorf(pc.y));
            // a level smaller than its window: top-left corners past the last in-range one are not "near"
            const int cxm = min(ogx[l] + kWW[l] - 2, lvW[l] - 1) - ogx[l], rym = min(ogy[l] + kWH[l] - 2, lvH[l] - 1) - ogy[l];
            const bool near = in & ((uint32_t)(cx - ogx[l]) <= (uint32_t)cxm) & ((uint32_t)(cy - ogy[l]) <= (uint32_t)rym);
            nb |= near ? (1u << (2 * l + LP)) : 0u;
            farmask |= (in & !near) ? (1u << (4 * LP + l)) : 0u;
          }
        }
        fm = farmask << (8 * h);                             // points 2h, 2h + 1
        fm |= (uint32_t)dppi<0xB1>((int)fm);                   // quad_perm [1,0,3,2]: the other lane of the pair
      }

      v2f acc[4][2];                                         // quarter t ^ cq of the pixel: channels (0,1), (2,3) of this lane's piece in it
#pragma unroll
      for (int t = 0; t < 4; ++t) { acc[t][0] = v2f{0.f, 0.f}; acc[t][1] = v2f{0.f, 0.f}; }

      // ---- far samples: raw buffer loads, one far sample per pair and step, in two corner rows ---------------------------
      auto far_step = [&]() __attribute__((always_inline)) {
        const bool has = fm != 0u;
        const int idx = has ? __builtin_ctz(fm) : 0;
        fm &= fm - 1u;
        const int fl_ = idx & 3, pt = idx >> 2;                // level and point of the far sample; lane pt >> 1 of the pair prepared it
        const int src = ((ln & ~1) | (pt >> 1)) << 2;          // byte address of the preparing lane for ds_bpermute
        const bool c1 = (fl_ & 1) != 0, c2 = (fl_ & 2) != 0, lp1 = (pt & 1) != 0;
        const int4 lv = *reinterpret_cast<const int4*>(&mt.lvl[fl_][0]);   // the far sample's level: H, W, first pixel
        const int fH_ = lv.x, fW_ = lv.y, fS_ = lv.z;
        // every lane evaluates its own candidate (level fl_, local point pt & 1), the pair pulls the preparing lane's
        const float qx = sel4(c1, c2, lp1 ? lc[0][2] : lc[0][0], lp1 ? lc[1][2] : lc[1][0], lp1 ? lc[2][2] : lc[2][0], lp1 ? lc[3][2] : lc[3][0]);
        const float qy = sel4(c1, c2, lp1 ? lc[0][3] : lc[0][1], lp1 ? lc[1][3] : lc[1][1], lp1 ? lc[2][3] : lc[2][1], lp1 ? lc[3][3] : lc[3][1]);
        const v2f pc = __builtin_elementwise_fma(v2f{qx, qy}, v2f{(float)fW_, (float)fH_}, v2f{-0.5f, -0.5f});   // == coord(fl_, pt & 1)
        const float qa = sel4(c1, c2, lp1 ? sa[0].y : sa[0].x, lp1 ? sa[1].y : sa[1].x, lp1 ? sa[2].y : sa[2].x, lp1 ? sa[3].y : sa[3].x);
        // pairs without a far sample left run along with zero weights: their stand-in coordinates must be finite
        const uint32_t hm = has ? 0xffffffffu : 0u;
        const float fxv = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)__float_as_uint(pc.x)) & hm);
        const float fyv = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)__float_as_uint(pc.y)) & hm);
        const float fav = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)__float_as_uint(qa)) & hm);
        const uint32_t rowG = mul_u24_s((uint32_t)fW_, pixB);
        const float xf = floorf(fxv), yf = floorf(fyv);
        const float lw = fxv - xf, lh = fyv - yf;
        const int fx0 = (int)xf, fy0 = (int)yf;              // in range or 0 for the stand-ins
        const bool t_ok = has & (fy0 >= 0), b_ok = has & (fy0 + 1 <= fH_ - 1), l_ok = fx0 >= 0, r_ok = fx0 + 1 <= fW_ - 1;
        const float wt = (1.f - lh) * fav, wb = lh * fav;
        // 24-bit multiply-adds (pixel index < 2^24, pitch < 2^24) on the CLAMPED top-left pixel: with fy0 or fx0 = -1 the
        // live corners sit in row / column 0, and a 24-bit product of a negative index is not what a 32-bit one wraps to
        const int cy = max(fy0, 0), cx = max(fx0, 0);
        const uint32_t o0 = mad_u24_s(mad_u24((uint32_t)cy, (uint32_t)fW_, (uint32_t)(fS_ + cx)), pixB, 0u);
        const uint32_t dx = fx0 >= 0 ? pixB : 0u, dy = fy0 >= 0 ? rowG : 0u;   // step to the right / bottom neighbour
        const uint32_t o1 = (t_ok & l_ok) ? o0 : kOobOffset;
        const uint32_t o2 = (t_ok & r_ok) ? o0 + dx : kOobOffset;
        const uint32_t o3 = (b_ok & l_ok) ? o0 + dy : kOobOffset;
        const uint32_t o4 = (b_ok & r_ok) ? o0 + dy + dx : kOobOffset;
        auto row = [&](uint32_t oL, uint32_t oR, float wrow) __attribute__((always_inline)) {
          f32x4 L[4], R[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {                         // (kOobOffset + off stays out of range)
            L[t] = buffer_load_f32x4(vsrc, oL + (off0 ^ (32u * t)), hoff);
            R[t] = buffer_load_f32x4(vsrc, oR + (off0 ^ (32u * t)), hoff);
          }
          const float wl = wrow * (1.f - lw), wr = wrow * lw;
          const v2f WL = {wl, wl}, WR = {wr, wr};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            acc[t][0] = __builtin_elementwise_fma(WL, v2f{L[t][0], L[t][1]}, acc[t][0]);
            acc[t][1] = __builtin_elementwise_fma(WL, v2f{L[t][2], L[t][3]}, acc[t][1]);
            acc[t][0] = __builtin_elementwise_fma(WR, v2f{R[t][0], R[t][1]}, acc[t][0]);
            acc[t][1] = __builtin_elementwise_fma(WR, v2f{R[t][2], R[t][3]}, acc[t][1]);
          }
        };
        row(o1, o2, wt);
        row(o3, o4, wb);
      };
      // ---- stage the four windows: LDS-DMA, one instruction = 8 consecutive slots (1 KB) of ONE level per wave.
      // Straight-line code with the same number of instructions in every wave (a wave without a chunk left in a
      // level issues an out-of-range one into the all-zero region, which costs no memory access) ----------------------
      if (pass == 0) {
        const uint32_t chunk = (uint32_t)(ln & 7) * 16u;
        const int sub = ln >> 3;
        auto stage_level = [&](auto ltag) __attribute__((always_inline)) {
          constexpr int LV = decltype(ltag)::value;
          constexpr int WW = kWW[LV], C0 = kBase[LV] / 8, C1 = kBase[LV + 1] / 8;
          constexpr int kSteps = (C1 - C0 + kWaves - 1) / kWaves;
          constexpr int kDR = (8 * kWaves) / WW, kDC = (8 * kWaves) % WW;
          const int Hs = lvH[LV], Ws = lvW[LV], xS = ogx[LV] + lvS[LV], oy = ogy[LV], ox = ogx[LV];
          int i = C0 + wv;                                   // this wave's first chunk of the level
          int subv = sub;
          asm volatile("" : "+v"(subv));                     // opaque: the level's start is computed HERE
          const int rel = 8 * wv + subv;                     // slot of this lane in the level's window
          int r = (int)(((float)rel + 0.5f) * (1.f / WW)), c = rel - r * WW;
#pragma unroll
          for (int t = 0; t < kSteps; ++t, i += kWaves) {
            const bool have = i < C1;                        // wave-uniform
            const int y = oy + r;
            const bool inside = have & ((unsigned)y < (unsigned)Hs) & ((unsigned)(ox + c) < (unsigned)Ws);
            // pixel index < 2^24 and pixel pitch M * 128 < 2^24 by win4_forward_ok: two full-rate 24-bit multiply-adds
            const uint32_t pix = mad_u24_s((uint32_t)y, (uint32_t)Ws, (uint32_t)(xS + c));
            const uint32_t in_off = mad_u24_s(pix, pixB, chunk);
            const uint32_t offv = inside ? in_off : kOobOffset;
            const int dst = have ? i * 1024 : kZeroOff;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vsrc, (__attribute__((address_space(3))) void*)(smem + dst), 16,
                                                     offv, hoff, 0, 0);
            if (t + 1 < kSteps) {
              c += kDC; r += kDR;
              while (kDC != 0 && c >= WW) { c -= WW; r += 1; }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        stage_level(std::integral_constant<int, 0>{});
        stage_level(std::integral_constant<int, 1>{});
        stage_level(std::integral_constant<int, 2>{});
        stage_level(std::integral_constant<int, 3>{});
        W4_STAMP(8);
      }
      // far steps while the windows travel (the first wait for far loads covers the wave's own DMA instructions, which
      // are older in the same queue)
      while (__ballot(fm != 0u)) far_step();
      W4_STAMP(9);
      if (pass == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of the windows has landed
        W4_STAMP(10);
        lds_barrier();                                       // #3 ... and everybody else's
        W4_STAMP(11);
        if (ln < 16 && wv == 0) (&mt.sum[0][0])[ln] = 0;      // the next item's sums (everybody has read this item's)
      }

      // ---- near samples: 4 levels x 4 points x 4 corners x 4 quarters from the LDS windows ------------------------------
      const uint32_t zero_first = smem_base + kZeroOff + 128u * (uint32_t)ce;        // parity ce; the other parity: ^ 128
      // this lane's two points on level LV (the reference's bilinear weights with the attention weight folded in)
      auto prepare = [&](auto ltag, Smp (&s)[2]) __attribute__((always_inline)) {
        constexpr int LV = decltype(ltag)::value;
#pragma unroll
        for (int LP = 0; LP < 2; ++LP) {
          bool in_;
          const v2f xy = coord(LV, LP, in_);
          const v2f fl = {floorf(xy.x), floorf(xy.y)};
          v2f fr = xy - fl;                                  // (fx, fy); inf - inf / NaN for poisoned locations ...
          fr.x = fmaxf(fr.x, 0.f); fr.y = fmaxf(fr.y, 0.f);  // ... which must not turn the zero weights of dead samples into NaN
          const v2f om = v2f{1.f, 1.f} - fr;                 // (1 - fx, 1 - fy)
          const int cx = cvt_i32(fl.x) - ogx[LV], ry = cvt_i32(fl.y) - ogy[LV];
          const bool near = ((nb >> (2 * LV + LP)) & 1u) != 0u;
          const uint32_t sw = (uint32_t)(cx ^ ce) & 1u;      // 1: the right-hand pixel has this pair's first parity
          const float an = near ? (LP ? sa[LV].y : sa[LV].x) : 0.f;   // dead and far samples: all four weights 0
          const v2f gx = sw ? v2f{fr.x, om.x} : v2f{om.x, fr.x};      // x factors of the (first, second) pixel
          const v2f wtb = v2f{om.y, fr.y} * an;              // (top, bottom) row weight x attention weight
          s[LP].wT = gx * wtb.x;
          s[LP].wB = gx * wtb.y;
          const uint32_t tl = smem_base + (uint32_t)(kBase[LV] * 128) + (uint32_t)(__mul24(ry, kWW[LV]) + cx) * 128u;
          s[LP].aF = near ? tl + (sw << 7) : zero_first;
          s[LP].aS = near ? tl + 128u - (sw << 7) : (zero_first ^ 128u);
        }
      };
      // Half rows (one pixel of a corner row = this lane's four 16-byte pieces, 16 registers) through a ring of three register sets
      struct Half { f32x4 r[4]; };
      struct Adr { uint32_t f0, s0; };                        // first / second pixel of the top row + this lane's offset for t = 0
      // half J of the sample (level LV, point PT): 0 = top row / first pixel, 1 = top / second, 2 = bottom / first, 3 = bottom / second
      auto fetch_half = [&](auto ltag, auto ptag, auto jtag, const Smp (&s)[2], Half& hh, Adr& ad) __attribute__((always_inline)) {
        constexpr int LV = decltype(ltag)::value, PT = decltype(ptag)::value, J = decltype(jtag)::value;
        constexpr int HO = PT >> 1, LP = PT & 1;
        constexpr int kRow = kWW[LV] * 8;                    // one window row, in 16-byte units
        if constexpr (J == 0) {
          ad.f0 = pb<HO>(s[LP].aF) + off0;
#pragma unroll
          for (int t = 0; t < 4; ++t) hh.r[t] = reinterpret_cast<lds4>((uintptr_t)(ad.f0 ^ (32u * t)))[0];
        } else if constexpr (J == 1) {
          ad.s0 = pb<HO>(s[LP].aS) + off0;
#pragma unroll
          for (int t = 0; t < 4; ++t) hh.r[t] = reinterpret_cast<lds4>((uintptr_t)(ad.s0 ^ (32u * t)))[0];
        } else if constexpr (J == 2) {
#pragma unroll
          for (int t = 0; t < 4; ++t) hh.r[t] = reinterpret_cast<lds4>((uintptr_t)(ad.f0 ^ (32u * t)))[kRow];
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) hh.r[t] = reinterpret_cast<lds4>((uintptr_t)(ad.s0 ^ (32u * t)))[kRow];
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      auto consume_half = [&](auto ptag, auto jtag, const Smp (&s)[2], const Half& hh) __attribute__((always_inline)) {
        constexpr int PT = decltype(ptag)::value, J = decltype(jtag)::value;
        constexpr int HO = PT >> 1, LP = PT & 1;
        const float w = pbf<HO>(J == 0 ? s[LP].wT.x : J == 1 ? s[LP].wT.y : J == 2 ? s[LP].wB.x : s[LP].wB.y);
        const v2f W2 = {w, w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[t][0] = __builtin_elementwise_fma(W2, v2f{hh.r[t][0], hh.r[t][1]}, acc[t][0]);
          acc[t][1] = __builtin_elementwise_fma(W2, v2f{hh.r[t][2], hh.r[t][3]}, acc[t][1]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(acc[t][0]), "+v"(acc[t][1]));   // pins the FMAs here
        __builtin_amdgcn_sched_barrier(0);
      };
      {
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        Half h0, h1, h2;
        Adr ad;
        Smp s0[2], s1[2];
        // generated sequence: consume half g, request half g + 3 (g = 16 * level + 4 * point + half); the next level's samples
        // are prepared just before their first half is requested
        prepare(I0{}, s0);
        fetch_half(I0{}, I0{}, I0{}, s0, h0, ad);
        fetch_half(I0{}, I0{}, I1{}, s0, h1, ad);
        fetch_half(I0{}, I0{}, I2{}, s0, h2, ad);
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
        prepare(I1{}, s1);
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
        prepare(I2{}, s0);
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
        prepare(I3{}, s1);
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

      W4_STAMP(13);                                          // LDS pass done
      if (live) {   // a pair writes 4 x 32 contiguous bytes
        float* op = out + pair_img * 32 + pair * 32u;
#pragma unroll
        for (int t = 0; t < 4; ++t)
          __builtin_nontemporal_store(f32x4{acc[t][0].x, acc[t][0].y, acc[t][1].x, acc[t][1].y},
                                      reinterpret_cast<f32x4*>(reinterpret_cast<char*>(op) + (off0 ^ (32u * t))));
      }
#ifdef MSDA_WIN4_PROF
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      W4_STAMP(15);
#endif
    }
  }
}

#ifdef MSDA_WIN4_PROF
extern "C" int msda_debug_read_prof4(void* dst, int nblocks) {
  if (nblocks > kProfBlocks) nblocks = kProfBlocks;
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_win4_prof), (size_t)nblocks * kWaves * kProfSlots * 8, 0, hipMemcpyDeviceToHost);
}
#endif

bool win4_forward_ok(const Dims& d) {
  // (the last condition keeps the work-item index, and item + 0.5, exact in float: the kernel splits it into (image,
  // tile) with a reciprocal; the grid's y extent is the number of workgroups per head)
  return d.D == 32 && d.P == 4 && d.L == 4 && d.Lq == d.S && d.S >= 1024 && d.M <= 65535 &&
         (int64_t)d.S * d.M * 128 < (int64_t)kOobOffset && d.N <= 65535 &&
         (int64_t)d.N * ((d.S + 127) / 128) < ((int64_t)1 << 22);
}

int launch_forward_win4(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn,
                        const Dims& d, float* out, hipStream_t stream) {
  static std::atomic<uint64_t> lds_opted_in{0};
  const void* fn = reinterpret_cast<const void*>(msda_fwd_win4);
  if (int rc = ensure_dynamic_lds(fn, kLdsBytes, lds_opted_in)) return rc;
  // persistent grid: two resident workgroups per CU (77 KB of LDS each) spread over the heads, static stride over the head's
  // items; head m = blockIdx.x, so that (by the observed round-robin placement of the linear workgroup id) XCD m % 8 only
  // touches head m's slice of `value` when M is a multiple of 8.  MSDA_WIN4_WGS=n: n workgroups per head (A/B switch).
  static const int wgs_env = std::getenv("MSDA_WIN4_WGS") ? std::atoi(std::getenv("MSDA_WIN4_WGS")) : 0;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  int K = wgs_env > 0 ? wgs_env : (2 * cus + d.M - 1) / d.M;
  const int items = d.N * ((d.S + 127) / 128);             // at least the tile count of any pyramid whose level 0 holds <= ~3/4 of the pixels
  if (K > items) K = items;
  if (K < 1) K = 1;
  if (K > 65535) K = 65535;
  hipLaunchKernelGGL(msda_fwd_win4, dim3((unsigned)d.M, (unsigned)K), dim3(kT), kLdsBytes, stream, value, shapes, lsi, loc,
                     attn, d, out);
  return (int)hipGetLastError();
}

}  // namespace msda
