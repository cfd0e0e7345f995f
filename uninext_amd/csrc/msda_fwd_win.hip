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
//   windows    = per level a WH x WW block of head m's value rows (128 B per pixel), 14x22 / 10x14 / 8x10 / 7x8
//                pixels = 584 slots (592 with chunk padding) = 74 KB, so that TWO 512-thread workgroups fit a CU.  A window is placed where
//                the tile's own samples fall: the mean top-left corner of the in-range samples of the first 128
//                queries, reduced over the workgroup with integer LDS atomics.  Pixels outside the image are staged
//                as zeros (out-of-range raw buffer offsets), so the zero padding of border samples needs no masks.
//                Staging is LDS-DMA (`buffer_load_dwordx4 ... lds`): no registers, no ds_write, and it stays in
//                flight while the wave does other work.
//   lane roles = a QUAD of lanes owns one (query, head) pair; lane k of the quad prepares the four points of level
//                k (x first, then y; corner weights with the attention weight folded in) and accumulates channels
//                8k .. 8k+7.  Prepared samples never touch memory: the consumer lanes read them straight out of the
//                preparing lane's registers with DPP quad_perm broadcasts (folded into the FMA / address add).
//   LDS banks  = a ds_read_b128 is served in four groups of 16 lanes = 4 quads; a quad reads 4 x 16 B spaced 32 B
//                apart = 16 of a pixel's 32 banks.  The four quads of a group take four different (16-byte half,
//                pixel parity) orders over the two x-adjacent corners of a bilinear row, so that in every
//                instruction the group covers all 64 banks exactly once, for any sample position.
//   far        = an in-range sample with a corner outside its window takes raw buffer loads (invalid corners at an
//                out-of-range offset), in a pass that runs while the window DMA is still in flight.  Correctness
//                never depends on where the windows are; only speed does (uniform-random locations are all far).
//
// All geometry comes from the int64 shape tensors on the device; the host only knows S, so the grid has
// ceil(S / 128) workgroups per (image, head) -- at least the number of tiles of any pyramid whose level 0 holds
// <= ~3/4 of the pixels -- and a workgroup walks tiles g, g + G, ... (one tile, or none, at the R50 shapes).
#include <type_traits>

#include "msda_common.hpp"

namespace msda {
namespace {

constexpr int kT = 512, kWaves = kT / 64, kQuads = kT / 4;
constexpr int kTH = 8, kTW = 16;
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
  int sum[4][4];                                                // per level: sum x0, sum y0, count, unused
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
      const unsigned blk_ = blockIdx.y * gridDim.x + blockIdx.x;                                             \
      if (blk_ < (unsigned)kProfBlocks) g_win_prof[blk_ * kProfSlots + (i)] = __builtin_amdgcn_s_memrealtime(); \
    }                                                                                                        \
  } while (0)
#else
#define WIN_STAMP(i) do { } while (0)
#endif

template <int SRC>
__device__ __forceinline__ uint32_t qb(uint32_t v) {   // value held by lane SRC of this lane's quad
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, SRC * 0x55, 0xF, 0xF, true);
}
template <int SRC>
__device__ __forceinline__ float qbf(float v) { return __uint_as_float(qb<SRC>(__float_as_uint(v))); }
// acc += (w as held by lane SRC of the quad) * d: the broadcast rides on the FMA's DPP operand (the compiler folds DPP
// into adds but not into v_fmac, so this one is spelled out)
template <int SRC>
__device__ __forceinline__ void fma_qb(float& acc, float w, float d) {
  asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[%3,%3,%3,%3] row_mask:0xf bank_mask:0xf bound_ctrl:1"
               : "+v"(acc) : "v"(w), "v"(d), "n"(SRC));
}

// One prepared sample in the preparing lane's registers.
//   near: w = (first-top, second-top, first-bottom, second-bottom) corner weights, a0 / a1 = LDS byte address of the
//         first / second pixel of the top row ("first" = the pixel whose slot parity this quad reads first)
//   far : w = (TL, TR, BL, BR) weights, a0 = byte offset of the top-left pixel in `value` (image-relative, modular),
//         a1 = corner validity bits
//   dead: w = 0, a0 / a1 = the zero region
struct Smp {
  float w[4];
  uint32_t a0, a1;
};

}  // namespace

__global__ void __launch_bounds__(kT, 4)
msda_fwd_win(const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
             const float* __restrict__ loc, const float* __restrict__ attn, Dims d, int G, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Meta& mt = *reinterpret_cast<Meta*>(smem + kMetaOff);
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = lane & 3, pq = lane >> 2;                 // lane of the quad = level it prepares; quad of the wave
  const int cls_a = (pq >> 1) & 1, cls_e = (pq >> 2) & 1; // bank class of the quad: half read first, parity read first
  const uint32_t c0 = (uint32_t)(32 * k + 16 * cls_a);    // this lane's first 16 bytes inside a pixel (second: ^ 16)
  const int M = d.M, b = blockIdx.y;
  const int m = blockIdx.x % M, g = blockIdx.x / M;

  // ---- launch constants straight from the shape tensors (uniform addresses: scalar loads, no LDS table, no barrier) --
  int lvH[4], lvW[4], lvS[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    lvH[l] = (int)shapes[2 * l];
    lvW[l] = (int)shapes[2 * l + 1];
    lvS[l] = (int)lsi[l];
  }
  const int TY = (lvH[0] + kTH - 1) / kTH, TX = (lvW[0] + kTW - 1) / kTW;
  const int ntiles = TY * TX;
  if (g >= ntiles) return;                                 // over-provisioned part of the grid
  // the all-zero region, the placement sums of the first tile
  for (int o = tid * 16; o < kZeroBytes; o += kT * 16) *reinterpret_cast<f32x4*>(smem + kZeroOff + o) = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tid < 16) (&mt.sum[0][0])[tid] = 0;
  __syncthreads();                                         // the sums are zero before any wave adds to them

  // this lane's own level (it prepares level k)
  const int myH = k == 0 ? lvH[0] : k == 1 ? lvH[1] : k == 2 ? lvH[2] : lvH[3];
  const int myW = k == 0 ? lvW[0] : k == 1 ? lvW[1] : k == 2 ? lvW[2] : lvW[3];
  const int myS = k == 0 ? lvS[0] : k == 1 ? lvS[1] : k == 2 ? lvS[2] : lvS[3];
  const int myWH = k == 0 ? kWH[0] : k == 1 ? kWH[1] : k == 2 ? kWH[2] : kWH[3];
  const int myWW = k == 0 ? kWW[0] : k == 1 ? kWW[1] : k == 2 ? kWW[2] : kWW[3];
  const uint32_t myWin = smem_base + 128u * (uint32_t)(k == 0 ? kBase[0] : k == 1 ? kBase[1] : k == 2 ? kBase[2] : kBase[3]);
  const uint32_t zero_first = smem_base + kZeroOff + 128u * (uint32_t)cls_e;        // parity cls_e
  const uint32_t zero_second = smem_base + kZeroOff + 128u * (uint32_t)(cls_e ^ 1);
  // tile -> query rectangle of level k along one axis: pixels [f(t), f(t + 1)) with f(t) = ceil(t * T * n / n0 - 1/2),
  // i.e. the pixels whose centre falls into the tile.  Evaluated in float: any monotone f with f(0) = 0 gives an exact
  // partition as long as every workgroup evaluates the same expression, which is all that correctness needs.
  const float fxs = (float)(kTW * myW) / (float)lvW[0], fys = (float)(kTH * myH) / (float)lvH[0];

  const uint32_t pixB = (uint32_t)M * 128u;
  const uint32_t myRowG = (uint32_t)myW * pixB;
  const __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(value) + (int64_t)b * d.S * M * 32, 0, (int)((uint32_t)d.S * pixB), 0x00020000);
  const uint32_t hoff = (uint32_t)m * 128u;
  const int64_t pair_img = (int64_t)b * d.Lq * M;           // first (query, head) pair of image b

  for (int tile = g; tile < ntiles; tile += G) {
    WIN_STAMP(0);
    const int ty = (int)(((float)tile + 0.5f) / (float)TX), tx = tile - ty * TX;
    // ---- tile geometry: lane k owns level k's query rectangle, the quad shares it by DPP ---------------------------
    int gxs, gys, gnx, gny;
    {
      const int xs = min(max((int)ceilf((float)tx * fxs - 0.5f), 0), myW);
      const int xe = tx == TX - 1 ? myW : min(max((int)ceilf((float)(tx + 1) * fxs - 0.5f), xs), myW);
      const int ys = min(max((int)ceilf((float)ty * fys - 0.5f), 0), myH);
      const int ye = ty == TY - 1 ? myH : min(max((int)ceilf((float)(ty + 1) * fys - 0.5f), ys), myH);
      gxs = xs; gys = ys; gnx = xe - xs; gny = ye - ys;
    }
    const int cnt = gnx * gny;
    const int c1 = (int)qb<0>((uint32_t)cnt), c2 = c1 + (int)qb<1>((uint32_t)cnt), c3 = c2 + (int)qb<2>((uint32_t)cnt);
    const int nq = __builtin_amdgcn_readfirstlane(c3 + (int)qb<3>((uint32_t)cnt));   // queries of this tile
    WIN_STAMP(1);

    // a round = the next 128 queries of the tile, one per quad
    auto query_of = [&](int qi, bool live) __attribute__((always_inline)) -> int64_t {
      const int ql = live ? (qi >= c1 ? 1 : 0) + (qi >= c2 ? 1 : 0) + (qi >= c3 ? 1 : 0) : 0;
      const int j = qi - (ql == 0 ? 0 : ql == 1 ? c1 : ql == 2 ? c2 : c3);
      const int nx = ql == 0 ? (int)qb<0>((uint32_t)gnx) : ql == 1 ? (int)qb<1>((uint32_t)gnx) : ql == 2 ? (int)qb<2>((uint32_t)gnx) : (int)qb<3>((uint32_t)gnx);
      const int xs = ql == 0 ? (int)qb<0>((uint32_t)gxs) : ql == 1 ? (int)qb<1>((uint32_t)gxs) : ql == 2 ? (int)qb<2>((uint32_t)gxs) : (int)qb<3>((uint32_t)gxs);
      const int ys = ql == 0 ? (int)qb<0>((uint32_t)gys) : ql == 1 ? (int)qb<1>((uint32_t)gys) : ql == 2 ? (int)qb<2>((uint32_t)gys) : (int)qb<3>((uint32_t)gys);
      const int Wq = ql == 0 ? lvW[0] : ql == 1 ? lvW[1] : ql == 2 ? lvW[2] : lvW[3];
      const int Sq = ql == 0 ? lvS[0] : ql == 1 ? lvS[1] : ql == 2 ? lvS[2] : lvS[3];
      const int yy = (int)(((float)j + 0.5f) / (float)max(nx, 1));
      const int q = Sq + (ys + yy) * Wq + xs + (j - yy * nx);
      return pair_img + (int64_t)(live ? q : 0) * M + m;
    };
    // lane k: the four points of level k -- 32 B of locations, 16 B of weights
    f32x4 lcA, lcB, at;
    auto fetch = [&](int64_t pair, bool live) __attribute__((always_inline)) {
      lcA = lcB = at = f32x4{0.f, 0.f, 0.f, 0.f};
      if (live) {
        const f32x4* lp = reinterpret_cast<const f32x4*>(loc + pair * 32 + 8 * k);
        lcA = __builtin_nontemporal_load(lp);
        lcB = __builtin_nontemporal_load(lp + 1);
        at = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(attn + pair * 16 + 4 * k));
      }
    };
    int qi = wv * 16 + pq;
    bool live = qi < nq;
    int64_t pair = query_of(qi, live);
    fetch(pair, live);
    int myOx = 0, myOy = 0;                                    // window origin of level k, known after the first round

    for (int q0 = 0; q0 < nq; q0 += kQuads) {
      // sample coordinates (the reference's arithmetic, cuh:282-288 and :38-46)
      float sx[4], sy[4], sa[4];
      bool inr[4];
      int x0[4], y0[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float lx = p == 0 ? lcA[0] : p == 1 ? lcA[2] : p == 2 ? lcB[0] : lcB[2];
        const float ly = p == 0 ? lcA[1] : p == 1 ? lcA[3] : p == 2 ? lcB[1] : lcB[3];
        sa[p] = at[p];
        sx[p] = lx * (float)myW - 0.5f;
        sy[p] = ly * (float)myH - 0.5f;
        inr[p] = live && (sy[p] > -1.f) && (sx[p] > -1.f) && (sy[p] < (float)myH) && (sx[p] < (float)myW);
        x0[p] = inr[p] ? (int)floorf(sx[p]) : 0;
        y0[p] = inr[p] ? (int)floorf(sy[p]) : 0;
      }

      if (q0 == 0) {
        WIN_STAMP(2);                                          // loc / attn have arrived, sample coordinates done
        // ---- window placement: mean top-left corner of this tile's in-range samples, per level ---------------------
        int ax = 0, ay = 0, an = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          ax += inr[p] ? x0[p] : 0;
          ay += inr[p] ? y0[p] : 0;
          an += inr[p] ? 1 : 0;
        }
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) {                   // over the 16 quads of the wave (same k)
          ax += __shfl_xor(ax, o, 64);
          ay += __shfl_xor(ay, o, 64);
          an += __shfl_xor(an, o, 64);
        }
        if (pq == 0 && an > 0) {
          atomicAdd(&mt.sum[k][0], ax);
          atomicAdd(&mt.sum[k][1], ay);
          atomicAdd(&mt.sum[k][2], an);
        }
        __syncthreads();
        WIN_STAMP(3);
        {
          const int4 sm = *reinterpret_cast<const int4*>(&mt.sum[k][0]);
          myOx = gxs - 3; myOy = gys - 3;
          if (sm.z > 0) {   // v_rcp_f32: every lane of the workgroup evaluates the same expression on the same sums
            const float inv = __builtin_amdgcn_rcpf((float)sm.z);
            myOx = (int)floorf((float)sm.x * inv + 0.5f) - (myWW - 2) / 2;
            myOy = (int)floorf((float)sm.y * inv + 0.5f) - (myWH - 2) / 2;
          }
          myOx = max(-1, min(myOx, myW + 1 - myWW));
          myOy = max(-1, min(myOy, myH + 1 - myWH));
        }
        // ---- stage the four windows: LDS-DMA, one instruction = 8 consecutive slots (1 KB) of ONE level per wave -------
        {
          int ogx[4], ogy[4];
          ogx[0] = __builtin_amdgcn_readfirstlane((int)qb<0>((uint32_t)myOx)); ogy[0] = __builtin_amdgcn_readfirstlane((int)qb<0>((uint32_t)myOy));
          ogx[1] = __builtin_amdgcn_readfirstlane((int)qb<1>((uint32_t)myOx)); ogy[1] = __builtin_amdgcn_readfirstlane((int)qb<1>((uint32_t)myOy));
          ogx[2] = __builtin_amdgcn_readfirstlane((int)qb<2>((uint32_t)myOx)); ogy[2] = __builtin_amdgcn_readfirstlane((int)qb<2>((uint32_t)myOy));
          ogx[3] = __builtin_amdgcn_readfirstlane((int)qb<3>((uint32_t)myOx)); ogy[3] = __builtin_amdgcn_readfirstlane((int)qb<3>((uint32_t)myOy));
          const uint32_t chunk = (uint32_t)(lane & 7) * 16u;
          const int sub = lane >> 3;
          for (int i = wv; i < kSlots / 8; i += kWaves) {        // i, and with it the level, is wave-uniform
            const int sl = (i >= kBase[1] / 8 ? 1 : 0) + (i >= kBase[2] / 8 ? 1 : 0) + (i >= kBase[3] / 8 ? 1 : 0);
            const int rel = 8 * i + sub - (sl == 0 ? kBase[0] : sl == 1 ? kBase[1] : sl == 2 ? kBase[2] : kBase[3]);
            const int ww = sl == 0 ? kWW[0] : sl == 1 ? kWW[1] : sl == 2 ? kWW[2] : kWW[3];
            const int r = (int)(((float)rel + 0.5f) * (sl == 0 ? 1.f / kWW[0] : sl == 1 ? 1.f / kWW[1] : sl == 2 ? 1.f / kWW[2] : 1.f / kWW[3]));
            const int y = (sl == 0 ? ogy[0] : sl == 1 ? ogy[1] : sl == 2 ? ogy[2] : ogy[3]) + r;
            const int x = (sl == 0 ? ogx[0] : sl == 1 ? ogx[1] : sl == 2 ? ogx[2] : ogx[3]) + rel - r * ww;
            const int Hs = sl == 0 ? lvH[0] : sl == 1 ? lvH[1] : sl == 2 ? lvH[2] : lvH[3];
            const int Ws = sl == 0 ? lvW[0] : sl == 1 ? lvW[1] : sl == 2 ? lvW[2] : lvW[3];
            const int Ss = sl == 0 ? lvS[0] : sl == 1 ? lvS[1] : sl == 2 ? lvS[2] : lvS[3];
            const bool inside = (unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws;   // also false for the pad slots' rows
            const uint32_t off = inside ? (uint32_t)(Ss + y * Ws + x) * pixB + chunk : kOobOffset;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vsrc, (__attribute__((address_space(3))) void*)(smem + i * 1024), 16,
                                                     off, hoff, 0, 0);
          }
        }
        WIN_STAMP(4);                                          // window DMA issued
      }

      // ---- prepare this lane's four samples ---------------------------------------------------------------------------
      Smp smp[4];
      uint32_t farmask = 0;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float a = sa[p];
        const float fx = sx[p] - floorf(sx[p]), fy = sy[p] - floorf(sy[p]);
        const float wt = (1.f - fy) * a, wb = fy * a;
        const float wTL = wt * (1.f - fx), wTR = wt * fx, wBL = wb * (1.f - fx), wBR = wb * fx;
        const int cx = x0[p] - myOx, ry = y0[p] - myOy;
        const bool near = inr[p] && (unsigned)cx <= (unsigned)(myWW - 2) && (unsigned)ry <= (unsigned)(myWH - 2);
        const bool far = inr[p] && !near;
        const bool swap = ((cx & 1) != cls_e);
        const uint32_t tl = myWin + (uint32_t)(ry * myWW + cx) * 128u;
        const bool t_ok = y0[p] >= 0, b_ok = y0[p] + 1 <= myH - 1, l_ok = x0[p] >= 0, r_ok = x0[p] + 1 <= myW - 1;
        const uint32_t bits = (t_ok && l_ok ? 1u : 0u) | (t_ok && r_ok ? 2u : 0u) | (b_ok && l_ok ? 4u : 0u) | (b_ok && r_ok ? 8u : 0u);
        const uint32_t goff = (uint32_t)(myS + y0[p] * myW + x0[p]) * pixB;
        smp[p].w[0] = near ? (swap ? wTR : wTL) : far ? wTL : 0.f;
        smp[p].w[1] = near ? (swap ? wTL : wTR) : far ? wTR : 0.f;
        smp[p].w[2] = near ? (swap ? wBR : wBL) : far ? wBL : 0.f;
        smp[p].w[3] = near ? (swap ? wBL : wBR) : far ? wBR : 0.f;
        smp[p].a0 = near ? (swap ? tl + 128u : tl) : far ? goff : zero_first;
        smp[p].a1 = near ? (swap ? tl : tl + 128u) : far ? bits : zero_second;
        farmask |= far ? (1u << p) : 0u;
      }

      f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};   // channels at c0 and at c0 ^ 16
      if (q0 == 0) WIN_STAMP(5);                               // samples prepared

      // ---- far samples: raw buffer loads, one far sample per quad and step (overlaps the window DMA in round 0) ----------
      {
        uint32_t fm = farmask << (4 * k);                      // the pair's 16 samples: bit 4 * level + point
        fm |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)fm, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
        fm |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)fm, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
        const uint32_t gfirst = 32u * (uint32_t)k + 16u * (uint32_t)cls_a;
        while (__ballot(fm != 0u)) {
          const bool has = fm != 0u;
          const int idx = has ? __builtin_ctz(fm) : 0;
          fm &= fm - 1u;
          const int ps = idx & 3;
          const int src = ((lane & ~3) | (idx >> 2)) << 2;     // byte address of the preparing lane for ds_bpermute
          // the preparing lane's sample `ps`: every lane selects its own candidate, the quad pulls the right one
          const float cw0 = ps == 0 ? smp[0].w[0] : ps == 1 ? smp[1].w[0] : ps == 2 ? smp[2].w[0] : smp[3].w[0];
          const float cw1 = ps == 0 ? smp[0].w[1] : ps == 1 ? smp[1].w[1] : ps == 2 ? smp[2].w[1] : smp[3].w[1];
          const float cw2 = ps == 0 ? smp[0].w[2] : ps == 1 ? smp[1].w[2] : ps == 2 ? smp[2].w[2] : smp[3].w[2];
          const float cw3 = ps == 0 ? smp[0].w[3] : ps == 1 ? smp[1].w[3] : ps == 2 ? smp[2].w[3] : smp[3].w[3];
          const uint32_t ca0 = ps == 0 ? smp[0].a0 : ps == 1 ? smp[1].a0 : ps == 2 ? smp[2].a0 : smp[3].a0;
          const uint32_t ca1 = ps == 0 ? smp[0].a1 : ps == 1 ? smp[1].a1 : ps == 2 ? smp[2].a1 : smp[3].a1;
          const uint32_t off = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)ca0);
          const uint32_t bits = has ? (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)ca1) : 0u;
          const uint32_t rowG = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)myRowG);
          const float w1 = has ? __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)__float_as_uint(cw0))) : 0.f;
          const float w2 = has ? __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)__float_as_uint(cw1))) : 0.f;
          const float w3 = has ? __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)__float_as_uint(cw2))) : 0.f;
          const float w4 = has ? __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)__float_as_uint(cw3))) : 0.f;
          const uint32_t o1 = (bits & 1u) ? off + gfirst : kOobOffset;
          const uint32_t o2 = (bits & 2u) ? off + pixB + gfirst : kOobOffset;
          const uint32_t o3 = (bits & 4u) ? off + rowG + gfirst : kOobOffset;
          const uint32_t o4 = (bits & 8u) ? off + rowG + pixB + gfirst : kOobOffset;
          const f32x4 d1a = buffer_load_f32x4(vsrc, o1, hoff), d1b = buffer_load_f32x4(vsrc, o1 ^ 16u, hoff);
          const f32x4 d2a = buffer_load_f32x4(vsrc, o2, hoff), d2b = buffer_load_f32x4(vsrc, o2 ^ 16u, hoff);
          const f32x4 d3a = buffer_load_f32x4(vsrc, o3, hoff), d3b = buffer_load_f32x4(vsrc, o3 ^ 16u, hoff);
          const f32x4 d4a = buffer_load_f32x4(vsrc, o4, hoff), d4b = buffer_load_f32x4(vsrc, o4 ^ 16u, hoff);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            accA[c] = fmaf(w4, d4a[c], fmaf(w3, d3a[c], fmaf(w2, d2a[c], fmaf(w1, d1a[c], accA[c]))));
            accB[c] = fmaf(w4, d4b[c], fmaf(w3, d3b[c], fmaf(w2, d2b[c], fmaf(w1, d1b[c], accB[c]))));
          }
          asm volatile("" : "+v"(accA), "+v"(accB));
        }
        // far samples are done: the LDS pass sees them as dead
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const bool f = ((farmask >> p) & 1u) != 0u;
          smp[p].w[0] = f ? 0.f : smp[p].w[0];
          smp[p].w[1] = f ? 0.f : smp[p].w[1];
          smp[p].w[2] = f ? 0.f : smp[p].w[2];
          smp[p].w[3] = f ? 0.f : smp[p].w[3];
          smp[p].a0 = f ? zero_first : smp[p].a0;
          smp[p].a1 = f ? zero_second : smp[p].a1;
        }
      }

      if (q0 == 0) {
        WIN_STAMP(6);                                          // far pass done
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of the windows has landed
        __syncthreads();                                     // ... and everybody else's
        WIN_STAMP(7);                                          // windows complete
      }

      // ---- the next round's locations and weights travel while this round reads the LDS ----------------------------------
      const int64_t cur_pair = pair;
      const bool cur_live = live;
      if (q0 + kQuads < nq) {
        qi += kQuads;
        live = qi < nq;
        pair = query_of(qi, live);
        fetch(pair, live);
      }

      // ---- near samples: 16 samples x 4 corners x 2 halves from the LDS windows ---------------------------------------
      auto lds_step = [&](auto ltag, int p) __attribute__((always_inline)) {
        constexpr int LV = decltype(ltag)::value;
        constexpr int kRow = kWW[LV] * 8;                     // one window row, in 16-byte units
        const uint32_t aF = qb<LV>(smp[p].a0) + c0, aS = qb<LV>(smp[p].a1) + c0;
        const lds4 pF = reinterpret_cast<lds4>((uintptr_t)aF), pF2 = reinterpret_cast<lds4>((uintptr_t)(aF ^ 16u));
        const lds4 pS = reinterpret_cast<lds4>((uintptr_t)aS), pS2 = reinterpret_cast<lds4>((uintptr_t)(aS ^ 16u));
        const f32x4 tFa = pF[0], tFb = pF2[0], tSa = pS[0], tSb = pS2[0];
        const f32x4 bFa = pF[kRow], bFb = pF2[kRow], bSa = pS[kRow], bSb = pS2[kRow];
        float accA_[4] = {accA[0], accA[1], accA[2], accA[3]}, accB_[4] = {accB[0], accB[1], accB[2], accB[3]};
        // in load order, so that the waits on the LDS returns are progressive
#pragma unroll
        for (int c = 0; c < 4; ++c) fma_qb<LV>(accA_[c], smp[p].w[0], tFa[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) fma_qb<LV>(accB_[c], smp[p].w[0], tFb[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) fma_qb<LV>(accA_[c], smp[p].w[1], tSa[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) fma_qb<LV>(accB_[c], smp[p].w[1], tSb[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) fma_qb<LV>(accA_[c], smp[p].w[2], bFa[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) fma_qb<LV>(accB_[c], smp[p].w[2], bFb[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) fma_qb<LV>(accA_[c], smp[p].w[3], bSa[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) fma_qb<LV>(accB_[c], smp[p].w[3], bSb[c]);
        accA = f32x4{accA_[0], accA_[1], accA_[2], accA_[3]};
        accB = f32x4{accB_[0], accB_[1], accB_[2], accB_[3]};
        __builtin_amdgcn_sched_barrier(0);                    // one sample in flight per wave: 4 waves / SIMD hide the LDS
      };
#pragma unroll
      for (int p = 0; p < 4; ++p) lds_step(std::integral_constant<int, 0>{}, p);
#pragma unroll
      for (int p = 0; p < 4; ++p) lds_step(std::integral_constant<int, 1>{}, p);
#pragma unroll
      for (int p = 0; p < 4; ++p) lds_step(std::integral_constant<int, 2>{}, p);
#pragma unroll
      for (int p = 0; p < 4; ++p) lds_step(std::integral_constant<int, 3>{}, p);

      if (q0 == 0) WIN_STAMP(8);                               // LDS pass of the first round done
      if (cur_live) {
        float* op = out + cur_pair * 32 + 8 * k;
        __builtin_nontemporal_store(accA, reinterpret_cast<f32x4*>(op + 4 * cls_a));
        __builtin_nontemporal_store(accB, reinterpret_cast<f32x4*>(op + 4 * (cls_a ^ 1)));
      }
    }
    WIN_STAMP(9);                                              // all rounds done (wave 0)
    if (tile + G < ntiles) {                                   // odd pyramids only: another tile for this workgroup
      __syncthreads();                                         // everybody is done with the windows and the sums
      if (tid < 16) (&mt.sum[0][0])[tid] = 0;
      __syncthreads();
    }
  }
}

#ifdef MSDA_WIN_PROF
extern "C" int msda_debug_read_prof(void* dst, int nblocks) {
  if (nblocks > kProfBlocks) nblocks = kProfBlocks;
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_win_prof), (size_t)nblocks * kProfSlots * 8, 0, hipMemcpyDeviceToHost);
}
#endif

bool win_forward_ok(const Dims& d) {
  return d.D == 32 && d.P == 4 && d.L == 4 && d.Lq == d.S && d.S >= 1024 &&
         (int64_t)d.S * d.M * 128 < (int64_t)kOobOffset && d.N <= 65535;
}

int launch_forward_win(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn,
                       const Dims& d, float* out, hipStream_t stream) {
  static std::atomic<uint64_t> lds_opted_in{0};
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(msda_fwd_win), kLdsBytes, lds_opted_in)) return rc;
  const int G = (d.S + 127) / 128;
  hipLaunchKernelGGL(msda_fwd_win, dim3((unsigned)(d.M * G), (unsigned)d.N), dim3(kT), kLdsBytes, stream, value, shapes,
                     lsi, loc, attn, d, G, out);
  return (int)hipGetLastError();
}

}  // namespace msda
