// msda_bwd_dst -- MSDeformAttn backward for decoder-style calls (few queries, many pixels) with grad_value summed on the
// DESTINATION side: fp32, D = 32, L = P = 4.  gfx950 only.  Backward variant 8 (round 6); does, for these calls, the work of
// ops/src/cuda/ms_deform_im2col_cuda.cuh:301-403 / :406-920.
//
// Why it exists: msda_bwd_dec is (vector work) + (L2 atomics), added up -- 31 us of the first (a half wave per (query, level)
// unit: the sample preparation runs 32 times redundantly) and 45 us of the second (563 k direct full-line atomics on levels 0 / 1,
// up to 307 k flush atomics of the 16 query slices; profiles/r05_timing_ablations.txt) -- and its coarse-level sums are fixed point.
// Here every corner is added in LDS, in float64:
//
//   work slot  = (image b, head m, REGION of the image): the pyramid is cut into nty x ntx regions -- level l into tiles of
//                ceil(H_l / nty) x ceil(W_l / ntx) pixels, region (ry, rx) owning tile (ry, rx) of EVERY level (about 12 x 16 pixels of
//                level 0, 6 x 8 of level 1, 3 x 4, 2 x 2 at the R50 shapes: 256 pixels x 32 channels of float64 sums = 64 KB of LDS,
//                two 512-thread workgroups per CU; ds_add_f64 is native, one rounding per element at the end -- no fixed-point
//                scale, no bound, nothing to state about dynamic range).  Sampling locations are normalised, so a query's samples
//                fall into the same region on all four levels (or next to it): the slots are balanced and need no query slices.
//   scan       = every wave walks its share of the (image, head)'s (query, level) units LANE-PARALLEL -- a quad of lanes reads the
//                whole 128-byte line of a (query, head)'s locations -- and files the CANDIDATES (top-left corner in the region's
//                tile of that level or one pixel before it; six compares per sample) into a wave-private list in LDS.
//                (The first generation of this kernel had one LEVEL per slot: a (query, head, level)'s 32 bytes sit in a line of their
//                own, and 77 tiles of level 0 each pulled all of them through the L1: profiles/r06_backward_decoder_dst.txt.)
//   process    = ten candidates at a time: ten lanes run the reference's per-sample arithmetic (cuh:282-288 / :38-46) once each and
//                stage the result; a half wave per record, lane = channel, then adds weight x attention x upstream gradient into
//                the corners' LDS sums (ds_add_f64), and the record's OWNER -- the region that holds the in-image pixel nearest to
//                the sample's top-left corner -- also gathers the four corner values and writes the sample's grad_attn_weight /
//                grad_sampling_loc (cuh:87-159's formulas; every element written once, by exactly one workgroup).  The loads of
//                the five record pairs of a batch travel together.
//   flush      = touched pixels leave once, as one full-line float atomic each (read and clear of the LDS sums).
//   grid       = persistent, two workgroups per CU; slots are DRAWN from a per-launch counter, the next slot's draw in the
//                shadow of the current slot; (image, head)-major, so that the workgroups running together share the lines they scan.
#include <algorithm>

#include "msda_common.hpp"

namespace msda {
namespace {

constexpr int kDstThreads = 512, kDstWaves = kDstThreads / 64;
constexpr int kDstPx = 256;                                          // pixels of a region, all levels together
constexpr int kDstTH0 = 12, kDstTW0 = 16;                            // level-0 tile the region grid starts from
#ifndef DST_SCAN
#define DST_SCAN 3
#endif
#ifndef DST_BATCH
#define DST_BATCH 5
#endif
constexpr int kDstScan = DST_SCAN;                                   // scan steps whose loads travel together
constexpr int kDstBatch = DST_BATCH;                                 // pairs of records whose loads travel together
constexpr int kDstStage = 2 * kDstBatch;
constexpr int kDstCap = 128;                                         // candidates per wave list; worked off when a round might overflow it
constexpr int kDstAccBytes = kDstPx * 32 * 8;                        // 64 KB
constexpr int kDstListBytes = kDstWaves * kDstCap * 3 * 4;           // 12 KB: {unit << 2 | point, x, y}
constexpr int kDstStageBytes = kDstWaves * kDstStage * 5 * 4;        // staged records: {unit, point, flags | h_low | w_low | lh | lw}
struct DstLevel { int H, W, S0, y0, x0, th, tw, base; };             // per slot and level: geometry, this region's tile, its first sum
constexpr int kDstLevelOff = kDstAccBytes + kDstListBytes + kDstStageBytes;
constexpr int kDstMapOff = kDstLevelOff + 4 * (int)sizeof(DstLevel);
constexpr int kDstNextOff = kDstMapOff + kDstPx * 4;
constexpr int kDstLds = kDstNextOff + 16;
static_assert(kDstLds <= 80 * 1024, "two workgroups per CU");

__device__ unsigned g_dst_tickets[64 * 16];                         // one slot counter per launch in flight (64 B apart)

__device__ __forceinline__ float dst_half_sum(float f) {   // over the 32 lanes of a half wave; every lane gets the total
  f += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(f), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  f += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(f), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  f += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(f), 0x141, 0xF, 0xF, true));   // row_half_mirror
  f += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(f), 0x140, 0xF, 0xF, true));   // row_mirror
  f += __shfl_xor(f, 16, 64);
  return f;
}

// staged record, word 0: bits 0..3 = corner 1..4 (top-left, top-right, bottom-left, bottom-right) is a valid pixel of this region's
// tile; bit 4 = this region owns the sample (writes its grad_attn_weight / grad_sampling_loc); bits 5..6 = point; 7..8 = level;
// 9.. = query
constexpr int kOwner = 16;

}  // namespace

__global__ void __launch_bounds__(kDstThreads, 4)
msda_bwd_dst(const float* __restrict__ grad_out, const float* __restrict__ value, const int64_t* __restrict__ shapes,
             const int64_t* __restrict__ lsi, const float* __restrict__ loc, const float* __restrict__ attn, Dims d,
             float* __restrict__ grad_value, float* __restrict__ grad_loc, float* __restrict__ grad_attn, unsigned* __restrict__ ticket) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* const acc = reinterpret_cast<double*>(smem);                   // [256 pixels][32 channels]
  const int tid = threadIdx.x, lane = tid & 63, ln = lane & 31, half = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  int* const lst = reinterpret_cast<int*>(smem + kDstAccBytes) + wv * (kDstCap * 3);       // this wave's candidates: three arrays of kDstCap
  int* const cA = lst;
  float* const cX = reinterpret_cast<float*>(lst + kDstCap);
  float* const cY = reinterpret_cast<float*>(lst + 2 * kDstCap);
  int* const stg = reinterpret_cast<int*>(smem + kDstAccBytes + kDstListBytes) + wv * (kDstStage * 5);   // staged records: five arrays of kDstStage
  int* const sA = stg;
  int* const sH = stg + kDstStage;
  int* const sW = stg + 2 * kDstStage;
  float* const sLh = reinterpret_cast<float*>(stg + 3 * kDstStage);
  float* const sLw = reinterpret_cast<float*>(stg + 4 * kDstStage);
  DstLevel* const lv = reinterpret_cast<DstLevel*>(smem + kDstLevelOff);
  int* const pixmap = reinterpret_cast<int*>(smem + kDstMapOff);       // sum slot -> pixel of the (image's) pyramid, -1: none
  int* const s_next = reinterpret_cast<int*>(smem + kDstNextOff);      // the workgroup's next slot
  const int M = d.M;

  // ---- the region grid, from the shape tensors (uniform: scalar loads and arithmetic) ------------------------------------------
  int H[4], W[4], S0[4], TH[4], TW[4], BASE[5];
#pragma unroll
  for (int l = 0; l < 4; ++l) { H[l] = max((int)shapes[2 * l], 1); W[l] = max((int)shapes[2 * l + 1], 1); S0[l] = (int)lsi[l]; }
  int nty = (H[0] + kDstTH0 - 1) / kDstTH0, ntx = (W[0] + kDstTW0 - 1) / kDstTW0;
  for (int it = 0; it < 4096; ++it) {   // (any pyramid: finer grids until a region's tiles fit the sums; 1 x 1 tiles need four)
    int sum = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      TH[l] = (H[l] + nty - 1) / nty;
      TW[l] = (W[l] + ntx - 1) / ntx;
      sum += TH[l] * TW[l];
    }
    if (sum <= kDstPx) break;
    int hm = 0, wm = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) { hm = max(hm, TH[l]); wm = max(wm, TW[l]); }
    if (hm >= wm) ++nty; else ++ntx;
  }
  BASE[0] = 0;
#pragma unroll
  for (int l = 0; l < 4; ++l) BASE[l + 1] = BASE[l] + TH[l] * TW[l];
  const int nreg = nty * ntx, total = d.N * M * nreg;
  const uint32_t ps32 = (uint32_t)M * 32u;
  const __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(value), 0, (int)((uint32_t)d.N * (uint32_t)d.S * ps32 * 4u), 0x00020000);
  const int nunits = d.Lq * 4;                                           // (query, level) units of an (image, head)

  for (int o = tid * 2; o < kDstPx * 32; o += kDstThreads * 2) *reinterpret_cast<double2*>(acc + o) = make_double2(0.0, 0.0);

  // Slots are DRAWN (ticket != nullptr: one counter per launch, the draw of the next slot in the current one's shadow) or dealt out
  // by stride (under stream capture: replays of one graph on two streams would share the counter).
  int slot = blockIdx.x;
  if (ticket) {
    if (tid == 0) *s_next = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    slot = *s_next;
  }
  while (slot < total) {
    unsigned drawn = 0;
    if (ticket && tid == 0) drawn = atomicAdd(ticket, 1u);               // the next slot: used at the end of this one
    const int bm = slot / nreg, reg = slot - bm * nreg;
    const int b = bm / M, m = bm - b * M;
    const int ry = reg / ntx, rx = reg - ry * ntx;
    const int64_t pair0 = (int64_t)b * d.Lq * M + m;                     // (query 0, head m) of image b; the next query is M pairs on
    const float* const loc_s = loc + pair0 * 32;
    const float* const attn_s = attn + pair0 * 16;
    const float* const go_s = grad_out + pair0 * 32;
    float* const ga_s = grad_attn + pair0 * 16;
    float* const gl_s = grad_loc + pair0 * 32;
    const uint32_t img_off = (uint32_t)b * (uint32_t)d.S * ps32 + (uint32_t)m * 32u;   // (pixel 0, head m) of image b in value / grad_value

    // ---- the slot's tables: per level the geometry and this region's tile; per sum slot its pixel -------------------------------
    if (tid < 4) {
      const int l = tid;
      DstLevel e;
      e.H = l == 0 ? H[0] : l == 1 ? H[1] : l == 2 ? H[2] : H[3];
      e.W = l == 0 ? W[0] : l == 1 ? W[1] : l == 2 ? W[2] : W[3];
      e.S0 = l == 0 ? S0[0] : l == 1 ? S0[1] : l == 2 ? S0[2] : S0[3];
      e.th = l == 0 ? TH[0] : l == 1 ? TH[1] : l == 2 ? TH[2] : TH[3];
      e.tw = l == 0 ? TW[0] : l == 1 ? TW[1] : l == 2 ? TW[2] : TW[3];
      e.base = l == 0 ? BASE[0] : l == 1 ? BASE[1] : l == 2 ? BASE[2] : BASE[3];
      e.y0 = ry * e.th;
      e.x0 = rx * e.tw;
      lv[l] = e;
    }
    if (tid < kDstPx) {
      const int i = tid;
      const int l = (i >= BASE[1] ? 1 : 0) + (i >= BASE[2] ? 1 : 0) + (i >= BASE[3] ? 1 : 0);
      const int Hl = l == 0 ? H[0] : l == 1 ? H[1] : l == 2 ? H[2] : H[3], Wl = l == 0 ? W[0] : l == 1 ? W[1] : l == 2 ? W[2] : W[3];
      const int Sl = l == 0 ? S0[0] : l == 1 ? S0[1] : l == 2 ? S0[2] : S0[3];
      const int th = l == 0 ? TH[0] : l == 1 ? TH[1] : l == 2 ? TH[2] : TH[3], tw = l == 0 ? TW[0] : l == 1 ? TW[1] : l == 2 ? TW[2] : TW[3];
      const int bs = l == 0 ? BASE[0] : l == 1 ? BASE[1] : l == 2 ? BASE[2] : BASE[3];
      const int j = i - bs, py = j / tw, px = j - py * tw;
      const int gy = ry * th + py, gx = rx * tw + px;
      pixmap[i] = (i < BASE[4] && gy < Hl && gx < Wl) ? Sl + gy * Wl + gx : -1;
    }
    __syncthreads();                                                     // (also: the previous slot's flush has cleared the sums)

    // ---- process: kDstStage candidates of this wave's list at a time ----------------------------------------------------------------
    auto process = [&](int cnt) __attribute__((always_inline)) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");            // the list's writes (this wave's own) are behind us
      for (int r0 = 0; r0 < cnt; r0 += kDstStage) {
        // stage: one lane per candidate runs the reference's per-sample arithmetic once
        if (lane < kDstStage && r0 + lane < cnt) {
          const int a = cA[r0 + lane];
          const int l = (a >> 2) & 3;
          const DstLevel e = lv[l];
          const Sample<float> t = make_sample<float>(cX[r0 + lane], cY[r0 + lane], e.H, e.W);
          int flags = 0;
          if (t.in_range) {
            const bool ry0 = (unsigned)(t.h_low - e.y0) < (unsigned)e.th, ry1 = (unsigned)(t.h_low + 1 - e.y0) < (unsigned)e.th;
            const bool rx0 = (unsigned)(t.w_low - e.x0) < (unsigned)e.tw, rx1 = (unsigned)(t.w_low + 1 - e.x0) < (unsigned)e.tw;
            flags = (t.ok1 && ry0 && rx0 ? 1 : 0) | (t.ok2 && ry0 && rx1 ? 2 : 0) | (t.ok3 && ry1 && rx0 ? 4 : 0) | (t.ok4 && ry1 && rx1 ? 8 : 0);
            const int oy = max(t.h_low, 0), ox = max(t.w_low, 0);       // (in range: h_low <= H - 1, w_low <= W - 1)
            if ((unsigned)(oy - e.y0) < (unsigned)e.th && (unsigned)(ox - e.x0) < (unsigned)e.tw) flags |= kOwner;
          }
          // (flags 0: a candidate that touches nothing here)
          sA[lane] = ((a >> 4) << 9) | (l << 7) | ((a & 3) << 5) | flags;
          sH[lane] = t.h_low; sW[lane] = t.w_low;
          sLh[lane] = t.lh; sLw[lane] = t.lw;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // request: upstream gradient, attention weight and (owner) corner values of kDstBatch record pairs
        float g[kDstBatch], at[kDstBatch], v[kDstBatch][4];
#pragma unroll
        for (int k = 0; k < kDstBatch; ++k) {
          const int j = 2 * k + half;
          const bool live = r0 + j < cnt;
          const int a = live ? sA[j] : 0, hlo = sH[j], wlo = sW[j];
          const int l = (a >> 7) & 3, p = (a >> 5) & 3, q = a >> 9;
          const DstLevel e = lv[l];
          const bool own = (a & kOwner) != 0;
          const bool any = (a & 31) != 0;
          g[k] = any ? go_s[(int64_t)q * (M * 32) + ln] : 0.f;
          at[k] = any ? attn_s[(int64_t)q * (M * 16) + l * 4 + p] : 0.f;
          // (all in 32-bit arithmetic: with h_low or w_low = -1 the top-left offset wraps and its neighbours wrap back)
          const uint32_t ob = (img_off + (uint32_t)(e.S0 + hlo * e.W + wlo) * ps32 + (uint32_t)ln) * 4u;
          const bool tp = hlo >= 0, bt = hlo + 1 <= e.H - 1, lf = wlo >= 0, rt = wlo + 1 <= e.W - 1;   // (make_sample's rule)
          const uint32_t rowb = (uint32_t)e.W * ps32 * 4u, pxb = ps32 * 4u;
          v[k][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vsrc, (own && tp && lf) ? ob : kOobOffset, 0, 0));
          v[k][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vsrc, (own && tp && rt) ? ob + pxb : kOobOffset, 0, 0));
          v[k][2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vsrc, (own && bt && lf) ? ob + rowb : kOobOffset, 0, 0));
          v[k][3] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vsrc, (own && bt && rt) ? ob + rowb + pxb : kOobOffset, 0, 0));
        }
        // consume
#pragma unroll
        for (int k = 0; k < kDstBatch; ++k) {
          if (r0 + 2 * k >= cnt) break;                                  // (wave-uniform)
          const int j = 2 * k + half;
          const bool live = r0 + j < cnt;
          const int a = live ? sA[j] : 0, hlo = sH[j], wlo = sW[j];
          const float lh = sLh[j], lw = sLw[j];
          const int flags = a & 31, l = (a >> 7) & 3, p = (a >> 5) & 3, q = a >> 9;
          const DstLevel e = lv[l];
          const float hh = 1.f - lh, hw_ = 1.f - lw;
          const float w1 = hh * hw_, w2 = hh * lw, w3 = lh * hw_, w4 = lh * lw;
          const float tgv = g[k] * at[k];
          // the corners inside the region's tile
          double* const at_p = acc + ((e.base + (hlo - e.y0) * e.tw + (wlo - e.x0)) * 32 + ln);
          if (flags & 1) unsafeAtomicAdd(at_p, (double)(w1 * tgv));
          if (flags & 2) unsafeAtomicAdd(at_p + 32, (double)(w2 * tgv));
          if (flags & 4) unsafeAtomicAdd(at_p + e.tw * 32, (double)(w3 * tgv));
          if (flags & 8) unsafeAtomicAdd(at_p + e.tw * 32 + 32, (double)(w4 * tgv));
          const bool own = (flags & kOwner) != 0;
          if (__ballot(own)) {                                           // wave-uniform (the sums need every lane of a half)
            float pa = g[k] * (w1 * v[k][0] + w2 * v[k][1] + w3 * v[k][2] + w4 * v[k][3]);
            float pw = tgv * (hh * (v[k][1] - v[k][0]) + lh * (v[k][3] - v[k][2]));
            float ph = tgv * (hw_ * (v[k][2] - v[k][0]) + lw * (v[k][3] - v[k][1]));
            pa = dst_half_sum(pa);
            pw = dst_half_sum(pw);
            ph = dst_half_sum(ph);
            if (own && ln == 0) {
              ga_s[(int64_t)q * (M * 16) + l * 4 + p] = pa;
              gl_s[(int64_t)q * (M * 32) + l * 8 + 2 * p] = (float)e.W * pw;
              gl_s[(int64_t)q * (M * 32) + l * 8 + 2 * p + 1] = (float)e.H * ph;
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // the staging area is free again
      }
    };

    // ---- scan: one (query, level) unit per lane and step -- a quad reads a (query, head)'s whole line of locations ----------------
    int cnt = 0;
    {
      const DstLevel e = lv[lane & 3];                                   // this lane's level (the step is a multiple of 4)
      const float fH = (float)e.H, fW = (float)e.W;
      const float fy0 = (float)(e.y0 - 1), fy1 = (float)(e.y0 + e.th), fx0 = (float)(e.x0 - 1), fx1 = (float)(e.x0 + e.tw);
      // A batch = kDstScan steps whose loads travel together; its rounds -- (step k, point p) -- append while the list has room.  A
      // round that might overflow the list stops the walk: the list is worked off at ONE place below (the body of process() is long:
      // twelve inlined copies cost 56 spilled registers), the batch is loaded again (its registers are not kept across process()) and
      // the walk resumes at that round.  The same place works off what is left after the last batch.
      int base = wv * 64, start = 0;
      bool more = true;
      while (more) {
        bool full = false;
        if (base < nunits) {
          f32x4 la[kDstScan], lb[kDstScan];
#pragma unroll
          for (int k = 0; k < kDstScan; ++k) {
            const int u = base + k * kDstThreads + lane;
            const float nan = __builtin_nanf("");
            la[k] = lb[k] = f32x4{nan, nan, nan, nan};                   // (past the end: never a candidate)
            if (u < nunits) {
              const f32x4* lp = reinterpret_cast<const f32x4*>(loc_s + (int64_t)(u >> 2) * (M * 32) + (u & 3) * 8);
              la[k] = lp[0];
              lb[k] = lp[1];
            }
          }
#pragma unroll
          for (int k = 0; k < kDstScan; ++k) {
            const int u = base + k * kDstThreads + lane;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              if (k * 4 + p >= start && !full && base + k * kDstThreads < nunits) {   // (wave-uniform)
                const float x = p == 0 ? la[k][0] : p == 1 ? la[k][2] : p == 2 ? lb[k][0] : lb[k][2];
                const float y = p == 0 ? la[k][1] : p == 1 ? la[k][3] : p == 2 ? lb[k][1] : lb[k][3];
                const float h_im = y * fH - 0.5f, w_im = x * fW - 0.5f;  // (make_sample's expressions)
                if (reg == 0) {                                          // (uniform) samples outside the level: zero gradients, written by region 0
                  const bool in_range = (h_im > -1.f) && (w_im > -1.f) && (h_im < fH) && (w_im < fW);
                  if (u < nunits && !in_range) {
                    const int64_t qo = (int64_t)(u >> 2) * M;
                    ga_s[qo * 16 + (u & 3) * 4 + p] = 0.f;
                    *reinterpret_cast<f32x2*>(gl_s + qo * 32 + (u & 3) * 8 + 2 * p) = f32x2{0.f, 0.f};
                  }
                }
                const bool match = (h_im >= fy0) && (h_im < fy1) && (w_im >= fx0) && (w_im < fx1);
                const unsigned long long mask = __ballot(match);
                if (mask) {
                  const int n = __builtin_popcountll(mask);
                  if (cnt + n > kDstCap) {
                    full = true;
                    start = k * 4 + p;
                  } else {
                    if (match) {
                      const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                      cA[pos] = (u << 2) | p; cX[pos] = x; cY[pos] = y;
                    }
                    cnt += n;
                  }
                }
              }
            }
          }
          if (!full) { base += kDstThreads * kDstScan; start = 0; }
        } else {
          more = false;
        }
        if (full || (!more && cnt > 0)) { process(cnt); cnt = 0; }
      }
    }
    __syncthreads();

    // ---- flush: a half wave per sum slot, sixteen each: read and CLEAR the sums, one full-line atomic per touched pixel -------------
    {
      constexpr int kPer = kDstPx / (kDstThreads / 32);                 // 16
#pragma unroll
      for (int i = 0; i < kPer; ++i) {
        const int px = (tid >> 5) + i * (kDstThreads / 32);
        const float fv = (float)acc[px * 32 + ln];
        acc[px * 32 + ln] = 0.0;
        const int pix = pixmap[px];
        const unsigned long long any = __ballot(fv != 0.f) >> (tid & 32) & 0xffffffffull;
        if (any != 0 && pix >= 0) atomic_add(grad_value + (size_t)(img_off + (uint32_t)pix * ps32 + (uint32_t)ln), fv);
      }
    }
    if (ticket && tid == 0) {
      *s_next = (int)drawn;
      if (drawn == (unsigned)total + gridDim.x - 1u) *ticket = 0u;       // the launch's last draw: the counter is ready for the next launch
    }
    __syncthreads();                                                     // the sums are zero, the tables free, before the next slot
    slot = ticket ? *s_next : slot + (int)gridDim.x;
  }
}

bool dst_backward_ok(const Dims& d) {
  return d.D == 32 && d.L == 4 && d.P == 4 && d.Lq >= 1 && d.Lq < (1 << 21) && d.M <= 65535 && d.N <= 65535 &&
         (int64_t)d.N * d.S * d.M * 128 < (int64_t)kOobOffset;   // (32-bit byte offsets into value / grad_value)
}

int launch_backward_dst(const float* grad_out, const float* value, const int64_t* shapes, const int64_t* lsi,
                        const float* loc, const float* attn, const Dims& d, float* grad_value, float* grad_loc,
                        float* grad_attn, hipStream_t stream) {
  static std::atomic<uint64_t> lds_opted_in{0};
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(msda_bwd_dst), kDstLds, lds_opted_in)) return rc;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  static const int per_cu = ab_env_int("MSDA_BWD_DST_PER_CU", 2);
  // the launch's slot counter: a ring of 64 per device (zero in the code object; the launch's last draw zeroes it again), none under
  // stream capture
  static std::atomic<unsigned> seq{0};
  unsigned* ticket = nullptr;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusActive; }
  if (cs == hipStreamCaptureStatusNone) {
    void* base = nullptr;
    if (hipGetSymbolAddress(&base, HIP_SYMBOL(g_dst_tickets)) == hipSuccess) ticket = static_cast<unsigned*>(base) + 16 * (seq.fetch_add(1, std::memory_order_relaxed) % 64u);
    else (void)hipGetLastError();
  }
  hipLaunchKernelGGL(msda_bwd_dst, dim3((unsigned)(std::max(cus, 1) * std::max(per_cu, 1))), dim3(kDstThreads), kDstLds, stream, grad_out,
                     value, shapes, lsi, loc, attn, d, grad_value, grad_loc, grad_attn, ticket);
  return (int)hipGetLastError();
}

}  // namespace msda
