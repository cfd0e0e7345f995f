// msda_bwd_dst -- MSDeformAttn backward for decoder-style calls (few queries, many pixels) with grad_value summed on the
// DESTINATION side: fp32, D = 32, L = P = 4.  gfx950 only.  Backward variant 8 (round 6); does, for these calls, the work of
// ops/src/cuda/ms_deform_im2col_cuda.cuh:301-403 / :406-920.
//
// Why it exists: msda_bwd_dec is (vector work) + (L2 atomics), added up -- 31 us of the first (a half wave per (query, level)
// unit: the sample preparation runs 32 times redundantly) and 45 us of the second (563 k direct full-line atomics on levels 0 / 1,
// up to 307 k flush atomics of the 16 query slices; profiles/r05_timing_ablations.txt) -- and its coarse-level sums are fixed point.
// Here every corner is added in LDS, in float64:
//
//   work slot  = (image b, head m, level l, 16 x 16 pixel TILE of that level, slice of the queries).  The workgroup keeps the
//                tile's 256 x 32 sums in LDS as float64 (64 KB: two 512-thread workgroups per CU; ds_add_f64 is native, one
//                rounding per element at the end -- no fixed-point scale, no bound, nothing to state about dynamic range).
//   scan       = every wave walks its share of the slice's samples of level l LANE-PARALLEL, in two passes: a handful of
//                instructions per sample pick the CANDIDATES (top-left corner in the tile's rows / columns or the one before), the
//                reference's per-sample arithmetic (cuh:282-288 / :38-46) then runs once per candidate and files the samples that
//                have a corner in the tile -- or that the tile OWNS: the in-image pixel nearest to the sample's top-left corner is
//                the tile's -- as records of a wave-private list in LDS.
//   process    = a half wave per record, lane = channel: weight x attention x upstream gradient into the corners' LDS sums
//                (ds_add_f64); the owner also gathers the four corner values and writes the sample's grad_attn_weight /
//                grad_sampling_loc (cuh:87-159's formulas; every element written once, by exactly one workgroup).  The loads of
//                five pairs of records travel together.
//   flush      = touched pixels leave once, as one full-line float atomic each (read and clear of the LDS sums): ~390 k per call
//                at the R50 training shapes against ~870 k of msda_bwd_dec.
//   grid       = persistent, two workgroups per CU; slots are DRAWN from a per-launch counter, level 3 first (the sliced, heavier
//                ones), the next slot's draw in the shadow of the current slot.
//
// The slices of a level are chosen on the device from the shape tensors (the host only knows S and Lq): a tile's expected
// records, 4 Lq x 1.25 / tiles, over a target of 256 per workgroup, at most 16.
//
// Where it stands (profiles/r06_backward_decoder_dst.txt): 94.6 us against msda_bwd_dec's 89.0 (Lq = 1100, launch + memset) -- NOT the
// automatic choice.  A slot is a chain of memory round trips (~2 us each on cold inputs) that two workgroups per CU do not hide:
// level-0 slot 10.1 us = scan 3.6 (bound by the L1's line fills: the 32 bytes of a (query, head, level) sit in a 128-byte line of
// their own) + process 2.3 + barrier 1.4 + flush 1.8; 2352 slots over 512 workgroups.  What it offers today is the NUMERICS: float64
// sums, one rounding per element, no fixed point anywhere -- msda_hip_set_variant(1, 8).
#include <algorithm>

#include "msda_common.hpp"

namespace msda {
namespace {

constexpr int kDstThreads = 512, kDstWaves = kDstThreads / 64;
constexpr int kDstTile = 16, kDstTilePx = kDstTile * kDstTile;
constexpr int kDstRecCap = 96;                                       // records per wave list; a list is worked off when a step would overflow it
#ifndef DST_SCAN
#define DST_SCAN 5
#endif
#ifndef DST_BATCH
#define DST_BATCH 5
#endif
#ifndef DST_TARGET
#define DST_TARGET 256
#endif
constexpr int kDstTargetRecords = DST_TARGET, kDstMaxSlices = 16;
constexpr int kDstScan = DST_SCAN;                                          // scan steps whose loads travel together
constexpr int kDstBatch = DST_BATCH;                                         // pairs of records whose loads travel together
constexpr int kDstAccBytes = kDstTilePx * 32 * 8;                    // 64 KB
constexpr int kDstRecBytes = kDstWaves * kDstRecCap * 5 * 4;         // 15 KB
constexpr int kDstLds = kDstAccBytes + kDstRecBytes + 16;
static_assert(kDstLds <= 80 * 1024, "two workgroups per CU");

__device__ __forceinline__ float dst_half_sum(float f) {   // over the 32 lanes of a half wave; every lane gets the total
  f += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(f), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  f += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(f), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  f += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(f), 0x141, 0xF, 0xF, true));   // row_half_mirror
  f += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(f), 0x140, 0xF, 0xF, true));   // row_mirror
  f += __shfl_xor(f, 16, 64);
  return f;
}

// record flags: bits 0..3 = corner 1..4 (top-left, top-right, bottom-left, bottom-right) is a valid pixel of this tile;
// bit 4 = this tile owns the sample (writes its grad_attn_weight / grad_sampling_loc); bits 5..6 = point; bits 8.. = query - q0
constexpr int kOwner = 16;

__device__ unsigned g_dst_tickets[64 * 16];                         // one slot counter per launch in flight (64 B apart)


}  // namespace

__global__ void __launch_bounds__(kDstThreads, 4)
msda_bwd_dst(const float* __restrict__ grad_out, const float* __restrict__ value, const int64_t* __restrict__ shapes,
             const int64_t* __restrict__ lsi, const float* __restrict__ loc, const float* __restrict__ attn, Dims d,
             float* __restrict__ grad_value, float* __restrict__ grad_loc, float* __restrict__ grad_attn, unsigned* __restrict__ ticket) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* const acc = reinterpret_cast<double*>(smem);                   // [256 pixels][32 channels]
  int* const s_next = reinterpret_cast<int*>(smem + kDstAccBytes + kDstRecBytes);   // the workgroup's next slot
  const int tid = threadIdx.x, lane = tid & 63, ln = lane & 31, half = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  int* const rec = reinterpret_cast<int*>(smem + kDstAccBytes) + wv * (kDstRecCap * 5);   // this wave's list: five arrays of kDstRecCap
  int* const rA = rec;
  int* const rH = rec + kDstRecCap;
  int* const rW = rec + 2 * kDstRecCap;
  float* const rLh = reinterpret_cast<float*>(rec + 3 * kDstRecCap);
  float* const rLw = reinterpret_cast<float*>(rec + 4 * kDstRecCap);
  const int M = d.M, NM = d.N * d.M;

  // ---- the launch's slots, from the shape tensors (uniform: scalar loads) ----------------------------------------------------
  int H[4], W[4], S0[4], TX[4], NT[4], NS[4], CNT[4];
  int total = 0;
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    H[l] = (int)shapes[2 * l]; W[l] = (int)shapes[2 * l + 1]; S0[l] = (int)lsi[l];
    TX[l] = (W[l] + kDstTile - 1) / kDstTile;
    NT[l] = TX[l] * ((H[l] + kDstTile - 1) / kDstTile);
    const int want = (5 * d.Lq + NT[l] * kDstTargetRecords - 1) / max(NT[l] * kDstTargetRecords, 1);   // 4 Lq x 1.25 expected records
    NS[l] = max(1, min(min(want, kDstMaxSlices), (d.Lq + 31) / 32));
    CNT[l] = NT[l] * NS[l] * NM;
    total += CNT[l];
  }
  const uint32_t ps32 = (uint32_t)M * 32u;
  const __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(value), 0, (int)((uint32_t)d.N * (uint32_t)d.S * ps32 * 4u), 0x00020000);

  for (int o = tid * 2; o < kDstTilePx * 32; o += kDstThreads * 2) *reinterpret_cast<double2*>(acc + o) = make_double2(0.0, 0.0);
  __syncthreads();

  // Slots are DRAWN (ticket != nullptr: one counter per launch, include the draw of the next slot in the current one's shadow) --
  // they differ 2 x in length and a static deal left the last workgroup 20 us behind the median -- or dealt out by stride (under
  // stream capture: replays of one graph on two streams would share the counter).
  int slot = blockIdx.x;
  if (ticket) {
    if (tid == 0) *s_next = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    slot = *s_next;
  }
  while (slot < total) {
    unsigned drawn = 0;
    if (ticket && tid == 0) drawn = atomicAdd(ticket, 1u);               // the next slot: used at the end of this one
    // slots in the order level 3, 2, 1, 0; within a level (tile, slice) major, (image, head) minor
    int l = 3, idx = slot;
    if (idx >= CNT[3]) { idx -= CNT[3]; l = 2; if (idx >= CNT[2]) { idx -= CNT[2]; l = 1; if (idx >= CNT[1]) { idx -= CNT[1]; l = 0; } } }
    const bool l0 = (l & 1) != 0, l1 = (l & 2) != 0;
    const int Hl = l1 ? (l0 ? H[3] : H[2]) : (l0 ? H[1] : H[0]), Wl = l1 ? (l0 ? W[3] : W[2]) : (l0 ? W[1] : W[0]);
    const int Sl = l1 ? (l0 ? S0[3] : S0[2]) : (l0 ? S0[1] : S0[0]);
    const int txs = l1 ? (l0 ? TX[3] : TX[2]) : (l0 ? TX[1] : TX[0]), nsl = l1 ? (l0 ? NS[3] : NS[2]) : (l0 ? NS[1] : NS[0]);
    const int bm = idx % NM, ts = idx / NM;
    const int b = bm / M, m = bm - b * M;
    const int tile = ts / nsl, sl = ts - tile * nsl;
    const int ty = tile / txs, tx = tile - ty * txs;
    const int y0 = ty * kDstTile, x0 = tx * kDstTile;
    const int qper = (d.Lq + nsl - 1) / nsl;
    const int q0 = sl * qper, q1 = min(d.Lq, q0 + qper);
    const int ns = max(q1 - q0, 0) * 4;                                  // samples of this slice on level l: 4 x query + point
    const int64_t pair0 = ((int64_t)b * d.Lq + q0) * M + m;              // the slice's first (query, head) pair; the next query is M pairs on
    const float* const loc_s = loc + pair0 * 32 + l * 8;
    const float* const attn_s = attn + pair0 * 16 + l * 4;
    const float* const go_s = grad_out + pair0 * 32;
    float* const ga_s = grad_attn + pair0 * 16 + l * 4;
    float* const gl_s = grad_loc + pair0 * 32 + l * 8;
    const uint32_t lvl_off = ((uint32_t)b * (uint32_t)d.S + (uint32_t)Sl) * ps32 + (uint32_t)m * 32u;

    // ---- process: a half wave per record of this wave's list, kDstBatch pairs of records at a time: their upstream gradients and
    // (owner) corner values are requested together, then consumed -- one memory round trip per batch, not per record ---------------
    auto process = [&](int cnt) __attribute__((always_inline)) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");            // the list's writes (this wave's own) are behind us
      for (int r0 = 0; r0 < cnt; r0 += 2 * kDstBatch) {
        float g[kDstBatch], at_[kDstBatch], v[kDstBatch][4];
#pragma unroll
        for (int k = 0; k < kDstBatch; ++k) {
          const int r = r0 + 2 * k + half;
          const bool live = r < cnt;
          const int rr = live ? r : r0;
          const int a = rA[rr], hlo = rH[rr], wlo = rW[rr];
          const bool own = live && (a & kOwner) != 0;
          g[k] = live ? go_s[(int64_t)(a >> 8) * (M * 32) + ln] : 0.f;
          at_[k] = live ? attn_s[(int64_t)(a >> 8) * (M * 16) + ((a >> 5) & 3)] : 0.f;   // (the weight travels with the gradient: the scan does not touch attn_weight)
          // (all in 32-bit arithmetic: with h_low or w_low = -1 the top-left offset wraps and its neighbours wrap back)
          const uint32_t ob = (lvl_off + (uint32_t)(hlo * Wl + wlo) * ps32 + (uint32_t)ln) * 4u;
          const bool tp = hlo >= 0, bt = hlo + 1 <= Hl - 1, lf = wlo >= 0, rt = wlo + 1 <= Wl - 1;   // (make_sample's rule)
          const uint32_t rowb = (uint32_t)Wl * ps32 * 4u, pxb = ps32 * 4u;
          v[k][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vsrc, (own && tp && lf) ? ob : kOobOffset, 0, 0));
          v[k][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vsrc, (own && tp && rt) ? ob + pxb : kOobOffset, 0, 0));
          v[k][2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vsrc, (own && bt && lf) ? ob + rowb : kOobOffset, 0, 0));
          v[k][3] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vsrc, (own && bt && rt) ? ob + rowb + pxb : kOobOffset, 0, 0));
        }
#pragma unroll
        for (int k = 0; k < kDstBatch; ++k) {
          if (r0 + 2 * k >= cnt) break;                                  // (wave-uniform)
          const int r = r0 + 2 * k + half;
          const bool live = r < cnt;
          const int rr = live ? r : r0;
          const int a = rA[rr], hlo = rH[rr], wlo = rW[rr];
          const float lh = rLh[rr], lw = rLw[rr], at = at_[k];
          const int flags = live ? (a & 31) : 0, p = (a >> 5) & 3, ql = a >> 8;
          const float hh = 1.f - lh, hw_ = 1.f - lw;
          const float w1 = hh * hw_, w2 = hh * lw, w3 = lh * hw_, w4 = lh * lw;
          const float tgv = g[k] * at;
          // the corners inside the tile
          double* const at_p = acc + (((hlo - y0) * kDstTile + (wlo - x0)) * 32 + ln);
          if (flags & 1) unsafeAtomicAdd(at_p, (double)(w1 * tgv));
          if (flags & 2) unsafeAtomicAdd(at_p + 32, (double)(w2 * tgv));
          if (flags & 4) unsafeAtomicAdd(at_p + kDstTile * 32, (double)(w3 * tgv));
          if (flags & 8) unsafeAtomicAdd(at_p + kDstTile * 32 + 32, (double)(w4 * tgv));
          const bool own = (flags & kOwner) != 0;
          if (__ballot(own)) {                                           // wave-uniform (the sums need every lane of a half)
            float pa = g[k] * (w1 * v[k][0] + w2 * v[k][1] + w3 * v[k][2] + w4 * v[k][3]);
            float pw = tgv * (hh * (v[k][1] - v[k][0]) + lh * (v[k][3] - v[k][2]));
            float ph = tgv * (hw_ * (v[k][2] - v[k][0]) + lw * (v[k][3] - v[k][1]));
            pa = dst_half_sum(pa);
            pw = dst_half_sum(pw);
            ph = dst_half_sum(ph);
            if (own && ln == 0) {
              ga_s[(int64_t)ql * (M * 16) + p] = pa;
              gl_s[(int64_t)ql * (M * 32) + 2 * p] = (float)Wl * pw;
              gl_s[(int64_t)ql * (M * 32) + 2 * p + 1] = (float)Hl * ph;
            }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");            // the list is free again
    };

    // ---- scan, pass 2: the candidates of this wave's list become records (the reference's per-sample arithmetic, once per candidate)
    auto refine = [&](int cnt) __attribute__((always_inline)) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      for (int c = lane; c < cnt; c += 64) {
        const int sq = rA[c];
        const Sample<float> t = make_sample<float>(rLh[c], rLw[c], Hl, Wl);
        int flags = 0;
        if (t.in_range) {
          const bool ry0 = (unsigned)(t.h_low - y0) < (unsigned)kDstTile, ry1 = (unsigned)(t.h_low + 1 - y0) < (unsigned)kDstTile;
          const bool rx0 = (unsigned)(t.w_low - x0) < (unsigned)kDstTile, rx1 = (unsigned)(t.w_low + 1 - x0) < (unsigned)kDstTile;
          flags = (t.ok1 && ry0 && rx0 ? 1 : 0) | (t.ok2 && ry0 && rx1 ? 2 : 0) | (t.ok3 && ry1 && rx0 ? 4 : 0) | (t.ok4 && ry1 && rx1 ? 8 : 0);
          const int oy = max(t.h_low, 0), ox = max(t.w_low, 0);         // (in range: h_low <= H - 1, w_low <= W - 1)
          if ((unsigned)(oy - y0) < (unsigned)kDstTile && (unsigned)(ox - x0) < (unsigned)kDstTile) flags |= kOwner;
        }
        rA[c] = ((sq >> 2) << 8) | ((sq & 3) << 5) | flags;             // (flags 0: a candidate that touches nothing here -- process skips it)
        rH[c] = t.h_low; rW[c] = t.w_low;
        rLh[c] = t.lh; rLw[c] = t.lw;
      }
    };
    // ---- scan, pass 1: one sample per lane and step, kDstScan steps' locations and weights requested together; a CANDIDATE is a sample
    // whose top-left corner lies in the tile's rows / columns or the one before them (a superset of what pass 2 keeps): 77 tiles of level
    // 0 walk the same 4400 samples of their (image, head), so what every sample costs every tile has to be a handful of instructions ----
    int cnt = 0;
    const float fy0 = (float)(y0 - 1), fy1 = (float)(y0 + kDstTile), fx0 = (float)(x0 - 1), fx1 = (float)(x0 + kDstTile);
    const float fH = (float)Hl, fW = (float)Wl;
    // A batch = kDstScan steps whose loads travel together; its steps append while the list has room.  A step that might overflow the
    // list stops the walk: the list is worked off at ONE place below (the body of process() is long; inlined once per step and once
    // behind the loop it cost 23 spilled registers), the batch is loaded again (its registers are not kept across process()) and the walk
    // resumes at that step.  The same place works off what is left after the last batch.
    int base = wv * 64, start = 0;
    bool more = true;
    while (more) {
      bool full = false;
      if (base < ns) {
        float sx[kDstScan], sy[kDstScan];
#pragma unroll
        for (int k = 0; k < kDstScan; ++k) {
          const int s = base + k * kDstThreads + lane;
          sx[k] = sy[k] = __builtin_nanf("");                            // (past the end: never a candidate)
          if (s < ns) {
            const int64_t qo = (int64_t)(s >> 2) * M;
            const f32x2 xy = *reinterpret_cast<const f32x2*>(loc_s + qo * 32 + (s & 3) * 2);
            sx[k] = xy[0]; sy[k] = xy[1];
          }
        }
#pragma unroll
        for (int k = 0; k < kDstScan; ++k) {
          if (k >= start && !full && base + k * kDstThreads < ns) {      // (wave-uniform)
            const int s = base + k * kDstThreads + lane;
            const float h_im = sy[k] * fH - 0.5f, w_im = sx[k] * fW - 0.5f;  // (make_sample's expressions)
            if (tile == 0) {                                             // (uniform) samples outside the level: zero gradients, written by the level's first tile
              const bool in_range = (h_im > -1.f) && (w_im > -1.f) && (h_im < fH) && (w_im < fW);
              if (s < ns && !in_range) {
                const int64_t qo = (int64_t)(s >> 2) * M;
                ga_s[qo * 16 + (s & 3)] = 0.f;
                *reinterpret_cast<f32x2*>(gl_s + qo * 32 + (s & 3) * 2) = f32x2{0.f, 0.f};
              }
            }
            const bool match = (h_im >= fy0) && (h_im < fy1) && (w_im >= fx0) && (w_im < fx1);
            const unsigned long long mask = __ballot(match);
            if (mask) {
              const int n = __builtin_popcountll(mask);
              if (cnt + n > kDstRecCap) {
                full = true;
                start = k;
              } else {
                if (match) {
                  const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                  rA[pos] = s; rLh[pos] = sx[k]; rLw[pos] = sy[k];
                }
                cnt += n;
              }
            }
          }
        }
        if (!full) { base += kDstThreads * kDstScan; start = 0; }
      } else {
        more = false;
      }
      if (full || (!more && cnt > 0)) { refine(cnt); process(cnt); cnt = 0; }
    }
    __syncthreads();

    // ---- flush: a half wave per pixel of the tile, sixteen pixels each: read and CLEAR the sums, then all reads of grad_value, then
    // all writes -----------------------------------------------------------------------------------------------------------------------
    {
      constexpr int kPer = kDstTilePx / (kDstThreads / 32);             // 16
      float fv[kPer];
      bool fo[kPer];
#pragma unroll
      for (int i = 0; i < kPer; ++i) {
        const int px = (tid >> 5) + i * (kDstThreads / 32);
        fv[i] = (float)acc[px * 32 + ln];
        acc[px * 32 + ln] = 0.0;
        const unsigned long long any = __ballot(fv[i] != 0.f) >> (tid & 32) & 0xffffffffull;
        fo[i] = any != 0 && y0 + (px >> 4) < Hl && x0 + (px & 15) < Wl;
      }
#pragma unroll
      for (int i = 0; i < kPer; ++i) {
        const int px = (tid >> 5) + i * (kDstThreads / 32);
        if (fo[i]) atomic_add(grad_value + (size_t)(lvl_off + (uint32_t)((y0 + (px >> 4)) * Wl + x0 + (px & 15)) * ps32 + (uint32_t)ln), fv[i]);
      }
    }
    if (ticket && tid == 0) {
      *s_next = (int)drawn;
      if (drawn == (unsigned)total + gridDim.x - 1u) *ticket = 0u;       // the launch's last draw: the counter is ready for the next launch
    }
    __syncthreads();                                                     // the sums are zero again before the next slot adds
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

