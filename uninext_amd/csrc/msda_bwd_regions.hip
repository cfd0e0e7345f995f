// msda_bwd_regions -- MSDeformAttn backward for encoder-shaped calls whose samples do NOT stay near their queries.
// fp32, D = 32, L = P = 4.  gfx950 only.  Replaces, for these calls, the work of ops/src/cuda/ms_deform_im2col_cuda.cuh:
// 301-403 (the col2im kernels) -- same arithmetic per corner (cuh:87-159), a different owner for every sum.
//
// Why: every other backward kernel here is organised around the QUERY: what lands near the query's tile is combined in LDS,
// the rest leaves as full-line L2 atomics, one per bilinear corner, at ~11 G lines/s.  At far fractions of 0.4 / 0.9 that is
// 9 / 21 M atomics = 0.6 / 1.8 ms of a 0.9 / 2.1 ms launch (profiles/r03_backward_window.txt), and no window that fits the
// LDS contains sigma = 6 px offsets.  Here the DESTINATION owns the sum:
//
//   regions   = every level is cut into regions of 16x16 / 8x16 / 4x8 / 2x4 pixels (halved from level to level: each level
//               receives a quarter of the samples, so the bins come out even -- profiles/r03_backward_region_bins.txt).
//               bin = (image, head, level, region).
//   filing    = msda_bwd_regions_file<false> counts, per bin, the samples with an in-image corner in the region (a sample
//               lands in 1, 2 or 4 bins: the regions of its valid corner rows x those of its valid corner columns);
//               msda_bwd_regions_scan turns the counts into bin starts; msda_bwd_regions_file<true> writes one 32-byte
//               record (query, where the sample sits in the region, its four corner weights) per (sample, bin).  Both file kernels combine in LDS first: a
//               workgroup's 256 queries of one head touch few bins, and each bin costs it ONE global atomic.
//   gradients = grad_sampling_loc / grad_attn_weight come from msda_bwd_tiled's query-side pass with everything that
//               concerns grad_value compiled out (msda_bwd_tiled_nogv).
//   regions   = msda_bwd_regions_add: a workgroup per bin reads the bin's records in order, a half wave per record (lane =
//               channel: one coalesced 128-byte row of grad_output), adds the corners that lie INSIDE the region into a
//               float64 LDS tile (ds_add_f64 is native on gfx950, no fixed-point scale, no bound to respect) and ADDS the
//               tile to grad_value with a plain read-modify-write: every pixel is touched exactly once, by its owner -- no
//               global atomic (the C ABI accumulates into grad_value: include/msda_hip.h).
//
// The result does not depend on where samples fall, only the record count does.  tools/proto/owner_computes_ref.py is the
// numpy restatement the filing rule was checked with.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <mutex>

#include "msda_common.hpp"

namespace msda {
namespace {

constexpr int kRT = 256;                 // threads per workgroup of every kernel here
constexpr int kHistCap = 4096;           // bins of one head that the file kernels combine in LDS (more: straight global atomics)
constexpr int kTilePixels = 256;         // largest region: 16 x 16 pixels

// One (sample, bin) record: the query, where the sample's top-left corner sits in the region (pixel offset, may be negative:
// the corner itself is then outside), which of its four corners the region owns, and their weights with the attention weight
// folded in -- everything msda_bwd_regions_add needs besides the query's grad_output row.
struct __attribute__((aligned(32))) Rec {
  uint32_t q;
  uint32_t pk;        // bits 0..15: top-left pixel offset in the region + kOffBias; bits 16..19: corners inside (TL, TR, BL, BR)
  float w[4];         // (1 - lh)(1 - lw), (1 - lh) lw, lh (1 - lw), lh lw  --  each x attention weight   (cuh:113-158)
  uint32_t pad[2];
};
constexpr int kOffBias = 64;

// regions of a 4-level pyramid: sizes 16x16, 8x16, 4x8, 2x4 (log2: rows 4 - l, columns 4, 4, 3, 2)
struct Geom {
  int H[4], W[4], RY[4], RX[4], base[5];
};
__device__ __forceinline__ int row_shift(int l) { return 4 - l; }
__device__ __forceinline__ int col_shift(int l) { return l == 0 ? 4 : 5 - l; }
__device__ __forceinline__ void load_geom(const int64_t* __restrict__ shapes, Geom& g) {
  int acc = 0;
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    g.H[l] = (int)shapes[2 * l];
    g.W[l] = (int)shapes[2 * l + 1];
    g.RY[l] = (g.H[l] + (1 << row_shift(l)) - 1) >> row_shift(l);
    g.RX[l] = (g.W[l] + (1 << col_shift(l)) - 1) >> col_shift(l);
    g.base[l] = acc;
    acc += g.RY[l] * g.RX[l];
  }
  g.base[4] = acc;
}

// the bins (relative to the head's first one) of one in-range sample at (x, y) of level l: regions of its valid corner rows x
// regions of its valid corner columns (cuh:38-46: a corner is valid iff its row and its column are inside the image)
__device__ __forceinline__ int sample_bins(const Geom& g, int l, int y0, int x0, int (&bins)[4], int (&rys)[4], int (&rxs)[4]) {
  int rA = y0 >= 0 ? y0 >> row_shift(l) : -1, rB = y0 + 1 <= g.H[l] - 1 ? (y0 + 1) >> row_shift(l) : -1;
  int cA = x0 >= 0 ? x0 >> col_shift(l) : -1, cB = x0 + 1 <= g.W[l] - 1 ? (x0 + 1) >> col_shift(l) : -1;
  if (rA < 0) { rA = rB; rB = -1; }
  if (cA < 0) { cA = cB; cB = -1; }
  if (rB == rA) rB = -1;
  if (cB == cA) cB = -1;
  int n = 0;
  const int first = g.base[l], RX = g.RX[l];
  if (rA >= 0 && cA >= 0) { bins[n] = first + rA * RX + cA; rys[n] = rA; rxs[n++] = cA; }
  if (rA >= 0 && cB >= 0) { bins[n] = first + rA * RX + cB; rys[n] = rA; rxs[n++] = cB; }
  if (rB >= 0 && cA >= 0) { bins[n] = first + rB * RX + cA; rys[n] = rB; rxs[n++] = cA; }
  if (rB >= 0 && cB >= 0) { bins[n] = first + rB * RX + cB; rys[n] = rB; rxs[n++] = cB; }
  return n;
}

// WRITE = false: counts[bin] += samples filed under the bin.  WRITE = true: the records, at starts[bin] + a slot taken from
// cursors[bin].  Grid (ceil(Lq / 256), M, N): a workgroup = 256 consecutive queries of one head of one image.
template <bool WRITE>
__global__ void __launch_bounds__(kRT)
msda_bwd_regions_file(const int64_t* __restrict__ shapes, const float* __restrict__ loc, const float* __restrict__ attn, Dims d,
                      int hist_cap, uint32_t* __restrict__ counts, const uint32_t* __restrict__ starts,
                      uint32_t* __restrict__ cursors, Rec* __restrict__ recs) {
  __shared__ uint32_t hist[kHistCap];                      // per bin of the head: this workgroup's records; then its local cursor
  __shared__ uint32_t basep[WRITE ? kHistCap : 1];         // WRITE: first slot of this workgroup's records in the bin
  Geom g;
  load_geom(shapes, g);
  const int RL = g.base[4];
  const bool fast = RL <= hist_cap;                          // (hist_cap <= kHistCap; smaller in tests of the other path)
  const int tid = threadIdx.x, m = blockIdx.y, b = blockIdx.z;
  const int q = blockIdx.x * kRT + tid;
  const uint32_t head = (uint32_t)(b * d.M + m) * (uint32_t)RL;
  if (fast) {
    for (int i = tid; i < RL; i += kRT) hist[i] = 0u;
    __syncthreads();
  }
  const float* const lp = loc + (((int64_t)b * d.Lq + q) * d.M + m) * 32;    // [4 levels][4 points][x, y]
  const float* const ap = attn + (((int64_t)b * d.Lq + q) * d.M + m) * 16;

  // sweep: every (sample, bin) of this thread's query; `emit(local bin, record)`
  auto sweep = [&](auto&& emit) __attribute__((always_inline)) {
    if (q >= d.Lq) return;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const float4 la = *reinterpret_cast<const float4*>(lp + 8 * l), lb = *reinterpret_cast<const float4*>(lp + 8 * l + 4);
      const float4 aw = *reinterpret_cast<const float4*>(ap + 4 * l);
      const float lx[4] = {la.x, la.z, lb.x, lb.z}, ly[4] = {la.y, la.w, lb.y, lb.w}, a4[4] = {aw.x, aw.y, aw.z, aw.w};
      const float fW = (float)g.W[l], fH = (float)g.H[l];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float x = lx[p] * fW - 0.5f, y = ly[p] * fH - 0.5f;            // cuh:282-288 / :303-306
        if (!(y > -1.f && x > -1.f && y < fH && x < fW)) continue;
        const float yf = floorf(y), xf = floorf(x);
        const int y0 = (int)yf, x0 = (int)xf;
        int bins[4], rys[4], rxs[4];
        const int n = sample_bins(g, l, y0, x0, bins, rys, rxs);
        Rec r;
        r.q = (uint32_t)q;
        r.pad[0] = r.pad[1] = 0u;
        if constexpr (WRITE) {
          const float lh = y - yf, lw = x - xf, hh = 1.f - lh, hw = 1.f - lw, a = a4[p];
          r.w[0] = hh * hw * a; r.w[1] = hh * lw * a; r.w[2] = lh * hw * a; r.w[3] = lh * lw * a;
        }
        const int rs = row_shift(l), cs = col_shift(l);
        for (int k = 0; k < n; ++k) {
          if constexpr (WRITE) {
            const int yr = y0 - (rys[k] << rs), xr = x0 - (rxs[k] << cs);     // top-left corner relative to the region: -1 .. size - 1
            // a corner the region owns: inside the region AND inside the image (past the last row / column in an edge region)
            const bool t_in = (unsigned)yr < (1u << rs), b_in = (unsigned)(yr + 1) < (1u << rs) && y0 + 1 <= g.H[l] - 1;
            const bool l_in = (unsigned)xr < (1u << cs), r_in = (unsigned)(xr + 1) < (1u << cs) && x0 + 1 <= g.W[l] - 1;
            const uint32_t mask = (t_in && l_in ? 1u : 0u) | (t_in && r_in ? 2u : 0u) | (b_in && l_in ? 4u : 0u) | (b_in && r_in ? 8u : 0u);
            r.pk = (uint32_t)(yr * (1 << cs) + xr + kOffBias) | (mask << 16);
          }
          emit(bins[k], r);
        }
      }
    }
  };

  if (!fast) {                                               // a head with more bins than the LDS table: one global atomic per record
    sweep([&](int bin, const Rec& r) __attribute__((always_inline)) {
      if constexpr (WRITE) {
        const uint32_t slot = atomicAdd(&cursors[head + bin], 1u);
        recs[starts[head + bin] + slot] = r;
      } else {
        atomicAdd(&counts[head + bin], 1u);
      }
    });
    return;
  }
  sweep([&](int bin, const Rec&) __attribute__((always_inline)) { atomicAdd(&hist[bin], 1u); });
  __syncthreads();
  for (int i = tid; i < RL; i += kRT) {
    const uint32_t n = hist[i];
    if constexpr (WRITE) {
      basep[i] = n ? starts[head + i] + atomicAdd(&cursors[head + i], n) : 0u;
      hist[i] = 0u;
    } else {
      if (n) atomicAdd(&counts[head + i], n);
    }
  }
  if constexpr (WRITE) {
    __syncthreads();
    sweep([&](int bin, const Rec& r) __attribute__((always_inline)) {
      const uint32_t slot = atomicAdd(&hist[bin], 1u);
      recs[basep[bin] + slot] = r;
    });
  }
}

// starts = exclusive scan of counts over the N * M * RL bins (+ the total behind them); counts and cursors are left at zero
// for the file kernel / the next call.  One workgroup of 1024 threads.
__global__ void __launch_bounds__(1024)
msda_bwd_regions_scan(const int64_t* __restrict__ shapes, Dims d, uint32_t* __restrict__ counts, uint32_t* __restrict__ starts,
                      uint32_t* __restrict__ cursors) {
  __shared__ uint32_t part[1024];
  Geom g;
  load_geom(shapes, g);
  const int nbins = d.N * d.M * g.base[4];
  const int tid = threadIdx.x, per = (nbins + 1023) / 1024;
  const int lo = min(tid * per, nbins), hi = min(lo + per, nbins);
  uint32_t s = 0;
  for (int i = lo; i < hi; ++i) s += counts[i];
  part[tid] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {                       // inclusive scan of the partial sums
    const uint32_t v = tid >= o ? part[tid - o] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = part[tid] - s;
  for (int i = lo; i < hi; ++i) {
    const uint32_t c = counts[i];
    starts[i] = run;
    run += c;
    counts[i] = 0u;
    cursors[i] = 0u;
  }
  if (tid == 1023) starts[nbins] = part[1023];
}

// One bin at a time per workgroup (persistent grid): records -> float64 LDS tile -> plain stores.  A half wave per record (lane
// = channel: the 32 lanes add 256 contiguous bytes, every LDS bank once), kAddUnroll records of a half wave in flight -- the
// records are read in order but their grad_output rows are a gather, and one row at a time per half wave is what a first
// version spent 1.6 of its 2.0 ms on.
constexpr int MSDA_REGIONS_UNROLL = 4;
constexpr int kAddT = 1024, kAddHalfWaves = kAddT / 32, kAddUnroll = MSDA_REGIONS_UNROLL;
__global__ void __launch_bounds__(kAddT)
msda_bwd_regions_add(const float* __restrict__ grad_out, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                     Dims d, const uint32_t* __restrict__ starts, const Rec* __restrict__ recs, float* __restrict__ grad_value) {
  __shared__ double tile[kTilePixels * 32];                  // [pixel of the region][channel]: 64 KB
  Geom g;
  load_geom(shapes, g);
  const int RL = g.base[4], nbins = d.N * d.M * RL;
  const int tid = threadIdx.x, c = tid & 31, hw = tid >> 5;
  for (int bin = blockIdx.x; bin < nbins; bin += gridDim.x) {
    const int bm = bin / RL, rel = bin - bm * RL;
    const int b = bm / d.M, m = bm - b * d.M;
    const int l = (rel >= g.base[1] ? 1 : 0) + (rel >= g.base[2] ? 1 : 0) + (rel >= g.base[3] ? 1 : 0);
    const int H = l == 0 ? g.H[0] : l == 1 ? g.H[1] : l == 2 ? g.H[2] : g.H[3], W = l == 0 ? g.W[0] : l == 1 ? g.W[1] : l == 2 ? g.W[2] : g.W[3];
    const int RX = l == 0 ? g.RX[0] : l == 1 ? g.RX[1] : l == 2 ? g.RX[2] : g.RX[3];
    const int first = l == 0 ? g.base[0] : l == 1 ? g.base[1] : l == 2 ? g.base[2] : g.base[3];
    const int rs = row_shift(l), cs = col_shift(l);
    const int ry = (rel - first) / RX, rx = (rel - first) - ry * RX;
    const int y_lo = ry << rs, x_lo = rx << cs;              // first pixel of the region
    const int npx = 1 << (rs + cs);
    for (int i = tid; i < npx * 32; i += kAddT) tile[i] = 0.0;
    __syncthreads();
    const uint32_t r0 = starts[bin], r1 = starts[bin + 1];
    const float* const go_head = grad_out + ((int64_t)b * d.Lq * d.M + m) * 32 + c;
    const int row = 32 << cs;                                  // doubles from a tile row to the next
    // two loads deep: the queries of the NEXT iteration's records travel with this iteration's grad_output rows and weights,
    // so that an iteration costs one memory latency, not two (record, then the row it names)
    uint32_t qn[kAddUnroll];
    auto load_queries = [&](uint32_t base) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < kAddUnroll; ++k) {
        const uint32_t i = base + (uint32_t)(k * kAddHalfWaves + hw);
        qn[k] = i < r1 ? recs[i].q : 0xffffffffu;
      }
    };
    load_queries(r0);
    for (uint32_t base = r0; base < r1; base += kAddHalfWaves * kAddUnroll) {
      uint32_t pk[kAddUnroll];
      float w[kAddUnroll][4], gq[kAddUnroll];
#pragma unroll
      for (int k = 0; k < kAddUnroll; ++k) {
        const uint32_t i = base + (uint32_t)(k * kAddHalfWaves + hw);
        const bool have = qn[k] != 0xffffffffu;
        gq[k] = have ? go_head[(int64_t)qn[k] * d.M * 32] : 0.f;
        const uint4 ra = have ? *reinterpret_cast<const uint4*>(&recs[i]) : make_uint4(0u, 0u, 0u, 0u);   // mask 0: nothing to add
        const uint2 rb = have ? *reinterpret_cast<const uint2*>(&recs[i].w[2]) : make_uint2(0u, 0u);
        pk[k] = ra.y;
        w[k][0] = __uint_as_float(ra.z); w[k][1] = __uint_as_float(ra.w); w[k][2] = __uint_as_float(rb.x); w[k][3] = __uint_as_float(rb.y);
      }
      load_queries(base + kAddHalfWaves * kAddUnroll);
#pragma unroll
      for (int k = 0; k < kAddUnroll; ++k) {
        double* const t00 = tile + ((int)(pk[k] & 0xffffu) - kOffBias) * 32 + c;
        const float gk = gq[k];
        if (pk[k] & (1u << 16)) __hip_atomic_fetch_add(t00, (double)(w[k][0] * gk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (pk[k] & (2u << 16)) __hip_atomic_fetch_add(t00 + 32, (double)(w[k][1] * gk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (pk[k] & (4u << 16)) __hip_atomic_fetch_add(t00 + row, (double)(w[k][2] * gk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (pk[k] & (8u << 16)) __hip_atomic_fetch_add(t00 + row + 32, (double)(w[k][3] * gk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    __syncthreads();
    const int64_t lvl_first = lsi[l];
    float* const gv_head = grad_value + ((int64_t)b * d.S * d.M + m) * 32 + c;
    for (int p = hw; p < npx; p += kAddHalfWaves) {
      const int cy = y_lo + (p >> cs), cx = x_lo + (p & ((1 << cs) - 1));
      const int64_t pix = lvl_first + (int64_t)cy * W + cx;
      // (+=: the C ABI ACCUMULATES into grad_value, include/msda_hip.h, as every other kernel here and the reference's atomics do;
      // this workgroup owns the pixel, so a plain read-modify-write is the accumulation)
      if (cy < H && cx < W && pix < d.S) gv_head[pix * d.M * 32] += (float)tile[p * 32 + c];
    }
    __syncthreads();                                           // the tile is zeroed for the next bin
  }
}

// ---- workspace: per device, grown on demand, handed from stream to stream behind an event ---------------------------------
struct Workspace {
  std::mutex mu;
  char* buf = nullptr;
  size_t bytes = 0;
  size_t table_bytes = 0;          // capacity of each of the three bin tables; the layout depends on the CAPACITY, never on the call
  hipEvent_t done = nullptr;
  hipStream_t last = nullptr;
  bool used = false;
};
constexpr int kMaxDevices = 64;
Workspace g_ws[kMaxDevices];

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// bins <= N * M * S (a region holds at least one pixel); records <= 4 per sample
inline size_t table_bytes(const Dims& d) { return align256(((size_t)d.N * d.M * d.S + 1) * 4); }
inline size_t record_bytes(const Dims& d) { return align256((size_t)d.N * d.Lq * d.M * 16 * 4 * sizeof(Rec)); }

// a workspace the CALLER lent to the next backward call of this thread (msda_hip_backward_ws_f32): consumed by that call
thread_local void* t_call_ws = nullptr;
thread_local size_t t_call_ws_bytes = 0;

}  // namespace

size_t regions_workspace_bytes(const Dims& d) {
  return regions_backward_ok(d) ? 3 * table_bytes(d) + record_bytes(d) : 0;
}
void set_call_workspace(void* p, size_t bytes) { t_call_ws = p; t_call_ws_bytes = bytes; }

bool regions_backward_ok(const Dims& d) {
  // encoder-shaped (the query-side pass is msda_bwd_tiled's), 4 levels; record slots and bins are 32-bit
  return tiled_backward_ok(d) && d.L == 4 && d.P == 4 && (int64_t)d.N * d.Lq * d.M * 64 < (int64_t)1 << 32 &&
         (int64_t)d.N * d.M * d.S < (int64_t)1 << 28;
}

int launch_backward_regions(const float* grad_out, const float* value, const int64_t* shapes, const int64_t* lsi,
                            const float* loc, const float* attn, const Dims& d, float* grad_value, float* grad_loc,
                            float* grad_attn, hipStream_t stream, const char** kernel_name) {
  // *kernel_name: the kernel that actually ran -- msda_bwd_tiled when no workspace could be had (stream capture, allocation
  // failure): the caller's diagnostic must not say msda_bwd_regions then
  auto tiled_instead = [&]() {
    if (kernel_name) *kernel_name = "msda_bwd_tiled";
    return launch_backward_tiled(grad_out, value, shapes, lsi, loc, attn, d, grad_value, grad_loc, grad_attn, stream);
  };
  if (kernel_name) *kernel_name = "msda_bwd_regions";
  static const int hist_cap = std::max(0, std::min(ab_env_int("MSDA_BWD_REGIONS_HIST", kHistCap), kHistCap));   // (the path without the LDS table)
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= kMaxDevices) return (int)hipErrorInvalidDevice;
  const size_t b_counts = table_bytes(d), b_recs = record_bytes(d);
  void* const lent = t_call_ws;
  const size_t lent_bytes = t_call_ws_bytes;
  t_call_ws = nullptr;
  t_call_ws_bytes = 0;
  const bool own = !(lent && lent_bytes >= 3 * b_counts + b_recs && ((uintptr_t)lent & 255) == 0);
  if (own) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) (void)hipGetLastError();
    if (cap != hipStreamCaptureStatusNone)   // no allocation and no cross-stream hand-over inside a capture
      return tiled_instead();
  }
  Workspace& ws = g_ws[dev];
  std::unique_lock<std::mutex> lock(ws.mu, std::defer_lock);
  char* base;
  size_t tb;
  if (!own) {
    // the caller's buffer (stream-ordered by the caller's allocator): tables laid out by this call's sizes, nothing survives
    base = static_cast<char*>(lent);
    tb = b_counts;
  } else {
    lock.lock();
    if (!ws.done && hipEventCreateWithFlags(&ws.done, hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
    // The tables sit at offsets that depend on the workspace's CAPACITY (a first version laid them out by the CALL's sizes: a
    // smaller call after a larger one then found its `counts` inside the larger call's tables, not zero, and its records went
    // wherever the garbage starts pointed -- found by the named-workload suite, which changes shapes from test to test).
    if (ws.table_bytes < b_counts || ws.bytes < 3 * ws.table_bytes + b_recs) {
      const size_t ntb = std::max(ws.table_bytes, b_counts), rb = std::max(ws.bytes > 3 * ws.table_bytes ? ws.bytes - 3 * ws.table_bytes : 0, b_recs);
      if (ws.buf) {
        if (hipError_t e = hipDeviceSynchronize(); e != hipSuccess) return (int)e;   // (grows a few times per process at most)
        (void)hipFree(ws.buf);
        ws.buf = nullptr;
        ws.bytes = ws.table_bytes = 0;
      }
      if (hipMalloc(reinterpret_cast<void**>(&ws.buf), 3 * ntb + rb) != hipSuccess) {
        // ~0.7 GB outside the caller's allocator: when the device cannot spare it the step must not die of a bare hip error
        // code mid-training -- the query-centric kernel needs no workspace (slower on these patterns, same result)
        (void)hipGetLastError();
        ws.buf = nullptr;
        lock.unlock();
        return tiled_instead();
      }
      ws.bytes = 3 * ntb + rb;
      ws.table_bytes = ntb;
      ws.used = false;
    } else if (ws.used && ws.last != stream) {
      if (hipError_t e = hipStreamWaitEvent(stream, ws.done, 0); e != hipSuccess) return (int)e;
    }
    base = ws.buf;
    tb = ws.table_bytes;
  }
  uint32_t* const counts = reinterpret_cast<uint32_t*>(base);
  uint32_t* const starts = reinterpret_cast<uint32_t*>(base + tb);
  uint32_t* const cursors = reinterpret_cast<uint32_t*>(base + 2 * tb);
  Rec* const recs = reinterpret_cast<Rec*>(base + 3 * tb);
  // counts of this call's bins start at zero whatever an earlier (possibly aborted) call left: 1.4 MB, ~2 us
  if (hipError_t e = hipMemsetAsync(counts, 0, b_counts, stream); e != hipSuccess) return (int)e;

  // the query side: msda_bwd_q (round 4; MSDA_BWD_REGIONS_Q=0: msda_bwd_tiled with its grad_value half compiled out, A/B)
  static const bool q_pass = ab_env_int("MSDA_BWD_REGIONS_Q", 1) != 0;
  if (q_pass && q_backward_ok(d)) {
    if (int rc = launch_backward_q(grad_out, value, shapes, lsi, loc, attn, d, grad_loc, grad_attn, stream)) return rc;
  } else if (int rc = launch_backward_tiled_nogv(grad_out, value, shapes, lsi, loc, attn, d, grad_loc, grad_attn, stream)) {
    return rc;
  }
  const dim3 fgrid((unsigned)((d.Lq + kRT - 1) / kRT), (unsigned)d.M, (unsigned)d.N);
  hipLaunchKernelGGL(msda_bwd_regions_file<false>, fgrid, dim3(kRT), 0, stream, shapes, loc, attn, d, hist_cap, counts,
                     static_cast<const uint32_t*>(starts), cursors, recs);
  hipLaunchKernelGGL(msda_bwd_regions_scan, dim3(1), dim3(1024), 0, stream, shapes, d, counts, starts, cursors);
  hipLaunchKernelGGL(msda_bwd_regions_file<true>, fgrid, dim3(kRT), 0, stream, shapes, loc, attn, d, hist_cap, counts,
                     static_cast<const uint32_t*>(starts), cursors, recs);
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  hipLaunchKernelGGL(msda_bwd_regions_add, dim3((unsigned)(2 * cus)), dim3(kAddT), 0, stream, grad_out, shapes, lsi, d,
                     static_cast<const uint32_t*>(starts), static_cast<const Rec*>(recs), grad_value);
  const int rc = (int)hipGetLastError();
  if (rc == 0 && own) {
    (void)hipEventRecord(ws.done, stream);
    ws.last = stream;
    ws.used = true;
  }
  return rc;
}

}  // namespace msda
