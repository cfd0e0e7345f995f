// msda_fwd_tiled -- LDS-tiled MSDeformAttn forward for encoder-style calls (Lq == S: every pixel of the
// pyramid is a query, raster order per level), fp32, D = 32, L <= 4, L*P <= 16.  gfx950 only.
//
// Why: the 4-corner x 16-sample gather moves 8 KB per (query, head) for 128 B of output; through the
// vector L1 (64 B/clk/CU) that alone costs > 70 us per encoder call at the R50 shapes, 3-4x the HBM time of
// the compulsory bytes.  LDS delivers 256 B/clk/CU, so the sampled neighbourhood is staged in LDS:
//
//   work item = (image b, head m, 8x8 tile of level-0 pixels).  The tile's queries are the pixels of EVERY
//   level whose centre falls into the tile's normalised rectangle (a partition of all S queries).  For each
//   level a WH x WW window of head m's value rows (128 B per pixel) around the tile is copied into LDS; the
//   window position follows the mean sampling offset of the tile's own queries (deformable-attention heads
//   look in a preferred direction), measured from the data.  Samples whose corners fall inside a window
//   are read from LDS; the rest (far offsets) take raw buffer loads -- correctness never depends on the
//   window heuristics, and the tile partition does not assume anything about where queries look.
//
//   gather lane mapping (wave64 = 2 pairs x 4 corners x 8 16-byte chunks): ds_read_b128 is served in four
//   16-lane groups {0-3,12-15,20-27} {4-11,16-19,28-31} {32-35,44-47,52-59} {36-43,48-51,60-63}
//   (MI355X_MICROARCH.md, LDS).  Lanes are given roles such that each group reads the two x-adjacent
//   corners of one bilinear row = 256 contiguous bytes = all 64 banks exactly once: conflict-free for any
//   sample position.  Each lane accumulates "its" corner over the 16 samples; the four corners are summed
//   once per query through a 1 KB LDS exchange.  Corner weights and addresses are produced once per sample
//   by a setup role (32 lanes per pair: sample x row), parked in LDS records, re-read by the quad lane that
//   owns the sample (4 samples per lane) and broadcast inside the quad with DPP quad_perm -- so the hot loop
//   per sample is: v_add(dpp) address, ds_read_b128, 4 v_fmac (dpp weight).
//
// A persistent grid (2 workgroups per CU, ~77 KB LDS each) walks the items head-minor, so that (observed
// block->XCD round-robin) XCD x keeps working on head x.  All geometry is derived on the device from the
// int64 shape tensors: the host never needs the level shapes (no sync, graph-capturable).
#include "msda_common.hpp"

namespace msda {

constexpr int kTile = 8;                       // level-0 pixels per tile side
constexpr int kWinSlots = 576;                 // 72 KiB of 128-byte pixel slots for the windows
constexpr int kTiledMaxL = 4;
constexpr int kTiledLP = 16;
constexpr int kMaxTileQ = 128;                 // queries handled per table round
constexpr int kRecBytes = 4 * 1024;            // 1 KiB per wave: sample records, then the corner exchange
constexpr int kWinMargin = 7;

struct TiledMeta {
  int H[kTiledMaxL], W[kTiledMaxL], start[kTiledMaxL];
  int WH[kTiledMaxL], WW[kTiledMaxL], slot[kTiledMaxL];  // window geometry, constant per launch
  int ys[kTiledMaxL], xs[kTiledMaxL], ny[kTiledMaxL], nx[kTiledMaxL], cum[kTiledMaxL + 1];
  float gcy[kTiledMaxL], gcx[kTiledMaxL];                  // tile centre in level coordinates
  float dev[kTiledMaxL][4];                                // sum dy, sum dx, count, unused
  int oy[kTiledMaxL], ox[kTiledMaxL];
  int TY, TX, nq;
  int qtab[kMaxTileQ];
};

constexpr int kTiledLdsBytes = kWinSlots * 128 + kRecBytes + ((sizeof(TiledMeta) + 15) / 16) * 16;

__device__ __forceinline__ int ceil_div_signed(int a, int b) {  // b > 0
  return a >= 0 ? (a + b - 1) / b : -((-a) / b);
}

typedef const f32x4 __attribute__((address_space(3)))* lds_f32x4_ptr;  // 32-bit LDS address, no base add

template <int K>
__device__ __forceinline__ int quad_bcast(int v) {  // value of lane K of the caller's quad
  return __builtin_amdgcn_mov_dpp(v, K * 0x55, 0xF, 0xF, true);
}
template <int K>
__device__ __forceinline__ float quad_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), K * 0x55, 0xF, 0xF, true));
}

__global__ void __launch_bounds__(kBlock, 2)
msda_fwd_tiled(const float* __restrict__ value, const int64_t* __restrict__ shapes,
               const int64_t* __restrict__ lsi, const float* __restrict__ loc,
               const float* __restrict__ attn, Dims d, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* recs_all = smem + kWinSlots * 128;
  TiledMeta& mt = *reinterpret_cast<TiledMeta*>(smem + kWinSlots * 128 + kRecBytes);

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int L = d.L, P = d.P, LP = L * P, M = d.M;
  char* recs = recs_all + wv * 1024;

  // ---- once per launch: level table and window geometry ---------------------------------------
  if (tid == 0) {
    for (int l = 0; l < L; ++l) {
      mt.H[l] = (int)shapes[2 * l];
      mt.W[l] = (int)shapes[2 * l + 1];
      mt.start[l] = (int)lsi[l];
    }
    const int H0 = mt.H[0], W0 = mt.W[0];
    mt.TY = (H0 + kTile - 1) / kTile;
    mt.TX = (W0 + kTile - 1) / kTile;
    int total = 0;
    for (int margin = kWinMargin; margin >= 0; --margin) {
      total = 0;
      for (int l = 0; l < L; ++l) {
        const int ex = (kTile * mt.W[l] + W0 - 1) / W0, ey = (kTile * mt.H[l] + H0 - 1) / H0;
        mt.WW[l] = min(mt.W[l], ex + margin);
        mt.WH[l] = min(mt.H[l], ey + margin);
        total += mt.WW[l] * mt.WH[l];
      }
      if (total <= kWinSlots) break;
    }
    while (total > kWinSlots) {  // odd pyramids: give up the largest window (that level goes through L1/L2)
      int big = 0;
      for (int l = 1; l < L; ++l)
        if (mt.WW[l] * mt.WH[l] > mt.WW[big] * mt.WH[big]) big = l;
      total -= mt.WW[big] * mt.WH[big];
      mt.WW[big] = mt.WH[big] = 0;
    }
    int acc = 0;
    for (int l = 0; l < L; ++l) {
      mt.slot[l] = acc;
      acc += mt.WW[l] * mt.WH[l];
    }
  }
  __syncthreads();

  // ---- per-lane roles ---------------------------------------------------------------------------
  const int half = lane >> 5, t = lane & 31;
  // setup role: sample s_set, bilinear row cy_set of this half's pair
  const int s_set = t >> 1, cy_set = t & 1;
  const bool s_live = s_set < LP;
  int l_set = 0;
  for (int l = 1; l < L; ++l) l_set += (s_set >= l * P) ? 1 : 0;
  const int sH = mt.H[l_set], sW = mt.W[l_set], sStart = mt.start[l_set];
  const int sWH = mt.WH[l_set], sWW = mt.WW[l_set], sSlot = mt.slot[l_set];
  // gather role: corner (cy_g, cx_g), 16-byte chunk of the 128-byte pixel row
  const int quad = t >> 2, k = t & 3;
  const int cy_g = (0x96 >> quad) & 1, cx_g = (0xF0 >> quad) & 1, hf_g = (0xCC >> quad) & 1;
  const uint32_t chunk_off = (uint32_t)(hf_g * 4 + k) * 16u;
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int TY = mt.TY, TX = mt.TX;
  const int items = d.N * M * TY * TX;
  const uint32_t pix_bytes = (uint32_t)M * 128u;

  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int m = item % M;
    const int tile = (item / M) % (TY * TX);
    const int b = item / (M * TY * TX);
    const int ty = tile / TX, tx = tile % TX;

    // ---- P0: the tile's query rectangles per level (threads 0..L-1), then the query table ----
    if (tid < L) {
      const int l = tid;
      const int H0 = mt.H[0], W0 = mt.W[0], Hl = mt.H[l], Wl = mt.W[l];
      // level-l pixel x belongs to tile tx iff floor((2x+1) * W0 / (2 * kTile * Wl)) == tx
      int xs = ceil_div_signed(tx * 2 * kTile * Wl - W0, 2 * W0);
      int xe = ceil_div_signed((tx + 1) * 2 * kTile * Wl - W0, 2 * W0);
      int ys = ceil_div_signed(ty * 2 * kTile * Hl - H0, 2 * H0);
      int ye = ceil_div_signed((ty + 1) * 2 * kTile * Hl - H0, 2 * H0);
      xs = max(0, min(xs, Wl)); xe = max(xs, min(xe, Wl));
      ys = max(0, min(ys, Hl)); ye = max(ys, min(ye, Hl));
      if (tx == TX - 1) xe = Wl;
      if (ty == TY - 1) ye = Hl;
      mt.xs[l] = xs; mt.ys[l] = ys; mt.nx[l] = xe - xs; mt.ny[l] = ye - ys;
      // centre of the tile's normalised rectangle in level-l sample coordinates (x_im = loc * W - 0.5)
      const float x_lo = (float)(tx * kTile) / W0, x_hi = (float)min((tx + 1) * kTile, W0) / W0;
      const float y_lo = (float)(ty * kTile) / H0, y_hi = (float)min((ty + 1) * kTile, H0) / H0;
      mt.gcx[l] = 0.5f * (x_lo + x_hi) * Wl - 0.5f;
      mt.gcy[l] = 0.5f * (y_lo + y_hi) * Hl - 0.5f;
      mt.dev[l][0] = 0.f; mt.dev[l][1] = 0.f; mt.dev[l][2] = 0.f;
    }
    __syncthreads();
    if (tid == 0) {
      int acc = 0;
      for (int l = 0; l < L; ++l) {
        mt.cum[l] = acc;
        acc += mt.nx[l] * mt.ny[l];
      }
      for (int l = L; l <= kTiledMaxL; ++l) mt.cum[l] = acc;
      mt.nq = acc;
    }
    __syncthreads();
    const int nq = mt.nq;

    for (int qbase = 0; qbase < nq; qbase += kMaxTileQ) {
      const int nround = min(kMaxTileQ, nq - qbase);
      if (tid < nround) {
        const int qi = qbase + tid;
        int l = 0;
        for (int ll = 1; ll < L; ++ll) l += (qi >= mt.cum[ll]) ? 1 : 0;
        const int j = qi - mt.cum[l], nx = mt.nx[l];
        const int yy = (int)(((float)j + 0.5f) / (float)nx);
        const int xx = j - yy * nx;
        mt.qtab[tid] = mt.start[l] + (mt.ys[l] + yy) * mt.W[l] + mt.xs[l] + xx;
      }
      __syncthreads();

      // ---- P1: where do this tile's queries look?  mean (clamped) deviation from the tile centre ----
      if (qbase == 0) {
        const int l = tid & 3, i = tid >> 2;  // 64 query slots x 4 levels
        float sy = 0.f, sx = 0.f, sn = 0.f;
        if (l < L) {
          const int qloc = (i * nround) >> 6;
          const int64_t pair = ((int64_t)b * d.Lq + mt.qtab[qloc]) * M + m;
          const float* lp = loc + (pair * LP + l * P) * 2;
          const float Hl = (float)mt.H[l], Wl = (float)mt.W[l], gy = mt.gcy[l], gx = mt.gcx[l];
          for (int p = 0; p < P; ++p) {
            const float x = lp[2 * p] * Wl - 0.5f, y = lp[2 * p + 1] * Hl - 0.5f;
            const float dx = x - gx, dy = y - gy;
            if (fabsf(dx) <= 12.f && fabsf(dy) <= 12.f) {  // ignore far-away points (and NaNs)
              sx += dx; sy += dy; sn += 1.f;
            }
          }
        }
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) {
          sy += __shfl_xor(sy, o, 64);
          sx += __shfl_xor(sx, o, 64);
          sn += __shfl_xor(sn, o, 64);
        }
        if (lane < L) {
          atomicAdd(&mt.dev[lane][0], sy);
          atomicAdd(&mt.dev[lane][1], sx);
          atomicAdd(&mt.dev[lane][2], sn);
        }
        __syncthreads();
        if (tid < L) {
          const int l = tid;
          const float n = fmaxf(mt.dev[l][2], 1.f);
          const float cy = mt.gcy[l] + mt.dev[l][0] / n, cx = mt.gcx[l] + mt.dev[l][1] / n;
          // window covers [o, o + W?) pixels; bilinear touches floor(c) and floor(c)+1 -> centre on c + 0.5
          const int oy = (int)floorf(cy + 1.0f - 0.5f * (float)mt.WH[l]);
          const int ox = (int)floorf(cx + 1.0f - 0.5f * (float)mt.WW[l]);
          mt.oy[l] = max(0, min(oy, mt.H[l] - mt.WH[l]));
          mt.ox[l] = max(0, min(ox, mt.W[l] - mt.WW[l]));
        }
        __syncthreads();

        // ---- P3: copy the windows of head m into LDS (8 lanes x 16 B per pixel) --------------------
        const __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(value) + (int64_t)b * d.S * M * 32, 0, (int)((uint32_t)d.S * pix_bytes), 0x00020000);
        const uint32_t hoff = (uint32_t)m * 128u;
        const int nslots = mt.slot[L - 1] + mt.WW[L - 1] * mt.WH[L - 1];
        const int c8 = tid & 7;
        // all loads of the window copy are issued before the first LDS store (one memory latency per tile):
        // kWinSlots / 32 = 18 pixel slots per 8-lane group, 72 transient VGPRs
        constexpr int kStage = kWinSlots / (kBlock / 8);
        f32x4 sv[kStage];
#pragma unroll
        for (int j = 0; j < kStage; ++j) {
          const int p = (tid >> 3) + j * (kBlock / 8);
          sv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (p < nslots) {
            // the last level whose slot offset is <= p and whose window is non-empty (dropped levels share
            // their slot offset with the successor)
            int l = 0;
            for (int ll = 1; ll < L; ++ll)
              if (p >= mt.slot[ll] && mt.WW[ll] > 0) l = ll;
            const int rel = p - mt.slot[l], ww = mt.WW[l];
            const int r = (int)(((float)rel + 0.5f) / (float)ww);
            const int c = rel - r * ww;
            const uint32_t gpix = (uint32_t)(mt.start[l] + (mt.oy[l] + r) * mt.W[l] + mt.ox[l] + c);
            sv[j] = buffer_load_f32x4(vsrc, gpix * pix_bytes + (uint32_t)c8 * 16u, hoff);
          }
        }
#pragma unroll
        for (int j = 0; j < kStage; ++j) {
          const int p = (tid >> 3) + j * (kBlock / 8);
          if (p < nslots) *reinterpret_cast<f32x4*>(smem + p * 128 + c8 * 16) = sv[j];
        }
        __syncthreads();
      }

      // ---- P4: the tile's (query, head m) pairs, two per wave per iteration ------------------------
      const __amdgpu_buffer_rsrc_t vsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(value) + (int64_t)b * d.S * M * 32, 0, (int)((uint32_t)d.S * pix_bytes), 0x00020000);
      const uint32_t hoff = (uint32_t)m * 128u;
      const int oy = mt.oy[l_set], ox = mt.ox[l_set];
      const int niter = (nround + 7) >> 3;

      auto pair_of = [&](int it) -> int64_t {  // -1 when this half has no query in iteration `it`
        const int qi = it * 8 + wv * 2 + half;
        if (qi >= nround) return -1;
        return ((int64_t)b * d.Lq + mt.qtab[qi]) * M + m;
      };
      float2 cur_loc = make_float2(0.f, 0.f);
      float cur_a = 0.f;
      int64_t cur_pair = pair_of(0);
      if (cur_pair >= 0 && s_live) {
        cur_loc = *reinterpret_cast<const float2*>(loc + (cur_pair * LP + s_set) * 2);
        cur_a = attn[cur_pair * LP + s_set];
      }

      for (int it = 0; it < niter; ++it) {
        // prefetch the next iteration's sampling data
        const int64_t nxt_pair = (it + 1 < niter) ? pair_of(it + 1) : -1;
        float2 nxt_loc = make_float2(0.f, 0.f);
        float nxt_a = 0.f;
        if (nxt_pair >= 0 && s_live) {
          nxt_loc = *reinterpret_cast<const float2*>(loc + (nxt_pair * LP + s_set) * 2);
          nxt_a = attn[nxt_pair * LP + s_set];
        }

        // -- setup role: record {w0, addr0, w1, addr1} of (sample s_set, row cy_set)
        {
          float w0 = 0.f, w1 = 0.f;
          uint32_t a0 = 0u, a1 = 0u;
          if (cur_pair >= 0 && s_live) {
            const float x = cur_loc.x * (float)sW - 0.5f, y = cur_loc.y * (float)sH - 0.5f;
            const bool inr = (y > -1.f) && (x > -1.f) && (y < (float)sH) && (x < (float)sW);
            if (inr) {
              const float yf = floorf(y), xf = floorf(x);
              const float ly = y - yf, lx = x - xf;
              const int yy = (int)yf + cy_set, x0 = (int)xf;
              const float wy = (cy_set ? ly : 1.f - ly) * cur_a;
              const bool rowok = yy >= 0 && yy <= sH - 1;
              const bool ok0 = rowok && x0 >= 0, ok1 = rowok && x0 + 1 <= sW - 1;
              w0 = ok0 ? wy * (1.f - lx) : 0.f;
              w1 = ok1 ? wy * lx : 0.f;
              const int ry = yy - oy, c0 = x0 - ox;
              const bool rowin = (unsigned)ry < (unsigned)sWH;
              const uint32_t lds0 = (uint32_t)(sSlot + ry * sWW + c0) * 128u;
              const uint32_t gpix = (uint32_t)(sStart + yy * sW + x0);
              if (ok0) a0 = (rowin && (unsigned)c0 < (unsigned)sWW) ? lds0 : (0x80000000u | (gpix * pix_bytes));
              if (ok1)
                a1 = (rowin && (unsigned)(c0 + 1) < (unsigned)sWW) ? lds0 + 128u
                                                                    : (0x80000000u | ((gpix + 1u) * pix_bytes));
              // a dead corner next to a live LDS one re-reads the live pixel (same address = broadcast)
              if (!ok0 && ok1 && (int)a1 >= 0) a0 = a1;
              if (!ok1 && ok0 && (int)a0 >= 0) a1 = a0;
            }
          }
          u32x4 rec;
          rec[0] = __float_as_uint(w0); rec[1] = a0; rec[2] = __float_as_uint(w1); rec[3] = a1;
          *reinterpret_cast<u32x4*>(recs + ((half * 16 + s_set) * 2 + cy_set) * 16) = rec;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // -- gather role: own the records of samples k, k+4, k+8, k+12 for corner (cy_g, cx_g)
        float rw[4], gw[4];
        uint32_t ra[4], ga[4];
        bool any_global = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint2 r = *reinterpret_cast<const uint2*>(recs + ((half * 16 + k + 4 * i) * 2 + cy_g) * 16 + cx_g * 8);
          const bool g = (int)r.y < 0;
          rw[i] = g ? 0.f : __uint_as_float(r.x);
          ra[i] = (g ? 0u : r.y) + smem_base;  // the reader adds its own chunk offset after the broadcast
          gw[i] = g ? __uint_as_float(r.x) : 0.f;
          ga[i] = g ? (r.y & 0x7fffffffu) : kOobOffset;
          any_global |= g;
        }

        // far samples first: issue their raw buffer loads now (skipped wave-uniformly per sample when no lane
        // needs one; entries that are not flagged carry an out-of-range offset and weight 0), consume them
        // after the LDS gather so the memory latency overlaps it
        f32x4 fv[16];
        const bool far = __ballot(any_global) != 0;
        if (far) {
#define MSDA_TILED_GLOAD(i, kk)                                                         \
          {                                                                             \
            const uint32_t a = (uint32_t)quad_bcast<kk>((int)ga[i]);                    \
            fv[(i) * 4 + (kk)] = f32x4{0.f, 0.f, 0.f, 0.f};                             \
            if (__ballot(a < kOobOffset)) fv[(i) * 4 + (kk)] = buffer_load_f32x4(vsrc, a + chunk_off, hoff); \
          }
#define MSDA_TILED_GROW(i) MSDA_TILED_GLOAD(i, 0) MSDA_TILED_GLOAD(i, 1) MSDA_TILED_GLOAD(i, 2) MSDA_TILED_GLOAD(i, 3)
          MSDA_TILED_GROW(0) MSDA_TILED_GROW(1) MSDA_TILED_GROW(2) MSDA_TILED_GROW(3)
#undef MSDA_TILED_GROW
#undef MSDA_TILED_GLOAD
        }

        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#define MSDA_TILED_STEP(i, kk)                                                        \
        {                                                                             \
          const uint32_t a = (uint32_t)quad_bcast<kk>((int)ra[i]) + chunk_off;        \
          const float w = quad_bcast<kk>(rw[i]);                                      \
          const f32x4 v = *reinterpret_cast<lds_f32x4_ptr>((uintptr_t)a);             \
          acc[0] = fmaf(w, v[0], acc[0]); acc[1] = fmaf(w, v[1], acc[1]);             \
          acc[2] = fmaf(w, v[2], acc[2]); acc[3] = fmaf(w, v[3], acc[3]);             \
        }
#define MSDA_TILED_ROW(i) MSDA_TILED_STEP(i, 0) MSDA_TILED_STEP(i, 1) MSDA_TILED_STEP(i, 2) MSDA_TILED_STEP(i, 3)
        MSDA_TILED_ROW(0) MSDA_TILED_ROW(1) MSDA_TILED_ROW(2) MSDA_TILED_ROW(3)
#undef MSDA_TILED_ROW
#undef MSDA_TILED_STEP

        if (far) {
#define MSDA_TILED_GSTEP(i, kk)                                                       \
          {                                                                           \
            const float w = quad_bcast<kk>(gw[i]);                                    \
            const f32x4 v = fv[(i) * 4 + (kk)];                                       \
            acc[0] = fmaf(w, v[0], acc[0]); acc[1] = fmaf(w, v[1], acc[1]);           \
            acc[2] = fmaf(w, v[2], acc[2]); acc[3] = fmaf(w, v[3], acc[3]);           \
          }
#define MSDA_TILED_GROW(i) MSDA_TILED_GSTEP(i, 0) MSDA_TILED_GSTEP(i, 1) MSDA_TILED_GSTEP(i, 2) MSDA_TILED_GSTEP(i, 3)
          MSDA_TILED_GROW(0) MSDA_TILED_GROW(1) MSDA_TILED_GROW(2) MSDA_TILED_GROW(3)
#undef MSDA_TILED_GROW
#undef MSDA_TILED_GSTEP
        }

        // -- sum the four corners through LDS (records are dead now) and store 128 B per pair
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        *reinterpret_cast<f32x4*>(recs + ((half * 4 + cy_g * 2 + cx_g) * 8 + hf_g * 4 + k) * 16) = acc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (t < 8 && cur_pair >= 0) {
          const char* e = recs + (half * 4 * 8 + t) * 16;
          const f32x4 s0 = *reinterpret_cast<const f32x4*>(e), s1 = *reinterpret_cast<const f32x4*>(e + 128);
          const f32x4 s2 = *reinterpret_cast<const f32x4*>(e + 256), s3 = *reinterpret_cast<const f32x4*>(e + 384);
          *reinterpret_cast<f32x4*>(out + cur_pair * 32 + t * 4) = (s0 + s1) + (s2 + s3);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        cur_pair = nxt_pair; cur_loc = nxt_loc; cur_a = nxt_a;
      }
      __syncthreads();  // windows / table are rewritten by the next round or item
    }
  }
}

// Host side ---------------------------------------------------------------------------------------
bool tiled_forward_ok(const Dims& d) {
  return d.D == 32 && d.L <= kTiledMaxL && d.L * d.P <= kTiledLP && d.Lq == d.S &&
         (int64_t)d.S * d.M * 128 < (int64_t)kOobOffset && (int64_t)d.N * d.Lq * d.M < ((int64_t)1 << 40);
}

int launch_forward_tiled(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc,
                         const float* attn, const Dims& d, float* out, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {  // > 64 KiB of dynamic LDS needs the opt-in; idempotent, not a stream operation
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(msda_fwd_tiled),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kTiledLdsBytes);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  // persistent grid: 2 workgroups per CU on 256 CUs, a multiple of 8 so that item % M tracks blockIdx % 8
  const unsigned grid = 512;
  hipLaunchKernelGGL(msda_fwd_tiled, dim3(grid), dim3(kBlock), kTiledLdsBytes, stream, value, shapes, lsi, loc, attn, d,
                     out);
  return (int)hipGetLastError();
}

}  // namespace msda
