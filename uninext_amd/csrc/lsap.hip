// Linear sum assignment on the device with SciPy's exact tie behaviour -- see include/lsap_hip.h and
// oracle/lsap_oracle.py (the same algorithm, sequential, pinned against SciPy).
//
// One 1024-thread workgroup per problem.  The matrix is converted once to float64 in the algorithm's orientation
// (rows <= cols; a tall matrix is transposed) into the workspace, so every later scan is coalesced.  Per row the
// shortest augmenting path is grown step by step: the scan of the remaining columns (update of the tentative path
// costs, selection of the next column) is a parallel loop + one workgroup reduction whose comparator reproduces the
// sequential rule "lowest cost; among equals an unassigned column, the LAST unassigned one in scan order; otherwise
// the FIRST in scan order"; the swap-removal from the `remaining` list, the dual update and the augmentation follow
// the reference implementation literally.  No multiplications anywhere: the float64 sums are the ones SciPy forms.
#include "../../include/lsap_hip.h"

#include <cmath>

#include "msda_common.hpp"

namespace lsap {

constexpr int kThreads = 1024;

struct Problem {
  const float* cost;
  long long ld;
  int rows, cols;
  int64_t* row_ind;
  int64_t* col_ind;
  char* ws;
  int32_t* status;
};
struct Batch {
  Problem p[LSAP_HIP_MAX_BATCH];
};

struct Cand {
  double s;
  int it;       // position in `remaining`; -1 = none
  int unassigned;
};
__device__ __forceinline__ bool better(const Cand& a, const Cand& b) {   // is a preferred over b
  if (b.it < 0) return a.it >= 0;
  if (a.it < 0) return false;
  if (a.s != b.s) return a.s < b.s;
  if (a.unassigned != b.unassigned) return a.unassigned != 0;
  return a.unassigned ? a.it > b.it : a.it < b.it;
}

__host__ __device__ inline size_t align8(size_t x) { return (x + 7) & ~(size_t)7; }
// workspace layout (nr <= nc after the optional transpose)
struct Layout {
  size_t costd, u, v, spc, path, row4col, remaining, removed, col4row, sr, total;
  __host__ __device__ Layout(int nr, int nc) {
    size_t o = 0;
    costd = o; o += (size_t)nr * nc * 8;
    u = o; o += align8((size_t)nr * 8);
    v = o; o += (size_t)nc * 8;
    spc = o; o += (size_t)nc * 8;
    path = o; o += align8((size_t)nc * 4);
    row4col = o; o += align8((size_t)nc * 4);
    remaining = o; o += align8((size_t)nc * 4);
    removed = o; o += align8((size_t)nc * 4);
    col4row = o; o += align8((size_t)nr * 4);
    sr = o; o += align8((size_t)nr);
    total = o;
  }
};

__global__ void __launch_bounds__(kThreads)
lsap_kernel(Batch batch) {
  const Problem pr = batch.p[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool transpose = pr.cols < pr.rows;
  const int nr = transpose ? pr.cols : pr.rows, nc = transpose ? pr.rows : pr.cols;
  const Layout L(nr, nc);
  double* costd = reinterpret_cast<double*>(pr.ws + L.costd);
  double* u = reinterpret_cast<double*>(pr.ws + L.u);
  double* v = reinterpret_cast<double*>(pr.ws + L.v);
  double* spc = reinterpret_cast<double*>(pr.ws + L.spc);
  int* path = reinterpret_cast<int*>(pr.ws + L.path);
  int* row4col = reinterpret_cast<int*>(pr.ws + L.row4col);
  int* remaining = reinterpret_cast<int*>(pr.ws + L.remaining);
  int* removed = reinterpret_cast<int*>(pr.ws + L.removed);
  int* col4row = reinterpret_cast<int*>(pr.ws + L.col4row);
  unsigned char* SR = reinterpret_cast<unsigned char*>(pr.ws + L.sr);

  __shared__ Cand red[kThreads / 64];
  __shared__ int sh_bad, sh_i, sh_sink, sh_nrem, sh_nremoved;
  __shared__ double sh_min;

  // ---- float64 copy in the algorithm's orientation; reject NaN / -inf as SciPy does ---------------------------
  if (tid == 0) sh_bad = 0;
  __syncthreads();
  {
    int bad = 0;
    const long long total = (long long)nr * nc;
    if (!transpose) {
      for (long long e = tid; e < total; e += kThreads) {
        const int i = (int)(e / nc), j = (int)(e - (long long)i * nc);
        const double c = (double)pr.cost[i * pr.ld + j];
        bad |= (c != c) || (c == -INFINITY);
        costd[e] = c;
      }
    } else {   // internal (i, j) = original (j, i): read along the original rows (coalesced), write transposed
      for (long long e = tid; e < total; e += kThreads) {
        const int jo = (int)(e / nr), io = (int)(e - (long long)jo * nr);   // original row jo, original column io
        const double c = (double)pr.cost[jo * pr.ld + io];
        bad |= (c != c) || (c == -INFINITY);
        costd[(long long)io * nc + jo] = c;
      }
    }
    if (bad) sh_bad = 1;
  }
  for (int i = tid; i < nr; i += kThreads) { u[i] = 0.0; col4row[i] = -1; }
  for (int j = tid; j < nc; j += kThreads) { v[j] = 0.0; row4col[j] = -1; path[j] = -1; }
  __syncthreads();
  if (sh_bad) {
    if (tid == 0) *pr.status = 1;
    return;
  }

  for (int cur = 0; cur < nr; ++cur) {
    // ---- shortest augmenting path from row `cur` -----------------------------------------------------------
    for (int j = tid; j < nc; j += kThreads) { remaining[j] = nc - 1 - j; spc[j] = INFINITY; }
    for (int i = tid; i < nr; i += kThreads) SR[i] = 0;
    if (tid == 0) { sh_i = cur; sh_sink = -1; sh_nrem = nc; sh_nremoved = 0; sh_min = 0.0; }
    __syncthreads();
    while (true) {
      const int i = sh_i, nrem = sh_nrem;
      const double min_val = sh_min, ui = u[i];
      const double* crow = costd + (long long)i * nc;
      Cand best{INFINITY, -1, 0};
      for (int it = tid; it < nrem; it += kThreads) {
        const int j = remaining[it];
        const double r = min_val + crow[j] - ui - v[j];
        double s = spc[j];
        if (r < s) { path[j] = i; spc[j] = r; s = r; }
        const Cand c{s, it, row4col[j] == -1 ? 1 : 0};
        if (better(c, best)) best = c;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        Cand other;
        other.s = __shfl_xor(best.s, o, 64);
        other.it = __shfl_xor(best.it, o, 64);
        other.unassigned = __shfl_xor(best.unassigned, o, 64);
        if (better(other, best)) best = other;
      }
      if (lane == 0) red[wv] = best;
      __syncthreads();
      if (tid == 0) {
        Cand b = red[0];
        for (int w = 1; w < kThreads / 64; ++w)
          if (better(red[w], b)) b = red[w];
        SR[i] = 1;
        if (b.it < 0 || b.s == INFINITY) {   // infeasible
          sh_sink = -2;
        } else {
          sh_min = b.s;
          const int j = remaining[b.it];
          if (row4col[j] == -1) sh_sink = j; else sh_i = row4col[j];
          removed[sh_nremoved++] = j;
          remaining[b.it] = remaining[nrem - 1];
          sh_nrem = nrem - 1;
        }
      }
      __syncthreads();
      if (sh_sink != -1) break;
    }
    if (sh_sink == -2) {
      if (tid == 0) *pr.status = 2;
      return;
    }
    // ---- dual update (with the assignment BEFORE the augmentation) ---------------------------------------------
    const double min_val = sh_min;
    for (int i = tid; i < nr; i += kThreads) {
      if (i == cur) u[i] += min_val;
      else if (SR[i]) u[i] += min_val - spc[col4row[i]];
    }
    const int nremoved = sh_nremoved;
    for (int k = tid; k < nremoved; k += kThreads) {
      const int j = removed[k];
      v[j] -= min_val - spc[j];
    }
    __syncthreads();
    // ---- augment -----------------------------------------------------------------------------------------
    if (tid == 0) {
      int j = sh_sink;
      while (true) {
        const int i = path[j];
        row4col[j] = i;
        const int t = col4row[i];
        col4row[i] = j;
        j = t;
        if (i == cur) break;
      }
    }
    __syncthreads();
  }

  // ---- result, row indices ascending -------------------------------------------------------------------------
  if (!transpose) {
    for (int i = tid; i < nr; i += kThreads) { pr.row_ind[i] = i; pr.col_ind[i] = col4row[i]; }
  } else {   // internal row i is original column i, col4row[i] the original row: sort by it (distinct values)
    for (int i = tid; i < nr; i += kThreads) {
      const int key = col4row[i];
      int rank = 0;
      for (int k = 0; k < nr; ++k) rank += col4row[k] < key;
      pr.row_ind[rank] = key;
      pr.col_ind[rank] = i;
    }
  }
  if (tid == 0) *pr.status = 0;
}

}  // namespace lsap

extern "C" {

int dynmask_set_error(int code, const char* what);   // msda_capi.hip (shared last-error slot)

size_t lsap_hip_workspace_bytes(int rows, int cols) {
  if (rows <= 0 || cols <= 0) return 8;
  const int nr = rows < cols ? rows : cols, nc = rows < cols ? cols : rows;
  return lsap::Layout(nr, nc).total;
}

int lsap_hip_batch_f32(int count, const float* const* cost, const long long* ld, const int* rows, const int* cols,
                       int64_t* const* row_ind, int64_t* const* col_ind, void* const* workspace, int32_t* const* status,
                       void* stream) {
  if (count < 0 || count > LSAP_HIP_MAX_BATCH) return dynmask_set_error(LSAP_ERR_BAD_DIMS, "lsap: batch size out of range");
  if (count == 0) return 0;
  if (!cost || !ld || !rows || !cols || !row_ind || !col_ind || !workspace || !status)
    return dynmask_set_error(LSAP_ERR_NULL_POINTER, "lsap: null pointer argument");
  lsap::Batch b;
  int n = 0;
  for (int k = 0; k < count; ++k) {
    if (rows[k] < 0 || cols[k] < 0 || (rows[k] > 0 && cols[k] > 0 && ld[k] < cols[k]))
      return dynmask_set_error(LSAP_ERR_BAD_DIMS, "lsap: bad dimensions");
    if (!status[k]) return dynmask_set_error(LSAP_ERR_NULL_POINTER, "lsap: null pointer argument");
    if (rows[k] == 0 || cols[k] == 0) continue;   // nothing to assign; status is written by the memset below
    if (!cost[k] || !row_ind[k] || !col_ind[k] || !workspace[k])
      return dynmask_set_error(LSAP_ERR_NULL_POINTER, "lsap: null pointer argument");
    b.p[n++] = lsap::Problem{cost[k], ld[k], rows[k], cols[k], row_ind[k], col_ind[k], static_cast<char*>(workspace[k]), status[k]};
  }
  for (int k = 0; k < count; ++k)
    if (rows[k] == 0 || cols[k] == 0) {
      const hipError_t e = hipMemsetAsync(status[k], 0, sizeof(int32_t), (hipStream_t)stream);
      if (e != hipSuccess) return dynmask_set_error((int)e, hipGetErrorString(e));
    }
  if (n == 0) return 0;
  hipLaunchKernelGGL(lsap::lsap_kernel, dim3((unsigned)n), dim3(lsap::kThreads), 0, (hipStream_t)stream, b);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : dynmask_set_error((int)e, hipGetErrorString(e));
}

int lsap_hip_f32(const float* cost, long long ld, int rows, int cols, int64_t* row_ind, int64_t* col_ind, void* workspace,
                 int32_t* status, void* stream) {
  const float* c[1] = {cost};
  int64_t* r[1] = {row_ind};
  int64_t* cc[1] = {col_ind};
  void* w[1] = {workspace};
  int32_t* s[1] = {status};
  return lsap_hip_batch_f32(1, c, &ld, &rows, &cols, r, cc, w, s, stream);
}

}  // extern "C"
