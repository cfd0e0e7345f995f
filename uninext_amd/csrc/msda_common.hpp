// Shared device/host helpers for the gfx950 MSDeformAttn kernels (internal, not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cstdlib>

namespace msda {

constexpr int kBlock = 256;      // 4 wave64 per workgroup
constexpr int kMaxLP = 128;      // lane-group kernels keep a per-sample (H, W, start) table in LDS
constexpr int kLevelTableBytes = 3 * kMaxLP * 4;  // multiple of 16: the records behind it stay aligned
constexpr uint32_t kOobOffset = 0x80000000u;  // buffer offset that is always past num_records (< 2 GiB)

// GCC-style vector: the type __builtin_amdgcn_raw_buffer_load_b128 returns (an ext_vector_type
// typedef silently converts through a scalar splat on this compiler).
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));
typedef float f32x4 __attribute__((__vector_size__(16)));
typedef float f32x2 __attribute__((__vector_size__(8)));

// 16-byte raw buffer load as 4 floats.  NB: never __builtin_bit_cast a single vector ELEMENT
// (`bit_cast(float, v[i])` reads element 0 for every i on this compiler); cast the whole vector.
__device__ __forceinline__ f32x4 buffer_load_f32x4(__amdgpu_buffer_rsrc_t rsrc, uint32_t voffset, uint32_t soffset) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, 0));
}

struct Dims {
  int N, S, M, D, L, Lq, P;
};

// One bilinear sample of level (H, W): the reference's per-sample arithmetic
// (ops/src/cuda/ms_deform_im2col_cuda.cuh:282-288 and :38-46).
template <typename T>
struct Sample {
  T lh, lw, hh, hw;          // fractional parts and their complements
  int h_low, w_low;          // top-left corner (may be -1)
  bool in_range;             // the reference's `h_im > -1 && w_im > -1 && h_im < H && w_im < W`
  bool ok1, ok2, ok3, ok4;   // per-corner validity (top-left, top-right, bottom-left, bottom-right)
};

template <typename T>
__device__ __forceinline__ Sample<T> make_sample(T loc_w, T loc_h, int H, int W) {
  Sample<T> s;
  const T h_im = loc_h * (T)H - (T)0.5;
  const T w_im = loc_w * (T)W - (T)0.5;
  s.in_range = (h_im > (T)-1) && (w_im > (T)-1) && (h_im < (T)H) && (w_im < (T)W);
  const T hf = floor(h_im), wf = floor(w_im);
  // Out-of-range samples may carry huge / NaN coordinates: keep the int conversion defined.
  s.h_low = s.in_range ? (int)hf : 0;
  s.w_low = s.in_range ? (int)wf : 0;
  s.lh = h_im - hf;
  s.lw = w_im - wf;
  s.hh = (T)1 - s.lh;
  s.hw = (T)1 - s.lw;
  const bool t = s.in_range && s.h_low >= 0, bt = s.in_range && s.h_low + 1 <= H - 1;
  const bool lf = s.w_low >= 0, rt = s.w_low + 1 <= W - 1;
  s.ok1 = t && lf;
  s.ok2 = t && rt;
  s.ok3 = bt && lf;
  s.ok4 = bt && rt;
  return s;
}

__device__ __forceinline__ void atomic_add(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add(double* p, double v) { unsafeAtomicAdd(p, v); }

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Launchers implemented in msda_fwd.hip / msda_bwd.hip.  `kernel_name` receives a static string.
template <typename T>
int launch_forward(int variant, const T* value, const int64_t* shapes, const int64_t* lsi, const T* loc,
                   const T* attn, const Dims& d, T* out, hipStream_t stream, const char** kernel_name);

template <typename T>
int launch_backward(int variant, const T* grad_out, const T* value, const int64_t* shapes, const int64_t* lsi,
                    const T* loc, const T* attn, const Dims& d, T* grad_value, T* grad_loc, T* grad_attn,
                    hipStream_t stream, const char** kernel_name);

// More than 64 KiB of dynamic LDS needs an opt-in that HIP keeps PER DEVICE; `done` (one per kernel) remembers the
// device ordinals already opted in, so a process that drives several GPUs gets the attribute on each of them.
// Not a stream operation (safe under graph capture); racing threads at worst set the attribute twice.
inline int ensure_dynamic_lds(const void* kernel, int bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return 0;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  done.fetch_or(bit, std::memory_order_release);
  return 0;
}

// A/B hooks of the kernels (tile shapes, grid forms, alternative passes): the PRODUCT library reads no such environment
// variable -- ab_env_int folds to its default at compile time; `make experiments` (-DMSDA_EXPERIMENTS) builds the library that
// honours them, to repeat the measurements recorded under profiles/.  The supported switches are the ones listed in
// INTEGRATION.md ("Environment").
#ifdef MSDA_EXPERIMENTS
inline int ab_env_int(const char* name, int dflt) {
  const char* e = std::getenv(name);
  return e ? std::atoi(e) : dflt;
}
#else
constexpr int ab_env_int(const char*, int dflt) { return dflt; }
#endif

// Variant numbering shared with include/msda_hip.h.
enum Variant { kAuto = 0, kGeneric = 1, kLaneGroup = 2, kTiled = 3, kTiledL0 = 4, kTiledL0Big = 5, kLaneGroupCL = 6,
               kLaneGroupL3 = 7, kLaneGroupP = 8, kWin = 9, kWin2 = 10, kWin3 = 11, kWin4 = 12, kWinL = 13, kWinP = 14, kNumVariants = 15 };

// msda_fwd.hip: forward with the MSDeformAttn prologue (softmax + sampling locations) fused in.
bool fused_forward_ok(const Dims& d, int ref_dim);
bool fused_forward_hm_ok(const Dims& d, int ref_dim);   // head-major value: encoder-sized calls only
int launch_forward_fused(int variant, const float* value, int head_major, const int64_t* shapes, const int64_t* lsi,
                         const float* ref_points, int ref_dim, const float* offsets, const float* logits, const Dims& d,
                         float* out, hipStream_t stream, const char** kernel_name);

// msda_bwd_tiled.hip: backward with grad_value privatised in LDS (fp32, D = 32, P = 4, Lq == S).
bool tiled_backward_ok(const Dims& d);
int launch_backward_tiled(const float* grad_out, const float* value, const int64_t* shapes, const int64_t* lsi,
                          const float* loc, const float* attn, const Dims& d, float* grad_value, float* grad_loc,
                          float* grad_attn, hipStream_t stream);

int launch_backward_tiled_nogv(const float* grad_out, const float* value, const int64_t* shapes, const int64_t* lsi,
                               const float* loc, const float* attn, const Dims& d, float* grad_loc, float* grad_attn,
                               hipStream_t stream);   // grad_sampling_loc / grad_attn_weight only

// msda_bwd_regions.hip: encoder backward with grad_value summed on the DESTINATION side (fp32, D = 32, L = P = 4, Lq == S):
// no global atomics, time independent of where the samples fall
bool regions_backward_ok(const Dims& d);
// msda_bwd_q.hip: grad_sampling_loc / grad_attn_weight alone, in msda_fwd_lg3's gather structure (the query-side pass of msda_bwd_regions)
bool q_backward_ok(const Dims& d);
int launch_backward_q(const float* grad_out, const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc,
                      const float* attn, const Dims& d, float* grad_loc, float* grad_attn, hipStream_t stream);
size_t regions_workspace_bytes(const Dims& d);              // include/msda_hip.h: msda_hip_backward_workspace_bytes
void set_call_workspace(void* p, size_t bytes);              // lent to the next backward call of this thread (nullptr: none)
int launch_backward_regions(const float* grad_out, const float* value, const int64_t* shapes, const int64_t* lsi,
                            const float* loc, const float* attn, const Dims& d, float* grad_value, float* grad_loc,
                            float* grad_attn, hipStream_t stream, const char** kernel_name = nullptr);

// msda_bwd_win.hip: encoder backward with value AND gradient windows in LDS (fp32, D = 32, L = P = 4, Lq == S).
bool win_backward_ok(const Dims& d);
// msda_bwd_dec.hip: decoder-style calls (fp32, D = 32, L = P = 4) with LDS accumulators for the coarse levels
bool dec_backward_ok(const Dims& d);
int launch_backward_dec(const float* grad_out, const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc,
                        const float* attn, const Dims& d, float* grad_value, float* grad_loc, float* grad_attn, hipStream_t stream);
int launch_backward_win(const float* grad_out, const float* value, const int64_t* shapes, const int64_t* lsi,
                        const float* loc, const float* attn, const Dims& d, float* grad_value, float* grad_loc,
                        float* grad_attn, hipStream_t stream);
// msda_bwd_dst.hip (round 6): decoder-style calls with grad_value summed on the destination side (a workgroup owns a 16 x 16 pixel tile
// of one level; float64 sums in LDS; no L2 atomics but the slice sums of the coarse levels)
bool dst_backward_ok(const Dims& d);
int launch_backward_dst(const float* grad_out, const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc,
                        const float* attn, const Dims& d, float* grad_value, float* grad_loc, float* grad_attn, hipStream_t stream);

// msda_fwd_win.hip: encoder forward with LDS windows on all four levels (fp32, D = 32, L = P = 4, Lq == S).
bool win_forward_ok(const Dims& d);
bool win_forward_auto(const Dims& d, hipStream_t stream);   // auto dispatch: take the window kernel for this call? (consumes the call context)
// experiments/msda_bwd_win2.hip (`make experiments`): the same partition with ONE window set per workgroup (value windows, then accumulators): two workgroups per CU
bool win2_backward_ok(const Dims& d);
int launch_backward_win2(const float* grad_out, const float* value, const int64_t* shapes, const int64_t* lsi,
                         const float* loc, const float* attn, const Dims& d, float* grad_value, float* grad_loc,
                         float* grad_attn, hipStream_t stream);
int backward_site_choice(const Dims& d);                     // backward of an encoder-shaped call: 1 = msda_bwd_win (the site's forward calls reported near samples), 2 = msda_bwd_regions (they reported far ones), 0 = no report to go by (consumes the context)
void set_call_context(int slot, unsigned flags);            // include/msda_hip.h: msda_hip_set_call_context
void drop_call_context();
int launch_forward_win_fused(const float* value, int head_major, const int64_t* shapes, const int64_t* lsi,
                             const float* ref_points, int ref_dim, const float* offsets, const float* logits, const Dims& d,
                             float* out, hipStream_t stream);
int forward_locality(double* far_fraction);      // reports so far (0: none yet) and the last one's far fraction
void reset_call_site(int slot);                  // include/msda_hip.h: msda_hip_reset_call_site (< 0: every slot of the current device)
int launch_forward_win(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn,
                       const Dims& d, float* out, hipStream_t stream);

// experiments/msda_fwd_winl.hip (round 6): the window kernel with scalar level constants, a pair's points split over two lanes (same preconditions as msda_fwd_win).
bool winl_forward_ok(const Dims& d);
int launch_forward_winl(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn,
                        const Dims& d, float* out, hipStream_t stream);

// experiments/msda_fwd_winp.hip (round 6): the pipelined window kernel (one 12-wave workgroup per CU, two window sets; same preconditions).
bool winp_forward_ok(const Dims& d);
int launch_forward_winp(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn,
                        const Dims& d, float* out, hipStream_t stream);

// msda_fwd_win2.hip: the one-pass, 11-wave generation of the window kernel (same preconditions).
bool win2_forward_ok(const Dims& d);
// msda_fwd_win3.hip: the persistent, software-pipelined generation (two window sets, one workgroup per CU; same preconditions).
bool win3_forward_ok(const Dims& d);
// msda_fwd_win4.hip: TWO lanes per (query, head) pair instead of a quad (same preconditions).
bool win4_forward_ok(const Dims& d);
int launch_forward_win4(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn,
                        const Dims& d, float* out, hipStream_t stream);
int launch_forward_win3(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn,
                        const Dims& d, float* out, hipStream_t stream);
int launch_forward_win2(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn,
                        const Dims& d, float* out, hipStream_t stream);

// msda_fwd_tiled.hip: LDS-tiled encoder forward (fp32, D = 32, Lq == S).
bool tiled_forward_ok(const Dims& d);
int launch_forward_tiled(int flavour, const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc,
                         const float* attn, const Dims& d, float* out, hipStream_t stream);

}  // namespace msda
