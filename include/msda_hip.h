/*
 * include/msda_hip.h -- C ABI of libmsda_hip.so, the MI355X (gfx950) implementation of
 * UNINEXT's MultiScaleDeformableAttention operator.
 *
 * This is the drop-in boundary: the entry points below are what the reference's pybind
 * module `MultiScaleDeformableAttention` (ops/src/vision.cpp:13-16) would bind instead of
 * its CUDA implementation.  ops/ = projects/UNINEXT/uninext/models/deformable_detr/ops/.
 *
 *   msda_hip_forward_{f32,f64}   replaces ms_deform_attn_cuda_forward
 *                                (ops/src/cuda/ms_deform_attn_cuda.cu:20-80) and the kernel
 *                                it launches (ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299,
 *                                923-954).
 *   msda_hip_backward_{f32,f64}  replaces ms_deform_attn_cuda_backward
 *                                (ops/src/cuda/ms_deform_attn_cuda.cu:83-153) and the
 *                                col2im dispatcher + kernels (…cuda.cuh:301-920, 956-1327).
 *
 * Conventions
 *   - Plain pointers and sizes only; no torch / ATen types.  All data pointers are DEVICE
 *     pointers on the current HIP device, contiguous, row-major:
 *       value  [batch, spatial_size, num_heads, channels]
 *       spatial_shapes [num_levels, 2] int64 (H, W)       -- device memory, as in the reference
 *       level_start_index [num_levels] int64              -- device memory
 *       sampling_loc [batch, num_query, num_heads, num_levels, num_point, 2]  (x, y) in [0,1]
 *       attn_weight  [batch, num_query, num_heads, num_levels, num_point]
 *       output / grad_output [batch, num_query, num_heads*channels]
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).  Kernels
 *     are only ENQUEUED on it: no allocation, no synchronisation, no host<->device copy,
 *     so the calls are hipGraph-capturable (the reference launches on the current stream
 *     the same way, ms_deform_attn_cuda.cu:65,135).
 *   - Forward writes every element of `output` (no pre-zeroing needed; the reference's
 *     at::zeros at ms_deform_attn_cuda.cu:54 is redundant).
 *   - Backward OVERWRITES grad_sampling_loc and grad_attn_weight and ACCUMULATES into
 *     grad_value with float atomics: the caller must zero grad_value first (the reference
 *     does so with at::zeros_like, ms_deform_attn_cuda.cu:121).  Summation order of
 *     grad_value is therefore not deterministic, exactly as in the reference.
 *   - Return value: 0 on success, a negative MSDA_ERR_* for rejected arguments, a positive
 *     hipError_t if the launch failed (the reference only printf()s launch errors,
 *     ms_deform_im2col_cuda.cuh:948-952).  msda_hip_last_error() returns a thread-local
 *     human-readable message for the last non-zero return.
 *   - The whole batch is processed in one launch; the reference's im2col_step chunking
 *     (ms_deform_attn_cuda.cu:50-52,61) has no numerical effect and is validated by the
 *     Python binding only.
 */
#ifndef MSDA_HIP_H_
#define MSDA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSDA_HIP_ABI_VERSION 2

#define MSDA_ERR_NULL_POINTER (-1)
#define MSDA_ERR_BAD_DIMS (-2)      /* a dimension <= 0 (except batch/num_query == 0, which is a no-op) */
#define MSDA_ERR_TOO_LARGE (-3)     /* a per-image extent does not fit the 32-bit offsets the kernels use */
#define MSDA_ERR_BAD_VARIANT (-4)   /* MSDA_HIP_FWD_VARIANT / msda_hip_set_variant names an unknown kernel */
#define MSDA_ERR_UNSUPPORTED (-5)   /* this entry point has no kernel for the given geometry: use the unfused one */

int msda_hip_abi_version(void);
const char* msda_hip_last_error(void);

int msda_hip_forward_f32(const float* value, const int64_t* spatial_shapes,
                         const int64_t* level_start_index, const float* sampling_loc,
                         const float* attn_weight, int batch, int spatial_size, int num_heads,
                         int channels, int num_levels, int num_query, int num_point,
                         float* output, void* stream);

int msda_hip_forward_f64(const double* value, const int64_t* spatial_shapes,
                         const int64_t* level_start_index, const double* sampling_loc,
                         const double* attn_weight, int batch, int spatial_size, int num_heads,
                         int channels, int num_levels, int num_query, int num_point,
                         double* output, void* stream);

/*
 * Numerics of grad_value in msda_hip_backward_f32.  The reference adds every contribution with a float atomic
 * (ms_deform_im2col_cuda.cuh:129-156): each add rounds to the accumulated value of ITS pixel.  On encoder-shaped calls
 * (num_query == spatial_size, channels 32, 4 points) variant 0 runs msda_bwd_tiled, which combines the adds of an 8 x 16
 * tile of queries in LDS in 32-bit FIXED POINT with one power-of-two scale per (image, head, tile), taken from a bound
 * that cannot overflow: (queries of the tile, <= 256) * max|grad_output| * max over pairs of sum|attn_weight| over
 * the tile.  Every add therefore rounds (to nearest, unbiased) to at most  bound / 2^30  <=  2^-22 * max|grad_output|
 * of the tile for softmaxed weights -- 2.4e-7 of the tile's LARGEST upstream gradient, whatever the size of the pixel's
 * own gradient: contributions much smaller than that next to a large one in the same tile lose relative precision
 * (worst case over the ~1300 adds a level-3 pixel receives 3e-4 of that maximum, ~5e-6 in practice).  Non-finite
 * inputs switch the tile to float atomics, so NaN / Inf propagate as in the reference.  When the reference's
 * per-pixel rounding is required, pin the float-atomic kernel: msda_hip_set_variant(1, 1) or MSDA_HIP_BWD_VARIANT=1
 * (msda_bwd_generic, 2.05 ms instead of 0.35 ms per encoder call).  grad_sampling_loc and grad_attn_weight are
 * plain fp32 in every kernel.  tests/test_msda_parity_gpu.py holds the bound on inputs with 8 decades of dynamic range.
 * The same holds for msda_bwd_win (encoder-shaped calls of a call site with near samples) and, on the rows of the two
 * coarsest levels it keeps in LDS, for msda_bwd_dec -- variant 0 on every other fp32 call with channels 32, 4 levels x 4
 * points and 64 .. 16384 queries (the decoder): one scale per (image, head, slice of <= min(ceil(num_query / 16), 256)
 * queries) from (4 * queries of the slice) * max|grad_output| * max|attn_weight|, i.e. steps of <= 2^-20 * max|grad_output|
 * of the slice (<= 2^-21 for num_query <= 2048).  The finer levels take float atomics there as in the reference; calls with
 * more queries than that (and not encoder-shaped) take msda_bwd_generic, float atomics throughout.
 * msda_bwd_regions (backward variant 6) has no fixed point: every pixel's corners are added in float64 by the one workgroup
 * that owns the pixel and rounded to float ONCE (each term is the float product weight x attention x grad_output as in the
 * reference); its grad_value differs from the exact sum by one float rounding.
 * msda_bwd_dst (backward variant 8, round 6; fp32 calls with channels 32 and 4 levels x 4 points, meant for the decoder's shapes)
 * is its counterpart for few queries on many pixels: float64 sums per 16 x 16 pixel tile in LDS, one rounding per element and
 * workgroup, the few workgroups of a coarse tile meeting in float atomics; 6-12 % slower than msda_bwd_dec at the R50 shapes and
 * therefore not variant 0's choice: msda_hip_set_variant(1, 8) selects it where the fixed point above is unwanted.
 */
int msda_hip_backward_f32(const float* grad_output, const float* value,
                          const int64_t* spatial_shapes, const int64_t* level_start_index,
                          const float* sampling_loc, const float* attn_weight, int batch,
                          int spatial_size, int num_heads, int channels, int num_levels,
                          int num_query, int num_point, float* grad_value,
                          float* grad_sampling_loc, float* grad_attn_weight, void* stream);

/*
 * Workspace of the backward.  One kernel family needs scratch memory: msda_bwd_regions (encoder-shaped fp32 calls whose call
 * site reported far samples, or backward variant 6) files one 32-byte record per (sample, destination region) before it sums
 * them -- three bin tables plus up to four records per sample, ~0.7 GB at the R50 training shapes.
 *
 *   msda_hip_backward_workspace_bytes(dims...)   the bytes a call with these sizes may need: 0 when no kernel it can take
 *                                                uses a workspace (decoder-shaped calls, fp64, other channel counts).
 *   msda_hip_backward_ws_f32(..., workspace, workspace_bytes, stream)
 *                                                msda_hip_backward_f32 with a workspace LENT by the caller for this call:
 *                                                device memory, 256-byte aligned, at least ..._workspace_bytes; contents
 *                                                undefined before and after; it must stay valid until the work enqueued on
 *                                                `stream` by this call has completed (a stream-ordered allocator such as
 *                                                PyTorch's caching allocator gives exactly that).  NULL / too small / not
 *                                                aligned: the library's own workspace, as for msda_hip_backward_f32.
 *
 * Without a lent workspace the library keeps ONE buffer per device, allocated (hipMalloc) at the first call that needs
 * it, grown with a device synchronisation when a larger call arrives, and handed from stream to stream behind an event;
 * inside a stream capture such a call takes msda_bwd_tiled instead (no allocation in a capture).  With a lent workspace
 * none of that happens: no allocation, no synchronisation, capturable.
 */
size_t msda_hip_backward_workspace_bytes(int batch, int spatial_size, int num_heads, int channels, int num_levels,
                                         int num_query, int num_point);
int msda_hip_backward_ws_f32(const float* grad_output, const float* value,
                             const int64_t* spatial_shapes, const int64_t* level_start_index,
                             const float* sampling_loc, const float* attn_weight, int batch,
                             int spatial_size, int num_heads, int channels, int num_levels,
                             int num_query, int num_point, float* grad_value,
                             float* grad_sampling_loc, float* grad_attn_weight, void* workspace,
                             size_t workspace_bytes, void* stream);

int msda_hip_backward_f64(const double* grad_output, const double* value,
                          const int64_t* spatial_shapes, const int64_t* level_start_index,
                          const double* sampling_loc, const double* attn_weight, int batch,
                          int spatial_size, int num_heads, int channels, int num_levels,
                          int num_query, int num_point, double* grad_value,
                          double* grad_sampling_loc, double* grad_attn_weight, void* stream);

/*
 * Forward with the elementwise prologue of MSDeformAttn.forward folded in
 * (ops/modules/ms_deform_attn.py:99-112).  Instead of normalised sampling locations and
 * softmaxed weights it takes the RAW Linear outputs and the reference points:
 *   reference_points [batch, num_query, num_levels, ref_dim]   ref_dim 2: (x, y); 4: (cx, cy, w, h)
 *   sampling_offsets [batch, num_query, num_heads * num_levels * num_point * 2]
 *   attn_logits      [batch, num_query, num_heads * num_levels * num_point]
 * and computes  attn = softmax over (levels x points);  loc = ref + off / (W_l, H_l)  (ref_dim 2)  or
 * loc = ref_xy + off / num_point * ref_wh * 0.5  (ref_dim 4)  inside the kernel.  fp32, channels == 32,
 * num_levels * num_point == 16 only: other geometries return MSDA_ERR_UNSUPPORTED and the caller uses
 * msda_hip_forward_f32 after the PyTorch prologue.  Forward only (inference).
 */
int msda_hip_forward_fused_f32(const float* value, const int64_t* spatial_shapes,
                               const int64_t* level_start_index, const float* reference_points, int ref_dim,
                               const float* sampling_offsets, const float* attn_logits, int batch,
                               int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                               int num_point, float* output, void* stream);

/*
 * The same with `value` in HEAD-MAJOR layout [batch, num_heads, spatial_size, channels] -- what
 * linear_hip_packed_hm_f32 (include/linear_hip.h) writes for the value projection.  A head's pixels are then 128 bytes
 * apart instead of num_heads * 128, which the gather likes better (6-15 % at the R50 shapes).  Encoder-sized calls
 * only (num_levels == num_point == 4, num_query >= 1024); otherwise MSDA_ERR_UNSUPPORTED.
 */
int msda_hip_forward_fused_hm_f32(const float* value_head_major, const int64_t* spatial_shapes,
                                  const int64_t* level_start_index, const float* reference_points, int ref_dim,
                                  const float* sampling_offsets, const float* attn_logits, int batch, int spatial_size,
                                  int num_heads, int channels, int num_levels, int num_query, int num_point,
                                  float* output, void* stream);

/*
 * Training-side prologue (the autograd half).  The fused forward entry points above take the RAW Linear
 * outputs; with these two the backward of such a call needs neither sampling_loc nor attn_weight kept from the forward:
 *
 *   msda_hip_prologue_f32            sampling_loc [N, Lq, M, L, P, 2] and attn_weight [N, Lq, M, L, P] from reference_points
 *                                    [N, Lq, L, ref_dim], sampling_offsets [N, Lq, M * L * P * 2] and attn_logits [N, Lq, M * L * P]:
 *                                    exactly ops/modules/ms_deform_attn.py:99-112 (softmax = exp(x - max) / sum; loc = ref + off /
 *                                    (W_l, H_l) for ref_dim 2, ref_xy + ((off / P) * ref_wh) * 0.5 for ref_dim 4), one kernel.
 *   msda_hip_prologue_backward_f32   grad_sampling_offsets and grad_attn_logits from grad_sampling_loc / grad_attn_weight (as
 *                                    msda_hip_backward_f32 returns them) and the forward's attn_weight: softmax backward
 *                                    w (g - sum w g) and the location chain rule, one kernel; grad_reference_points [N, Lq, L,
 *                                    ref_dim] when the pointer is not NULL (a second small kernel, fixed summation order over heads
 *                                    and points).  grad_sampling_offsets may BE grad_sampling_loc and grad_attn_logits may BE
 *                                    grad_attn_weight (same sizes; in place).
 * Device pointers, contiguous fp32, kernels only enqueued on `stream`.  num_levels * num_point <= 64.  uninext_amd.functions.
 * MSDeformAttnFusedFunction strings them together: forward = msda_hip_forward_fused[_hm]_f32, backward = prologue ->
 * msda_hip_backward_f32 -> prologue backward.
 */
int msda_hip_prologue_f32(const int64_t* spatial_shapes, const float* reference_points, int ref_dim,
                          const float* sampling_offsets, const float* attn_logits, int batch, int num_heads,
                          int num_levels, int num_query, int num_point, float* sampling_loc, float* attn_weight,
                          void* stream);
int msda_hip_prologue_backward_f32(const int64_t* spatial_shapes, const float* reference_points, int ref_dim,
                                   const float* sampling_offsets, const float* attn_weight,
                                   const float* grad_sampling_loc, const float* grad_attn_weight, int batch,
                                   int num_heads, int num_levels, int num_query, int num_point,
                                   float* grad_sampling_offsets, float* grad_attn_logits,
                                   float* grad_reference_points, void* stream);

/*
 * Host-pointer (CPU) variants -- SURVEY.md 8(b)(i).  Same argument order and tensor layouts as the device entry
 * points, but every pointer is HOST memory and the work runs on `num_threads` host threads (<= 0: all hardware
 * threads).  The reference has no CPU implementation (ops/src/cpu/ms_deform_attn_cpu.cpp:17-41 are AT_ERROR stubs and
 * ops/src/ms_deform_attn.h:35-38 raises for CPU tensors); these cover BASELINE configs[0], the model's plumbing on a
 * GPU-less box, which the reference can only run through the grid_sample composition
 * ms_deform_attn_core_pytorch (ops/functions/ms_deform_attn_func.py:43-63).  Synchronous; no HIP call is made, so
 * they work without a GPU.  Backward ACCUMULATES into grad_value (zero it first) and overwrites the other two, like
 * the device entry points; its summation order is deterministic (one thread per (image, head) slice of grad_value).
 * Implemented in uninext_amd/csrc/msda_host.cpp.
 */
int msda_host_forward_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                          const float* sampling_loc, const float* attn_weight, int batch, int spatial_size,
                          int num_heads, int channels, int num_levels, int num_query, int num_point,
                          float* output, int num_threads);
int msda_host_forward_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                          const double* sampling_loc, const double* attn_weight, int batch, int spatial_size,
                          int num_heads, int channels, int num_levels, int num_query, int num_point,
                          double* output, int num_threads);
int msda_host_backward_f32(const float* grad_output, const float* value, const int64_t* spatial_shapes,
                           const int64_t* level_start_index, const float* sampling_loc, const float* attn_weight,
                           int batch, int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                           int num_point, float* grad_value, float* grad_sampling_loc, float* grad_attn_weight,
                           int num_threads);
int msda_host_backward_f64(const double* grad_output, const double* value, const int64_t* spatial_shapes,
                           const int64_t* level_start_index, const double* sampling_loc, const double* attn_weight,
                           int batch, int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                           int num_point, double* grad_value, double* grad_sampling_loc, double* grad_attn_weight,
                           int num_threads);

/* Host threads the last msda_host_* call of the calling thread actually ran on (diagnostic; 0 before any call).  With
 * num_threads <= 0 that is min(hardware threads, independent units, one per ~256 rows of work): rows = batch * num_query
 * for the forward, (image, head) slices of num_query rows each for the backward. */
int msda_host_last_num_threads(void);

/*
 * Kernel selection (tuning / A-B measurement only; results are identical up to fp32
 * summation order).  which: 0 = forward, 1 = backward.  variant: 0 = automatic (default),
 * 1 = generic one-thread-per-output kernel, 2 = lane-group gather kernel, higher numbers as
 * listed by msda_hip_variant_name().  Returns 0 or MSDA_ERR_BAD_VARIANT.  Numbers are stable across builds; the kernels
 * that lost their A/B (forward 3-6, 8, 10-12) are compiled only into the experiments build (`make -C uninext_amd/csrc
 * experiments` -> libmsda_hip_exp.so): in the default library their names read "exp:<kernel>" and selecting one returns
 * MSDA_ERR_BAD_VARIANT.  Backward variants: 1..6; anything else is MSDA_ERR_BAD_VARIANT.
 */
int msda_hip_set_variant(int which, int variant);
int msda_hip_get_variant(int which);
const char* msda_hip_variant_name(int which, int variant); /* NULL past the last variant */

/*
 * Name of the kernel the last forward (which = 0) / backward (which = 1) call of this
 * process actually launched -- lets tests assert that the intended HIP path ran.
 */
const char* msda_hip_last_kernel(int which);

/*
 * Automatic choice of the fp32 forward kernel on the encoder shape (num_query == spatial_size, channels 32, 4 levels
 * x 4 points).  The LDS-window kernel (msda_fwd_win) is faster than the gather kernel (msda_fwd_lg3) while the samples
 * of a query stay within a few pixels of it and slower when they do not, and only the sampling locations tell.
 *
 * msda_hip_set_call_context(call_site, flags) describes the NEXT forward or backward call made from the calling thread
 * (operator or fused entry point; the context is consumed by that call):
 *   call_site   0..63: the caller's slot.  The choice is made PER CALL SITE -- the six encoder layers of a model each
 *               pass their own and each converge on their own kernel.  < 0: no automatic choice (gather kernel).
 *   flags       MSDA_CTX_GEOMETRY_CHECKED  the caller vouches that sum_l H_l * W_l == spatial_size and that
 *                                          level_start_index holds the prefix sums of H_l * W_l.  PRECONDITION of the
 *                                          window kernels (they enumerate the queries from the level shapes); without it
 *                                          variant 0 takes the gather kernel, which walks 0..num_query-1 for any shapes.
 *                                          (Pinned window variants 9 / 10 assume it; they never read or write outside
 *                                          the tensors, but rows >= sum H_l * W_l would be left unwritten.)
 *               MSDA_CTX_DETERMINISTIC     pin the gather kernel (one kernel, no history).
 * Without a context variant 0 takes the gather kernel: a plain msda_hip_forward_f32 call is history-free.
 *
 * With a context: every reporting launch of the window kernel counts the samples that missed their tile's windows into
 * a counter of its own and its last workgroup stores (count, sequence number) in host-mapped memory.  The report of a
 * launch is consumed at the call site's SECOND call after it, behind a wait on an event recorded after that launch
 * (normally complete long before), so the kernel a call takes depends on the call sequence only, never on timing: window
 * kernel (every 8th launch reporting once the first reports are in) until a report says far fraction > 0.33, then the
 * gather kernel with every 64th call sent through the window kernel to refresh the report.  Two runs of the same call sequence are bitwise equal; the two kernels differ from each
 * other in fp32 summation order only.  Under a stream capture (events cannot be waited for) and with
 * MSDA_HIP_FWD_ADAPTIVE=0 in the environment variant 0 takes the gather kernel.
 *
 * Backward (variant 0, fp32, encoder shape): msda_bwd_win -- value and gradient windows in LDS -- when the call carries a
 * context (geometry vouched for, not deterministic) and the FORWARD calls of that call site have last reported a far
 * fraction <= 0.05; msda_bwd_regions -- grad_value summed on the destination side, no global atomics, time independent of
 * the locations -- when they have last reported one > 0.33; msda_bwd_tiled otherwise (no context, no forward yet, in
 * between).  A backward call launches no report and waits for nothing: its choice follows the call sequence of the site's
 * forward calls.  msda_bwd_regions keeps a per-device workspace (bin tables + up to four 32-byte records per sample, ~0.7 GB
 * at the R50 training shapes, allocated at its first call, handed from stream to stream behind an event; inside a stream
 * capture the call takes msda_bwd_tiled instead).
 *
 * msda_hip_forward_locality: number of reports consumed so far on the call site used last on the current device (it
 * waits for the launches made so far on that site) and, in *far_fraction (may be NULL), the far fraction of the latest.
 *
 * msda_hip_reset_call_site(call_site): the slot (< 0: every slot) of the current device forgets what its calls have reported --
 * its next call is a first call again.  For a process that changes the model or the checkpoint behind a slot (the slots are
 * process-wide: 64 of them, and callers that pass no site share the derived ones, uninext_amd/ext.py).
 */
#define MSDA_CTX_GEOMETRY_CHECKED 1u
#define MSDA_CTX_DETERMINISTIC 2u
void msda_hip_set_call_context(int call_site, unsigned flags);
int msda_hip_forward_locality(double* far_fraction);
void msda_hip_reset_call_site(int call_site);

#ifdef __cplusplus
}
#endif
#endif /* MSDA_HIP_H_ */
