/*
 * include/lsap_hip.h -- C ABI of the on-device linear sum assignment of UNINEXT's Hungarian matcher on MI355X
 * (gfx950), part of libmsda_hip.so.  SURVEY.md 8(f) rank 4.
 *
 * Replaces the `C.cpu()` + `scipy.optimize.linear_sum_assignment(c[i])` step of HungarianMatcherVL.forward
 * (projects/UNINEXT/uninext/models/deformable_detr/matcher.py:500-503; same call at :257) -- a device->host copy, a
 * host sync and, for the encoder-proposal matching with 22 223 rows, a long single-threaded solve per image and step.
 * The kernel runs SciPy's algorithm (shortest augmenting paths after D. F. Crouse, IEEE TAES 52(4), 2016;
 * scipy/optimize/rectangular_lsap) in float64 with the same scan order and tie rules, so the assignment is the one
 * SciPy returns, index for index (tests: SciPy itself and oracle/lsap_oracle.py on random, integer and constant
 * matrices).  One 1024-thread workgroup per problem: the column scan of every augmenting-path step is parallel, the
 * steps are sequential as in the algorithm.
 *
 * cost      [rows, cols] fp32 on the device, row stride `ld` elements (a column slice of a wider matrix is fine)
 * row_ind / col_ind   int64 [min(rows, cols)] on the device: the assigned pairs, row_ind ascending (SciPy's order)
 * workspace device buffer of lsap_hip_workspace_bytes(rows, cols) bytes, 8-byte aligned, private to the call until the
 *           stream reaches the end of the kernel
 * status    int32 on the device, written by the kernel: 0 ok, 1 the matrix contains NaN or -inf, 2 infeasible
 *           (SciPy raises ValueError in both cases; here the caller decides when to look)
 * Up to LSAP_HIP_MAX_BATCH problems go into one launch (one workgroup = one CU each).  Kernels are only enqueued.
 * Returns 0, a negative LSAP_ERR_*, or a positive hipError_t; the message is available from msda_hip_last_error().
 */
#ifndef LSAP_HIP_H_
#define LSAP_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSAP_ERR_NULL_POINTER (-1)
#define LSAP_ERR_BAD_DIMS (-2)
#define LSAP_HIP_MAX_BATCH 32

size_t lsap_hip_workspace_bytes(int rows, int cols);

int lsap_hip_f32(const float* cost, long long ld, int rows, int cols, int64_t* row_ind, int64_t* col_ind,
                 void* workspace, int32_t* status, void* stream);

/* `count` problems in one launch: arrays of length count on the HOST holding the per-problem arguments of lsap_hip_f32 */
int lsap_hip_batch_f32(int count, const float* const* cost, const long long* ld, const int* rows, const int* cols,
                       int64_t* const* row_ind, int64_t* const* col_ind, void* const* workspace, int32_t* const* status,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LSAP_HIP_H_ */
