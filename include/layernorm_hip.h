/*
 * include/layernorm_hip.h -- residual add + LayerNorm of the transformer layers around the MSDeformAttn operator on
 * MI355X (gfx950), part of libmsda_hip.so.
 *
 *     out[r, :] = LayerNorm(x[r, :] + residual[r, :]) * gamma + beta          (biased variance, eps inside the sqrt)
 *
 * replaces `src = src + dropout(src2); src = self.normN(src)` of DeformableTransformerEncoderLayer
 * (projects/UNINEXT/uninext/models/deformable_detr/deformable_transformer_dino.py:356-357, 364-365) at inference
 * (dropout is the identity): one pass over the three tensors instead of two kernels and an intermediate.
 * fp32; `features` must be a multiple of 4 and at most 4096; `residual` may be NULL (plain LayerNorm).  Device
 * pointers, contiguous rows; `stream` is a hipStream_t as void*; the kernel is only enqueued.  Returns 0, a negative
 * LAYERNORM_ERR_*, or a positive hipError_t; the message is available from msda_hip_last_error().
 */
#ifndef LAYERNORM_HIP_H_
#define LAYERNORM_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

#define LAYERNORM_ERR_NULL_POINTER (-1)
#define LAYERNORM_ERR_BAD_DIMS (-2)
#define LAYERNORM_ERR_UNSUPPORTED (-5)

int add_layernorm_hip_f32(const float* x, const float* residual, const float* gamma, const float* beta, float eps,
                          long long rows, int features, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LAYERNORM_HIP_H_ */
