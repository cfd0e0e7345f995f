/*
 * include/linear_hip.h -- C ABI of the fp32 Linear layers around the MSDeformAttn operator on MI355X (gfx950), part
 * of libmsda_hip.so.
 *
 * MSDeformAttn.forward (projects/UNINEXT/uninext/models/deformable_detr/ops/modules/ms_deform_attn.py:95-116) runs
 * four nn.Linear layers around the sampling kernel: value_proj (+ masked_fill of padded tokens, :95-97),
 * sampling_offsets (:99), attention_weights (:100) and output_proj (:114); at the R50 shapes (44 446 tokens x 256)
 * they take 2/3 of the layer's time.  These entry points compute
 *     out[m, n] = (row_mask && row_mask[m]) ? 0 : bias[n] + sum_k x[m, k] * weight[n, k]
 * with split-bf16 products on the matrix cores: every fp32 operand is split into two bf16 halves (x = hi + lo, 16
 * mantissa bits kept) and a product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation --
 * ~2e-5 of the output scale, inside the 1e-4 parity bound of the path.  The weight is split and re-ordered ONCE
 * (linear_hip_pack_weight_f32) into a caller-owned device buffer of linear_hip_packed_weight_bytes(n, k) bytes.
 *
 * All pointers are device pointers, contiguous, row-major; `bias` and `row_mask` (one byte per row, non-zero = write
 * zeros: the reference's masked_fill(input_padding_mask[..., None], 0)) may be NULL; `stream` is a hipStream_t as
 * void*; kernels are only enqueued.  in_features must be a multiple of 64.  Returns 0, a negative LINEAR_ERR_*, or a
 * positive hipError_t; the message is available from msda_hip_last_error().
 */
#ifndef LINEAR_HIP_H_
#define LINEAR_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LINEAR_ERR_NULL_POINTER (-1)
#define LINEAR_ERR_BAD_DIMS (-2)
#define LINEAR_ERR_UNSUPPORTED (-5)   /* in_features not a multiple of 64 */

size_t linear_hip_packed_weight_bytes(int out_features, int in_features);   /* 0 if unsupported */
/* weight [out_features, in_features] fp32 (nn.Linear layout) -> packed */
int linear_hip_pack_weight_f32(const float* weight, int out_features, int in_features, void* packed, void* stream);
/* x [rows, in_features] -> out [rows, out_features] */
int linear_hip_packed_f32(const float* x, const void* packed, const float* bias, const uint8_t* row_mask,
                          long long rows, int in_features, int out_features, float* out, void* stream);

/*
 * The same with the output written HEAD-MAJOR for msda_hip_forward_fused_hm_f32 (include/msda_hip.h): x is
 * [images * rows_per_image, in_features], out is [images, out_features / 32, rows_per_image, 32] -- the value
 * projection of MSDeformAttn with its `view(N, S, heads, 32)` transposed to (N, heads, S, 32) at no extra cost.
 */
int linear_hip_packed_hm_f32(const float* x, const void* packed, const float* bias, const uint8_t* row_mask,
                             long long rows, int in_features, int out_features, int rows_per_image, float* out,
                             void* stream);

/*
 * Extended form for the layers around the attention (DeformableTransformerEncoderLayer.forward,
 * deformable_transformer_dino.py:354-370): the input is x + x_add when x_add != NULL (`with_pos_embed(src, pos)`, :363,
 * folded into the operand load) and `activation` 1 applies ReLU in the epilogue (`activation(linear1(src))`, :355).
 */
int linear_hip_packed_ex_f32(const float* x, const float* x_add, const void* packed, const float* bias,
                             const uint8_t* row_mask, long long rows, int in_features, int out_features, int activation,
                             float* out, void* stream);

/*
 * Two Linear layers on the same input as ONE product: W is the row-wise concatenation [W_a; W_b] packed as a
 * [out_features, in_features] weight, bias the concatenation of the two biases (or NULL); columns [0, split_col) are
 * written to out_a [rows, split_col], the rest to out_b [rows, out_features - split_col].  MSDeformAttn.forward
 * computes sampling_offsets(query) and attention_weights(query) from the same query (ops/modules/ms_deform_attn.py:
 * 99-100): one pass over query (+ x_add = the positional embedding) instead of two.  split_col must be a multiple of
 * 128.  Every output element is the same sum of products as with the two separate calls.
 */
int linear_hip_packed_split_f32(const float* x, const float* x_add, const void* packed, const float* bias, long long rows,
                                int in_features, int out_features, int split_col, float* out_a, float* out_b, void* stream);

/*
 * Linear followed by the residual add and LayerNorm of the transformer layer, in the Linear's epilogue:
 *     out[m, :] = LayerNorm(residual[m, :] + bias + x[m, :] W^T) * gamma + beta
 * (`src = src + dropout(linear2(...)); src = norm2(src)` and the attention's output_proj + norm1,
 * deformable_transformer_dino.py:355-357, 363-365).  out_features must be 256 (a workgroup holds whole rows);
 * `residual`, `gamma`, `beta`, `bias` may be NULL.
 */
int linear_hip_packed_ln_f32(const float* x, const void* packed, const float* bias, const float* residual,
                             const float* gamma, const float* beta, float eps, long long rows, int in_features,
                             int out_features, float* out, void* stream);

/*
 * The whole feed-forward block of the transformer layer in one kernel (d_model == 256, d_ffn % 128 == 0):
 *     out[m, :] = f(residual[m, :] + bias2 + relu(x[m, :] W1^T + bias1) W2^T),   f = LayerNorm(.) * gamma + beta when
 * layer_norm != 0, identity otherwise
 * (`src2 = linear2(dropout(activation(linear1(src)))); src = norm2(src + dropout(src2))`,
 * deformable_transformer_dino.py:354-357, dropout = identity at inference).  The [rows, d_ffn] hidden activations
 * never leave the CU; products and accumulation order are those of linear_hip_packed_ex_f32 (ReLU) followed by
 * linear_hip_packed_ln_f32, so the result is bitwise the two-call result.  packed1 / packed2: linear_hip_pack_weight_f32
 * of W1 [d_ffn, d_model] and W2 [d_model, d_ffn]; bias1, bias2, residual, gamma, beta may be NULL.
 */
int linear_hip_packed_ffn_f32(const float* x, const void* packed1, const float* bias1, const void* packed2,
                              const float* bias2, const float* residual, const float* gamma, const float* beta, float eps,
                              int layer_norm, long long rows, int d_model, int d_ffn, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LINEAR_HIP_H_ */
