/*
 * include/patch_embed_hip.h -- C ABI of the backbone patch-embedding convolutions of UNINEXT on MI355X (gfx950), part
 * of libmsda_hip.so.  SURVEY.md 8(f) rank 3.
 *
 * A convolution whose kernel size equals its stride, without padding, is a GEMM over non-overlapping patches:
 *     out[b, py, px, e] = bias[e] + sum_{c, ky, kx} x[b, c, py*k + ky, px*k + kx] * weight[e, c, ky, kx]
 * with M = B * (H / k) * (W / k) rows, N = E columns, K = C * k * k.  Replaces
 *   - ViT   PatchEmbed.proj + permute   projects/UNINEXT/uninext/backbone/utils.py:177-186   (k = 16, 3 -> 768/1280,
 *                                       constructed at uninext/backbone/vit.py:291)         channels_last = 1
 *   - ConvNeXt stem conv                projects/UNINEXT/uninext/backbone/convnext.py:80     (k = 4, 3 -> 96/192/...)
 *   - ConvNeXt downsample convs         projects/UNINEXT/uninext/backbone/convnext.py:87     (k = 2, C -> 2C)
 * by one implicit-GEMM kernel: the patches are never materialised (no im2col buffer); 128 x 128 x 16 tiles are staged
 * through double-buffered LDS and multiplied with v_mfma_f32_32x32x2_f32 -- exact fp32 (an fmaf chain in k order), so
 * results match nn.Conv2d to fp32 round-off; there is no TF32-like mode on gfx950.
 *
 * Rows/columns that do not fill a tile are masked; H and W need not be multiples of k (the remainder is ignored, as
 * nn.Conv2d does).  All pointers are device pointers, contiguous fp32; `bias` may be NULL; `stream` is a hipStream_t
 * as void*; the kernel is only enqueued.  Returns 0, a negative PATCH_EMBED_ERR_*, or a positive hipError_t; the
 * message is available from msda_hip_last_error().
 */
#ifndef PATCH_EMBED_HIP_H_
#define PATCH_EMBED_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PATCH_EMBED_ERR_NULL_POINTER (-1)
#define PATCH_EMBED_ERR_BAD_DIMS (-2)
#define PATCH_EMBED_ERR_UNSUPPORTED (-5)   /* patch not in {2, 4, 8, 16} or C * patch^2 not a multiple of 16 */

/*
 * x       [batch, in_chans, height, width]
 * weight  [embed_dim, in_chans, patch, patch]        (nn.Conv2d layout)
 * bias    [embed_dim] or NULL
 * out     channels_last != 0: [batch, height / patch, width / patch, embed_dim]   (PatchEmbed.forward, after its permute)
 *         channels_last == 0: [batch, embed_dim, height / patch, width / patch]   (nn.Conv2d)
 */
int patch_embed_hip_f32(const float* x, const float* weight, const float* bias, int batch, int in_chans, int height,
                        int width, int embed_dim, int patch, int channels_last, float* out, void* stream);

/*
 * Fast path for inference with fixed weights: split-bf16 products.  Every fp32 operand is split into two bf16 halves
 * (x = hi + lo, 16 mantissa bits kept) and a product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32
 * accumulation: ~2e-5 of the output scale (inside the 1e-4 parity bound of this path) at 3/16 of the matrix-pipe time
 * of the exact kernel.  The weights are split and re-ordered ONCE by patch_embed_hip_pack_weight_f32 into a
 * caller-owned device buffer of patch_embed_hip_packed_weight_bytes(...) bytes (0 = unsupported geometry: patch not
 * in {2, 4, 8, 16} or in_chans * patch^2 not a multiple of 48); patch_embed_hip_packed_f32 then takes that buffer in
 * place of `weight`.  Same argument meaning and layouts as patch_embed_hip_f32.
 */
size_t patch_embed_hip_packed_weight_bytes(int embed_dim, int in_chans, int patch);
int patch_embed_hip_pack_weight_f32(const float* weight, int embed_dim, int in_chans, int patch, void* packed, void* stream);
int patch_embed_hip_packed_f32(const float* x, const void* packed, const float* bias, int batch, int in_chans, int height,
                               int width, int embed_dim, int patch, int channels_last, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PATCH_EMBED_HIP_H_ */
