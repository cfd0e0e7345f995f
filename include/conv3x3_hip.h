/*
 * include/conv3x3_hip.h -- C ABI of the 3x3 convolutions of UNINEXT's static mask head on MI355X (gfx950), part of
 * libmsda_hip.so.  SURVEY.md 8(f) rank 2, second half ("the static MaskHeadSmallConv 3x3 convs as MFMA
 * implicit-GEMM", ~31 GFLOP per 800x1333 image).
 *
 * Replaces torch.nn.Conv2d(cin, cout, 3, padding=1) followed by F.relu in MaskHeadSmallConv.forward
 * (projects/UNINEXT/uninext/models/ddetrs_dn.py:941-953 constructors; :991-993 lay3, :1002-1004 lay4,
 * :1016-1018 jia_dcn, :1020-1025 lay1 / lay2; same class in models/ddetrs.py:670) by one implicit-GEMM kernel:
 *     out[b, n, y, x] = act(bias[n] + sum_{c, ky, kx} in[b, c, y + ky - 1, x + kx - 1] * weight[n, c, ky, kx])
 * with zero padding; M = B*H*W pixels, N = cout, K = 9*cin.  The shifted input windows are read in place (no im2col
 * buffer), tiles are staged through double-buffered LDS and multiplied with v_mfma_f32_32x32x2_f32 -- exact fp32.
 *
 * All pointers are device pointers, contiguous fp32, NCHW; `bias` may be NULL; `stream` is a hipStream_t as void*;
 * the kernel is only enqueued.  Returns 0, a negative CONV3X3_ERR_*, or a positive hipError_t; the message is
 * available from msda_hip_last_error().
 */
#ifndef CONV3X3_HIP_H_
#define CONV3X3_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CONV3X3_ERR_NULL_POINTER (-1)
#define CONV3X3_ERR_BAD_DIMS (-2)
#define CONV3X3_ERR_UNSUPPORTED (-5)   /* 9 * cin is not a multiple of 16 */

/*
 * in      [batch, cin, height, width]
 * weight  [cout, cin, 3, 3]
 * bias    [cout] or NULL
 * relu    != 0: out = max(0, conv + bias)
 * precision  0: exact fp32 products (v_mfma_f32_32x32x2_f32), bitwise an fmaf chain in k order;
 *            1: split-bf16 products -- each fp32 operand is split into two bf16 halves (16 mantissa bits kept) and a
 *               product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: ~2e-5 of the
 *               output scale (inside the 1e-4 parity bound of this path) at 3/16 of the matrix-pipe time.
 * out     [batch, cout, height, width]
 */
int conv3x3_hip_f32(const float* in, const float* weight, const float* bias, int batch, int cin, int height, int width,
                    int cout, int relu, int precision, float* out, void* stream);

/*
 * Fast path of precision 1 for inference with fixed weights: the weights are split into bf16 hi / lo halves and
 * re-ordered ONCE (conv3x3_hip_pack_weight_f32 into a caller-owned device buffer of
 * conv3x3_hip_packed_weight_bytes(cout, cin) bytes; cin must be a multiple of 16), and conv3x3_hip_packed_f32 runs the
 * convolution from that buffer: 8 x 16 pixel tiles whose 10 x 18 halo is staged once per 16 input channels and
 * serves all nine taps, weight fragments loaded straight from the packed buffer into registers.  Same numerics as
 * precision 1 above (split-bf16 products, fp32 accumulation).
 */
size_t conv3x3_hip_packed_weight_bytes(int cout, int cin);   /* 0 if the geometry is unsupported */
int conv3x3_hip_pack_weight_f32(const float* weight, int cout, int cin, void* packed, void* stream);
int conv3x3_hip_packed_f32(const float* in, const void* packed, const float* bias, int batch, int cin, int height,
                           int width, int cout, int relu, float* out, void* stream);

/*
 * The EXACT fp32 convolution in the same halo structure (round 4): weights re-ordered once by
 * conv3x3_hip_pack_weight_exact_f32 into [cin / 16][9 taps][cout padded to 128][16] fp32 -- position 8 h + s of a chunk holds
 * channel 2 s + h, the order in which v_mfma_f32_32x32x2_f32 consumes them (one float per lane and operand: lane (row, h) feeds
 * channel 2 s + h in k-step s) -- conv3x3_hip_packed_exact_weight_bytes(cout, cin) bytes; conv3x3_hip_packed_exact_f32 stages
 * the fp32 halo once per 16 input channels and runs 288 MFMAs per wave between two barriers.  Every output element is one
 * chain of fp32 fused multiply-adds over (chunk, tap, k-step, h): exact fp32 arithmetic in a fixed order, bitwise repeatable,
 * within fp32 round-off of precision 0 above and of any library convolution.  cin % 16 == 0.
 */
size_t conv3x3_hip_packed_exact_weight_bytes(int cout, int cin);   /* 0 if the geometry is unsupported */
int conv3x3_hip_pack_weight_exact_f32(const float* weight, int cout, int cin, void* packed, void* stream);
int conv3x3_hip_packed_exact_f32(const float* in, const void* packed, const float* bias, int batch, int cin, int height,
                                 int width, int cout, int relu, float* out, void* stream);

/*
 * out = skip + nearest-neighbour up-sampling of `low` to skip's size: the FPN-style merges of MaskHeadSmallConv.forward
 * (`x[-2] + F.interpolate(fused_x, size=..., mode="nearest")`, ddetrs_dn.py:1001,1012) in one pass instead of an
 * interpolate kernel, an add kernel and an intermediate.  skip / out [batch, channels, height, width], low
 * [batch, channels, low_h, low_w]; source index = min(floor(dst * (low / size)), low - 1) in fp32, as PyTorch computes it.
 */
int upsample_add_hip_f32(const float* skip, const float* low, int batch, int channels, int height, int width, int low_h,
                         int low_w, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CONV3X3_HIP_H_ */
