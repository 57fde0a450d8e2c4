/* yolo2_hip_lab.h -- test and lab instrumentation of libyolo2hip.so.
 *
 * NOT part of the drop-in boundary (include/yolo2_hip.h): these entry points exist so that tests/ can compare the
 * kernel-layout weight images the device packers write with the host packers word for word (tests/test_gpu_prep.py,
 * tests/test_host_prep.py).  A caller of network_predict_* never needs them; they may change without an ABI bump. */
#ifndef YOLO2_HIP_LAB_H
#define YOLO2_HIP_LAB_H

#include "yolo2_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test hook: the packed weight image of conv layer i as it sits on the device: which = 0 k-major FP32 panels,
 * 1 Winograd U, 2 int8 / bf16 units, 3 XNOR sign words; XNOR layers also 4 = int32 count thresholds of the sign-only
 * epilogue [Mpad] + the number of filters without one, 5 = mean[M], 6 = bias[M]; 7 = the weights as three bf16 pieces
 * (conv_f32_x3.hip), 8 = the row-transformed weights as three bf16 pieces (conv_f32_row3.hip).  Returns its size in bytes (0 = none);
 * copies it when dst_host != NULL (dst_bytes >= size). */
long long yl_debug_layer_packed(yl_network *net, int i, int which, void *dst_host, long long dst_bytes);
/* Test hook (host only, no GPU needed): the Winograd weight transform U = G g G^T of a 3x3 layer
 * (weights[m][c][3][3]) packed the way the kernel reads it.  tiling must be 32 (conv_f32_wino32.hip):
 * [m/32][c/4][xi 16][half 2][m 32][kk 2] with channel = panel*4 + 2*kk + half.  (16 / 64 selected round 3's
 * alternative kernels, removed in round 4: YL_ERR_ARG.)  dst == NULL returns the number of floats needed. */
long long yl_debug_wino_pack(const float *weights, int c, int m, int tiling, float *dst, long long dst_floats);
/* Test hook (host only): the weights of a convolution (weights[m][c][size][size], c % 16 == 0) as conv_f32_x3.hip reads them:
 * every weight as three bf16 numbers whose sum is the weight exactly, [panel][piece 3][k-octet 2][Mpad][8] with panel =
 * (channel / 16) * size^2 + tap, Mpad = m rounded up to 128, zero padded.  dst == NULL returns the number of bytes needed. */
long long yl_debug_x3_pack(const float *weights, int c, int m, int size, void *dst, long long dst_bytes);
/* Test hook (host only): the weights of a 3x3 / stride-1 / pad-1 convolution (weights[m][c][3][3], c % 16 == 0) as
 * conv_f32_row3.hip reads them: the row transform U = G g of every filter row (U0 = g0, U1 = (g0 + g1 + g2) / 2,
 * U2 = (g0 - g1 + g2) / 2, U3 = g2, formed in double and rounded once) as three bf16 numbers whose sum is U exactly,
 * [group = (channel / 16) * 3 + ky][plane 4][piece 3][k-octet 2][Mpad][8], Mpad = m rounded up to 128, zero padded.
 * dst == NULL returns the number of bytes needed. */
long long yl_debug_row3_pack(const float *weights, int c, int m, void *dst, long long dst_bytes);

#ifdef __cplusplus
}
#endif

#endif
