/*
 * oracle/fast_oracle.c -- a FAST form of the oracle's INT8 convolution for full-size parity runs.
 *
 * TEST INFRASTRUCTURE ONLY (liboracle_fast.so); the product never links or calls it.
 *
 * oracle_conv_int8 (yolo2_oracle.c, pinned bit for bit against the reference's
 * forward_convolutional_layer_q, src/yolov2_forward_network_quantized.c:527-631) walks one output
 * at a time with a k -> (c,ky,kx) decode per multiply: ~10 minutes for yolov3 at 608x608.  The
 * accumulation is INTEGER (int8 x int8 products summed in int32, no overflow: |acc| <= 9*1024*127^2),
 * so any summation order gives the same acc32 -- here the loops are tap-outermost with a
 * vectorisable row loop and filters run on OpenMP threads.  Everything that is floating point
 * (the input quantisation and the acc -> o16 -> y epilogue) is the same scalar code in the same
 * order as in oracle_conv_int8.  tests/test_oracle_pin.py::test_fast_int8_oracle_equals_oracle
 * requires bit-equality (accumulators and outputs) with oracle_conv_int8 on small shapes.
 *
 * Build: gcc -O3 -fno-fast-math -ffp-contract=off -fopenmp -fPIC -shared (no -march: the library
 * travels to another host).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ACT_LEAKY 7

static int max_abs_i(int src, int max_val)
{
    if (abs(src) > abs(max_val)) src = (src > 0) ? max_val : -max_val;
    return src;
}

void oracle_conv_int8_fast(const float *in, const int8_t *weights_int8, const float *biases, float *out,
                           int32_t *acc16_out, int batch, int c, int h, int w, int n, int size, int stride,
                           int pad, int act, float input_quant_multipler, float weights_quant_multipler)
{
    const int out_h = (h + 2 * pad - size) / stride + 1;
    const int out_w = (w + 2 * pad - size) / stride + 1;
    const int K = c * size * size;
    const size_t inputs = (size_t)c * h * w;
    const size_t ohw = (size_t)out_h * out_w;
    int8_t *xq = (int8_t *)malloc(inputs);
    const float ALPHA1 = 32 / (input_quant_multipler * weights_quant_multipler);
    for (int b = 0; b < batch; ++b) {
        const float *im = in + (size_t)b * inputs;
#pragma omp parallel for schedule(static)
        for (long long z = 0; z < (long long)inputs; ++z) {
            const float t = im[z] * input_quant_multipler;
            const int16_t src = (int16_t)(int32_t)t;       /* as gcc/x86-64 compiles `int16_t = float` */
            xq[z] = (int8_t)max_abs_i(src, 127);
        }
        float *o = out + (size_t)b * n * ohw;
#pragma omp parallel
        {
            int32_t *acc = (int32_t *)malloc(ohw * sizeof(int32_t));
#pragma omp for schedule(dynamic, 1)
            for (int f = 0; f < n; ++f) {
                memset(acc, 0, ohw * sizeof(int32_t));
                for (int ci = 0; ci < c; ++ci)
                    for (int ky = 0; ky < size; ++ky)
                        for (int kx = 0; kx < size; ++kx) {
                            const int wv = weights_int8[(size_t)f * K + ((size_t)ci * size + ky) * size + kx];
                            if (wv == 0) continue;
                            /* ox range with 0 <= ox*stride - pad + kx < w */
                            int lo = 0;
                            while (lo < out_w && lo * stride - pad + kx < 0) ++lo;
                            int hi = out_w;
                            while (hi > lo && (hi - 1) * stride - pad + kx >= w) --hi;
                            for (int oy = 0; oy < out_h; ++oy) {
                                const int iy = oy * stride - pad + ky;
                                if (iy < 0 || iy >= h) continue;
                                const int8_t *row = xq + ((size_t)ci * h + iy) * w - pad + kx;
                                int32_t *a = acc + (size_t)oy * out_w;
                                if (stride == 1) {
                                    for (int ox = lo; ox < hi; ++ox) a[ox] += wv * row[ox];
                                } else {
                                    for (int ox = lo; ox < hi; ++ox) a[ox] += wv * row[ox * stride];
                                }
                            }
                        }
                for (size_t p = 0; p < ohw; ++p) {
                    const int16_t o16 = (int16_t)max_abs_i(acc[p] / 32, 256 * 128 - 1);
                    const size_t oi = (size_t)f * ohw + p;
                    if (acc16_out) acc16_out[(size_t)b * n * ohw + oi] = o16;
                    float y = o16 * ALPHA1;
                    y += biases[f];
                    if (act == ACT_LEAKY) y = (y > 0) ? y : y / 10;
                    o[oi] = y;
                }
            }
            free(acc);
        }
    }
    free(xq);
}

/*
 * FLOAT64 GROUND TRUTH of the FP32 convolution (forward_convolutional_layer_cpu, FP32 branch,
 * src/yolov2_forward_network.c:204-261): the same function of the same float weights / biases, with the layer
 * input, every product, the whole sum, the bias add and the activation carried in double.  Products of two
 * floats are exact in double and a double sum over K <= 9216 terms is good to ~1e-13 relative, so the summation
 * order is immaterial here (tap-outermost, filters on OpenMP threads).  Used by tests/ and tools/parity_layers.py
 * to measure how far the reference's own builds (scalar gemm_nn, AVX gemm_nn) and the HIP kernels each sit from
 * the exact result -- the yardstick VERDICT round 2 (next-round item 2) asks for.  Leaky = .1*x in double
 * (src/additionally.h:91), logistic = 1./(1.+exp(-x)) (:84).
 */
#include <math.h>

void oracle_conv_f64(const double *in, const float *weights, const float *biases, double *out,
                     int batch, int c, int h, int w, int n, int size, int stride, int pad, int act)
{
    const int out_h = (h + 2 * pad - size) / stride + 1;
    const int out_w = (w + 2 * pad - size) / stride + 1;
    const int K = c * size * size;
    const size_t inputs = (size_t)c * h * w;
    const size_t ohw = (size_t)out_h * out_w;
    for (int b = 0; b < batch; ++b) {
        const double *im = in + (size_t)b * inputs;
        double *o = out + (size_t)b * n * ohw;
#pragma omp parallel for schedule(dynamic, 1)
        for (int f = 0; f < n; ++f) {
            double *acc = o + (size_t)f * ohw;
            memset(acc, 0, ohw * sizeof(double));
            for (int ci = 0; ci < c; ++ci)
                for (int ky = 0; ky < size; ++ky)
                    for (int kx = 0; kx < size; ++kx) {
                        const double wv = weights[(size_t)f * K + ((size_t)ci * size + ky) * size + kx];
                        int lo = 0;
                        while (lo < out_w && lo * stride - pad + kx < 0) ++lo;
                        int hi = out_w;
                        while (hi > lo && (hi - 1) * stride - pad + kx >= w) --hi;
                        for (int oy = 0; oy < out_h; ++oy) {
                            const int iy = oy * stride - pad + ky;
                            if (iy < 0 || iy >= h) continue;
                            const double *row = im + ((size_t)ci * h + iy) * w - pad + kx;
                            double *a = acc + (size_t)oy * out_w;
                            if (stride == 1) {
                                for (int ox = lo; ox < hi; ++ox) a[ox] += wv * row[ox];
                            } else {
                                for (int ox = lo; ox < hi; ++ox) a[ox] += wv * row[ox * stride];
                            }
                        }
                    }
            for (size_t p = 0; p < ohw; ++p) {
                double y = acc[p] + (double)biases[f];
                if (act == ACT_LEAKY) y = (y > 0) ? y : .1 * y;
                else if (act == 0 /* LOGISTIC */) y = 1. / (1. + exp(-y));
                acc[p] = y;
            }
        }
    }
}
