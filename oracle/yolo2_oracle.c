/*
 * oracle/yolo2_oracle.c -- CPU RESTATEMENT of the reference hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (yolo2_light_amd/,
 * libyolo2hip.so) may include, link or call this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do, as the checker.
 *
 * Each function restates, in plain scalar C with the same types and the same
 * evaluation order, what the reference's CPU path computes (scalar build,
 * `-O2 -fno-fast-math`, no AVX/OpenMP -- SURVEY 8c/A19).  Pinning: the
 * reference ships no golden vectors, so tests/test_oracle_pin.py
 * pins every function here bit-for-bit against the reference's own functions
 * run in this container/GPU box through oracle/_ref/libyolo2ref.so (built by
 * oracle/Makefile from the unmodified reference sources), and against the
 * fixtures under tests/golden/ generated from that library.
 *
 * Build:  gcc -O2 -fno-fast-math -ffp-contract=off -fPIC -shared (oracle/Makefile)
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ACTIVATION  src/additionally.h:68-70 */
enum { ACT_LOGISTIC = 0, ACT_RELU, ACT_RELIE, ACT_LINEAR, ACT_RAMP, ACT_TANH, ACT_PLSE, ACT_LEAKY, ACT_ELU, ACT_LOGGY,
       ACT_STAIR, ACT_HARDTAN, ACT_LHTAN };

/* activate()  src/additionally.h:132-165 and the *_activate bodies :72-105: the same C expressions on a float
 * argument (literals with a decimal point are doubles, exp() is the double function) */
static float activate(float x, int a)
{
    switch (a) {
    case ACT_LINEAR: return x;
    case ACT_LOGISTIC: return 1. / (1. + exp(-x));
    case ACT_LOGGY: return 2. / (1. + exp(-x)) - 1;
    case ACT_RELU: return x * (x > 0);
    case ACT_ELU: return (x >= 0) * x + (x < 0) * (exp(x) - 1);
    case ACT_RELIE: return (x > 0) ? x : .01 * x;
    case ACT_RAMP: return x * (x > 0) + .1 * x;
    case ACT_LEAKY: return (x > 0) ? x : .1 * x;
    case ACT_TANH: return (exp(2 * x) - 1) / (exp(2 * x) + 1);
    case ACT_PLSE:
        if (x < -4) return .01 * (x + 4);
        if (x > 4) return .01 * (x - 4) + 1;
        return .125 * x + .5;
    case ACT_STAIR: {
        int n = floor(x);
        if (n % 2 == 0) return floor(x / 2.);
        else return (x - n) + floor(x / 2.);
    }
    case ACT_HARDTAN:
        if (x < -1) return -1;
        if (x > 1) return 1;
        return x;
    case ACT_LHTAN:
        if (x < 0) return .001 * x;
        if (x > 1) return .001 * (x - 1) + 1;
        return x;
    }
    return 0;
}

/* forward_convolutional_layer_cpu, FP32 branch
 *   src/yolov2_forward_network.c:38 (zero fill), :205 im2col_cpu_custom -> im2col_cpu
 *   (src/additionally.c:39-62, zero padding), :207-210 gemm_nn (src/additionally.c:1272-1286:
 *   for k ascending: C[j] += A[k]*B[k][j], separate multiply and add),
 *   :243-252 += bias, :261 activate_array_cpu_custom. */
void oracle_conv_f32(const float *in, const float *weights, const float *biases, float *out,
                     int batch, int c, int h, int w, int n, int size, int stride, int pad, int act)
{
    const int out_h = (h + 2 * pad - size) / stride + 1;
    const int out_w = (w + 2 * pad - size) / stride + 1;
    const int K = c * size * size;
    for (int b = 0; b < batch; ++b) {
        const float *im = in + (size_t)b * c * h * w;
        float *o = out + (size_t)b * n * out_h * out_w;
        for (int f = 0; f < n; ++f) {
            for (int oy = 0; oy < out_h; ++oy) {
                for (int ox = 0; ox < out_w; ++ox) {
                    float acc = 0;
                    for (int k = 0; k < K; ++k) {
                        const int kx = k % size;
                        const int ky = (k / size) % size;
                        const int ci = k / size / size;
                        const int iy = ky + oy * stride - pad;
                        const int ix = kx + ox * stride - pad;
                        float v = 0;
                        if (iy >= 0 && ix >= 0 && iy < h && ix < w) v = im[ix + w * (iy + h * ci)];
                        const float a_part = 1 * weights[(size_t)f * K + k];   /* ALPHA*A[i*lda+k] */
                        acc += a_part * v;
                    }
                    acc += biases[f];
                    o[((size_t)f * out_h + oy) * out_w + ox] = activate(acc, act);
                }
            }
        }
    }
}

/* forward_convolutional_layer_q   src/yolov2_forward_network_quantized.c:527-631
 *   :554-560  int16_t src = x*in_mult (float->int16: x86 cvttss2si then low 16 bits); max_abs(src,127)
 *   :186-209  im2col_cpu_int8 (zero padding)
 *   :469-491  gemm_nn_int8_int16: int32 accumulate over all k, then C += max_abs(acc / 32, 32767)
 *   :596-616  y = o16 * (32 / (in_mult*w_mult)); y += bias
 *   :623-627  leaky as y/10
 * The reference processes batch item 0 only; the batched semantics here are
 * "B independent B=1 runs" (SURVEY A14).  acc16_out (optional) receives the
 * int16-clamped accumulators as int32. */
static int max_abs_i(int src, int max_val)
{
    if (abs(src) > abs(max_val)) src = (src > 0) ? max_val : -max_val;
    return src;
}

void oracle_conv_int8(const float *in, const int8_t *weights_int8, const float *biases, float *out,
                      int32_t *acc16_out, int batch, int c, int h, int w, int n, int size, int stride,
                      int pad, int act, float input_quant_multipler, float weights_quant_multipler)
{
    const int out_h = (h + 2 * pad - size) / stride + 1;
    const int out_w = (w + 2 * pad - size) / stride + 1;
    const int K = c * size * size;
    const size_t inputs = (size_t)c * h * w;
    int8_t *xq = (int8_t *)malloc(inputs);
    const float ALPHA1 = 32 / (input_quant_multipler * weights_quant_multipler);
    for (int b = 0; b < batch; ++b) {
        const float *im = in + (size_t)b * inputs;
        for (size_t z = 0; z < inputs; ++z) {
            const float t = im[z] * input_quant_multipler;
            /* what gcc/x86-64 emits for `int16_t src = float`: cvttss2si (32-bit) then truncate */
            const int16_t src = (int16_t)(int32_t)t;
            xq[z] = (int8_t)max_abs_i(src, 127);
        }
        float *o = out + (size_t)b * n * out_h * out_w;
        for (int f = 0; f < n; ++f) {
            for (int oy = 0; oy < out_h; ++oy) {
                for (int ox = 0; ox < out_w; ++ox) {
                    int32_t acc = 0;
                    for (int k = 0; k < K; ++k) {
                        const int kx = k % size;
                        const int ky = (k / size) % size;
                        const int ci = k / size / size;
                        const int iy = ky + oy * stride - pad;
                        const int ix = kx + ox * stride - pad;
                        int8_t v = 0;
                        if (iy >= 0 && ix >= 0 && iy < h && ix < w) v = xq[ix + w * (iy + h * ci)];
                        const int16_t a_part = 1 * weights_int8[(size_t)f * K + k];
                        acc += a_part * v;
                    }
                    const int16_t o16 = (int16_t)max_abs_i(acc / 32, 256 * 128 - 1);
                    const size_t oi = ((size_t)f * out_h + oy) * out_w + ox;
                    if (acc16_out) acc16_out[(size_t)b * n * out_h * out_w + oi] = o16;
                    float y = o16 * ALPHA1;
                    y += biases[f];
                    if (act == ACT_LEAKY) y = (y > 0) ? y : y / 10;
                    o[oi] = y;
                }
            }
        }
    }
    free(xq);
}

/* forward_convolutional_layer_cpu, XNOR bit branch (taken iff xnor && stride==1 && pad==1)
 *   src/yolov2_forward_network.c:116-203; bit = (x > 0) (src/additionally.c:132,1354,1544);
 *   im2col'd zero padding is bit 0 => behaves as -1 and IS counted in K;
 *   gemm_nn_custom_bin_mean_transposed  src/additionally.c:1504-1534:
 *       count = #matching bits over the K real positions ; C = (2*count - K) * mean
 *   then += bias (:243-252), activation (:261).  Both the c%32==0 and c%32!=0
 *   sub-branches compute this same function.
 *   weights = BN-fused float weights (sign source); mean_arr from binarize_weights (:113-126). */
void oracle_conv_xnor(const float *in, const float *weights, const float *mean_arr, const float *biases,
                      float *out, int32_t *count_out, int batch, int c, int h, int w, int n, int act)
{
    const int K = c * 9;
    for (int b = 0; b < batch; ++b) {
        const float *im = in + (size_t)b * c * h * w;
        float *o = out + (size_t)b * n * h * w;
        for (int f = 0; f < n; ++f) {
            const float mean_val = mean_arr[f];
            for (int oy = 0; oy < h; ++oy) {
                for (int ox = 0; ox < w; ++ox) {
                    int count = 0;
                    for (int ci = 0; ci < c; ++ci)
                        for (int ky = 0; ky < 3; ++ky)
                            for (int kx = 0; kx < 3; ++kx) {
                                const int iy = oy + ky - 1, ix = ox + kx - 1;
                                int xb = 0;
                                if (iy >= 0 && ix >= 0 && iy < h && ix < w) xb = im[ix + w * (iy + h * ci)] > 0;
                                const int wb = weights[((size_t)f * c + ci) * 9 + ky * 3 + kx] > 0;
                                count += (xb == wb);
                            }
                    const size_t oi = ((size_t)f * h + oy) * w + ox;
                    if (count_out) count_out[(size_t)b * n * h * w + oi] = count;
                    float v = (2 * count - K) * mean_val;
                    v += biases[f];
                    o[oi] = activate(v, act);
                }
            }
        }
    }
}

/* binarize_cpu  src/additionally.c:128-134 (input of an xnor conv on the FP32 fallback path,
 * src/yolov2_forward_network.c:46-49) and binarize_weights :113-126 given the per-filter mean */
void oracle_binarize(const float *in, float *out, size_t n)
{
    for (size_t i = 0; i < n; ++i) out[i] = (in[i] > 0) ? 1 : -1;
}

void oracle_binarize_weights(const float *weights, const float *mean_arr, int n, int size, float *binary)
{
    for (int f = 0; f < n; ++f)
        for (int i = 0; i < size; ++i)
            binary[(size_t)f * size + i] = (weights[(size_t)f * size + i] > 0) ? mean_arr[f] : -mean_arr[f];
}

/* forward_maxpool_layer_avx (scalar)  src/additionally.c:1448-1482: window origin -pad/2 */
void oracle_maxpool(const float *src, float *dst, int size, int w, int h, int out_w, int out_h, int c,
                    int pad, int stride, int batch)
{
    const int w_offset = -pad / 2, h_offset = -pad / 2;
    for (int b = 0; b < batch; ++b)
        for (int k = 0; k < c; ++k)
            for (int i = 0; i < out_h; ++i)
                for (int j = 0; j < out_w; ++j) {
                    const int out_index = j + out_w * (i + out_h * (k + c * b));
                    float max = -FLT_MAX;
                    for (int n = 0; n < size; ++n)
                        for (int m = 0; m < size; ++m) {
                            const int cur_h = h_offset + i * stride + n;
                            const int cur_w = w_offset + j * stride + m;
                            const int index = cur_w + w * (cur_h + h * (k + b * c));
                            const int valid = (cur_h >= 0 && cur_h < h && cur_w >= 0 && cur_w < w);
                            const float val = valid ? src[index] : -FLT_MAX;
                            max = (val > max) ? val : max;
                        }
                    dst[out_index] = max;
                }
}

/* forward_shortcut_layer_cpu / shortcut_cpu  src/yolov2_forward_network.c:444-449 / 410-434
 * (w1,h1,c1) = dims of `add`, (w2,h2,c2) = dims of in/out */
void oracle_shortcut(const float *in, const float *add, float *out, int batch, int w1, int h1, int c1,
                     int w2, int h2, int c2, int act)
{
    const size_t total = (size_t)batch * w2 * h2 * c2;
    memcpy(out, in, total * sizeof(float));
    int stride = w1 / w2, sample = w2 / w1;
    if (stride < 1) stride = 1;
    if (sample < 1) sample = 1;
    const int minw = (w1 < w2) ? w1 : w2, minh = (h1 < h2) ? h1 : h2, minc = (c1 < c2) ? c1 : c2;
    for (int b = 0; b < batch; ++b)
        for (int k = 0; k < minc; ++k)
            for (int j = 0; j < minh; ++j)
                for (int i = 0; i < minw; ++i) {
                    const int out_index = i * sample + w2 * (j * sample + h2 * (k + c2 * b));
                    const int add_index = i * stride + w1 * (j * stride + h1 * (k + c1 * b));
                    out[out_index] += add[add_index];
                }
    for (size_t i = 0; i < total; ++i) out[i] = activate(out[i], act);
}

/* forward_upsample_layer_cpu / upsample_cpu  src/yolov2_forward_network.c:398-407 / 380-395 */
void oracle_upsample(const float *in, float *out, int batch, int c, int h, int w, int stride, float scale)
{
    for (int b = 0; b < batch; ++b)
        for (int k = 0; k < c; ++k)
            for (int j = 0; j < h * stride; ++j)
                for (int i = 0; i < w * stride; ++i) {
                    const int in_index = b * w * h * c + k * w * h + (j / stride) * w + i / stride;
                    const int out_index = b * w * h * c * stride * stride + k * w * h * stride * stride + j * w * stride + i;
                    out[out_index] = scale * in[in_index];
                }
}

/* forward_yolo_layer_cpu  src/yolov2_forward_network.c:453-472 (entry_index src/additionally.c:4200) */
void oracle_yolo(const float *in, float *out, int batch, int n, int classes, int wh)
{
    const int per = 4 + classes + 1;
    const size_t outputs = (size_t)n * per * wh;
    memcpy(out, in, outputs * batch * sizeof(float));
    for (int b = 0; b < batch; ++b)
        for (int a = 0; a < n; ++a) {
            float *p = out + b * outputs + (size_t)a * wh * per;
            for (int i = 0; i < 2 * wh; ++i) p[i] = activate(p[i], ACT_LOGISTIC);
            p += 4 * (size_t)wh;
            for (int i = 0; i < (1 + classes) * wh; ++i) p[i] = activate(p[i], ACT_LOGISTIC);
        }
}

/* forward_region_layer_cpu / softmax_cpu  src/yolov2_forward_network.c:511-575 / 476-492 */
void oracle_region(const float *in, float *out, int batch, int n, int classes, int coords, int wh, int softmax)
{
    const int size = coords + classes + 1;
    const int layers = size * n;
    const size_t outputs = (size_t)layers * wh;
    for (int b = 0; b < batch; ++b)
        for (int c = 0; c < layers; ++c)
            for (int i = 0; i < wh; ++i)
                out[b * outputs + (size_t)i * layers + c] = in[b * outputs + (size_t)c * wh + i];
    for (int b = 0; b < batch; ++b)
        for (int i = 0; i < wh * n; ++i) {
            float *p = out + b * outputs + (size_t)size * i;
            const float x = p[4];
            p[4] = 1.0F / (1.0F + expf(-x));
            if (softmax) {
                float *cl = p + 5;
                float sum = 0, largest = -FLT_MAX;
                for (int k = 0; k < classes; ++k) if (cl[k] > largest) largest = cl[k];
                for (int k = 0; k < classes; ++k) {
                    const float e = expf(cl[k] / 1 - largest / 1);
                    sum += e;
                    cl[k] = e;
                }
                for (int k = 0; k < classes; ++k) cl[k] /= sum;
            }
        }
}

/* forward_region_layer_cpu with l.softmax_tree (YOLO9000): softmax_tree / softmax_cpu per group
 * src/yolov2_forward_network.c:494-507, :556-562; groups as read_tree builds them (src/additionally.c:1895) */
void oracle_region_tree(const float *in, float *out, int batch, int n, int classes, int coords, int wh,
                        const int *group_size, int groups)
{
    const int size = coords + classes + 1;
    const size_t outputs = (size_t)size * n * wh;
    oracle_region(in, out, batch, n, classes, coords, wh, 0);
    for (int b = 0; b < batch; ++b)
        for (int i = 0; i < wh * n; ++i) {
            float *cl = out + b * outputs + (size_t)size * i + coords + 1;
            int count = 0;
            for (int g = 0; g < groups; ++g) {
                const int gs = group_size[g];
                float sum = 0, largest = -FLT_MAX;
                for (int k = count; k < count + gs; ++k) if (cl[k] > largest) largest = cl[k];
                for (int k = count; k < count + gs; ++k) {
                    const float e = expf(cl[k] / 1 - largest / 1);
                    sum += e;
                    cl[k] = e;
                }
                for (int k = count; k < count + gs; ++k) cl[k] /= sum;
                count += gs;
            }
        }
}

/* forward_reorg_layer_cpu  src/yolov2_forward_network.c:337-373 */
void oracle_reorg(const float *x, float *out, int batch, int out_c, int out_h, int out_w, int stride)
{
    const int in_c = out_c / (stride * stride);
    for (int b = 0; b < batch; ++b)
        for (int k = 0; k < out_c; ++k)
            for (int j = 0; j < out_h; ++j)
                for (int i = 0; i < out_w; ++i) {
                    const int in_index = i + out_w * (j + out_h * (k + out_c * b));
                    const int c2 = k % in_c;
                    const int offset = k / in_c;
                    const int w2 = i * stride + offset % stride;
                    const int h2 = j * stride + offset / stride;
                    const int out_index = w2 + out_w * stride * (h2 + out_h * stride * (c2 + in_c * b));
                    out[in_index] = x[out_index];
                }
}

/* Image front end of test_detector_cpu (src/main.c:187-189):
 *   load_image_stb's conversion  src/additionally.c:3095-3103   im[k][j][i] = (float)u8[(j*w+i)*c+k] / 255.
 *   resize_image                 src/additionally.c:3021-3064   two-pass bilinear stretch: rows first
 *     into `part` (w x im.h), then columns; the last column/row copy the edge sample, and the
 *     vertical pass adds its second tap as a separate rounded product.
 * pix = HWC u8 [sh][sw][sc]; out = CHW float [sc][h][w]. */
void oracle_load_resized_u8(const unsigned char *pix, int sw, int sh, int sc, int w, int h, float *out)
{
    float *im = (float *)malloc(sizeof(float) * (size_t)sw * sh * sc);
    float *part = (float *)malloc(sizeof(float) * (size_t)w * sh * sc);
    for (int k = 0; k < sc; ++k)
        for (int j = 0; j < sh; ++j)
            for (int i = 0; i < sw; ++i)
                im[i + sw * j + (size_t)sw * sh * k] = (float)pix[k + sc * i + (size_t)sc * sw * j] / 255.;
    const float w_scale = (float)(sw - 1) / (w - 1);
    const float h_scale = (float)(sh - 1) / (h - 1);
    for (int k = 0; k < sc; ++k)
        for (int r = 0; r < sh; ++r)
            for (int c = 0; c < w; ++c) {
                const float *row = im + (size_t)k * sh * sw + (size_t)r * sw;
                float val;
                if (c == w - 1 || sw == 1) val = row[sw - 1];
                else {
                    const float sx = c * w_scale;
                    const int ix = (int)sx;
                    const float dx = sx - ix;
                    val = (1 - dx) * row[ix] + dx * row[ix + 1];
                }
                part[(size_t)k * sh * w + (size_t)r * w + c] = val;
            }
    for (int k = 0; k < sc; ++k)
        for (int r = 0; r < h; ++r) {
            const float sy = r * h_scale;
            const int iy = (int)sy;
            const float dy = sy - iy;
            float *dst = out + (size_t)k * h * w + (size_t)r * w;
            const float *p0 = part + (size_t)k * sh * w + (size_t)iy * w;
            for (int c = 0; c < w; ++c) dst[c] = (1 - dy) * p0[c];
            if (r == h - 1 || sh == 1) continue;
            for (int c = 0; c < w; ++c) dst[c] += dy * p0[w + c];
        }
    free(im);
    free(part);
}

/* entropy_calibration  src/yolov2_forward_network_quantized.c:1292-1400 -- the INT8 input-multiplier
 * search of the calibration tool (network_calibrate_cpu, src/yolov2_forward_network.c:784-786, calls
 * it with bin_width 1/16 and 4096 bins on every conv layer's input):
 *   H[b] = number of elements with lround(|x| / bin_width) == b (saturated at max_bin-1), float counts;
 *   for i = 128 .. max_bin-1: P = H[0..i) with the tail mass added to P[i-1]; Q = P squeezed into
 *   128 bins and expanded back (empty P bins stay empty); m[i] = KL(P || Q); the best i gives
 *   threshold = (i + 0.5) * bin_width and multiplier = 127 / threshold.
 * Types and evaluation order are the reference's (float accumulators, the uint64 outlier count that
 * goes through float on every add, log in double). */
float oracle_entropy_calibration(const float *src, size_t size, float bin_width, int max_bin)
{
    float *m_array = (float *)calloc(max_bin, sizeof(float));
    float *H = (float *)calloc(max_bin, sizeof(float));
    float *P = (float *)calloc(max_bin, sizeof(float));
    float *Q = (float *)calloc(max_bin, sizeof(float));
    float qQ[128];
    uint64_t qcount[128];
    {
        const int last_bin = max_bin - 1;
        for (size_t j = 0; j < size; ++j) {
            const int bin = (int)lround(fabs(src[j]) / bin_width);
            H[bin >= last_bin ? last_bin : bin]++;
        }
    }
    for (int i = 128; i < max_bin; ++i) {
        uint64_t outliers = 0;
        const int last_bin = i - 1;
        for (int j = 0; j < max_bin; ++j) {
            if (j <= last_bin) P[j] = H[j];
            else outliers += H[j];                 /* uint64 + float: evaluated in float, stored back */
        }
        const float expand = i / 128.0F;
        for (int j = 0; j < 128; ++j) { qQ[j] = 0; qcount[j] = 0; }
        for (int j = 0; j < i; ++j) {
            int qb = (int)lround(j / expand);
            if (qb > 127) qb = 127;
            qQ[qb] += P[j];
            if (P[j] != 0) qcount[qb]++;
        }
        for (int j = 0; j < i; ++j) Q[j] = 0;
        for (int j = 0; j < i; ++j) {
            int qb = (int)lround(j / expand);
            if (qb > 127) qb = 127;
            if (P[j] != 0) Q[j] = qQ[qb] / qcount[qb];
        }
        P[last_bin] += outliers;
        float sum_P = 0, sum_Q = 0;
        for (int j = 0; j < i; ++j) { sum_P += P[j]; sum_Q += Q[j]; }
        for (int j = 0; j < i; ++j) { P[j] /= sum_P; Q[j] /= sum_Q; }
        for (int j = 0; j < i; ++j) m_array[i] += P[j] * (log((P[j] + FLT_MIN) / (Q[j] + FLT_MIN)));
    }
    float m_index = 128, min_m = FLT_MAX;
    for (int i = 128; i < max_bin; ++i)
        if (m_array[i] < min_m) { min_m = m_array[i]; m_index = i; }
    const float threshold = (m_index + 0.5) * bin_width;
    const float multiplier = 127 / threshold;
    free(H); free(P); free(Q); free(m_array);
    return multiplier;
}

/* yolov2_fuse_conv_batchnorm  src/additionally.c:67-109 (epsilon outside the sqrt) */
void oracle_fuse_bn(float *weights, float *biases, const float *scales, const float *mean, const float *var,
                    int n, int filter_size)
{
    for (int f = 0; f < n; ++f) {
        biases[f] = biases[f] - scales[f] * mean[f] / (sqrtf(var[f]) + .000001f);
        for (int i = 0; i < filter_size; ++i) {
            const size_t wi = (size_t)f * filter_size + i;
            weights[wi] = weights[wi] * scales[f] / (sqrtf(var[f]) + .000001f);
        }
    }
}

/* binarize_weights + get_mean_array  src/additionally.c:113-126, 188-194 */
void oracle_binary_mean(const float *weights, int n, int size, float *mean_arr)
{
    for (int f = 0; f < n; ++f) {
        float mean = 0;
        for (int i = 0; i < size; ++i) mean += fabs(weights[(size_t)f * size + i]);
        mean = mean / size;
        const float bw0 = (weights[(size_t)f * size] > 0) ? mean : -mean;
        mean_arr[f] = fabs(bw0);
    }
}

/* get_multiplier(arr, size, 8)/4 and the weight quantisation loop
 * src/yolov2_forward_network_quantized.c:35-87, 1429-1446 */
float oracle_quantize_weights(const float *weights, size_t size, int8_t *weights_int8)
{
    const int number_of_ranges = 32;
    const float start_range = 1.F / 65536;
    int count[32];
    memset(count, 0, sizeof(count));
    for (size_t i = 0; i < size; ++i) {
        const float w = weights[i];
        float cur_range = start_range;
        for (int j = 0; j < number_of_ranges; ++j) {
            if (fabs(cur_range) <= w && w < fabs(cur_range * 2)) count[j]++;
            cur_range *= 2;
        }
    }
    int max_count_range = 0, index_max_count = 0;
    for (int j = 0; j < number_of_ranges; ++j) {
        int counter = 0;
        for (int i = j; i < (j + 8) && i < number_of_ranges; ++i) counter += count[i];
        if (max_count_range < counter) { max_count_range = counter; index_max_count = j; }
    }
    const float multiplier = 1 / (start_range * powf(2., (float)index_max_count));
    const float wm = multiplier / 4;
    for (size_t i = 0; i < size; ++i) {
        const float w = weights[i] * wm;
        weights_int8[i] = max_abs_i(w, 127);
    }
    return wm;
}
