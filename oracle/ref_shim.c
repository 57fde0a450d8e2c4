/*
 * oracle/ref_shim.c -- TEST INFRASTRUCTURE, never part of the product path.
 *
 * A thin handle-based wrapper that is compiled TOGETHER WITH the unmodified
 * reference sources (where they lie, /root/reference/src/{additionally,box,
 * yolov2_forward_network,yolov2_forward_network_quantized}.c) into
 * oracle/_ref/libyolo2ref*.so by oracle/Makefile.  It lets the python tests
 * and bench.py's cpu_baseline leg drive the *real* reference CPU path through
 * ctypes without knowing the layout of `network` / `layer`
 * (src/additionally.h:409-763), which is why it includes the reference header
 * at build time instead of restating the structs.
 *
 * Call sequence mirrored from test_detector_cpu (src/main.c:156-229):
 *   parse_network_cfg -> load_weights_upto_cpu -> yolov2_fuse_conv_batchnorm
 *   -> calculate_binary_weights -> [quantinization_and_get_multipliers]
 *   -> network_predict_cpu | network_predict_quantized
 *   -> get_network_boxes -> do_nms_sort
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
#include <time.h>

#include "additionally.h"
#include "box.h"

extern int gpu_index;                         /* src/additionally.c:22 */
void calculate_binary_weights(network net);   /* src/additionally.c:306 */
void do_nms_sort(detection *dets, int total, int classes, float thresh); /* src/box.c:296 */

typedef struct ref_net {
    network net;
    int quantized;
} ref_net;

/* the reference prints a table per parse and one line per conv per quantized
 * inference (src/yolov2_forward_network_quantized.c:1039): silence fd 1/2. */
static int g_quiet = 1;
static int saved_out = -1, saved_err = -1;
static void hush(void)
{
    if (!g_quiet) return;
    fflush(stdout); fflush(stderr);
    int nul = open("/dev/null", O_WRONLY);
    saved_out = dup(1); saved_err = dup(2);
    dup2(nul, 1); dup2(nul, 2);
    close(nul);
}
static void unhush(void)
{
    if (!g_quiet) return;
    fflush(stdout); fflush(stderr);
    dup2(saved_out, 1); dup2(saved_err, 2);
    close(saved_out); close(saved_err);
}

void ref_set_quiet(int q) { g_quiet = q; }

ref_net *ref_load(const char *cfg, const char *weights, int batch, int quantized)
{
    ref_net *r = (ref_net *)calloc(1, sizeof(ref_net));
    gpu_index = -1;                      /* only main() does this */
    hush();
    r->net = parse_network_cfg((char *)cfg, batch, quantized);
    if (weights && weights[0]) load_weights_upto_cpu(&r->net, (char *)weights, r->net.n);
    yolov2_fuse_conv_batchnorm(r->net);
    calculate_binary_weights(r->net);
    if (quantized) quantinization_and_get_multipliers(r->net);
    unhush();
    r->quantized = quantized;
    return r;
}

float *ref_predict(ref_net *r, float *input)
{
    float *out;
    hush();
    if (r->quantized) out = network_predict_quantized(r->net, input);
    else out = network_predict_cpu(r->net, input);
    unhush();
    return out;
}

/* wall-clock seconds for `iters` predictions (cpu_baseline leg of bench.py) */
double ref_time_predict(ref_net *r, float *input, int iters)
{
    struct timespec t0, t1;
    int i;
    hush();
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (i = 0; i < iters; ++i) {
        if (r->quantized) network_predict_quantized(r->net, input);
        else network_predict_cpu(r->net, input);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    unhush();
    return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}

int ref_num_layers(ref_net *r) { return r->net.n; }
int ref_batch(ref_net *r) { return r->net.batch; }
int ref_net_w(ref_net *r) { return r->net.w; }
int ref_net_h(ref_net *r) { return r->net.h; }
int ref_net_c(ref_net *r) { return r->net.c; }

/* info[24]: type, batch, w, h, c, n, size, stride, pad, out_w, out_h, out_c,
 *           outputs, inputs, activation, xnor, quantized, index, classes,
 *           coords, total, softmax, new_lda, batch_normalize */
void ref_layer_info(ref_net *r, int i, int *info)
{
    layer *l = &r->net.layers[i];
    info[0] = l->type;   info[1] = l->batch;  info[2] = l->w;      info[3] = l->h;
    info[4] = l->c;      info[5] = l->n;      info[6] = l->size;   info[7] = l->stride;
    info[8] = l->pad;    info[9] = l->out_w;  info[10] = l->out_h; info[11] = l->out_c;
    info[12] = l->outputs; info[13] = l->inputs; info[14] = l->activation;
    info[15] = l->xnor;  info[16] = l->quantized; info[17] = l->index;
    info[18] = l->classes; info[19] = l->coords; info[20] = l->total;
    info[21] = l->softmax; info[22] = (int)l->new_lda; info[23] = l->batch_normalize;
}

float *ref_layer_output(ref_net *r, int i) { return r->net.layers[i].output; }
float *ref_layer_weights(ref_net *r, int i) { return r->net.layers[i].weights; }
float *ref_layer_biases(ref_net *r, int i) { return r->net.layers[i].biases; }
int8_t *ref_layer_weights_int8(ref_net *r, int i) { return r->net.layers[i].weights_int8; }
float ref_layer_input_mult(ref_net *r, int i) { return r->net.layers[i].input_quant_multipler; }
float ref_layer_weights_mult(ref_net *r, int i) { return r->net.layers[i].weights_quant_multipler; }
unsigned char *ref_layer_bit_weights(ref_net *r, int i) { return (unsigned char *)r->net.layers[i].align_bit_weights; }
int ref_layer_bit_weights_size(ref_net *r, int i) { return (int)r->net.layers[i].align_bit_weights_size; }
float *ref_layer_mean_arr(ref_net *r, int i) { return r->net.layers[i].mean_arr; }
float *ref_layer_binary_weights(ref_net *r, int i) { return r->net.layers[i].binary_weights; }
int *ref_layer_route_inputs(ref_net *r, int i) { return r->net.layers[i].input_layers; }
int *ref_layer_mask(ref_net *r, int i) { return r->net.layers[i].mask; }
float *ref_layer_anchors(ref_net *r, int i) { return r->net.layers[i].biases; }

/* Detections exactly as main.c:228-229 obtains them.  The reference decodes
 * batch item 0 only (src/additionally.c:4213,4338); to get image `b` of a
 * batched run the YOLO/REGION outputs of item b are temporarily aliased into
 * the item-0 position (pointer swap only, no arithmetic is changed).
 * Row layout of `out` (stride 6+classes): x y w h objectness sort_class prob[classes].
 * Returns number of detections written (after do_nms_sort if nms > 0). */
int ref_get_detections(ref_net *r, int b, int w, int h, float thresh, float nms,
                       int relative, float *out, int max_rows, int *classes_out)
{
    network *net = &r->net;
    int n = net->n, i, j, nboxes = 0;
    float **saved = (float **)calloc(n, sizeof(float *));
    for (i = 0; i < n; ++i) {
        layer *l = &net->layers[i];
        saved[i] = l->output;
        if (l->type == YOLO || l->type == REGION) l->output = l->output + (size_t)b * l->outputs;
    }
    layer last = net->layers[n - 1];
    detection *dets = get_network_boxes(net, w, h, thresh, .5f, 0, relative, &nboxes, 0);
    if (nms > 0) do_nms_sort(dets, nboxes, last.classes, nms);
    int classes = last.classes;
    if (classes_out) *classes_out = classes;
    int stride = 6 + classes;
    int rows = nboxes < max_rows ? nboxes : max_rows;
    for (i = 0; i < rows; ++i) {
        float *o = out + (size_t)i * stride;
        o[0] = dets[i].bbox.x; o[1] = dets[i].bbox.y; o[2] = dets[i].bbox.w; o[3] = dets[i].bbox.h;
        o[4] = dets[i].objectness; o[5] = (float)dets[i].sort_class;
        for (j = 0; j < classes; ++j) o[6 + j] = dets[i].prob[j];
    }
    free_detections(dets, nboxes);
    for (i = 0; i < n; ++i) net->layers[i].output = saved[i];
    free(saved);
    return nboxes;
}

/* direct access to single reference ops for op-level pinning of the restatement */
void forward_maxpool_layer_avx(float *src, float *dst, int *indexes, int size, int w, int h, int out_w, int out_h, int c,
    int pad, int stride, int batch);          /* src/additionally.c:1448 */

void ref_maxpool(float *src, float *dst, int size, int w, int h, int out_w, int out_h, int c, int pad, int stride, int batch)
{
    int *idx = (int *)calloc((size_t)out_w * out_h * c * batch, sizeof(int));
    forward_maxpool_layer_avx(src, dst, idx, size, w, h, out_w, out_h, c, pad, stride, batch);
    free(idx);
}

/* entropy_calibration (src/yolov2_forward_network_quantized.c:1292), silenced */
float ref_entropy_calibration(float *src, size_t size, float bin_width, int max_bin)
{
    float m;
    hush();
    m = entropy_calibration(src, size, bin_width, max_bin);
    unhush();
    return m;
}

/* the reference's image front end exactly as test_detector_cpu drives it (src/main.c:187-189):
 * load_image(path, 0, 0, 3) [stb decode + HWC u8 -> CHW float /255., src/additionally.c:3068-3106]
 * then resize_image(im, w, h) [src/additionally.c:3021-3064].  out = float[3*h*w].
 * returns 0, or -1 if the file could not be decoded (the reference would exit(0)). */
int ref_load_resized(const char *path, int w, int h, float *out, int *src_w, int *src_h)
{
    FILE *f = fopen(path, "rb");
    image im, sized;
    if (!f) return -1;
    fclose(f);
    im = load_image((char *)path, 0, 0, 3);
    if (src_w) *src_w = im.w;
    if (src_h) *src_h = im.h;
    sized = resize_image(im, w, h);
    memcpy(out, sized.data, sizeof(float) * 3 * (size_t)w * h);
    free_image(im);
    free_image(sized);
    return 0;
}

#ifdef WITH_HIP_ADAPTOR
/* the drop-in under test: the reference's host code + integration/network_predict_hip.c
 * (the binding a maintainer would add) driving libyolo2hip.so */
float *network_predict_hip(network net, float *input);
void free_network_hip(void);

float *ref_predict_hip(ref_net *r, float *input)
{
    float *out;
    hush();
    out = network_predict_hip(r->net, input);
    unhush();
    return out;
}

void ref_free_hip(void) { free_network_hip(); }
#endif
