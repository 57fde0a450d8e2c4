/*
 * integration/network_predict_hip.c -- the reference-side binding of libyolo2hip.so.
 *
 * This is the file a maintainer of AlexeyAB/yolo2_light adds to src/ (plus one
 * `#elif defined(HIP)` arm at the three call sites src/main.c:199-219, :394-414,
 * src/additionally.c:4639-4659).  It is compiled TOGETHER WITH the reference's own
 * host code, built WITHOUT -DGPU so `layer` (952 B) and `network` (224 B) keep their CPU
 * layout (src/additionally.h:409-763), and links against libyolo2hip.so only through the
 * plain-C ABI of include/yolo2_hip.h.
 *
 *     float *network_predict_hip(network net, float *input);
 *
 * has exactly the contract of network_predict_cpu (src/yolov2_forward_network.c:632-646)
 * and network_predict_gpu_cudnn (src/yolov2_forward_network_gpu.cu:547-573):
 *   - `net` by value, `input` = host float[net.batch*net.c*net.h*net.w], CHW, [0,1];
 *   - returns net.layers[last non-COST].output (a borrowed host pointer);
 *   - fills the host `l.output` of every YOLO/REGION layer so that get_network_boxes
 *     (src/additionally.c:4403) and do_nms_sort (src/box.c:296) run unchanged;
 *   - `-quantized` follows net.quantized, XNOR follows l.xnor, using the weights the host
 *     prepared in main.c:160-171 (fused BN, weights_int8 + multipliers, mean_arr).
 * Errors follow the reference convention: error() = perror + exit (src/additionally.c:1595).
 *
 * Several GPUs: the reference's CLI selects one device with `-i <n>` (gpu_index, src/main.c:653-661).
 * With the environment variable YL_GPUS set ("all", or a comma list such as "0,1,2,3") the SAME call
 * splits net.batch over those devices of the node (yl_group_*: one host thread + stream per device,
 * weights replicated) -- l.output of every YOLO/REGION layer and the returned pointer cover the whole
 * batch exactly as on one device, so get_network_boxes / do_nms_sort still run unchanged.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "additionally.h"
#include "yolo2_hip.h"

extern int gpu_index;            /* src/additionally.c:22; -i <n> selects the device (main.c:653) */
void error(const char *s);       /* src/additionally.c:1595 */

static yl_network *g_hip_net = NULL;
static yl_group *g_hip_group = NULL;    /* set instead of a device image of g_hip_net when YL_GPUS is given */
static layer *g_hip_owner = NULL;       /* the network whose device image g_hip_net / g_hip_group holds */

static void hip_fail(const char *what)
{
    fprintf(stderr, "%s: %s\n", what, yl_last_error());
    error(what);
}

static yl_network *hip_build(network *net)
{
    int i;
    yl_network *h = NULL;
    if (yl_abi_version() != YL_ABI_VERSION) {       /* an older / newer libyolo2hip.so behind the same symbol names */
        fprintf(stderr, "libyolo2hip.so implements C-ABI version %d, this adaptor was built against %d\n",
                yl_abi_version(), YL_ABI_VERSION);
        error("network_predict_hip: ABI version mismatch");
    }
    yl_layer_desc *d = (yl_layer_desc *)calloc(net->n, sizeof(yl_layer_desc));
    for (i = 0; i < net->n; ++i) {
        layer *l = &net->layers[i];
        d[i].type = l->type;
        d[i].activation = l->activation;
        d[i].batch = l->batch; d[i].w = l->w; d[i].h = l->h; d[i].c = l->c;
        d[i].n = l->n; d[i].size = l->size; d[i].stride = l->stride; d[i].pad = l->pad;
        d[i].out_w = l->out_w; d[i].out_h = l->out_h; d[i].out_c = l->out_c;
        d[i].outputs = l->outputs; d[i].inputs = l->inputs;
        d[i].batch_normalize = l->batch_normalize;
        d[i].xnor = l->xnor;
        d[i].quantized = l->quantized;
        d[i].index = l->index;
        d[i].input_layers = l->input_layers; d[i].input_sizes = l->input_sizes;
        d[i].classes = l->classes; d[i].coords = l->coords; d[i].total = l->total; d[i].softmax = l->softmax;
        d[i].mask = l->mask;
        d[i].anchors = (l->type == YOLO || l->type == REGION) ? l->biases : NULL;
        d[i].scale = l->scale;
        if (l->type == CONVOLUTIONAL) {
            d[i].weights = l->weights; d[i].biases = l->biases;
            d[i].scales = l->scales; d[i].rolling_mean = l->rolling_mean; d[i].rolling_variance = l->rolling_variance;
            if (net->quantized) {
                d[i].weights_int8 = l->weights_int8;
                d[i].input_quant_multipler = l->input_quant_multipler;
                d[i].weights_quant_multipler = l->weights_quant_multipler;
            }
            d[i].mean_arr = l->xnor ? l->mean_arr : NULL;
        }
        d[i].output = l->output;
        if (l->type == REGION && l->softmax_tree) {
            d[i].tree_n = l->softmax_tree->n; d[i].tree_groups = l->softmax_tree->groups;
            d[i].tree_parent = l->softmax_tree->parent; d[i].tree_group_size = l->softmax_tree->group_size;
        }
    }
    if (yl_network_create_from_desc(d, net->n, net->batch, net->w, net->h, net->c, net->quantized,
                                    net->input_calibration, net->input_calibration_size, &h) != YL_OK)
        hip_fail("yl_network_create_from_desc");
    free(d);
    /* opt-in: bf16 operands for the FP32 convolutions (outside the FP32 path's 1e-4 contract) */
    {
        const char *bf = getenv("YL_BF16");
        if (bf && *bf && *bf != '0' && yl_network_set_precision(h, YL_PRECISION_BF16) != YL_OK) hip_fail("yl_network_set_precision");
    }
    /* conv+[shortcut] epilogue fusion and quantise-on-store: bit-identical head tensors, fewer kernels */
    if (yl_network_set_fusion(h, 1) != YL_OK) hip_fail("yl_network_set_fusion");
    {
        const char *gpus = getenv("YL_GPUS");
        int devs[64], n = 0;
        if (gpus && *gpus) {
            if (!strcmp(gpus, "all")) { for (n = 0; n < yl_device_count() && n < 64; ++n) devs[n] = n; }
            else { const char *p = gpus; while (*p && n < 64) { devs[n++] = atoi(p); p = strchr(p, ','); if (!p) break; ++p; } }
            if (n > net->batch) n = net->batch;          /* at least one image per device */
        }
        if (n > 1) {
            if (yl_group_create(h, devs, n, &g_hip_group) != YL_OK) hip_fail("yl_group_create");
            return h;                                    /* kept as the host model; the replicas own the devices */
        }
    }
    if (yl_network_to_device(h, gpu_index >= 0 ? gpu_index : 0) != YL_OK) hip_fail("yl_network_to_device");
    return h;
}

void free_network_hip(void);

float *network_predict_hip(network net, float *input)
{
    float *out;
    if (!g_hip_net || g_hip_owner != net.layers) {
        free_network_hip();
        g_hip_net = hip_build(&net);
        g_hip_owner = net.layers;
    }
    out = g_hip_group ? yl_group_predict(g_hip_group, input) : yl_network_predict(g_hip_net, input);
    if (!out) hip_fail("yl_network_predict");
    return out;
}

/* call before free_network(net) (src/additionally.c:2054) */
void free_network_hip(void)
{
    if (g_hip_group) yl_group_destroy(g_hip_group);
    g_hip_group = NULL;
    if (g_hip_net) yl_network_destroy(g_hip_net);
    g_hip_net = NULL;
    g_hip_owner = NULL;
}
