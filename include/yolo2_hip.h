/*
 * yolo2_hip.h -- C-ABI of libyolo2hip.so: the MI355X (gfx950) drop-in for the
 * device half of AlexeyAB/yolo2_light's inference hot path.
 *
 * Plain C, plain pointers and sizes only.  Every entry point names the
 * reference interface it replaces (paths relative to the reference tree).
 * The reference has no plugin registry: its back-end boundary is the
 * `network_predict_*` family (src/additionally.h:907-969) plus the one-time
 * model-prep passes the caller runs before it (src/main.c:160-171).  A
 * maintainer binds this library with the ~60-line adaptor shown in
 * INTEGRATION.md (`network_predict_hip(network, float*)`), built WITHOUT
 * -DGPU so `layer`/`network` keep their CPU layout.
 *
 * All functions return 0 on success and a negative code on failure unless
 * stated otherwise; yl_last_error() returns a thread-local message.  Nothing
 * here falls back to a CPU implementation: device entry points fail with
 * YL_ERR_DEVICE when no gfx950 device/kernels are available.
 */
#ifndef YOLO2_HIP_H
#define YOLO2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YL_OK            0
#define YL_ERR_ARG      -1
#define YL_ERR_IO       -2
#define YL_ERR_CFG      -3
#define YL_ERR_DEVICE   -4
#define YL_ERR_UNSUPPORTED -5
#define YL_ERR_STATE    -6

/* numeric values == reference LAYER_TYPE (src/additionally.h:376-403) so an
 * adaptor can pass `l.type` straight through */
enum {
    YL_CONVOLUTIONAL = 0, YL_MAXPOOL = 3, YL_ROUTE = 8, YL_SHORTCUT = 13,
    YL_REGION = 21, YL_YOLO = 22, YL_UPSAMPLE = 23, YL_REORG = 24, YL_BLANK = 25
};
/* numeric values == reference ACTIVATION (src/additionally.h:68-70) */
enum {
    YL_LOGISTIC = 0, YL_RELU = 1, YL_RELIE = 2, YL_LINEAR = 3, YL_RAMP = 4, YL_TANH = 5, YL_PLSE = 6, YL_LEAKY = 7,
    YL_ELU = 8, YL_LOGGY = 9, YL_STAIR = 10, YL_HARDTAN = 11, YL_LHTAN = 12
};

typedef struct yl_network yl_network;   /* opaque; owns host model + all device memory */

/* One layer as the reference's `layer` (src/additionally.h:409-684) describes
 * it, reduced to the fields the hot path reads.  Host pointers are only read
 * during yl_network_create_from_desc (a snapshot is taken). */
typedef struct yl_layer_desc {
    int type;                 /* l.type */
    int activation;           /* l.activation */
    int batch, w, h, c;       /* l.batch, l.w, l.h, l.c */
    int n;                    /* conv: filters; route: #inputs; yolo/region: #anchors of this head */
    int size, stride, pad;    /* l.size, l.stride, l.pad */
    int out_w, out_h, out_c;  /* l.out_w, l.out_h, l.out_c */
    int outputs, inputs;      /* l.outputs, l.inputs (per image) */
    int batch_normalize;      /* 0 once yolov2_fuse_conv_batchnorm has run */
    int xnor;                 /* l.xnor */
    int quantized;            /* l.quantized: the parser's GPU-path quantisation flag (src/additionally.c:3557-3559);
                                 only read under YL_QUANT_RULE_GPU */
    int index;                /* shortcut: absolute index of the `from` layer (l.index) */
    const int *input_layers;  /* route: l.input_layers[n] */
    const int *input_sizes;   /* route: l.input_sizes[n]  */
    int classes, coords, total, softmax;   /* yolo / region */
    const int *mask;          /* yolo: l.mask[n] */
    const float *anchors;     /* yolo: l.biases[2*total]; region: l.biases[2*n] */
    float scale;              /* upsample: l.scale */
    const float *weights;     /* conv: l.weights[n*c*size*size] */
    const float *biases;      /* conv: l.biases[n] */
    const float *scales, *rolling_mean, *rolling_variance;   /* conv BN (NULL if fused) */
    const int8_t *weights_int8;        /* conv: l.weights_int8 (NULL -> computed by yl_network_quantize) */
    float input_quant_multipler;       /* l.input_quant_multipler  */
    float weights_quant_multipler;     /* l.weights_quant_multipler */
    const float *mean_arr;             /* conv xnor: l.mean_arr[n] (NULL -> computed) */
    float *output;            /* host l.output[batch*outputs]: filled by yl_network_predict for
                                 YOLO/REGION layers and the last layer, like the reference GPU
                                 path does (src/yolov2_forward_network_gpu.cu:404-419,438,570) */
    /* region with l.softmax_tree (YOLO9000, src/additionally.h:352-364, read_tree src/additionally.c:1895):
     * tree_n = t->n (0 = no tree), tree_groups = t->groups, tree_parent = t->parent[tree_n],
     * tree_group_size = t->group_size[tree_groups] */
    int tree_n, tree_groups;
    const int *tree_parent, *tree_group_size;
} yl_layer_desc;

/* ABI version of this header.  Bumped whenever the signature or meaning of an existing entry point changes (a
 * caller built against an older header would still LINK against the same symbol name and pass garbage): 4 = round 4,
 * the first numbered one (round 3 had inserted mask_cap / anchors_cap into yl_network_layer_head without a marker).
 * Bindings compare yl_abi_version() with the YL_ABI_VERSION they were written against at load time
 * (integration/network_predict_hip.c, yolo2_light_amd/_lib.py) and refuse to run on a mismatch. */
#define YL_ABI_VERSION 4
int yl_abi_version(void);

/* thread-local human-readable message of the last failure */
const char *yl_last_error(void);

/* number of visible HIP devices (0 when there is no GPU); replaces nothing,
 * the reference selects the device with `-i` -> cuda_set_device (src/main.c:653-661) */
int yl_device_count(void);

/* wait for every stream of `device` (-1 = all visible devices): cudaDeviceSynchronize of the reference's GPU path
 * (src/yolov2_forward_network_gpu.cu uses it around its timers) */
int yl_device_synchronize(int device);

/* ------------------------------------------------------------------ *
 *  Host-side model build (L1 of the reference).  When the reference's
 *  own parser is the caller, use yl_network_create_from_desc instead.
 * ------------------------------------------------------------------ */

/* parse_network_cfg(filename, batch, quantized)       src/additionally.c:3955 */
int yl_network_create_from_cfg(const char *cfg_path, int batch, int quantized, yl_network **out);

/* Build from already-parsed (and possibly already-prepared) reference layers:
 * what the adaptor in INTEGRATION.md calls after main.c:160-171 has run.
 * input_calibration = net.input_calibration (src/additionally.c:3870-3885). */
int yl_network_create_from_desc(const yl_layer_desc *layers, int n_layers,
                                int batch, int w, int h, int c, int quantized,
                                const float *input_calibration, int input_calibration_size,
                                yl_network **out);

/* load_weights_upto_cpu(&net, filename, net.n)        src/additionally.c:3491 */
int yl_network_load_weights(yl_network *net, const char *weights_path);

/* load_weights_upto_cpu(&net, filename, cutoff) with the reference's tolerance: only layers [0, cutoff) are read and
 * a file that ends early is NOT an error (the reference never checks fread; backbone-only files load, the rest
 * keeps its initial values).  yl_network_load_weights above is strict: a short file is YL_ERR_IO. */
int yl_network_load_weights_upto(yl_network *net, const char *weights_path, int cutoff);

/* yolov2_fuse_conv_batchnorm(net)                     src/additionally.c:67 */
int yl_network_fuse_conv_batchnorm(yl_network *net);

/* calculate_binary_weights(net)                       src/additionally.c:306 */
int yl_network_calculate_binary_weights(yl_network *net);

/* quantinization_and_get_multipliers(net)             src/yolov2_forward_network_quantized.c:1402 */
int yl_network_quantize(yl_network *net);

/* The three passes above on the GPU: yolov2_fuse_conv_batchnorm, calculate_binary_weights' mean_arr
 * and -- for a network created with quantized != 0 -- quantinization_and_get_multipliers, in that order, with
 * explicitly rounded device arithmetic: folded weights / biases, mean_arr, weights_int8 and both multipliers are
 * bit-identical to the host passes (csrc/prep.hip).  Before yl_network_to_device; needs loaded weights. */
int yl_network_prepare_on_device(yl_network *net, int device);

/* Which convolutions `quantized` applies to.  The reference has two rules:
 *   YL_QUANT_RULE_CPU (default)  yolov2_forward_network_q: every conv with index >= 1 and a non-linear
 *                                activation (src/yolov2_forward_network_quantized.c:1036)
 *   YL_QUANT_RULE_GPU            forward_network_gpu_cudnn_quantized: the parser's `l.quantized` --
 *                                not the first layer, not linear, not 1x1, not strided beyond layer 1, and
 *                                nothing from the convolution two sections before the first [yolo] on
 *                                (src/additionally.c:3557-3559, 3996-4004).  The arithmetic of a quantised
 *                                layer is the CPU path's either way (that is the parity oracle).
 * Call before yl_network_to_device. */
#define YL_QUANT_RULE_CPU 0
#define YL_QUANT_RULE_GPU 1
int yl_network_set_quant_rule(yl_network *net, int rule);

/* free_network(net)                                   src/additionally.c:2054 */
void yl_network_destroy(yl_network *net);

/* ---- introspection (used by the parity tests and bench.py) ---- */
int yl_network_num_layers(const yl_network *net);
int yl_network_batch(const yl_network *net);
/* dims[3] = {w, h, c} of the network input */
int yl_network_input_dims(const yl_network *net, int *dims);
/* info[24]: type, batch, w, h, c, n, size, stride, pad, out_w, out_h, out_c,
 *           outputs, inputs, activation, xnor, quantized(int8 used), index,
 *           classes, coords, total, softmax, reserved, batch_normalize */
int yl_network_layer_info(const yl_network *net, int i, int *info);
/* YOLO/REGION layer i: mask[n] (yolo: l.mask; region: 0..n-1) and anchors[2*total] (l.biases);
 * either pointer may be NULL; mask_cap / anchors_cap = elements the caller's arrays hold, never more is written
 * (YL_ERR_ARG when a non-NULL array is too small).  Returns n (anchors of this head), < 0 on error;
 * anchors_len_out (may be NULL) receives the number of anchor floats the layer has. */
int yl_network_layer_head(const yl_network *net, int i, int *mask, int mask_cap, float *anchors, int anchors_cap,
                          int *anchors_len_out);
/* softmax tree of REGION layer i (l.softmax_tree, src/additionally.h:352-364): returns the number of groups
 * (0 = the layer has no tree, <0 on error); parent[classes] / group_size[groups] are copied when non-NULL */
int yl_network_layer_tree(const yl_network *net, int i, int *parent, int *group_size);
/* borrowed host pointers to the prepared parameters of conv layer i (NULL if absent) */
const float  *yl_network_layer_weights(const yl_network *net, int i);
const float  *yl_network_layer_biases(const yl_network *net, int i);
const int8_t *yl_network_layer_weights_int8(const yl_network *net, int i);
const float  *yl_network_layer_mean_arr(const yl_network *net, int i);
/* mult[2] = {input_quant_multipler, weights_quant_multipler} */
int yl_network_layer_quant_multipliers(const yl_network *net, int i, float *mult);
/* ALGORITHMIC HBM bytes layer i's kernels move per forward of the whole batch under the current fusion plan
 * (after yl_network_to_device): bytes[0] read, bytes[1] written -- every operand once, in the form it is stored
 * (FP32 NCHW, int8 NC/16HW16, sign words); what bench.py's roofline divides by the measured launch time */
int yl_network_layer_traffic(const yl_network *net, int i, double *bytes);
/* per-image FLOPs of the conv layers: sum 2*n*size^2*c*out_h*out_w (src/additionally.c:2903) */
double yl_network_flops_per_image(const yl_network *net);

/* ------------------------------------------------------------------ *
 *  Device side (replaces L3 device variant + L2 device kernels + the
 *  device half of L0 of the reference).
 * ------------------------------------------------------------------ */

/* Upload parameters and plan all activation buffers in HBM on `device`.
 * Replaces cuda_set_device + cuda_make_array/push_convolutional_layer calls
 * scattered through make_*_layer / fuse / binary_align_weights
 * (src/gpu.cu:97-266, src/additionally.c:92-96,281-299,2819-2881) and
 * init_gpu_int8x4 (src/yolov2_forward_network_gpu.cu). */
int yl_network_to_device(yl_network *net, int device);

/* Call BEFORE yl_network_to_device.  on = 1: a same-shape linear [shortcut] whose input is
 * the convolution right before it is folded into that convolution's epilogue
 * (shortcut.out = act(conv) + layers[from].out, bit-identical to the two-kernel result), as the
 * reference GPU path does for XNOR convs (calculate_binary_weights, src/additionally.c:326-339).
 * The folded conv's own output tensor is then NOT materialised (only legal -- and only done --
 * when no route/shortcut references it), so per-layer readback of that conv is undefined;
 * YOLO/REGION/route/shortcut outputs are unaffected.  Default off; ignored in debug mode. */
int yl_network_set_fusion(yl_network *net, int on);

/* float *network_predict_gpu_cudnn[_quantized](network net, float *input)
 *                                         src/yolov2_forward_network_gpu.cu:547,576
 * Same contract as network_predict_cpu (src/yolov2_forward_network.c:632):
 * `input` = host float[batch*c*h*w] CHW in [0,1]; returns a borrowed host
 * pointer to the last layer's output (NULL on failure); as a side effect the
 * host outputs of every YOLO/REGION layer are populated so that
 * get_network_boxes (src/additionally.c:4403) works unchanged.  Whether the
 * FP32, INT8 or XNOR convolution runs for a layer follows the reference CPU
 * rules (src/yolov2_forward_network_quantized.c:1036, src/yolov2_forward_network.c:116). */
float *yl_network_predict(yl_network *net, const float *input);

/* forward_network_gpu_cudnn(net, state)    src/yolov2_forward_network_gpu.cu:443
 * `input_dev` = device float[batch*c*h*w] already resident in HBM.  Asynchronous
 * on the network's stream (see yl_network_set_stream); no host copies. */
int yl_network_forward(yl_network *net, const float *input_dev);

/* HIP stream (hipStream_t as void*) the network launches on; NULL = its own stream. */
int yl_network_set_stream(yl_network *net, void *hip_stream);
/* block until everything queued by this network has finished (cudaDeviceSynchronize analogue) */
int yl_network_synchronize(yl_network *net);

/* cuda_pull_array(l.output_gpu, l.output, n)           src/gpu.cu:254
 * copies batch*outputs floats of layer i to dst_host (synchronous). */
int yl_network_layer_output(yl_network *net, int i, float *dst_host);
/* the same for ONE batch item: outputs floats of image `image` (large batches: 64 x 608x608 keeps 22 GB
 * of activations resident, a parity check wants three images of them) */
int yl_network_layer_output_image(yl_network *net, int i, int image, float *dst_host);
/* device pointer of layer i's output [batch][out_c][out_h][out_w] (borrowed); NULL (like the copies above:
 * YL_ERR_STATE) for tensors the fusion plan does not materialise */
const float *yl_network_layer_output_dev(const yl_network *net, int i);
/* device pointer of the network input staging buffer (net.input_state_gpu, src/additionally.c:4060) */
float *yl_network_input_dev(yl_network *net);

/* XNOR parity hook: integer match-count tensor of XNOR conv layer i from the
 * last forward (the `count` of gemm_nn_custom_bin_mean_transposed,
 * src/additionally.c:1504-1534, before `(2*count-K)*mean`).  Enabled by
 * yl_network_set_debug(net, 1) BEFORE yl_network_to_device. dst = int32[batch*outputs]. */
int yl_network_set_debug(yl_network *net, int on);
int yl_network_layer_xnor_counts(yl_network *net, int i, int32_t *dst_host);
/* INT8 parity hook: the int16-clamped accumulator of INT8 conv layer i
 * (`output_q`, src/yolov2_forward_network_quantized.c:550,486). dst = int32[batch*outputs]. */
int yl_network_layer_int8_acc(yl_network *net, int i, int32_t *dst_host);

/* Per-layer device timing with HIP events on the network's stream:
 * runs `iters` forwards, ms_per_layer[n_layers] = average ms of each layer's
 * kernels, *total_ms = average ms of a whole forward (events around the pass).
 * (The reference only has clock() around predict, src/main.c:197,220.) */
int yl_network_profile(yl_network *net, const float *input_dev, int iters,
                       float *ms_per_layer, float *total_ms);
/* The same measurement without a host sync inside the pass: yl_network_forward with a HIP
 * event recorded on the network's stream before every layer and after the last one, into
 * timing slot `slot` (0..63; one slot per step lets a bench record every step of its timed
 * region and read all of them afterwards).  yl_network_layer_times waits for the slot's
 * last event and reads its elapsed times. */
int yl_network_forward_timed(yl_network *net, const float *input_dev, int slot);
int yl_network_layer_times(yl_network *net, int slot, float *ms_per_layer, float *total_ms);
/* name of the kernel (template instance) layer i's last launch used, e.g.
 * "conv_f32_mfma<128x128,ks3>"; "" for layers that have not run or are pure aliases */
const char *yl_network_layer_kernel(const yl_network *net, int i);

/* ------------------------------------------------------------------ *
 *  Detections (L4).  The reference decodes batch item 0 only
 *  (src/additionally.c:4213,4338); `image` selects the batch item.
 * ------------------------------------------------------------------ */

/* get_network_boxes(&net, w, h, thresh, hier, map=0, relative, &num, letter)
 *   + do_nms_sort(dets, num, classes, nms)   src/additionally.c:4403, src/box.c:296
 * for batch item `image` of the last forward, computed ON THE DEVICE (K10 + K11, the same kernels
 * as yl_network_detect_batch; the decode of the whole batch is cached, so looping over the images
 * costs one pass).  rows[max_rows][6+classes]: x y w h objectness sort_class prob[classes], the
 * reference's rows bit for bit incl. order while the count stays <= YL_DETECT_MAX_CAP.  Returns the
 * number of detections the reference would return (may exceed max_rows), <0 on error.  Synchronous. */
int yl_network_get_boxes(yl_network *net, int image, int w, int h, float thresh,
                         int relative, int letter, float nms,
                         float *rows, int max_rows, int *classes_out);

/* (yl_network_get_boxes no longer needs it) D2H of every YOLO/REGION layer output (what src/yolov2_forward_network_gpu.cu:438
 * does per YOLO layer) after a yl_network_forward. */
int yl_network_pull_heads(yl_network *net);

/* Tuning/test hooks, PER NETWORK (two networks driven from two host threads share no launch state).
 * yl_network_set_conv_tile and yl_network_set_variant may be called at any time, also after yl_network_to_device: the
 * layer-fusion plan (conv + [shortcut], [yolo] folded into its head conv, a 2x2 / stride-2 [maxpool] written by the
 * convolution in front of it) is made at yl_network_to_device from the knobs of that moment, and a later change never
 * fails a forward pass -- where the newly selected kernel cannot write a planned pooled tensor, the convolution writes
 * its full tensor and the stand-alone pooling kernel runs behind it (same bits).  Weight images that a variant bit
 * selects (bit 5, yl_network_set_winograd) are packed at yl_network_to_device: set those BEFORE it.
 * yl_network_set_conv_tile: force the K1 kernel of every FP32 convolution of this network, any time:
 *   0 = built-in heuristic (default), 11..22 = direct implicit-GEMM tile 1..12 (conv_f32_mfma.hip),
 *   31 = Winograd F(2x2,3x3) (conv_f32_wino32.hip; a launch fails on layers it does not apply to),
 *   41 = LDS-free first-layer kernel (conv_f32_smallk.hip; C*size^2 <= 32 and filters <= 32 only),
 *   51..55 = the three-piece BF16 kernel's tiles (conv_f32_x3.hip), 61..70 = the row-wise Winograd kernel's tiles and
 *   schedules (conv_f32_row3.hip; 3x3 / stride 1 / pad 1 layers with C % 16 == 0 only).
 * yl_network_set_winograd: 0 = never pick Winograd heuristically and do not pack its weights, 1 = default;
 *   BEFORE yl_network_to_device.
 * yl_network_set_nms_mode: yl_network_detect_batch's suppression stage, 1 = one workgroup per
 *   (image, class) (default), 0 = one workgroup per image; same rows either way. */
int yl_network_set_conv_tile(yl_network *net, int cfg);
/* kernel-selection / schedule switches kept for same-box A/B measurements.  Bits 0-3 and 6-9 change the schedule only
 * (bit-identical results); bits 4, 5, 10 and 11 change WHICH kernel a layer takes (results within the FP32 contract):
 * bit 0 Winograd U panels by LDS-DMA, bit 1 Winograd epilogue prefetches the fused [shortcut] operand, bit 2 float4 B-panel
 * rows in the 1x1 direct kernel, bit 3 LDS-free first-layer kernel, bit 4 Winograd from 32 input channels up, bit 5 from 16,
 * bit 6 persistent Winograd workgroups (per-XCD tile counters), bit 7 the Winograd input transform in its register-shift
 * form (default: column masks folded into the transform), bit 8 sign-only XNOR layers evaluate the float epilogue instead of
 * comparing the match count with its threshold, bit 9 XNOR layers with >= 64 filters always run 64-filter workgroups
 * (default: 32 on shallow grids), bit 10 the direct FP32 layers with C % 16 == 0 and more than 32 filters on the BF16 matrix
 * pipe with every operand as the exact sum of three bf16 pieces (conv_f32_x3.hip; FP32 tensors, FP32-class accuracy),
 * bit 11 the 3x3 / stride-1 layers Winograd would take as ROW-WISE Winograd F(2,3) on the BF16 matrix pipe, three-piece
 * operands (conv_f32_row3.hip) instead of F(2x2,3x3) on the FP32 matrix instruction, bit 12 (A/B only) conv_f32_x3.hip without its
 * pinned schedule (same bits), bit 13 (off) conv_f32_row3.hip's 128 x 128 tile in its "view" form on maps up to 126 wide -- the
 * transformed rows staged once per channel block instead of once per filter row; same bits, +0.2 ... +0.6 % in the network,
 * bit 14 RGB first layers whose only output is the sign words of an XNOR network (<= 16 filters; with the 2x2 / stride-2 [maxpool]
 * behind them folded in) or the int8 units of an INT8 network (32 filters) on the FP32 matrix pipe (conv_f32_firstm.hip) instead of
 * the VALU (conv_f32_first.hip): the same fmaf chains, the same bits;
 * -1 = built-in default (bits 1-5, 10, 11 and 14) */
int yl_network_set_variant(yl_network *net, int bits);
/* Opt-in BF16 variant of the FP32 path (north_star (a) "FP32/BF16"; BEFORE yl_network_to_device): every FP32
 * convolution whose input has whole 8-channel groups runs on v_mfma_f32_32x32x16_bf16 with both operands rounded
 * to bf16 (nearest even) and FP32 accumulation / bias / activation (conv_bf16_mfma.hip); tensors between layers
 * stay FP32 wherever a [route]/[shortcut]/head reads them.  NOT inside the 1e-4 contract of the FP32 path
 * (operand rounding ~2^-9 relative): default is YL_PRECISION_FP32. */
#define YL_PRECISION_FP32 0
#define YL_PRECISION_BF16 1
/* YL_PRECISION_FP32_STRICT: every FP32 convolution on the direct implicit-GEMM kernel of the FP32 matrix instruction
 * (v_mfma_f32_32x32x2_f32, an fmaf chain in gemm_nn's k order per 32-deep block: conv_f32_mfma.hip) -- no Winograd
 * transform, no three-piece bf16 operands.  Equivalent to yl_network_set_winograd(0) + yl_network_set_variant(62).
 * The default FP32 path (K1r + K1x) is closer to a float64 evaluation than the reference's own builds but differs from
 * the reference's SCALAR build by more than 1e-4 relative on ~3e-3 of the yolov3-608 head elements; this mode keeps
 * that fraction at the level of the reference's own AVX-vs-scalar disagreement (5.7e-4 vs 3.5e-4,
 * tests/test_gpu_parity.py::test_fp32_error_vs_float64_truth) at ~0.55x the throughput (bench_detail.json
 * "strict_fp32"). */
#define YL_PRECISION_FP32_STRICT 2
int yl_network_set_precision(yl_network *net, int precision);
/* the same for the INT8 convolution (conv_i8_mfma.hip): 0 = heuristic, 1 = 64x128, 2 = 32x256, 3 = 128x128,
 * 4 = 128x256 (8 waves), 5 = 64x256, 6 / 7 = 128x128 / 64x128 with half-depth LDS panels */
int yl_network_set_int8_tile(yl_network *net, int cfg);
int yl_network_set_winograd(yl_network *net, int on);
int yl_network_set_nms_mode(yl_network *net, int mode);
/* Split K for grids below the chip (BEFORE yl_network_to_device; default 0 = off).  The reference runs one image on one device
 * (src/main.c:653-661) and has no analogue; config 3 at 8 GPUs is 8 images per GPU, where yolov3's 19 x 19 and 38 x 38 layers launch
 * fewer workgroups than the part has CUs and a layer's time is ONE workgroup's K loop.  With 1, an FP32 convolution on
 * conv_f32_x3.hip / conv_f32_row3.hip whose grid would leave CUs idle cuts its input channels into 2-4 contiguous ranges, one
 * workgroup set per range writes raw partial sums to a workspace, and a second kernel adds them IN RANGE ORDER, then bias,
 * activation and the fused [shortcut]: deterministic (run-to-run bit-identical), a different summation order than the unsplit
 * layer -- inside the FP32 contract (tests/test_gpu_splitk.py, against the oracle), but batch-B results are then no longer bit-equal
 * to batch-1 results of the same image (the number of ranges follows the grid).  Layers whose grid fills the chip are untouched. */
int yl_network_set_split_k(yl_network *net, int on);
/* Kernel-layout weight packing at yl_network_to_device (the reference packs on one host core:
 * binary_align_weights src/additionally.c:196-302, init_gpu_int8x4 src/yolov2_forward_network_quantized.c:1489):
 * 1 (default) = the prepared weights are uploaded as they are and packed by kernels on the device (csrc/pack.hip),
 * 0 = packed by host loops and uploaded inflated.  Bit-identical images either way.  BEFORE yl_network_to_device. */
int yl_network_set_device_pack(yl_network *net, int on);
/* (The test / lab instrumentation of the library -- packed-weight read-back, host-side packers -- is declared in
 * include/yolo2_hip_lab.h: not part of the drop-in boundary, nothing a caller of network_predict needs.) */

/* On-device detection compaction (new): threshold test
 * `objectness > thresh` (src/additionally.c:4341) and box decode
 * (get_yolo_box :4317 / get_region_box_cpu src/yolov2_forward_network.c:653)
 * executed on the GPU into a fixed-capacity record buffer that an RCCL gather
 * can ship: records_dev[batch][cap][6+classes] floats (row layout of
 * yl_network_get_boxes except column 5, which holds the record's position in the
 * reference's scan order; rows are in arbitrary order; boxes relative to the network
 * input, no NMS) and
 * counts_dev[batch] ints.  Asynchronous on the network's stream. */
int yl_network_compact_detections(yl_network *net, float thresh, int cap,
                                  float *records_dev, int *counts_dev);

/* Batched detections on the GPU (new).  For EVERY image b of the batch:
 *   get_network_boxes(&net, img_w[b], img_h[b], thresh, hier, 0, relative, &num, letter)
 *                                                       src/additionally.c:4403
 *   do_nms_sort(dets, num, classes, nms)                src/box.c:296-328
 * (the reference does both on the host and for batch item 0 only, src/additionally.c:4213,4338).
 * records_dev[batch][cap][6+classes] floats, row = x y w h objectness sort_class prob[classes];
 * counts_dev[batch] = number of detections of image b (rows beyond min(count, cap) are not
 * written).  While count <= cap the rows of image b are bit-identical, order included, to what
 * yl_network_get_boxes(net, b, ...) -- i.e. the reference -- returns: correct_yolo_boxes
 * (src/additionally.c:4281) and the class-by-class stable sort + greedy suppression are
 * replayed exactly (yolo2_light_amd/csrc/detect.hip); nms <= 0 skips the suppression.
 * img_w/img_h: host int[batch] with the source image sizes, or both NULL when relative=1 and
 * letter=0 (boxes relative to the network input).  cap <= YL_DETECT_MAX_CAP.
 * Asynchronous on the network's stream; reads the head outputs of the last forward. */
#define YL_DETECT_MAX_CAP 4096
int yl_network_detect_batch(yl_network *net, const int *img_w, const int *img_h, float thresh,
                            int relative, int letter, float nms, int cap,
                            float *records_dev, int *counts_dev);
/* the same, delivered to host buffers rows_host[batch][cap][6+classes], counts_host[batch]
 * (synchronous; only the filled rows are copied) */
int yl_network_get_boxes_batch(yl_network *net, const int *img_w, const int *img_h, float thresh,
                               int relative, int letter, float nms, int cap,
                               float *rows_host, int *counts_host);

/* ------------------------------------------------------------------ *
 *  Image front end on the GPU (new).
 * ------------------------------------------------------------------ */

/* What test_detector_cpu does per image on one host core (src/main.c:187-189):
 *   load_image_stb's  im[k][y][x] = (float)u8[(y*w+x)*c+k] / 255.     src/additionally.c:3095-3103
 *   resize_image(im, net.w, net.h)   two-pass bilinear stretch         src/additionally.c:3021-3064
 * fused into one kernel that writes batch slot `image` of the network input buffer
 * (yl_network_input_dev) -- bit-identical to the reference's `sized.data`.  `pixels` is the
 * decoder's output: HWC, 8 bits per channel, c == net.c.  The host variant stages through
 * pinned memory (3 bytes per source pixel over PCIe instead of 12 per network pixel) and is
 * asynchronous on the network's stream; `pixels_host` may be reused as soon as it returns.
 * Follow with yl_network_forward(net, yl_network_input_dev(net)). */
int yl_network_set_input_u8(yl_network *net, int image, const uint8_t *pixels_host, int w, int h, int c);
int yl_network_set_input_u8_dev(yl_network *net, int image, const uint8_t *pixels_dev, int w, int h, int c);
/* A whole batch of decoded frames in ONE call: frame i (HWC u8, w[i] x h[i] x c, pageable host memory is fine) -> batch slot
 * first + i, for i < count.  Same result as `count` calls of yl_network_set_input_u8 (bit-identical), without their per-frame host
 * cost: the frames are copied into the pinned slots by the library's host-thread pool, travel on the copy stream (the uploads of
 * step k+1 overlap the forward pass of step k), and the conversion + resize kernels are queued on the compute stream behind them.
 * Asynchronous like the per-frame call; the frames may be reused when it returns. */
int yl_network_set_input_u8_batch(yl_network *net, int first, int count, const uint8_t *const *pixels_host,
                                  const int *w, const int *h, int c);
/* copy of the device input buffer, float[batch*c*h*w] (synchronous; what `sized.data` /
 * the X argument of network_predict would hold, src/main.c:189,193) */
int yl_network_input_download(yl_network *net, float *dst_host);

/* ------------------------------------------------------------------ *
 *  INT8 calibration tool (new): produces the `input_calibration=` list of a cfg.
 * ------------------------------------------------------------------ */

/* `darknet detector calibrate` = validate_calibrate_valid (src/additionally.c:4902) ->
 * network_calibrate_cpu (src/yolov2_forward_network.c:731): for every image, before each
 * convolutional layer, entropy_calibration(state.input, l.inputs, 1/16, 4096)
 * (src/yolov2_forward_network_quantized.c:1292) picks the input multiplier 127/threshold that
 * minimises KL(P || Q); the per-layer values are averaged over the images.
 * Here: FP32 forward on the GPU for `n_images` images (host float CHW [0,1], a multiple of the batch),
 * the histogram of every conv layer's input counted on the GPU (exact integers), the KL scan on
 * the host with the reference's arithmetic, and the reference's averaging reproduced slot for slot
 * (its mean takes the previous conv layer's last-image value instead of the layer's own: documented
 * in runtime.hip).  multipliers[k] = value for the k-th convolutional layer, in layer order (the
 * tool then prints them followed by "16").  Returns the number of values, < 0 on error.
 * Difference: the reference's loop only knows CONV/MAXPOOL/ROUTE/REORG/REGION and skips every
 * other layer type (yolov3 cfgs calibrate on garbage there); this runs the real forward pass. */
int yl_network_calibrate(yl_network *net, const float *images_host, int n_images,
                         float *multipliers, int max_out);
/* the KL scan alone, from an exact histogram counts[max_bin] of lround(|x| / bin_width) (host only) */
float yl_entropy_from_histogram(const uint32_t *counts, int max_bin, float bin_width);

/* ------------------------------------------------------------------ *
 *  Several GPUs of one node, ONE host process (new).  The reference selects a single
 *  device (`-i <n>` -> cuda_set_device, src/main.c:653-661) and hands network_predict the whole batch;
 *  a group splits that batch of independent images over the listed devices (weights replicated, one
 *  host thread + HIP stream per device, no data-path collective in the forward pass) and gathers the
 *  fixed-capacity detection records on devices[0] with RCCL (ncclSend/ncclRecv over xGMI).
 * ------------------------------------------------------------------ */
typedef struct yl_group yl_group;

/* image range [first, first+count) of `rank` when global_batch images are split over n ranks: the first
 * global_batch % n ranks take one image more (host only) */
int yl_shard_range(int global_batch, int n, int rank, int *first, int *count);

/* `model`: a prepared HOST network (after load/fuse/quantise and any yl_network_set_* knob, NOT on a device)
 * whose batch is the GLOBAL batch; one replica per device is built and uploaded (in parallel).  The model may
 * be destroyed afterwards; host `output` pointers given through yl_layer_desc must stay valid (the replicas
 * write their slices of the heads into them).  devices[0] is the root of the gather. */
int yl_group_create(const yl_network *model, const int *devices, int n_devices, yl_group **out);
void yl_group_destroy(yl_group *g);
int yl_group_size(const yl_group *g);
int yl_group_shard(const yl_group *g, int rank, int *first, int *count);
/* borrowed replica of `rank` (introspection, yl_network_set_input_u8, per-layer readback, timing) */
yl_network *yl_group_member(yl_group *g, int rank);

/* network_predict contract on the global batch (see yl_network_predict): `input` = host
 * float[global_batch*c*h*w]; every rank stages, runs and pulls its shard concurrently; returns the host
 * pointer of the last layer's output for the global batch, NULL on failure. */
float *yl_group_predict(yl_group *g, const float *input);
/* asynchronous forward of every shard; inputs_dev[rank] = device pointer ON that rank's device, or NULL
 * (array or entry) for the replica's own input buffer (yl_network_input_dev / yl_network_set_input_u8) */
int yl_group_forward(yl_group *g, const float *const *inputs_dev);
int yl_group_synchronize(yl_group *g);
/* yl_network_detect_batch on every shard + RCCL gather: records_dev_root[global_batch][cap][6+classes] and
 * counts_dev_root[global_batch] live on devices[0]; img_w/img_h: host int[global_batch] or NULL.  Asynchronous;
 * the root replica's stream orders the result (yl_network_synchronize(yl_group_member(g, 0))). */
int yl_group_detect_batch(yl_group *g, const int *img_w, const int *img_h, float thresh, int relative, int letter,
                          float nms, int cap, float *records_dev_root, int *counts_dev_root);
/* the same, delivered to host buffers (synchronous; only the filled rows are copied) */
int yl_group_get_boxes_batch(yl_group *g, const int *img_w, const int *img_h, float thresh, int relative, int letter,
                             float nms, int cap, float *rows_host, int *counts_host);

#ifdef __cplusplus
}
#endif
#endif /* YOLO2_HIP_H */
