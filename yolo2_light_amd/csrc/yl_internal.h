// yl_internal.h -- host-side model representation shared by the cfg/prep code
// (plain C++) and the HIP runtime.  Field names follow the reference's `layer`
// (src/additionally.h:409-684) where the meaning is the same.
#pragma once
#include <cstdint>
#include <cstddef>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/yolo2_hip.h"
#include "kernels.h"

namespace yl {

// which convolution implementation a CONVOLUTIONAL layer runs with
enum ConvMode { CONV_F32 = 0, CONV_INT8 = 1, CONV_XNOR = 2, CONV_BF16 = 3 };
enum HostKind { HOST_NONE = 0, HOST_CALLER = 1, HOST_PINNED = 2 };

constexpr int PREDICT_MAX_SPLIT = 4;  // sub-batches of one yl_network_predict call
constexpr int ACT_FRONT_PAD = 64;      // floats (256 B) of readable, zeroed memory in front of every library-owned activation tensor
constexpr int ACT_TAIL_PAD = 16;       // floats of readable memory behind it: the 16-byte row loads of the Winograd kernels end up to two
                                       // columns beyond the last row of the last image (selected away, but the bytes must be mapped)

struct Layer {
    int type = YL_BLANK;
    int activation = YL_LINEAR;
    int batch = 0, w = 0, h = 0, c = 0;
    int n = 0, size = 0, stride = 1, pad = 0;
    int out_w = 0, out_h = 0, out_c = 0;
    int outputs = 0, inputs = 0;
    int batch_normalize = 0, xnor = 0;
    int gpu_quantized = 0;               // the reference parser's `l.quantized` (GPU-path quantisation rule)
    int index = 0;                       // shortcut `from`
    std::vector<int> input_layers, input_sizes;   // route
    int classes = 0, coords = 4, total = 0, softmax = 0;
    std::vector<int> mask;
    std::vector<float> anchors;
    float scale = 1.f;
    std::vector<int> tree_parent, tree_group_size;   // region with a softmax tree (read_tree, src/additionally.c:1895)

    // conv parameters (host)
    std::vector<float> weights, biases, scales, rolling_mean, rolling_variance;
    std::vector<int8_t> weights_int8;
    float input_quant_multipler = 0.f, weights_quant_multipler = 0.f;
    bool quant_ready = false;
    std::vector<float> mean_arr;         // xnor: per-filter mean(|w|)
    bool xnor_ready = false;
    int conv_mode = CONV_F32;

    // where pull_heads delivers this layer's tensor (heads / last layer only).  The HIP runtime is never handed
    // caller memory (staging.hip): HOST_CALLER destinations (yl_layer_desc.output = the reference's calloc'd
    // l.output) are reached through the network's own pinned block + memcpy, HOST_PINNED ones (library-allocated:
    // net.h_heads or a group's global tensors) take the DMA directly.
    float *host_output = nullptr;
    int   host_kind = HOST_NONE;
    bool  host_in_heads = false;         // host_output points into net.h_heads (dropped together with it)
    size_t h_head_off = 0;               // floats: this layer's region of net.h_heads

    // ---- device state (owned by runtime.hip) ----
    float *d_output = nullptr;           // [batch][out_c][out_h][out_w]
    bool  d_output_alias = false;        // route with one input aliases its source
    float *d_weights_t = nullptr;        // FP32: k-major packed [Kpad][Mpad]
    float *d_biases = nullptr;
    int   Kpad = 0, Mpad = 0;
    int   tapmajor = 0;                  // K order of d_weights_t (see conv_f32_mfma.hip)
    float *d_wino32_u = nullptr;         // FP32 3x3/1/1: Winograd-packed U (conv_f32_wino32.hip), else nullptr
    void *d_weights_x3 = nullptr;        // FP32, C % 16 == 0: the weights as three bf16 pieces (conv_f32_x3.hip), else nullptr
    void *d_weights_r3 = nullptr;        // FP32 3x3/1/1, C % 16 == 0: the row-transformed weights as three bf16 pieces (conv_f32_row3.hip), else nullptr
    unsigned *d_tile_ctr = nullptr;      //       8 work counters (one per XCD) of the persistent Winograd form, zero between launches
    size_t packed_bytes[6] = {0, 0, 0, 0, 0, 0};   // bytes of d_weights_t, d_wino32_u, d_weights_i8, d_weights_bits, d_weights_x3, d_weights_r3 (yl_debug_layer_packed)
    int8_t *d_weights_i8 = nullptr;      // INT8: [K16pad][Mpad][16] int8 units; BF16: [K8pad][Mpad][8] bf16 units
    int   Cpad = 0;                      // channels of the 16-byte-unit activation tensor (INT8: 16 per unit, BF16: 8)
    float bias_abs_max = 0.f;            // INT8: max |bias| and the smallest non-zero |bias| (-1 = a bias is not finite):
    float bias_abs_min_nz = 0.f;         //       the host's proof that the exact epilogue's corners cannot occur
    uint64_t *d_weights_bits = nullptr;  // XNOR: [Mpad][taps][Cw] 64-bit words
    float *d_mean = nullptr;
    int  *d_thr = nullptr;               // XNOR: [Mpad] count thresholds of the sign-only epilogue (+ 1 int: filters without one)
    bool  thr_ok = false;                //       every filter's result is a step function of the count (else d_thr is not used)
    int   Cw = 0;
    int32_t *d_debug = nullptr;          // xnor counts / int8 acc (debug mode)
    int  *d_tree = nullptr;              // region softmax tree: parent[classes] then group_size[groups]
    char kernel_name[64] = "";           // kernel instance of this layer's last launch
    int   fused_shortcut = -1;           // conv: index of the [shortcut] layer folded into its epilogue
    int   fused_yolo = -1;               // FP32 1x1 head conv: index of the [yolo] layer folded into its epilogue
    int   two_src_up = -1, two_src_other = -1;   // FP32 1x1 conv behind [route]([upsample], other): the two layers it can read directly (K1x)
    int   two_src_conv = -1;             // that [upsample] / [route]: the convolution that reads around them
    bool  two_src_skipped = false;       // ... and whether the last forward pass did (then this layer's tensor was not written)
    int   fused_pool = -1;               // FP32 conv (K1f / K1w): index of the 2x2 / stride-2 [maxpool] layer its epilogue also writes
    bool  pool_follows = false;          // FP32 conv in front of a 2x2 / stride-2 [maxpool] its kernel can fold in: keeps that kernel (K1f / K1w)
    bool  fused_into_conv = false;       // shortcut: produced by the preceding conv's epilogue
    bool  q_from_producer = false;       // INT8 conv: its quantised input is written by the producing conv
    bool  q_from_route = false;          // INT8 conv: its input is a multi-input [route], quantised source by source
    int   q_out_layer = -1;              // INT8 conv: also emits the quantised input of this later layer
    bool  binarize_input = false;        // xnor FP32 fallback: input -> +-1 before the conv
    bool  skip_f32_out = false;          // FP32 tensor of this layer has no reader and is not written
    // XNOR sign-domain fusion (yl_network_set_fusion): sign words travel between XNOR layers instead of FP32
    bool  bits_from_producer = false;    // XNOR conv: its packed input is written by the layer(s) before it
    int   bits_out_slot = -1;            // XNOR conv: also emits the sign words of its result into bit-ring slot (index % 3)
    int   pool_bits_mode = 0;            // maxpool: 1 = OR-pool sign words (slot i -> slot i+1), 2 = FP32 in -> pooled sign words
    bool  bits_pooled_by_producer = false;   // maxpool, mode 1: the convolution in front of it wrote the pooled words into slot i+1 in this pass
};

// arguments of the cached yl_network_get_boxes pass
struct DetKey {
    unsigned long long seq;
    int w, h;
    float thresh;
    int relative, letter;
    float nms;
};

struct Network {
    int batch = 1, w = 0, h = 0, c = 0;
    int quantized = 0;
    int quant_rule = YL_QUANT_RULE_CPU;  // which convolutions `quantized` applies to (yl_network_set_quant_rule)
    int precision = YL_PRECISION_FP32;   // opt-in BF16 operands for the FP32 convolutions (yl_network_set_precision)
    std::vector<float> input_calibration;
    std::vector<Layer> layers;
    bool weights_loaded = false;

    // device
    int device = -1;
    bool on_device = false;
    bool debug = false;
    bool fuse = false;                   // conv+shortcut epilogue fusion (yl_network_set_fusion)
    ConvF32Opts conv_opts;               // K1 kernel-selection knobs of THIS network (no process-global launch state)
    int i8_tile = 0;                     // K2 tile (0 = heuristic; yl_network_set_int8_tile)
    int nms_mode = 1;                    // 1 = one workgroup per (image, class), 0 = one per image
    bool split_k = false;                // yl_network_set_split_k: K ranges for FP32 convolutions whose grid leaves CUs idle
    float *d_ks_ws = nullptr;            //   partial sums: up to 4 ranges of the largest eligible layer's tensor
    size_t ks_ws_floats = 0;
    float *d_ks_zeros = nullptr;         //   the partial passes' bias (zeros, >= the widest layer's filters)
    unsigned long long forward_seq = 0;  // bumped by every forward: detection cache key
    void *stream = nullptr;              // hipStream_t
    bool own_stream = false;
    float *d_input = nullptr;
    int8_t *d_qbuf = nullptr;            // INT8: quantised NHWC activations scratch
    size_t qbuf_bytes = 0;
    uint64_t *d_bitbuf = nullptr;        // XNOR: channel-packed sign words, a ring of 3 slots (layer i reads slot i % 3)
    size_t bitbuf_bytes = 0;
    float *d_binbuf = nullptr;           // XNOR FP32 fallback: +-1 image scratch
    size_t binbuf_bytes = 0;
    bool device_pack = true;             // kernel-layout weight images are written by pack.hip's kernels (false: host loops)
    char *d_pack_src = nullptr;          // device scratch: one layer's prepared weights as they are (+ mean_arr)
    size_t pack_src_bytes = 0;
    void *h_pinned = nullptr;            // pinned staging for the input
    float *h_heads = nullptr;            // pinned: head / last-layer tensors (owned destinations and bounce regions)
    size_t h_heads_floats = 0;
    size_t pinned_bytes = 0;
    float *d_det_scratch = nullptr;      // batched detections: compacted records before NMS
    size_t det_scratch_bytes = 0;
    float *d_det_out = nullptr;          // batched detections: device staging of yl_network_get_boxes_batch
    size_t det_out_bytes = 0;
    int *d_det_counts = nullptr;         // [2][batch]: raw compaction counts, staged output counts
    float *h_det_rows = nullptr;         // pinned: the filled detection rows of the last decode, packed image after image
    size_t h_det_bytes = 0;
    int *h_det_counts = nullptr;         // pinned [batch]
    std::vector<int> det_counts;         // counts of the last decode (may exceed cap)
    std::vector<size_t> det_row_off;     // [batch + 1] float offsets into h_det_rows
    DetKey det_cache_key{};
    bool det_cache_valid = false;
    unsigned *d_det_meta = nullptr;      // [batch][1 + class words]: NMS `total` + class bitmap
    size_t det_meta_bytes = 0;
    uint8_t *h_u8 = nullptr;             // pinned staging of u8 source images, one region per batch slot
    uint8_t *d_u8 = nullptr;             // the same on the device
    size_t u8_stride = 0;                // bytes per slot
    std::vector<void *> u8_events;       // per slot: H2D of the slot's staging region has completed
    void *u8_resized = nullptr;          // hipEvent_t: the resize kernels of the last yl_network_set_input_u8_batch have read d_u8
    // yl_network_predict's pipeline (runtime.hip): the batch runs as up to PREDICT_MAX_SPLIT sub-batches; input of sub-batch k+1
    // travels on in_stream and the heads of sub-batch k-1 on out_stream while sub-batch k computes on `stream`
    void *in_stream = nullptr, *out_stream = nullptr;   // hipStream_t
    std::vector<void *> in_events;       // per sub-batch: its input has landed (in_stream)
    std::vector<void *> head_events;     // [sub-batch][layer]: the tensor is complete on `stream` (heads / last layer only, else nullptr)
    std::vector<void *> chunk_events;    // D2H chunks of the current predict call, in issue order
    int head_event_base = -1;            // >= 0: forward() records head_events[base + layer] (set by yl_network_predict)
    void *ev0 = nullptr, *ev1 = nullptr; // hipEvent_t pair for profiling
    std::vector<void *> layer_events;
};

void set_error(const std::string &msg);

// host_pool.cpp: pageable <-> pinned copies spread over a pool of host threads, overlapped with the DMA by the callers
struct HostCopyJob {
    std::atomic<int> pending{0};
};
void host_copy_async(HostCopyJob &job, void *dst, const void *src, size_t bytes);
void host_copy_wait(HostCopyJob &job);
unsigned host_copy_threads();

// staging.hip: caller memory <-> device through library-owned pinned chunks; both return when the bytes have landed
int stage_h2d(int device, void *dst_dev, const void *src_host, size_t bytes);
int stage_d2h(int device, void *dst_host, const void *src_dev, size_t bytes);   // producer stream already synchronised

// host_cfg.cpp
int parse_cfg_file(const char *path, int batch, int quantized, Network &net);
// host_prep.cpp
int load_weights_file(Network &net, const char *path, int cutoff = -1);
void fuse_conv_batchnorm(Network &net);
void calculate_binary_weights(Network &net);
void quantize_network(Network &net);
void select_conv_modes(Network &net);
float multiplier_from_range_counts(const int *count, int bits_length);
// prep.hip: the same three passes on the GPU, results bit-identical to the host passes
int prepare_on_device(Network &net, int device);
// pack.hip: the kernel-layout packers on the device (bit-identical to the host packers; dst pre-initialised by the caller)
int dev_pack_kmajor(const float *d_w, const float *d_mean, float *d_dst, int M, int C, int taps, int Mpad, int tapmajor, void *stream);
int dev_pack_wino(const float *d_w, float *d_dst, int C, int M, void *stream);
int dev_pack_i8_units(const int8_t *d_wq, int8_t *d_dst, int M, int C, int taps, int G, int Mpad, void *stream);
int dev_pack_bf16_units(const float *d_w, uint16_t *d_dst, int M, int C, int taps, int G, int Mpad, void *stream);
int dev_pack_x3(const float *d_w, void *d_dst, int M, int C, int taps, int Mpad, void *stream);
int dev_pack_row3(const float *d_w, void *d_dst, int M, int C, int Mpad, void *stream);
int dev_pack_xnor_words(const float *d_w, uint64_t *d_dst, int M, int C, int Cw, void *stream);
// host_calib.cpp
float entropy_from_counts(const uint32_t *counts, int max_bin, float bin_width);
}  // namespace yl

struct yl_network {
    yl::Network net;
};
