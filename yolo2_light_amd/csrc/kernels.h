// kernels.h -- launch interfaces of the hand-written gfx950 kernels.
// Every launcher is asynchronous on `stream` and returns hipError_t as int.
#pragma once
#include <cstdint>
#include <cstddef>

namespace yl {

// measured on MI355X, profiles/r2_ab_fp32_variants.txt: bit 1 +3.3 %, bit 2 +0.4 %, bit 3 +0.6 %, bit 4 +0.9 %, bit 0 -0.5 %.
// (Bits 5-7 selected round 3's three alternative Winograd kernels -- all-planes-per-wave on 16x16x4, its warp-specialised
// form, the 64-filter 8-wave tile: bit-identical, -2.1 / -17 / -4.4 % in the network, profiles/r3_ab_wino_kernels.txt --
// removed from the library in round 4; bit 5 has a new meaning since, bits 6-7 are ignored.)
// round 4: bit 5 (Winograd from 16 input channels: yolov3-tiny's 16 -> 32 layer at 208 x 208, 0.175 ms on the direct 32x256
// tile + 0.063 ms of [maxpool] -> 0.119 ms with the pooling folded into the Winograd epilogue; config 2 +7.8 %)
// bit 10 (round 4): the direct FP32 layers on the BF16 matrix pipe with three-piece operands (K1x, conv_f32_x3.hip): +1.8 ... +7 %
// on the step depending on the box (the kernel drives the chip into its power cap), same accuracy against float64
// bit 12 (round 5, A/B only, off): K1x without its pinned schedule
// bit 11 (round 5): the 3x3 / stride-1 layers as row-wise Winograd F(2,3) on the BF16 matrix pipe with three-piece operands (K1r,
// conv_f32_row3.hip) instead of F(2x2,3x3) on the FP32 matrix instruction (K1w): +12.7 % on the step with its first version,
// same box; +17 ... +19 % as shipped
// bit 13 (round 5, off): K1r's 128 x 128 tile in its view form where 128 + 2 TW <= 254 (conv_f32_row3v_kernel): fewer row loads / splits /
// LDS stores, lower power and a higher clock, more cycles -- +0.2 ... +0.6 % in the network (profiles/r5_ab_row3_view_form.txt)
// bit 14 (round 5): RGB first layers that emit sign words only / int8 units only on the FP32 matrix pipe (K1m, conv_f32_firstm.hip) instead
// of the VALU (K1f): the same fmaf chains, the same bits
constexpr int YL_VARIANT_DEFAULT = 2 | 4 | 8 | 16 | 32 | 1024 | 2048 | 16384;

// ---- K1: FP32 implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 ----
struct ConvF32Args {
    const float *in;      // [B][C][H][W]
    const float *wt;      // k-major packed weights [Kpad][Mpad]
    const float *bias;    // [M]
    const float *add;     // optional fused [shortcut]: out_add = act(conv) + add (nullptr = none)
    float *out_add;       // destination of the fused shortcut (the SHORTCUT layer's output)
    float *out;           // [B][M][OH][OW]; may be nullptr when only out_add / q_out is wanted
    int8_t *q_out = nullptr;   // optional quantised side output for the next INT8 conv: act_q[B][q_G][OH][OW][16]
    float q_mult = 0.f;   // that layer's input_quant_multipler
    int q_G = 0;          // its channel groups (Cpad/16); direct kernel only, needs M % 16 == 0
    uint64_t *bits_out = nullptr;   // optional sign words (x > 0) of the activated output, bits[B][1][OH][OW], for an XNOR
                          // convolution behind this layer; first-layer kernel only (conv_f32_smallk.hip, M <= 32)
    const float *in2 = nullptr;   // two-source 1x1 layers (conv_f32_x3.hip only, x3_two_source_ok): the layer's input is the channel
    int in2_C1 = 0, in2_up = 0;   // concatenation [upsample by in2_up of `in` (in2_C1 channels at H/up x W/up), `in2` (C - in2_C1 channels)]
    bool bits_pooled = false;  // bits_out receives the sign words of the 2x2 / stride-2 [maxpool] behind the layer, bits[B][1][OH/2][OW/2]
                          // (the OR of each window's words); conv_f32_firstm.hip only
    float *pool_out = nullptr; // optional fused [maxpool] 2x2 / stride 2 / pad 1 behind the layer (H, W even): [B][M][H/2][W/2],
                          // written by the epilogue next to (or instead of, out == nullptr) the full tensor; only the kernels
                          // conv_f32_pool_fusable() names have it (K1f: a lane owns a 2 x 4 patch; K1w: an F(2x2) tile is a window)
    int yolo_entries = 0; // > 0: the [yolo] layer that follows is folded into the epilogue (1x1 direct kernel): rows
                          // m with m % yolo_entries not in {2, 3} get logistic_activate, `out` is the YOLO layer's tensor
    int B, C, H, W, M, OH, OW;
    int K, Kpad, Mpad;
    int size, stride, pad;
    int act;              // YL_LINEAR / YL_LEAKY
    int tapmajor;         // K order of `wt`: 0 = (c,ky,kx) like im2col_cpu, 1 = (ky,kx,c) (needs C % 16 == 0)
    const float *wino32_u; // Winograd-packed weights (wino32_pack_weights) or nullptr: 3x3/1/1 layers only
    const void *x3_w = nullptr; // the weights as three bf16 pieces (x3_pack_weights, conv_f32_x3.hip) or nullptr
    const void *row3_w = nullptr; // 3x3 / 1 / 1 layers: the row-transformed weights U = G g as three bf16 pieces (row3_pack_weights,
                          // conv_f32_row3.hip) or nullptr
    // split K (yl_network_set_split_k; K1x / K1r only, every other kernel ignores it): the channel blocks are cut into `ksplit`
    // contiguous ranges, one workgroup set per range (gridDim.y), each writing its raw partial sums to ks_ws + range * B * M * OH * OW;
    // a second kernel adds the partials IN RANGE ORDER, then bias, activation and the fused [shortcut] (launch_splitk_finish):
    // run-to-run bit-stable, a different summation order than the unsplit layer (inside the FP32 contract, not bit-equal to it)
    int ksplit = 1;
    float *ks_ws = nullptr;        // ksplit * B * M * OH * OW floats
    const float *ks_zeros = nullptr;   // >= M zero floats (the partial passes' bias)
    bool in_front_pad = false;     // `in` has >= 4 readable bytes in front of it holding a FINITE value (library-owned tensors:
                           // yl_internal.h ACT_FRONT_PAD); the Winograd kernel then fetches left-edge patches one column early
                           // and folds the column masks into the transform instead of shifting registers
    unsigned *tile_ctr = nullptr;  // 8 zero-initialised device counters of this layer: work queues of the persistent Winograd
                           // form (variant bit 6), one per XCD; every launch leaves them at zero again
};
// per-network kernel-selection knobs (snapshotted in Network: two networks driven from two host
// threads, one per GPU, share no mutable launch state)
struct ConvF32Opts {
    int force_tile = 0;   // 0 = heuristic, 11..22 = direct tile 1..12, 31 = Winograd, 41 = small-K first-layer kernel, 51..54 = K1x tiles,
                          // 61..69 = K1r tiles (tuning / tests)
    int winograd = 1;     // Winograd F(2x2,3x3) for 3x3 / stride 1 / pad 1 layers with C >= 64
    // schedule variants kept switchable for same-box A/B runs (yl_network_set_variant): bit 0 Winograd U panels by
    // LDS-DMA, bit 1 Winograd epilogue requests the [shortcut] operand ahead of its LDS exchange, bit 2 1x1 direct
    // kernel loads the B panel as float4 rows, bit 3 LDS-free small-K kernel for the first layer (C*size^2 <= 32),
    // bit 4 Winograd from 32 input channels up (without it: from 64), bit 5 Winograd from 16 input channels up,
    // bit 6 persistent Winograd workgroups (two per CU draw tiles from per-XCD counters, the next tile's first loads
    // are issued in front of the epilogue), bit 7 Winograd input transform in its round-3 form (register shift + 24 selects
    // per patch instead of column masks folded into the transform; A/B switch), bit 8 XNOR layers between XNOR layers keep the float epilogue instead of the count
    // threshold (conv_xnor.hip; same bits either way), bit 9 XNOR layers always use 64-filter workgroups where the layer
    // has 64 filters (default: 32-filter workgroups on shallow grids).  (An 8-byte-access epilogue for odd map widths was measured and dropped: no gain,
    // profiles/r2_ab_fp32_variants.txt.)
    int variant = YL_VARIANT_DEFAULT;
};
// compute units of the CURRENT device, cached per device id with atomics (the tile heuristics of K1x / K1r / K1w ask per launch:
// two networks on two GPUs or host threads share no mutable launch state, and a mixed node gets each device's own count)
int device_cu_count();
// would launch_conv_f32 send this layer (a.pool_out ignored) to a kernel that can fold a 2x2 / stride-2 [maxpool]?
bool conv_f32_pool_fusable(const ConvF32Args &a, const ConvF32Opts &o);
// writes the name of the kernel instance it launched into name[name_len]
int launch_conv_f32(const ConvF32Args &a, const ConvF32Opts &o, void *stream, char *name, size_t name_len);
// K1w (conv_f32_wino32.hip): Winograd F(2x2,3x3) for 3x3 / stride 1 / pad 1 layers,
// 32 filters x 64 tiles per workgroup, two workgroups per CU
bool wino_applicable(int C, int M, int size, int stride, int pad);
bool wino32_fits(int B, int M, int H, int W);     // output tensor below 4 GB (32-bit byte offsets in the epilogue)
// K1x (conv_f32_x3.hip): the same FP32 convolution on the BF16 matrix pipe, every operand the exact sum of three bf16 pieces
bool x3_applicable(int C, int M, int size, int stride, int pad);
size_t x3_packed_bytes(int C, int M, int size);
void x3_pack_weights(const float *w, int C, int M, int size, void *dst);
bool x3_two_source_ok(const ConvF32Args &a);
// ... and whether launch_conv_f32 would run that form with these knobs (the runtime asks before it skips the [upsample] / [route] layers)
bool conv_f32_two_source_now(const ConvF32Args &a, const ConvF32Opts &o);
int launch_conv_f32_x3(const ConvF32Args &a, int tile, void *stream, char *name, size_t name_len, bool plain = false);
// K1r (conv_f32_row3.hip): 3x3 / stride 1 / pad 1 as row-wise Winograd F(2,3) on the BF16 matrix pipe, three-piece operands
bool row3_applicable(int C, int M, int size, int stride, int pad);
bool row3_fits(int B, int C, int M, int H, int W);   // the 32-bit lane offsets of launch_conv_f32_row3 cover this layer (else: K1w / K1x / the direct kernel)
size_t row3_packed_bytes(int C, int M);
void row3_pack_weights(const float *w, int C, int M, void *dst);
// tile: 0 = heuristic, 1..5 see conv_f32_row3.hip
int launch_conv_f32_row3(const ConvF32Args &a, int tile, void *stream, char *name, size_t name_len, bool view = false);
size_t wino32_packed_floats(int C, int M);
void wino32_pack_weights(const float *w, int C, int M, float *dst);
// K1s (conv_f32_smallk.hip): LDS-free kernel for first layers (C*size^2 <= 32, filters <= 32)
bool smallk_applicable(const ConvF32Args &a);
int launch_conv_f32_smallk(const ConvF32Args &a, void *stream, char *name, size_t name_len);
// K1f (conv_f32_first.hip): RGB 3x3/1/1 first layers with at most 16 filters on the VALU (4 pixels x 16 filters per
// lane, scalar weights), bit-identical to K1s; FP32 rows and / or sign words out
bool first_layer_valu_applicable(const ConvF32Args &a);
int launch_conv_f32_first(const ConvF32Args &a, void *stream, char *name, size_t name_len);
// K1m (conv_f32_firstm.hip): the same layer on the FP32 matrix pipe (K1f's bits) where the output is sign words only (<= 16 filters) or
// int8 units only (32 filters)
bool first_layer_mfma_applicable(const ConvF32Args &a);
int launch_conv_f32_firstm(const ConvF32Args &a, void *stream, char *name, size_t name_len);
int launch_conv_f32_wino32(const ConvF32Args &a, const float *u_packed, int variant, void *stream, char *name, size_t name_len);

// ---- K2: INT8 path ----
// K2a: x_q = clamp_abs((int16)(x*mult), 127), FP32 NCHW -> int8 NHWC(Cpad)   (quantized.c:554-560)
// (g_off, G_total): write this source's Cpad/16 groups at group offset g_off of a tensor with G_total groups
// (0, 0 = the source is the whole tensor)
int launch_quantize_nhwc(const float *in, int8_t *out, int B, int C, int H, int W, int Cpad,
                         float mult, void *stream, int g_off = 0, int G_total = 0, int up = 1);      // up > 1: `in` is the input of an [upsample] by `up`
struct ConvI8Args {
    const int8_t *in_q;   // [B][H][W][Cpad]
    const int8_t *w_q;    // [Mpad][size*size][Cpad]
    const float *bias;    // [M]
    float *out;           // [B][M][OH][OW]; may be nullptr when only out_add is wanted fp32
    int32_t *dbg;         // optional int16-clamped accumulators [B][M][OH][OW]
    const float *add;     // optional fused [shortcut] (see ConvF32Args)
    float *out_add;
    int8_t *q_out;        // optional quantised side output for the next INT8 conv: act_q[B][q_G][OH][OW][16]
    float q_mult;         // the next layer's input_quant_multipler
    int q_G;              // the next layer's channel groups (Cpad/16)
    int B, Cpad, H, W, M, Mpad, OH, OW;
    int size, stride, pad;
    int act;
    float alpha1;         // R_MULT / (in_mult * w_mult)   (quantized.c:596)
    // host-proved absence of the two data-dependent corners of the exact epilogue (conv_i8_mfma.hip): bit 0 = no
    // output can fall in 0 < |y| < 1e-30 (alpha1 and every non-zero bias >= 1e-20), bit 1 = no side-output operand
    // can reach |y * q_mult| >= 32768 (only without a fused [shortcut]: (32767*alpha1 + max|bias|) * q_mult < 32768)
    int no_corner = 0;
};
// tile: 0 = heuristic, 1..5 see conv_i8_mfma.hip (tuning / tests); writes the kernel instance name
int launch_conv_i8(const ConvI8Args &a, int tile, void *stream, char *name, size_t name_len);

// ---- K1b: opt-in BF16 variant of the FP32 path (conv_bf16_mfma.hip) ----
// FP32 NCHW -> bf16 units act_h[B][Cpad/8][H][W][8] (round to nearest even); (g_off, G_total) as launch_quantize_nhwc
int launch_pack_bf16(const float *in, void *out, int B, int C, int H, int W, int Cpad, void *stream,
                     int g_off = 0, int G_total = 0);
struct ConvBf16Args {
    const void *in_h;     // act_h[B][Cpad/8][H][W][8] bf16
    const void *w_h;      // [K8pad][Mpad][8] bf16, K8 = tap * (Cpad/8) + channel group
    const float *bias;    // [M]
    float *out;           // [B][M][OH][OW] FP32; may be nullptr when only out_add / h_out is wanted
    const float *add;     // optional fused [shortcut] (see ConvF32Args)
    float *out_add;
    void *h_out;          // optional bf16 side output for the next BF16 conv: act_h[B][h_G][OH][OW][8]
    int h_G;              // the next layer's channel groups (Cpad/8)
    int B, Cpad, H, W, M, Mpad, OH, OW;
    int size, stride, pad;
    int act;
};
int launch_conv_bf16(const ConvBf16Args &a, int tile, void *stream, char *name, size_t name_len);

// ---- K3: XNOR path ----
// K3a: sign bits of FP32 NCHW packed along channels -> [B][H][W][Cw] 64-bit words, bit = (x > 0)
int launch_pack_sign_bits(const float *in, uint64_t *out, int B, int C, int H, int W, int Cw, void *stream);
struct ConvXnorArgs {
    const uint64_t *in_bits;  // [B][H][W][Cw]
    const uint64_t *w_bits;   // [Mpad/2][Cw][2][9] (filter pairs interleaved per channel word); channel-pad bits = 1
    const float *mean;        // [M]
    const float *bias;        // [M]
    float *out;               // [B][M][H][W], or nullptr when only out_bits / out_add is wanted
    const float *add = nullptr;     // optional fused [shortcut]: out_add = act(conv) + add (the reference GPU path fuses
    float *out_add = nullptr;       // the same pair, src/additionally.c:326-339)
    uint64_t *out_bits = nullptr;   // optional sign words of the result for a following XNOR layer: [B][ceil(M/64)][H][W]
    int32_t *dbg;             // optional match counts
    const int *thr = nullptr; // optional [Mpad] count thresholds (launch_xnor_thresholds): sign of the result = (count >= thr[m])
    int B, C, Cw, H, W, M, Mpad;
    int act;
    int ft_mode = 0;          // filters per workgroup: 0 = by grid depth, 64 / 32 = forced (A/B runs, tests)
};
// writes the kernel instance name (filter tile, word width, threshold epilogue) into name[name_len] when name != nullptr
int launch_conv_xnor(const ConvXnorArgs &a, void *stream, char *name = nullptr, size_t name_len = 0);
// thr[m] = the smallest match count whose result (2*count - K) * mean[m] + bias[m] is > 0 (K + 1 if none), m < M;
// INT_MAX for the pad filters; *bad (device int, zeroed by the caller) counts filters whose result is NOT a step
// function of the count (then thr must not be used)
int launch_xnor_thresholds(const float *mean, const float *bias, int *thr, int *bad, int M, int Mpad, int K, void *stream);
// K3c: max-pooling in the sign domain (OR of the window's sign words), and FP32 -> pooled sign words in one pass
int launch_bit_maxpool(const uint64_t *in, uint64_t *out, int B, int Cw, int H, int W, int OH, int OW,
                       int size, int stride, int pad, void *stream);
int launch_maxpool_sign_pack(const float *in, uint64_t *out, int B, int C, int Cw, int H, int W, int OH, int OW,
                             int size, int stride, int pad, void *stream);

// ---- K4..K9: small coalesced layers ----
int launch_maxpool(const float *in, float *out, int B, int C, int H, int W, int OH, int OW,
                   int size, int stride, int pad, void *stream);
int launch_shortcut(const float *in, const float *add, float *out, int B,
                    int w1, int h1, int c1,   // dims of `add`
                    int w2, int h2, int c2,   // dims of in/out
                    int act, void *stream);
int launch_upsample(const float *in, float *out, int B, int C, int H, int W, int stride, float scale, void *stream);
int launch_copy_rows(const float *src, float *dst, int rows, int row_elems, size_t src_stride, size_t dst_stride, void *stream);
int launch_yolo(const float *in, float *out, int B, int n, int classes, int wh, void *stream);
// tree_group_size (device, groups entries) != nullptr: softmax per group of the class vector (softmax_tree, YOLO9000)
int launch_region(const float *in, float *out, int B, int n, int classes, int coords, int wh, int softmax, void *stream,
                  const int *tree_group_size = nullptr, int tree_groups = 0);
// x -> (x > 0 ? 1 : -1)   binarize_cpu, src/additionally.c:128-134
int launch_binarize(const float *in, float *out, size_t n, void *stream);
// x = activate(x, act) in place: activate_array_cpu_custom for activations other than LINEAR / LEAKY (activations.h)
int launch_activate(float *x, size_t n, int act, void *stream);
// second stage of a split-K convolution: out = act(ws[0] + ws[1] + ... (in this order) + bias) [, out_add = out + add]
int launch_splitk_finish(const float *ws, int parts, size_t part_stride, const float *bias, int B, int M, int OHW, int act,
                         const float *add, float *out, float *out_add, void *stream);
int launch_reorg(const float *in, float *out, int B, int out_c, int out_h, int out_w, int stride, void *stream);

// ---- K10: detection compaction ----
struct HeadDesc {
    const float *out;     // head layer output (device)
    int type;             // YL_YOLO / YL_REGION
    int w, h, n, classes, outputs;
    float anchors_w[16], anchors_h[16];   // already selected through mask[]
    const int *tree_parent = nullptr;     // REGION with a softmax tree: parent[classes] (device) -> hierarchical decode
};
int launch_compact(const HeadDesc *heads, int n_heads, int B, int netw, int neth, float thresh,
                   int cap, int row_stride, float *records, int *counts, void *stream);

// K11 (detect.hip): per-image correct_yolo_boxes + do_nms_sort over compacted records.
// rec_scratch[B][cap][6+classes] is consumed (its prob columns are zeroed in place); rows come out
// in rec_out in the reference's final order.  Source image sizes travel as a kernel argument
// (w | h << 16): mode 0 = network size, 1 = wh[0] for every image, 2 = wh[b].
constexpr int NMS_MAX_CAP = 4096;      // 30 bytes of LDS per record: 123 KB of the CU's 160 KB
constexpr int NMS_MAX_DIMS = 256;
struct ImgDims {
    int mode;
    uint32_t wh[NMS_MAX_DIMS];
};
// meta = unsigned[B][1 + (classes+31)/32] scratch (per image: `total`, class bitmap) for the
// (image, class)-parallel path (mode 1); nullptr or mode 0 = one workgroup per image
int launch_nms(float *rec_scratch, const int *counts, int B, int cap, int classes, float nms, int netw, int neth,
               const ImgDims &dims, int relative, int letter, float *rec_out, int *counts_out, unsigned *meta,
               int mode, void *stream);

// K13 (layers.hip): per-image histogram of lround(|x| / bin_width), saturated; hist = unsigned[batch][max_bin]
int launch_hist_abs(const float *x, size_t per_image, int batch, int max_bin, float bin_width, unsigned *hist, void *stream);

// K12 (preprocess.hip): HWC u8 [sh][sw][sc] -> resize_image'd CHW float [sc][h][w] in [0,1]
int launch_load_resize_u8(const uint8_t *pix, int sw, int sh, int sc, int w, int h, float *out, void *stream);

}  // namespace yl
