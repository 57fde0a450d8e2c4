// runtime.hip -- device runtime + the C-ABI of libyolo2hip.so.
//
// Replaces the reference's device variant of L3 (forward_network_gpu_cudnn,
// src/yolov2_forward_network_gpu.cu:443-491; network_predict_gpu_cudnn :547-573)
// and the device half of L0 (cuda_make_array / cuda_push_array / cuda_pull_array,
// src/gpu.cu:97-266).  Design: every layer output stays resident in HBM for the
// whole forward (route/shortcut index arbitrary earlier layers; 288 GB makes
// batch 64 at 608x608 a ~25 GB working set), all launches go to one HIP stream,
// there is no host synchronisation inside a forward, and no CPU fallback: every
// device entry point fails with YL_ERR_DEVICE when HIP reports an error.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "kernels.h"
#include "yl_internal.h"
#include "../../include/yolo2_hip_lab.h"

namespace yl {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

#define YL_HIP(expr)                                                                    \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));               \
            return YL_ERR_DEVICE;                                                       \
        }                                                                               \
    } while (0)

// stage_h2d / stage_d2h set the error text themselves
#define YL_STAGE(call)                                                                  \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != YL_OK) return rc_;                                                   \
    } while (0)

#define YL_LAUNCH(call, what)                                                           \
    do {                                                                                \
        int e_ = (call);                                                                \
        if (e_ != 0) {                                                                  \
            set_error(std::string(what) + ": " + hipGetErrorString((hipError_t)e_));    \
            return YL_ERR_DEVICE;                                                       \
        }                                                                               \
    } while (0)

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
// LINEAR and LEAKY live in the convolution kernels' epilogues; every other activation of activate()
// (src/additionally.h:132-165) is applied by a pass of its own behind a linear epilogue (layers.hip: activate_kernel)
static inline bool hot_activation(int a) { return a == YL_LINEAR || a == YL_LEAKY; }

// K ranges of an FP32 convolution on K1x / K1r (yl_network_set_split_k).  Measured on yolov3-608 at 8 images per GPU
// (profiles/r6_split_k_b8.txt): a split pays where ONE workgroup's K loop is the layer's time -- a grid of less than four 64 x 64
// workgroups per CU AND a deep K (>= 64 panels of 16 channels x taps: the 19 x 19 1x1 and 3x3 layers, the 3x3 / stride-2 layer in
// front of them), or a grid below one workgroup per CU from 32 panels on -- and costs where the second stage's pass over the
// partial sums outweighs it (the 38 x 38 and 76 x 76 layers: left alone).  2 .. 4 ranges, a divisor of the channel blocks, >= 4
// blocks (64 channels) each, as many as bring the grid to ~8 workgroups per CU.  A function of the layer and the batch only: the
// same network at the same batch always splits the same way.
static int split_k_parts(int B, int C, int M, int size, int OH, int OW, int n_cu)
{
    if ((C % 16) != 0 || C < 128 || M <= 32 || (size != 1 && size != 3)) return 1;
    const long long px = (long long)B * OH * OW;
    const long long nwg = (long long)((M + 63) / 64) * ((px + 63) / 64);
    const int cblocks = C / 16;
    const int depth = cblocks * size * size;
    if (nwg >= 4LL * n_cu) return 1;
    if (!(depth >= 64 || (nwg < n_cu && depth >= 32))) return 1;
    int want = (int)((8LL * n_cu + nwg - 1) / nwg);
    if (want > 4) want = 4;
    if (want < 2) want = 2;
    auto fits = [&](int s) { return s >= 2 && s <= 4 && cblocks % s == 0 && cblocks / s >= 4; };
    // (rounding DOWN where the wanted count does not divide the blocks: four ranges instead of two on the 19 x 19 3x3 layers measured
    //  0.150 vs 0.138 ms -- the second stage reads every range)
    for (int s = want; s >= 2; --s)
        if (fits(s)) return s;
    return 1;
}
// the shapes conv_f32_smallk.hip accepts (smallk_applicable, minus what only the launch knows)
// `batch`: both first-layer kernels address the input with 32-bit byte offsets (input tensor < 4 GiB) -- a batch
// beyond that must not get the sign-word plan, whose only producers they are (launch_conv_f32 would fail instead of
// falling back to the FP32 path)
static inline bool first_layer_kernel_takes(const Layer &l, int batch)
{
    const unsigned long long in_bytes = (unsigned long long)batch * l.c * l.h * l.w * sizeof(float);
    return l.size >= 1 && l.size <= 5 && l.size * l.size * l.c <= 32 && l.n <= 32 && !l.tapmajor && in_bytes < 0xFFFFFFFEull;
}
// the plan of the fusion pass below: FP32 first layer -> [maxpool] -> XNOR conv hands over sign words
static inline bool first_layer_sign_plan(const Network &net, const Layer &pp)
{
    return net.fuse && !net.debug && (net.conv_opts.variant & 8) && (net.conv_opts.force_tile == 0 || net.conv_opts.force_tile == 41) &&
           first_layer_kernel_takes(pp, net.batch);
}

// float -> bf16, round to nearest even (what v_cvt_pk_bf16_f32 does to the activations on the device)
static inline uint16_t f32_to_bf16_rne(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);      // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

static void free_device(Network &net)
{
    if (net.device >= 0) (void)hipSetDevice(net.device);
    // nothing of this network may still be queued or running when its buffers go away (asynchronous entry points:
    // yl_network_forward, _detect_batch, _set_input_u8); the stream is non-blocking, so be explicit
    if (net.stream) (void)hipStreamSynchronize((hipStream_t)net.stream);
    for (Layer &l : net.layers) {
        if (l.d_output && !l.d_output_alias) (void)hipFree(l.d_output - ACT_FRONT_PAD);
        l.d_output = nullptr;
        if (l.host_in_heads) { l.host_output = nullptr; l.host_kind = HOST_NONE; l.host_in_heads = false; }
        if (l.d_weights_t) (void)hipFree(l.d_weights_t);
        if (l.d_wino32_u) (void)hipFree(l.d_wino32_u);
        l.d_wino32_u = nullptr;
        if (l.d_weights_x3) (void)hipFree(l.d_weights_x3);
        l.d_weights_x3 = nullptr;
        if (l.d_weights_r3) (void)hipFree(l.d_weights_r3);
        l.d_weights_r3 = nullptr;
        if (l.d_tile_ctr) (void)hipFree(l.d_tile_ctr);
        l.d_tile_ctr = nullptr;
        if (l.d_biases) (void)hipFree(l.d_biases);
        if (l.d_weights_i8) (void)hipFree(l.d_weights_i8);
        if (l.d_weights_bits) (void)hipFree(l.d_weights_bits);
        if (l.d_mean) (void)hipFree(l.d_mean);
        if (l.d_thr) (void)hipFree(l.d_thr);
        l.d_thr = nullptr; l.thr_ok = false;
        if (l.d_debug) (void)hipFree(l.d_debug);
        if (l.d_tree) (void)hipFree(l.d_tree);
        l.d_tree = nullptr;
        l.d_weights_t = nullptr; l.d_biases = nullptr; l.d_weights_i8 = nullptr;
        l.d_weights_bits = nullptr; l.d_mean = nullptr; l.d_debug = nullptr;
    }
    if (net.d_pack_src) (void)hipFree(net.d_pack_src);
    net.d_pack_src = nullptr; net.pack_src_bytes = 0;
    if (net.d_input) (void)hipFree(net.d_input - ACT_FRONT_PAD);
    if (net.d_qbuf) (void)hipFree(net.d_qbuf);
    if (net.d_bitbuf) (void)hipFree(net.d_bitbuf);
    if (net.d_binbuf) (void)hipFree(net.d_binbuf);
    net.d_binbuf = nullptr;
    if (net.h_pinned) (void)hipHostFree(net.h_pinned);
    if (net.h_heads) (void)hipHostFree(net.h_heads);
    net.h_heads = nullptr; net.h_heads_floats = 0;
    if (net.h_u8) (void)hipHostFree(net.h_u8);
    if (net.d_u8) (void)hipFree(net.d_u8);
    for (void *e : net.u8_events) if (e) (void)hipEventDestroy((hipEvent_t)e);
    net.u8_events.clear();
    if (net.u8_resized) (void)hipEventDestroy((hipEvent_t)net.u8_resized);
    net.u8_resized = nullptr;
    net.h_u8 = nullptr; net.d_u8 = nullptr; net.u8_stride = 0;
    if (net.h_det_rows) (void)hipHostFree(net.h_det_rows);
    net.h_det_rows = nullptr; net.h_det_bytes = 0; net.det_cache_valid = false;
    if (net.h_det_counts) (void)hipHostFree(net.h_det_counts);
    net.h_det_counts = nullptr;
    if (net.d_det_scratch) (void)hipFree(net.d_det_scratch);
    if (net.d_det_out) (void)hipFree(net.d_det_out);
    if (net.d_det_counts) (void)hipFree(net.d_det_counts);
    if (net.d_det_meta) (void)hipFree(net.d_det_meta);
    net.d_det_meta = nullptr; net.det_meta_bytes = 0;
    if (net.d_ks_ws) (void)hipFree(net.d_ks_ws);
    if (net.d_ks_zeros) (void)hipFree(net.d_ks_zeros);
    net.d_ks_ws = nullptr; net.d_ks_zeros = nullptr; net.ks_ws_floats = 0;
    net.d_det_scratch = nullptr; net.d_det_out = nullptr; net.d_det_counts = nullptr;
    net.det_scratch_bytes = net.det_out_bytes = 0;
    net.d_input = nullptr; net.d_qbuf = nullptr; net.d_bitbuf = nullptr; net.h_pinned = nullptr;
    for (void *e : net.layer_events) if (e) (void)hipEventDestroy((hipEvent_t)e);
    net.layer_events.clear();
    for (void **st : {&net.in_stream, &net.out_stream}) {
        if (*st) { (void)hipStreamSynchronize((hipStream_t)*st); (void)hipStreamDestroy((hipStream_t)*st); }
        *st = nullptr;
    }
    for (void *e : net.in_events) if (e) (void)hipEventDestroy((hipEvent_t)e);
    net.in_events.clear();
    for (void *e : net.head_events) if (e) (void)hipEventDestroy((hipEvent_t)e);
    net.head_events.clear();
    for (void *e : net.chunk_events) if (e) (void)hipEventDestroy((hipEvent_t)e);
    net.chunk_events.clear();
    if (net.ev0) (void)hipEventDestroy((hipEvent_t)net.ev0);
    if (net.ev1) (void)hipEventDestroy((hipEvent_t)net.ev1);
    net.ev0 = net.ev1 = nullptr;
    if (net.own_stream && net.stream) (void)hipStreamDestroy((hipStream_t)net.stream);
    net.stream = nullptr; net.own_stream = false;
    net.on_device = false;
}

// ------------------------------------------------------------------ upload
// One layer's prepared weights, as they are, into the device scratch the pack kernels read (pack.hip).  The
// previous layer's pack kernels (same stream) must have finished with the scratch first.
static int pack_source_to_device(Network &net, const void *src, size_t bytes, const float *mean, int M)
{
    const size_t mean_off = (bytes + 255) & ~(size_t)255;
    const size_t need = mean_off + (mean ? sizeof(float) * (size_t)M : 0);
    YL_HIP(hipStreamSynchronize((hipStream_t)net.stream));
    if (need > net.pack_src_bytes) {
        if (net.d_pack_src) (void)hipFree(net.d_pack_src);
        net.d_pack_src = nullptr; net.pack_src_bytes = 0;
        YL_HIP(hipMalloc((void **)&net.d_pack_src, need + need / 4));
        net.pack_src_bytes = need + need / 4;
    }
    YL_STAGE(stage_h2d(net.device, net.d_pack_src, src, bytes));
    if (mean) YL_STAGE(stage_h2d(net.device, net.d_pack_src + mean_off, mean, sizeof(float) * (size_t)M));
    return YL_OK;
}

// Kernel-layout weight images of one convolution.  net.device_pack (default): the prepared weights travel once and
// pack.hip's kernels write the images; otherwise the host loops below build them (the reference implementation the
// device packers are checked against bit for bit, tests/test_gpu_prep.py) and the inflated images are uploaded.
static int upload_conv(Network &net, Layer &l)
{
    const int K = l.size * l.size * l.c;
    const int M = l.n;
    const int taps = l.size * l.size;
    void *s = net.stream;
    YL_HIP(hipMalloc((void **)&l.d_biases, sizeof(float) * M));
    YL_STAGE(stage_h2d(net.device, l.d_biases, l.biases.data(), sizeof(float) * M));
    for (size_t &b : l.packed_bytes) b = 0;
    if (l.conv_mode == CONV_F32) {
        // xnor conv outside the 3x3/stride-1/pad-1 bit path: the reference falls back to the FP32
        // GEMM on binarised operands (src/yolov2_forward_network.c:40-50,204-211): weights +-mean
        // (binarize_weights, src/additionally.c:113-126), input +-1 (binarize_cpu :128-134), ZERO padding.
        const bool xnor_fallback = l.xnor != 0;
        if (xnor_fallback && !l.xnor_ready) { set_error("XNOR layer without yl_network_calculate_binary_weights()"); return YL_ERR_STATE; }
        l.binarize_input = xnor_fallback;
        if (xnor_fallback) {
            const size_t bb = (size_t)net.batch * l.c * l.h * l.w * sizeof(float);
            if (bb > net.binbuf_bytes) net.binbuf_bytes = bb;
        }
        // k-major panel layout [Kpad][Mpad], zero padded
        // K order: (c,ky,kx) like im2col_cpu, or -- for the pipelined kernel when C % 16 == 0 --
        // tap-major (ky,kx,c) so that one BK panel shares a single tap (scalar decode once per panel)
        l.Kpad = round_up(K, 32);
        l.Mpad = round_up(M, 256);
        l.tapmajor = (l.size > 1 && l.size <= 5 && (l.c % 16) == 0) ? 1 : 0;
        // 3x3 / stride 1 / pad 1: also the Winograd F(2x2,3x3) form of the same weights (K1w).
        // Not for the xnor fallback: its +-mean weights would pick up G's halves and the layer is
        // specified by the reference as an exact +-1 GEMM.
        const bool wino = net.conv_opts.winograd && !xnor_fallback && wino_applicable(l.c, M, l.size, l.stride, l.pad) &&
                          l.out_h == l.h && l.out_w == l.w && l.h >= 4 && l.w >= 4;
        const size_t wt_floats = (size_t)l.Kpad * l.Mpad;
        const size_t u_floats = !wino ? 0 : wino32_packed_floats(l.c, M);
        YL_HIP(hipMalloc((void **)&l.d_weights_t, wt_floats * sizeof(float)));
        l.packed_bytes[0] = wt_floats * sizeof(float);
        if (wino) {
            YL_HIP(hipMalloc((void **)&l.d_wino32_u, u_floats * sizeof(float)));
            l.packed_bytes[1] = u_floats * sizeof(float);
            YL_HIP(hipMalloc((void **)&l.d_tile_ctr, 8 * sizeof(unsigned)));
            YL_HIP(hipMemsetAsync(l.d_tile_ctr, 0, 8 * sizeof(unsigned), (hipStream_t)s));
        }
        if (net.device_pack) {
            int rc = pack_source_to_device(net, l.weights.data(), sizeof(float) * (size_t)M * K, xnor_fallback ? l.mean_arr.data() : nullptr, M);
            if (rc != YL_OK) return rc;
            const float *d_src = reinterpret_cast<const float *>(net.d_pack_src);
            const float *d_mean = xnor_fallback ? reinterpret_cast<const float *>(net.d_pack_src + ((sizeof(float) * (size_t)M * K + 255) & ~(size_t)255)) : nullptr;
            YL_HIP(hipMemsetAsync(l.d_weights_t, 0, wt_floats * sizeof(float), (hipStream_t)s));
            YL_LAUNCH(dev_pack_kmajor(d_src, d_mean, l.d_weights_t, M, l.c, taps, l.Mpad, l.tapmajor, s), "pack_kmajor");
            if (wino) YL_LAUNCH(dev_pack_wino(d_src, l.d_wino32_u, l.c, M, s), "pack_wino");
        } else {
            std::vector<float> wt(wt_floats, 0.f);
            for (int m = 0; m < M; ++m)
                for (int c = 0; c < l.c; ++c)
                    for (int t = 0; t < taps; ++t) {
                        const int k_ref = c * taps + t;
                        // tap-major inside 16-channel blocks: k = ((c/16)*taps + t)*16 + c%16
                        const int k_dev = l.tapmajor ? (((c / 16) * taps + t) * 16 + (c % 16)) : k_ref;
                        float wv = l.weights[(size_t)m * K + k_ref];
                        if (xnor_fallback) wv = (wv > 0.f) ? l.mean_arr[m] : -l.mean_arr[m];
                        wt[(size_t)k_dev * l.Mpad + m] = wv;
                    }
            YL_STAGE(stage_h2d(net.device, l.d_weights_t, wt.data(), wt.size() * sizeof(float)));
            if (wino) {
                std::vector<float> u32(u_floats);
                wino32_pack_weights(l.weights.data(), l.c, M, u32.data());
                YL_STAGE(stage_h2d(net.device, l.d_wino32_u, u32.data(), u32.size() * sizeof(float)));
            }
        }
        // the split-operand form of the same weights for K1x (three bf16 pieces per weight, exact); on the device from the raw
        // weights the packers above already uploaded, or on the host -- bit-identical (test_gpu_prep.py)
        if (!xnor_fallback && x3_applicable(l.c, M, l.size, l.stride, l.pad)) {
            const size_t xb = x3_packed_bytes(l.c, M, l.size);
            YL_HIP(hipMalloc(&l.d_weights_x3, xb));
            l.packed_bytes[4] = xb;
            if (net.device_pack) {
                YL_HIP(hipMemsetAsync(l.d_weights_x3, 0, xb, (hipStream_t)s));
                YL_LAUNCH(dev_pack_x3(reinterpret_cast<const float *>(net.d_pack_src), l.d_weights_x3, M, l.c, taps, (int)(xb / ((size_t)(l.c / 16) * taps * 96)), s), "pack_x3");
            } else {
                std::vector<unsigned char> w3(xb);
                x3_pack_weights(l.weights.data(), l.c, M, l.size, w3.data());
                YL_STAGE(stage_h2d(net.device, l.d_weights_x3, w3.data(), xb));
            }
        }
        // 3x3 / stride 1 / pad 1: the row-transformed weights U = G g as three bf16 pieces for K1r (conv_f32_row3.hip)
        // (kept whenever Winograd packing is on, whatever variant bit 11 says now: yl_network_set_variant / _set_conv_tile may move a
        //  layer to K1r after yl_network_to_device and must never fail a forward pass; the memory this costs is in INTEGRATION.md)
        if (wino && row3_applicable(l.c, M, l.size, l.stride, l.pad)) {
            const size_t rb = row3_packed_bytes(l.c, M);
            YL_HIP(hipMalloc(&l.d_weights_r3, rb));
            l.packed_bytes[5] = rb;
            if (net.device_pack) {
                YL_HIP(hipMemsetAsync(l.d_weights_r3, 0, rb, (hipStream_t)s));
                YL_LAUNCH(dev_pack_row3(reinterpret_cast<const float *>(net.d_pack_src), l.d_weights_r3, M, l.c, (int)(rb / ((size_t)(l.c / 16) * 3 * 24 * 16)), s), "pack_row3");
            } else {
                std::vector<unsigned char> wr(rb);
                row3_pack_weights(l.weights.data(), l.c, M, wr.data());
                YL_STAGE(stage_h2d(net.device, l.d_weights_r3, wr.data(), rb));
            }
        }
    } else if (l.conv_mode == CONV_INT8) {
        if (!l.quant_ready) { set_error("INT8 layer without yl_network_quantize()"); return YL_ERR_STATE; }
        // k-major panels of 16-byte units [K16pad][Mpad][16], K16 index = tap*G + cg, zero padded;
        // G = channel groups of 16 rounded up to a power of two (shift/mask decode in the kernel)
        if (l.size > 5) { set_error("INT8 conv: size > 5 unsupported"); return YL_ERR_UNSUPPORTED; }
        int G = 1;
        while (G * 16 < l.c) G *= 2;
        l.Cpad = G * 16;
        l.Mpad = round_up(M, 128);
        const int K16 = taps * G;
        const int K16pad = round_up(K16, 8);
        const size_t bytes = (size_t)K16pad * l.Mpad * 16;
        YL_HIP(hipMalloc((void **)&l.d_weights_i8, bytes));
        l.packed_bytes[2] = bytes;
        if (net.device_pack) {
            int rc = pack_source_to_device(net, l.weights_int8.data(), (size_t)M * K, nullptr, M);
            if (rc != YL_OK) return rc;
            YL_HIP(hipMemsetAsync(l.d_weights_i8, 0, bytes, (hipStream_t)s));
            YL_LAUNCH(dev_pack_i8_units(reinterpret_cast<const int8_t *>(net.d_pack_src), l.d_weights_i8, M, l.c, taps, G, l.Mpad, s), "pack_i8_units");
        } else {
            std::vector<int8_t> wq(bytes, 0);
            for (int m = 0; m < M; ++m)
                for (int c = 0; c < l.c; ++c)
                    for (int t = 0; t < taps; ++t) {
                        const int g = t * G + c / 16;
                        wq[((size_t)g * l.Mpad + m) * 16 + (c % 16)] = l.weights_int8[((size_t)m * l.c + c) * taps + t];
                    }
            YL_STAGE(stage_h2d(net.device, l.d_weights_i8, wq.data(), wq.size()));
        }
        const size_t qb = (size_t)net.batch * l.h * l.w * l.Cpad;
        if (qb > net.qbuf_bytes) net.qbuf_bytes = qb;
        l.bias_abs_max = 0.f; l.bias_abs_min_nz = 3.0e38f;
        for (int m = 0; m < M; ++m) {
            const float b = fabsf(l.biases[m]);
            if (!(b <= 3.0e38f)) { l.bias_abs_max = -1.f; break; }          // inf / nan: no proof
            if (b > l.bias_abs_max) l.bias_abs_max = b;
            if (b != 0.f && b < l.bias_abs_min_nz) l.bias_abs_min_nz = b;
        }
    } else if (l.conv_mode == CONV_BF16) {
        // the INT8 layout with 8 bf16 channels per 16-byte unit: [K8pad][Mpad][8], K8 index = tap*G + cg
        int G = 1;
        while (G * 8 < l.c) G *= 2;
        l.Cpad = G * 8;
        l.Mpad = round_up(M, 128);
        const int K8 = taps * G;
        const int K8pad = round_up(K8, 8);
        const size_t elems = (size_t)K8pad * l.Mpad * 8;
        YL_HIP(hipMalloc((void **)&l.d_weights_i8, elems * sizeof(uint16_t)));
        l.packed_bytes[2] = elems * sizeof(uint16_t);
        if (net.device_pack) {
            int rc = pack_source_to_device(net, l.weights.data(), sizeof(float) * (size_t)M * K, nullptr, M);
            if (rc != YL_OK) return rc;
            YL_HIP(hipMemsetAsync(l.d_weights_i8, 0, elems * sizeof(uint16_t), (hipStream_t)s));
            YL_LAUNCH(dev_pack_bf16_units(reinterpret_cast<const float *>(net.d_pack_src), reinterpret_cast<uint16_t *>(l.d_weights_i8),
                                          M, l.c, taps, G, l.Mpad, s), "pack_bf16_units");
        } else {
            std::vector<uint16_t> wh(elems, 0);
            for (int m = 0; m < M; ++m)
                for (int c = 0; c < l.c; ++c)
                    for (int t = 0; t < taps; ++t) {
                        const int g = t * G + c / 8;
                        wh[((size_t)g * l.Mpad + m) * 8 + (c % 8)] = f32_to_bf16_rne(l.weights[((size_t)m * l.c + c) * taps + t]);
                    }
            YL_STAGE(stage_h2d(net.device, l.d_weights_i8, wh.data(), wh.size() * sizeof(uint16_t)));
        }
        const size_t qb = (size_t)net.batch * l.h * l.w * l.Cpad * 2;
        if (qb > net.qbuf_bytes) net.qbuf_bytes = qb;
    } else {   // CONV_XNOR
        if (!l.xnor_ready) { set_error("XNOR layer without yl_network_calculate_binary_weights()"); return YL_ERR_STATE; }
        // 64-bit sign words along channels, filter pairs interleaved: [Mpad/2][Cw][2][9]; bit = (w > 0) (src/additionally.c:123,1544);
        // channel-pad bits are 1 in the weights and 0 in the activations so they never match.
        l.Cw = (l.c + 63) / 64;
        l.Mpad = round_up(M, 64);
        const size_t words = (size_t)l.Mpad * l.Cw * 9 + 18;       // + one step: the kernel's last prefetch
        YL_HIP(hipMalloc((void **)&l.d_weights_bits, words * sizeof(uint64_t)));
        l.packed_bytes[3] = words * sizeof(uint64_t);
        if (net.device_pack) {
            int rc = pack_source_to_device(net, l.weights.data(), sizeof(float) * (size_t)M * K, nullptr, M);
            if (rc != YL_OK) return rc;
            YL_HIP(hipMemsetAsync(l.d_weights_bits, 0xFF, words * sizeof(uint64_t), (hipStream_t)s));
            YL_LAUNCH(dev_pack_xnor_words(reinterpret_cast<const float *>(net.d_pack_src), l.d_weights_bits, M, l.c, l.Cw, s), "pack_xnor_words");
        } else {
            std::vector<uint64_t> wb(words, ~0ull);
            for (int m = 0; m < M; ++m)
                for (int t = 0; t < 9; ++t)
                    for (int cw = 0; cw < l.Cw; ++cw) {
                        uint64_t word = 0;
                        for (int b = 0; b < 64; ++b) {
                            const int c = cw * 64 + b;
                            const bool bit = (c < l.c) ? (l.weights[((size_t)m * l.c + c) * 9 + t] > 0.f) : true;
                            if (bit) word |= (1ull << b);
                        }
                        wb[((size_t)(m / 2) * l.Cw + cw) * 18 + (m % 2) * 9 + t] = word;
                    }
            YL_STAGE(stage_h2d(net.device, l.d_weights_bits, wb.data(), wb.size() * sizeof(uint64_t)));
        }
        YL_HIP(hipMalloc((void **)&l.d_mean, sizeof(float) * M));
        YL_STAGE(stage_h2d(net.device, l.d_mean, l.mean_arr.data(), sizeof(float) * M));
        {   // thresholds of the sign-only epilogue (conv_xnor.hip), evaluated on the device with the kernel's own expression
            YL_HIP(hipMalloc((void **)&l.d_thr, sizeof(int) * ((size_t)l.Mpad + 1)));
            YL_HIP(hipMemsetAsync(l.d_thr + l.Mpad, 0, sizeof(int), (hipStream_t)s));
            YL_LAUNCH(launch_xnor_thresholds(l.d_mean, l.d_biases, l.d_thr, l.d_thr + l.Mpad, M, l.Mpad, 9 * l.c, s), "xnor_thresholds");
            YL_HIP(hipStreamSynchronize((hipStream_t)s));
            int bad = 1;
            YL_STAGE(stage_d2h(net.device, &bad, l.d_thr + l.Mpad, sizeof(int)));
            l.thr_ok = bad == 0;
        }
        size_t bb = (size_t)net.batch * l.h * l.w * l.Cw * sizeof(uint64_t);
        const size_t ob = (size_t)net.batch * l.h * l.w * ((M + 63) / 64) * sizeof(uint64_t);      // sign words of its result
        if (ob > bb) bb = ob;
        if (bb > net.bitbuf_bytes) net.bitbuf_bytes = bb;
    }
    if (net.debug && l.conv_mode != CONV_F32 && l.conv_mode != CONV_BF16) {
        YL_HIP(hipMalloc((void **)&l.d_debug, sizeof(int32_t) * (size_t)net.batch * l.outputs));
    }
    return YL_OK;
}

static int to_device(Network &net, int device)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        set_error("no HIP device visible (libyolo2hip has no CPU fallback)");
        return YL_ERR_DEVICE;
    }
    if (device < 0 || device >= count) { set_error("device index out of range"); return YL_ERR_ARG; }
    if (net.on_device) free_device(net);
    net.device = device;
    YL_HIP(hipSetDevice(device));
    if (!net.stream) {
        hipStream_t s;
        YL_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        net.stream = s; net.own_stream = true;
    }
    select_conv_modes(net);
    net.qbuf_bytes = 0; net.bitbuf_bytes = 0; net.binbuf_bytes = 0;
    const size_t in_elems = (size_t)net.batch * net.c * net.h * net.w;
    // every activation tensor the library owns has ACT_FRONT_PAD readable floats in front of it: the Winograd kernel reads
    // ONE float before a tensor (column -1 of the first patch row of the first image, masked to zero in the transform)
    YL_HIP(hipMalloc((void **)&net.d_input, (in_elems + ACT_FRONT_PAD + ACT_TAIL_PAD) * sizeof(float)));
    net.d_input += ACT_FRONT_PAD;            // (before the next fallible call: free_device frees d_input - ACT_FRONT_PAD)
    YL_HIP(hipMemsetAsync(net.d_input - ACT_FRONT_PAD, 0, ACT_FRONT_PAD * sizeof(float), (hipStream_t)net.stream));
    net.pinned_bytes = in_elems * sizeof(float);
    YL_HIP(hipHostMalloc(&net.h_pinned, net.pinned_bytes, hipHostMallocDefault));

    for (size_t i = 0; i < net.layers.size(); ++i) {
        Layer &l = net.layers[i];
        const size_t out_elems = (size_t)net.batch * l.outputs;
        if (l.type == YL_ROUTE && l.n == 1) {
            l.d_output = net.layers[l.input_layers[0]].d_output;     // pure alias, no copy
            l.d_output_alias = true;
        } else {
            l.d_output_alias = false;
            YL_HIP(hipMalloc((void **)&l.d_output, (out_elems + ACT_FRONT_PAD + ACT_TAIL_PAD) * sizeof(float)));
            l.d_output += ACT_FRONT_PAD;     // (before the next fallible call: free_device frees d_output - ACT_FRONT_PAD)
            YL_HIP(hipMemsetAsync(l.d_output - ACT_FRONT_PAD, 0, ACT_FRONT_PAD * sizeof(float), (hipStream_t)net.stream));      // finite: it is multiplied by 0
        }
        if (l.type == YL_CONVOLUTIONAL) {
            int rc = upload_conv(net, l);
            if (rc != YL_OK) return rc;
        }
        if (l.type == YL_REGION && !l.tree_parent.empty()) {
            // parents must precede their children (hierarchy_predictions multiplies in index order) and the groups
            // must tile the class vector: checked here so the kernels need no guards
            long long covered = 0;
            for (int g : l.tree_group_size) { if (g < 0) covered = -1; if (covered >= 0) covered += g; }
            bool ok = covered == (long long)l.classes;
            for (int j = 0; j < l.classes && ok; ++j) ok = l.tree_parent[j] < j;
            if (!ok) { set_error("region softmax tree: groups do not tile the classes or a parent follows its child"); return YL_ERR_CFG; }
            std::vector<int> t(l.tree_parent);
            t.insert(t.end(), l.tree_group_size.begin(), l.tree_group_size.end());
            YL_HIP(hipMalloc((void **)&l.d_tree, t.size() * sizeof(int)));
            YL_STAGE(stage_h2d(net.device, l.d_tree, t.data(), t.size() * sizeof(int)));
        }
    }
    // host side of the heads / last layer (what network_predict_* leaves in l.output): one pinned block.  A layer
    // without a caller destination owns its region of it; a caller destination (yl_layer_desc.output) is served
    // by a DMA into the region + memcpy, never by handing the caller's pointer to the runtime (staging.hip).
    {
        size_t total = 0;
        for (size_t i = 0; i < net.layers.size(); ++i) {
            Layer &l = net.layers[i];
            const bool is_head = (l.type == YL_YOLO || l.type == YL_REGION);
            if (!(is_head || i + 1 == net.layers.size()) || l.host_kind == HOST_PINNED) continue;
            l.h_head_off = total;
            total += ((size_t)net.batch * l.outputs + 63) & ~(size_t)63;
        }
        if (total) {
            YL_HIP(hipHostMalloc((void **)&net.h_heads, total * sizeof(float), hipHostMallocDefault));
            net.h_heads_floats = total;
            memset(net.h_heads, 0, total * sizeof(float));
        }
        for (size_t i = 0; i < net.layers.size(); ++i) {
            Layer &l = net.layers[i];
            const bool is_head = (l.type == YL_YOLO || l.type == YL_REGION);
            if (!(is_head || i + 1 == net.layers.size()) || l.host_kind != HOST_NONE) continue;
            l.host_output = net.h_heads + l.h_head_off;
            l.host_kind = HOST_PINNED;
            l.host_in_heads = true;
        }
        // yl_network_predict's pipeline: two copy streams (PCIe is full duplex), per sub-batch an "input landed" event and per
        // head / last layer an event that forward() records on the compute stream behind the kernel that completes the tensor
        for (void **st : {&net.in_stream, &net.out_stream})
            if (!*st) {
                hipStream_t cs;
                YL_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
                *st = cs;
            }
        const size_t nl = net.layers.size();
        net.in_events.assign(PREDICT_MAX_SPLIT, nullptr);
        net.head_events.assign(PREDICT_MAX_SPLIT * nl, nullptr);
        for (int k = 0; k < PREDICT_MAX_SPLIT; ++k) {
            hipEvent_t e;
            YL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            net.in_events[k] = e;
            for (size_t i = 0; i < nl; ++i) {
                const Layer &l = net.layers[i];
                if (!(l.type == YL_YOLO || l.type == YL_REGION || i + 1 == nl)) continue;
                YL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                net.head_events[k * nl + i] = e;
            }
        }
    }
    // an FP32 first layer that hands sign words to an XNOR convolution through a [maxpool] writes PRE-pool words
    // (one per pixel of its own output) into the ring: make the slots large enough before they are allocated
    for (size_t i = 0; i + 2 < net.layers.size(); ++i) {
        const Layer &pp = net.layers[i], &pool = net.layers[i + 1], &cons = net.layers[i + 2];
        if (pp.type == YL_CONVOLUTIONAL && pp.conv_mode == CONV_F32 && !pp.xnor && first_layer_sign_plan(net, pp) &&
            pool.type == YL_MAXPOOL && cons.type == YL_CONVOLUTIONAL && cons.conv_mode == CONV_XNOR) {
            const size_t bb = (size_t)net.batch * pp.out_h * pp.out_w * sizeof(uint64_t);
            if (bb > net.bitbuf_bytes) net.bitbuf_bytes = bb;
        }
    }
    // three quantised-activation buffers used round-robin (layer j reads ring[j % 3]): a producer
    // one or two layers earlier can write layer j's input while reading its own
    if (net.qbuf_bytes) YL_HIP(hipMalloc((void **)&net.d_qbuf, 3 * net.qbuf_bytes));
    if (net.bitbuf_bytes) {
        net.bitbuf_bytes = (net.bitbuf_bytes + 255) & ~(size_t)255;
        YL_HIP(hipMalloc((void **)&net.d_bitbuf, 3 * net.bitbuf_bytes));
        YL_HIP(hipMemsetAsync(net.d_bitbuf, 0, 3 * net.bitbuf_bytes, (hipStream_t)net.stream));       // pad bits start out 0; every writer stores whole 64-bit words (conv_xnor.hip), the ring is reused across layers
    }
    if (net.binbuf_bytes) YL_HIP(hipMalloc((void **)&net.d_binbuf, net.binbuf_bytes));
    // ---- split-K workspace (yl_network_set_split_k): 4 ranges of the largest tensor a split layer can have + a zero bias ----
    if (net.split_k) {
        size_t mx = 0, mfil = 0;
        for (const Layer &l : net.layers)
            if (l.type == YL_CONVOLUTIONAL && l.conv_mode == CONV_F32 && split_k_parts(net.batch, l.c, l.n, l.size, l.out_h, l.out_w, device_cu_count()) > 1) {
                mx = std::max(mx, (size_t)net.batch * l.outputs);
                mfil = std::max(mfil, (size_t)l.n);
            }
        if (mx) {
            net.ks_ws_floats = 4 * mx;
            YL_HIP(hipMalloc((void **)&net.d_ks_ws, net.ks_ws_floats * sizeof(float)));
            YL_HIP(hipMalloc((void **)&net.d_ks_zeros, (mfil + 256) * sizeof(float)));
            YL_HIP(hipMemsetAsync(net.d_ks_zeros, 0, (mfil + 256) * sizeof(float), (hipStream_t)net.stream));
        }
    }
    // ---- optional conv+shortcut fusion plan ----
    for (Layer &l : net.layers) { l.fused_shortcut = -1; l.fused_yolo = -1; l.fused_pool = -1; l.fused_into_conv = false;
                                  l.two_src_up = -1; l.two_src_other = -1; l.two_src_conv = -1; l.two_src_skipped = false; }
    if (net.fuse && !net.debug) {
        const int nl = (int)net.layers.size();
        for (int i = 1; i < nl; ++i) {
            Layer &sc = net.layers[i];
            Layer &cv = net.layers[i - 1];
            if (sc.type != YL_SHORTCUT || cv.type != YL_CONVOLUTIONAL) continue;
            if (sc.activation != YL_LINEAR) continue;
            if (!hot_activation(cv.activation)) continue;        // its activation is a pass of its own over the FP32 tensor
            if (!(sc.w == sc.out_w && sc.h == sc.out_h && sc.c == sc.out_c)) continue;   // same-shape add only
            if (sc.index == i - 1) continue;
            bool referenced = (i - 1 == nl - 1);
            for (int j = i + 1; j < nl && !referenced; ++j) {
                const Layer &o = net.layers[j];
                if (o.type == YL_SHORTCUT && o.index == i - 1) referenced = true;
                if (o.type == YL_ROUTE)
                    for (int id : o.input_layers) if (id == i - 1) referenced = true;
            }
            if (referenced) continue;
            cv.fused_shortcut = i;
            sc.fused_into_conv = true;
        }
    }
    // ---- [yolo] folded into the linear 1x1 FP32 head convolution in front of it: the activation runs in the conv's
    //      epilogue and the conv's own tensor (one full write + one full read of the head) is never materialised
    if (net.fuse && !net.debug) {
        const int nl = (int)net.layers.size();
        for (int i = 1; i < nl; ++i) {
            Layer &yo = net.layers[i];
            Layer &cv = net.layers[i - 1];
            if (yo.type != YL_YOLO || cv.type != YL_CONVOLUTIONAL || cv.conv_mode != CONV_F32 || cv.xnor) continue;
            if (cv.size != 1 || cv.pad != 0 || cv.stride != 1 || cv.activation != YL_LINEAR || cv.fused_shortcut >= 0) continue;
            if (cv.n != yo.n * (yo.classes + 5)) continue;
            bool referenced = false;
            for (int j = i + 1; j < nl && !referenced; ++j) {
                const Layer &o = net.layers[j];
                if (o.type == YL_SHORTCUT && o.index == i - 1) referenced = true;
                if (o.type == YL_ROUTE)
                    for (int id : o.input_layers) if (id == i - 1) referenced = true;
            }
            if (referenced) continue;
            cv.fused_yolo = i;
            yo.fused_into_conv = true;
        }
    }
    // ---- optional quantise-on-store plan (INT8): the producer of an INT8 conv's input emits the
    //      int8 NC/16HW16 tensor from its own epilogue; its FP32 tensor is skipped if nobody reads it
    for (Layer &l : net.layers) {
        l.q_from_producer = false; l.q_out_layer = -1; l.skip_f32_out = false; l.q_from_route = false;
        l.bits_from_producer = false; l.bits_out_slot = -1; l.pool_bits_mode = 0;
    }
    // a convolution in front of a 2x2 / stride-2 [maxpool] its kernel could fold in keeps that kernel whether or not fusion is on
    // (fused == unfused bit for bit needs the same summation order on both sides): K1r does not take it
    for (size_t j = 1; j < net.layers.size(); ++j) {
        const Layer &pl = net.layers[j];
        Layer &cv = net.layers[j - 1];
        cv.pool_follows = false;
        if (pl.type != YL_MAXPOOL || pl.size != 2 || pl.stride != 2 || pl.pad < 0 || pl.pad > 1) continue;
        if ((pl.h | pl.w) & 1 || pl.out_h != pl.h / 2 || pl.out_w != pl.w / 2) continue;
        if (cv.type != YL_CONVOLUTIONAL || cv.conv_mode != CONV_F32 || cv.xnor || cv.binarize_input || !hot_activation(cv.activation)) continue;
        ConvF32Args a;
        a.in = nullptr; a.wt = cv.d_weights_t; a.bias = cv.d_biases; a.add = nullptr; a.out_add = nullptr; a.out = nullptr;
        a.B = net.batch; a.C = cv.c; a.H = cv.h; a.W = cv.w; a.M = cv.n; a.OH = cv.out_h; a.OW = cv.out_w;
        a.K = cv.size * cv.size * cv.c; a.Kpad = cv.Kpad; a.Mpad = cv.Mpad;
        a.size = cv.size; a.stride = cv.stride; a.pad = cv.pad; a.act = cv.activation; a.tapmajor = cv.tapmajor;
        a.wino32_u = cv.d_wino32_u;
        ConvF32Opts o = net.conv_opts;
        o.force_tile = 0;
        cv.pool_follows = conv_f32_pool_fusable(a, o);
    }
    if (net.fuse && !net.debug) {
        const int nl = (int)net.layers.size();
        auto referenced_elsewhere = [&](int t, int consumer) {
            // is tensor of layer t read by anything except `consumer` taking it as its running input?
            if (t == nl - 1) return true;
            for (int j = t + 1; j < nl; ++j) {
                const Layer &o = net.layers[j];
                if (j == t + 1 && j != consumer && !(o.type == YL_ROUTE)) return true;      // running input of t+1
                if (o.type == YL_SHORTCUT && o.index == t) return true;
                if (o.type == YL_ROUTE) for (int id : o.input_layers) if (id == t) return true;
                if ((o.type == YL_YOLO || o.type == YL_REGION) && j == t + 1) return true;
            }
            return false;
        };
        for (int j = 1; j < nl; ++j) {
            Layer &cons = net.layers[j];
            if (cons.type != YL_CONVOLUTIONAL || !(cons.conv_mode == CONV_INT8 || cons.conv_mode == CONV_BF16)) continue;
            const int uc = cons.conv_mode == CONV_BF16 ? 8 : 16;            // channels per 16-byte unit
            // the tensor layer j consumes is the output of layer j-1
            int prod = j - 1;                                  // layer whose kernel writes that tensor
            const Layer &in_l = net.layers[j - 1];
            if (in_l.type == YL_SHORTCUT && in_l.fused_into_conv) prod = j - 2;
            Layer &pl = net.layers[prod];
            if (pl.type != YL_CONVOLUTIONAL) continue;
            if (!hot_activation(pl.activation) && pl.conv_mode != CONV_INT8) continue;      // side outputs are taken in the epilogue
            // producer kernels with a quantise-on-store epilogue: K2, and K1's direct kernel (never a layer
            // Winograd could take, never the xnor FP32 fallback): yolov3's layer 0 stops writing 3 GB of FP32
            const bool f32_direct = cons.conv_mode == CONV_INT8 && pl.conv_mode == CONV_F32 && !pl.xnor &&
                                    !wino_applicable(pl.c, pl.n, pl.size, pl.stride, pl.pad);
            if (pl.conv_mode != cons.conv_mode && !f32_direct) continue;
            if (prod == j - 1 && pl.fused_shortcut >= 0) continue;
            if (pl.q_out_layer >= 0) continue;
            if ((pl.n % uc) != 0 || cons.Cpad != pl.n) continue;          // no padded channel groups
            pl.q_out_layer = j;
            cons.q_from_producer = true;
            if (prod == j - 1) pl.skip_f32_out = !referenced_elsewhere(prod, j);
        }
        // ---- an INT8 conv behind a multi-input [route] nobody else reads: quantise the route's sources straight
        //      into the int8 tensor (channel groups at their offsets), the FP32 concatenation is not built
        for (int j = 1; j < nl; ++j) {
            Layer &cons = net.layers[j];
            Layer &rt = net.layers[j - 1];
            if (cons.type != YL_CONVOLUTIONAL || !(cons.conv_mode == CONV_INT8 || cons.conv_mode == CONV_BF16) || cons.q_from_producer) continue;
            const int uc = cons.conv_mode == CONV_BF16 ? 8 : 16;
            if (rt.type != YL_ROUTE || rt.d_output_alias || rt.n < 2 || referenced_elsewhere(j - 1, j)) continue;
            // channel groups beyond c (Cpad rounds the group count up to a power of two) are never written on this
            // path: harmless for int8 (anything x zero weights = 0), not for bf16 (stale NaN / Inf x 0 = NaN)
            bool ok = (cons.c % uc) == 0 && (cons.conv_mode == CONV_INT8 || cons.Cpad == cons.c);
            for (int k = 0; k < rt.n && ok; ++k) {
                const Layer &src = net.layers[rt.input_layers[k]];
                ok = (src.out_c % uc) == 0 && src.out_w == cons.w && src.out_h == cons.h &&
                     !(src.type == YL_CONVOLUTIONAL && (src.fused_shortcut >= 0 || src.skip_f32_out));
            }
            if (!ok) continue;
            cons.q_from_route = true;
            rt.skip_f32_out = true;
            // a source that is a nearest-neighbour [upsample] (scale 1) read by nothing but this [route]: quantising is pointwise, so the
            // int8 units are formed from the upsample's INPUT (quantize_nc16_kernel, up = stride) and its FP32 output is never written
            // (yolov3: layers 85 and 97, 0.07 + 0.13 ms of copies and 4x the quantiser's reads per step at 608 x 608, batch 64)
            if (cons.conv_mode == CONV_INT8) {
                for (int k = 0; k < rt.n; ++k) {
                    const int u = rt.input_layers[k];
                    Layer &up = net.layers[u];
                    if (up.type != YL_UPSAMPLE || up.scale != 1.f || up.stride < 2 || u < 1 || u == nl - 1) continue;
                    if ((up.out_h % up.stride) || (up.out_w % up.stride) || up.out_h / up.stride != up.h || up.out_w / up.stride != up.w) continue;
                    bool other = false;
                    for (int m = u + 1; m < nl && !other; ++m) {
                        const Layer &o = net.layers[m];
                        if (m == u + 1 && o.type != YL_ROUTE) other = true;                          // running input of u + 1
                        if (o.type == YL_SHORTCUT && o.index == u) other = true;
                        if (o.type == YL_ROUTE && m != j - 1) for (int id : o.input_layers) if (id == u) other = true;
                    }
                    if (!other) up.skip_f32_out = true;
                }
            }
        }
        // ---- sign-domain plan (XNOR): an XNOR convolution reads only (x > 0); sign(maxpool(x)) is the OR of
        //      the window's signs.  conv(xnor) [-> maxpool] -> conv(xnor) chains hand sign words over, the FP32
        //      tensors in between are written only where something else reads them.
        for (int j = 1; j < nl; ++j) {
            Layer &cons = net.layers[j];
            if (cons.type != YL_CONVOLUTIONAL || cons.conv_mode != CONV_XNOR) continue;
            Layer &prev = net.layers[j - 1];
            // (a producer with a rare activation finishes its FP32 tensor in a pass of its own: no epilogue sign words)
            if (prev.type == YL_CONVOLUTIONAL && prev.conv_mode == CONV_XNOR && prev.fused_shortcut < 0 && hot_activation(prev.activation)) {
                prev.bits_out_slot = j;                              // straight into this layer's input slot
                cons.bits_from_producer = true;
                prev.skip_f32_out = !referenced_elsewhere(j - 1, j);
            } else if (prev.type == YL_MAXPOOL && j >= 2) {
                Layer &pp = net.layers[j - 2];
                const bool pool_private = !referenced_elsewhere(j - 1, j);
                if (pp.type == YL_CONVOLUTIONAL && pp.conv_mode == CONV_XNOR && pp.fused_shortcut < 0 && pool_private && hot_activation(pp.activation) &&
                    !referenced_elsewhere(j - 2, j - 1)) {
                    pp.bits_out_slot = j - 1;                        // pre-pool sign words in the pool layer's slot
                    pp.skip_f32_out = true;
                    prev.pool_bits_mode = 1;
                    prev.skip_f32_out = true;
                    cons.bits_from_producer = true;
                } else if (pp.type == YL_CONVOLUTIONAL && pp.conv_mode == CONV_F32 && !pp.xnor && pp.fused_shortcut < 0 && pp.fused_yolo < 0 &&
                           pp.q_out_layer < 0 && pool_private && hot_activation(pp.activation) && !referenced_elsewhere(j - 2, j - 1) &&
                           first_layer_sign_plan(net, pp)) {
                    // FP32 first layer -> maxpool -> XNOR conv (tiny-yolo-obj_xnor.cfg layers 0-2: 1.4 GB of FP32 written and
                    // read back per batch of 128 just to take signs): conv_f32_smallk.hip emits the sign words itself
                    pp.bits_out_slot = j - 1;
                    pp.skip_f32_out = true;
                    prev.pool_bits_mode = 1;
                    prev.skip_f32_out = true;
                    cons.bits_from_producer = true;
                } else if (pool_private) {
                    prev.pool_bits_mode = 2;                         // FP32 in, pooled sign words out
                    prev.skip_f32_out = true;
                    cons.bits_from_producer = true;
                }
            }
        }
        // ---- [upsample] -> [route](upsampled, other) -> conv 1x1 (FP32): K1x reads the two tensors directly -- source 1 through the
        //      nearest-neighbour index map, source 2 as it is -- so neither the upsampled tensor nor the concatenation is written
        //      (yolov3 layers 85-87 and 97-99: 0.51 ms of copies per step at 608 x 608, batch 64).  The same values in the same channel
        //      order: the same bits.  Whether the kernel of the moment can do it is asked per forward pass (conv_f32_two_source_now).
        // (This plan runs BEFORE the [maxpool]-fusion plan below; nothing here depends on it: a 1x1 convolution is never
        //  pool-fusable (K1f / K1w are 3x3 kernels), a [maxpool] written by the convolution in front of it still HAS its tensor, and
        //  the pool plan drops a convolution's full-resolution tensor only where no other layer reads it -- a tensor this [route]
        //  reads is referenced.  tests/test_gpu_row3.py::test_two_source_plan_with_pool_fusion_around_it pins that.)
        auto tensor_written = [&](const Layer &x) {
            if (x.type == YL_MAXPOOL || x.type == YL_ROUTE || x.type == YL_UPSAMPLE) return !x.skip_f32_out;
            return !(x.type == YL_CONVOLUTIONAL && (x.fused_shortcut >= 0 || x.fused_yolo >= 0 || x.skip_f32_out));
        };
        for (int j = 2; j < nl; ++j) {
            Layer &cv = net.layers[j];
            Layer &rt = net.layers[j - 1];
            if (cv.type != YL_CONVOLUTIONAL || cv.conv_mode != CONV_F32 || cv.xnor || cv.binarize_input || !cv.d_weights_x3) continue;
            if (cv.size != 1 || cv.stride != 1 || cv.pad != 0 || cv.fused_yolo >= 0 || cv.q_out_layer >= 0 || cv.bits_out_slot >= 0) continue;
            if (rt.type != YL_ROUTE || rt.n != 2 || rt.d_output_alias || rt.skip_f32_out || referenced_elsewhere(j - 1, j)) continue;
            const int u = rt.input_layers[0], o = rt.input_layers[1];
            Layer &up = net.layers[u];
            const Layer &ot = net.layers[o];
            if (up.type != YL_UPSAMPLE || up.scale != 1.f || up.stride < 2 || u < 1 || up.skip_f32_out) continue;
            if (up.out_h != up.h * up.stride || up.out_w != up.w * up.stride || up.out_h != cv.h || up.out_w != cv.w) continue;
            if (ot.out_h != cv.h || ot.out_w != cv.w || up.out_c + ot.out_c != cv.c || !tensor_written(ot) || !tensor_written(net.layers[u - 1])) continue;
            bool other = false;                         // the upsampled tensor: read by this [route] only
            for (int m = u + 1; m < nl && !other; ++m) {
                const Layer &x = net.layers[m];
                if (m == u + 1 && x.type != YL_ROUTE) other = true;
                if (x.type == YL_SHORTCUT && x.index == u) other = true;
                if (x.type == YL_ROUTE && m != j - 1) for (int id : x.input_layers) if (id == u) other = true;
            }
            if (other) continue;
            cv.two_src_up = u; cv.two_src_other = o;
            up.two_src_conv = j; rt.two_src_conv = j;
        }
        // ---- 2x2 / stride-2 [maxpool] folded into the FP32 convolution in front of it (round 4): the kernels whose
        //      lanes finish whole pooling windows (K1f: a 2 x 4 patch per lane; K1w: an F(2x2) output tile) write the
        //      pooled tensor themselves; the full-resolution tensor is written only where something else reads it
        //      (yolov3-tiny: layer 8 also feeds a [route]).  forward_maxpool_layer_cpu, src/additionally.c:1448-1482:
        //      even H and W => window origin 0, no out-of-range taps, max is exact => the same bits as the two kernels.
        for (int j = 1; j < nl; ++j) {
            Layer &pl = net.layers[j];
            Layer &cv = net.layers[j - 1];
            if (pl.type != YL_MAXPOOL || pl.size != 2 || pl.stride != 2 || pl.pad < 0 || pl.pad > 1 || pl.pool_bits_mode != 0 || pl.skip_f32_out) continue;
            if ((pl.h | pl.w) & 1 || pl.out_h != pl.h / 2 || pl.out_w != pl.w / 2) continue;
            if (cv.type != YL_CONVOLUTIONAL || cv.conv_mode != CONV_F32 || cv.xnor || cv.binarize_input || !hot_activation(cv.activation)) continue;
            if (cv.fused_shortcut >= 0 || cv.fused_yolo >= 0 || cv.q_out_layer >= 0 || cv.bits_out_slot >= 0 || cv.skip_f32_out) continue;
            ConvF32Args a;
            a.in = nullptr; a.wt = cv.d_weights_t; a.bias = cv.d_biases; a.add = nullptr; a.out_add = nullptr; a.out = nullptr;
            a.B = net.batch; a.C = cv.c; a.H = cv.h; a.W = cv.w; a.M = cv.n; a.OH = cv.out_h; a.OW = cv.out_w;
            a.K = cv.size * cv.size * cv.c; a.Kpad = cv.Kpad; a.Mpad = cv.Mpad;
            a.size = cv.size; a.stride = cv.stride; a.pad = cv.pad; a.act = cv.activation; a.tapmajor = cv.tapmajor;
            a.wino32_u = cv.d_wino32_u;
            if (!conv_f32_pool_fusable(a, net.conv_opts)) continue;
            cv.fused_pool = j;
            pl.fused_into_conv = true;
            cv.skip_f32_out = !referenced_elsewhere(j - 1, j);
        }
    }
    // the pack kernels are done with the scratch once the stream is idle
    YL_HIP(hipStreamSynchronize((hipStream_t)net.stream));
    if (net.d_pack_src) (void)hipFree(net.d_pack_src);
    net.d_pack_src = nullptr; net.pack_src_bytes = 0;
    hipEvent_t e0, e1;
    YL_HIP(hipEventCreate(&e0));
    YL_HIP(hipEventCreate(&e1));
    net.ev0 = e0; net.ev1 = e1;
    net.on_device = true;
    return YL_OK;
}

// does the FP32 1x1 convolution `j` read [route]([upsample](x), y) from x and y directly in this pass?  (plan + the kernel-selection
// knobs of the moment; asked by the [upsample], the [route] and the convolution itself: one answer per pass)
static bool two_source_now(const Network &net, int j)
{
    const Layer &l = net.layers[j];
    if (l.two_src_up < 0) return false;
    const Layer &up = net.layers[l.two_src_up];
    ConvF32Args a;
    a.in = nullptr; a.wt = nullptr; a.bias = nullptr; a.add = nullptr; a.out_add = nullptr; a.out = nullptr;
    a.x3_w = l.d_weights_x3;
    a.in2 = net.layers[l.two_src_other].d_output; a.in2_C1 = up.out_c; a.in2_up = up.stride;
    a.B = net.batch; a.C = l.c; a.H = l.h; a.W = l.w; a.M = l.n; a.OH = l.out_h; a.OW = l.out_w;
    a.size = l.size; a.stride = l.stride; a.pad = l.pad; a.act = l.activation;
    return hot_activation(l.activation) && conv_f32_two_source_now(a, net.conv_opts);
}

// ------------------------------------------------------------------ forward
static int forward_layer(Network &net, size_t i, const float *input)
{
    Layer &l = net.layers[i];
    void *s = net.stream;
    const int B = net.batch;
    switch (l.type) {
    case YL_CONVOLUTIONAL: {
        // the -quantized convolution undoes nothing but LEAKY (src/yolov2_forward_network_quantized.c:623-627): there a
        // rare activation is simply not applied, as in the reference
        const bool post_act = !hot_activation(l.activation) && l.conv_mode != CONV_INT8;
        const int kernel_act = post_act ? YL_LINEAR : l.activation;
        if (l.conv_mode == CONV_F32) {
            ConvF32Args a;
            const float *conv_in = input;
            if (l.binarize_input) {
                YL_LAUNCH(launch_binarize(input, net.d_binbuf, (size_t)B * l.c * l.h * l.w, s), "binarize");
                conv_in = net.d_binbuf;
            }
            a.in = conv_in; a.wt = l.d_weights_t; a.bias = l.d_biases; a.add = nullptr; a.out_add = nullptr;
            a.out = l.skip_f32_out ? nullptr : l.d_output;
            if (l.two_src_up >= 0 && two_source_now(net, (int)i)) {      // [upsample] and [route] in front of it wrote nothing in this pass
                const Layer &up = net.layers[l.two_src_up];
                a.in = net.layers[l.two_src_up - 1].d_output;
                a.in2 = net.layers[l.two_src_other].d_output; a.in2_C1 = up.out_c; a.in2_up = up.stride;
            }
            if (l.q_out_layer >= 0) {
                const Layer &nx = net.layers[l.q_out_layer];
                a.q_out = net.d_qbuf + (l.q_out_layer % 3) * net.qbuf_bytes;
                a.q_mult = nx.input_quant_multipler;
                a.q_G = nx.Cpad / 16;
            }
            if (l.fused_shortcut >= 0) {
                // conv + [shortcut] in one pass (the reference GPU path fuses the same pair for XNOR
                // convs, src/additionally.c:326-339): shortcut.out = act(conv) + layers[index].out;
                // the conv's own tensor is not referenced by any other layer and is not written.
                Layer &sc = net.layers[l.fused_shortcut];
                a.add = net.layers[sc.index].d_output;
                a.out_add = sc.d_output;
                a.out = nullptr;
            }
            a.B = B; a.C = l.c; a.H = l.h; a.W = l.w; a.M = l.n; a.OH = l.out_h; a.OW = l.out_w;
            a.K = l.size * l.size * l.c; a.Kpad = l.Kpad; a.Mpad = l.Mpad;
            a.size = l.size; a.stride = l.stride; a.pad = l.pad; a.act = kernel_act;
            a.tapmajor = l.tapmajor;
            a.wino32_u = l.d_wino32_u;
            a.x3_w = l.d_weights_x3;
            a.row3_w = l.pool_follows ? nullptr : l.d_weights_r3;
            a.tile_ctr = l.d_tile_ctr;
            // the input tensor is library memory with the front pad (a caller's device pointer as the network input is not)
            a.in_front_pad = conv_in != net.d_binbuf && !(i == 0 && conv_in != net.d_input);
            if (l.fused_yolo >= 0) {
                const Layer &yo = net.layers[l.fused_yolo];
                a.yolo_entries = yo.classes + 5;
                a.out = yo.d_output;
            }
            // 2x2 / stride-2 [maxpool] written by this epilogue.  The plan was made at yl_network_to_device from the kernel-selection
            // knobs of that moment; if yl_network_set_conv_tile / yl_network_set_variant has since moved the layer to a kernel
            // without a pooled output (e.g. a forced direct tile), degrade instead of failing: the full tensor (always allocated)
            // is written and the stand-alone pooling kernel follows -- the same bits either way.
            bool pool_after = false;
            if (l.fused_pool >= 0) {
                a.pool_out = net.layers[l.fused_pool].d_output;
                if (!conv_f32_pool_fusable(a, net.conv_opts)) {
                    a.pool_out = nullptr;
                    a.out = l.d_output;
                    pool_after = true;
                }
            }
            if (l.bits_out_slot >= 0) {     // FP32 first layer -> [maxpool] -> XNOR conv: sign words instead of the FP32 tensor
                a.bits_out = net.d_bitbuf + (size_t)(l.bits_out_slot % 3) * (net.bitbuf_bytes / sizeof(uint64_t));
                // ... and where the kernel of the moment can OR the 2x2 / stride-2 windows itself (K1m), the POOLED words, straight into
                // the consumer's slot: the pooling layer then has nothing to do in this pass (decided per launch, like pool_after above)
                if (l.bits_out_slot == (int)i + 1 && i + 1 < net.layers.size()) {
                    Layer &pl = net.layers[i + 1];
                    pl.bits_pooled_by_producer = false;
                    if (pl.type == YL_MAXPOOL && pl.pool_bits_mode == 1 && pl.size == 2 && pl.stride == 2 && pl.pad >= 0 && pl.pad <= 1 &&
                        !((pl.h | pl.w) & 1) && pl.out_h == pl.h / 2 && pl.out_w == pl.w / 2 && net.conv_opts.force_tile == 0 &&
                        (net.conv_opts.variant & 8) && (net.conv_opts.variant & 16384)) {
                        a.bits_pooled = true;
                        if (first_layer_mfma_applicable(a)) {
                            a.bits_out = net.d_bitbuf + (size_t)((i + 2) % 3) * (net.bitbuf_bytes / sizeof(uint64_t));
                            pl.bits_pooled_by_producer = true;
                        } else a.bits_pooled = false;
                    }
                }
            }
            if (net.split_k && net.d_ks_ws && !a.in2 && !a.q_out && !a.bits_out && !a.pool_out && a.yolo_entries == 0 && hot_activation(kernel_act)) {
                const int parts = split_k_parts(B, l.c, l.n, l.size, l.out_h, l.out_w, device_cu_count());
                if (parts > 1 && (size_t)parts * B * l.outputs <= net.ks_ws_floats) { a.ksplit = parts; a.ks_ws = net.d_ks_ws; a.ks_zeros = net.d_ks_zeros; }
            }
            YL_LAUNCH(launch_conv_f32(a, net.conv_opts, s, l.kernel_name, sizeof(l.kernel_name)), "conv_f32");
            if (pool_after) {
                const Layer &pl = net.layers[l.fused_pool];
                YL_LAUNCH(launch_maxpool(l.d_output, pl.d_output, B, pl.c, pl.h, pl.w, pl.out_h, pl.out_w, pl.size, pl.stride, pl.pad, s), "maxpool");
            }
        } else if (l.conv_mode == CONV_INT8) {
            int8_t *q_in = net.d_qbuf + (i % 3) * net.qbuf_bytes;
            if (l.q_from_route) {
                const Layer &rt = net.layers[i - 1];
                int g_off = 0;
                for (int k = 0; k < rt.n; ++k) {
                    const Layer &src = net.layers[rt.input_layers[k]];
                    if (src.type == YL_UPSAMPLE && src.skip_f32_out)        // straight from the upsample's input (see the plan)
                        YL_LAUNCH(launch_quantize_nhwc(net.layers[rt.input_layers[k] - 1].d_output, q_in, B, src.out_c, l.h, l.w, src.out_c,
                                                       l.input_quant_multipler, s, g_off, l.Cpad / 16, src.stride), "quantize_route_up");
                    else
                    YL_LAUNCH(launch_quantize_nhwc(src.d_output, q_in, B, src.out_c, l.h, l.w, src.out_c, l.input_quant_multipler,
                                                   s, g_off, l.Cpad / 16), "quantize_route");
                    g_off += src.out_c / 16;
                }
            } else if (!l.q_from_producer)
                YL_LAUNCH(launch_quantize_nhwc(input, q_in, B, l.c, l.h, l.w, l.Cpad, l.input_quant_multipler, s),
                          "quantize_nhwc");
            ConvI8Args a;
            a.in_q = q_in; a.w_q = l.d_weights_i8; a.bias = l.d_biases; a.dbg = l.d_debug;
            a.out = l.skip_f32_out ? nullptr : l.d_output;
            a.add = nullptr; a.out_add = nullptr;
            a.q_out = nullptr; a.q_mult = 0.f; a.q_G = 0;
            if (l.fused_shortcut >= 0) {
                Layer &sc = net.layers[l.fused_shortcut];
                a.add = net.layers[sc.index].d_output;
                a.out_add = sc.d_output;       // always materialised: later shortcuts / routes read it
                a.out = nullptr;
            }
            if (l.q_out_layer >= 0) {
                const Layer &nx = net.layers[l.q_out_layer];
                a.q_out = net.d_qbuf + (l.q_out_layer % 3) * net.qbuf_bytes;
                a.q_mult = nx.input_quant_multipler;
                a.q_G = nx.Cpad / 16;
            }
            a.B = B; a.Cpad = l.Cpad; a.H = l.h; a.W = l.w; a.M = l.n; a.Mpad = l.Mpad; a.OH = l.out_h; a.OW = l.out_w;
            a.size = l.size; a.stride = l.stride; a.pad = l.pad; a.act = kernel_act;
            // float ALPHA1 = R_MULT / (l.input_quant_multipler * l.weights_quant_multipler);  (quantized.c:596)
            a.alpha1 = 32 / (l.input_quant_multipler * l.weights_quant_multipler);
            // corners of the exact epilogue that cannot occur in this layer (see ConvI8Args::no_corner): a non-zero
            // output is a sum of two floats of magnitude >= 1e-20 (>= 6e-28, never inside (0, 1e-30)); a side-output
            // operand is bounded by (32767 * alpha1 + max|bias|) * q_mult
            a.no_corner = 0;
            if (l.bias_abs_max >= 0.f && a.alpha1 >= 1e-20f && a.alpha1 <= 1e20f && l.bias_abs_min_nz >= 1e-20f) {
                a.no_corner |= 1;
                if (a.q_out && a.q_mult > 0.f && a.q_mult <= 1e20f &&
                    (32767.0 * (double)a.alpha1 + (double)l.bias_abs_max) * (double)a.q_mult * 1.00001 < 32768.0)
                    a.no_corner |= 2;
            }
            YL_LAUNCH(launch_conv_i8(a, net.i8_tile, s, l.kernel_name, sizeof(l.kernel_name)), "conv_i8");
        } else if (l.conv_mode == CONV_BF16) {
            int8_t *h_in = net.d_qbuf + (i % 3) * net.qbuf_bytes;
            if (l.q_from_route) {
                const Layer &rt = net.layers[i - 1];
                int g_off = 0;
                for (int k = 0; k < rt.n; ++k) {
                    const Layer &src = net.layers[rt.input_layers[k]];
                    YL_LAUNCH(launch_pack_bf16(src.d_output, h_in, B, src.out_c, l.h, l.w, src.out_c, s, g_off, l.Cpad / 8), "pack_bf16_route");
                    g_off += src.out_c / 8;
                }
            } else if (!l.q_from_producer)
                YL_LAUNCH(launch_pack_bf16(input, h_in, B, l.c, l.h, l.w, l.Cpad, s), "pack_bf16");
            ConvBf16Args a;
            a.in_h = h_in; a.w_h = l.d_weights_i8; a.bias = l.d_biases;
            a.out = l.skip_f32_out ? nullptr : l.d_output;
            a.add = nullptr; a.out_add = nullptr; a.h_out = nullptr; a.h_G = 0;
            if (l.fused_shortcut >= 0) {
                Layer &sc = net.layers[l.fused_shortcut];
                a.add = net.layers[sc.index].d_output;
                a.out_add = sc.d_output;
                a.out = nullptr;
            }
            if (l.q_out_layer >= 0) {
                a.h_out = net.d_qbuf + (l.q_out_layer % 3) * net.qbuf_bytes;
                a.h_G = net.layers[l.q_out_layer].Cpad / 8;
            }
            a.B = B; a.Cpad = l.Cpad; a.H = l.h; a.W = l.w; a.M = l.n; a.Mpad = l.Mpad; a.OH = l.out_h; a.OW = l.out_w;
            a.size = l.size; a.stride = l.stride; a.pad = l.pad; a.act = kernel_act;
            YL_LAUNCH(launch_conv_bf16(a, net.i8_tile, s, l.kernel_name, sizeof(l.kernel_name)), "conv_bf16");
        } else {
            auto ring = [&](int slot) { return net.d_bitbuf + (size_t)(slot % 3) * (net.bitbuf_bytes / sizeof(uint64_t)); };
            uint64_t *in_bits = ring((int)i);
            if (!l.bits_from_producer)
                YL_LAUNCH(launch_pack_sign_bits(input, in_bits, B, l.c, l.h, l.w, l.Cw, s), "pack_sign_bits");
            ConvXnorArgs a;
            a.in_bits = in_bits; a.w_bits = l.d_weights_bits; a.mean = l.d_mean; a.bias = l.d_biases;
            a.out = l.skip_f32_out ? nullptr : l.d_output; a.dbg = l.d_debug;
            a.thr = (l.thr_ok && (net.conv_opts.variant & 256) == 0) ? l.d_thr : nullptr;      // variant bit 8: A/B switch, float epilogue everywhere
            a.out_bits = l.bits_out_slot >= 0 ? ring(l.bits_out_slot) : nullptr;
            if (l.fused_shortcut >= 0) {        // conv_xnor + [shortcut] in one pass (src/additionally.c:326-339)
                Layer &sc = net.layers[l.fused_shortcut];
                a.add = net.layers[sc.index].d_output;
                a.out_add = sc.d_output;
                a.out = nullptr;
            }
            a.B = B; a.C = l.c; a.Cw = l.Cw; a.H = l.h; a.W = l.w; a.M = l.n; a.Mpad = l.Mpad; a.act = kernel_act;
            a.ft_mode = (net.conv_opts.variant & 512) ? 64 : 0;
            YL_LAUNCH(launch_conv_xnor(a, s, l.kernel_name, sizeof(l.kernel_name)), "conv_xnor");
        }
        if (post_act) YL_LAUNCH(launch_activate(l.d_output, (size_t)B * l.outputs, l.activation, s), "activate");
        break;
    }
    case YL_MAXPOOL: {
        if (l.fused_into_conv) break;          // written by the epilogue of the convolution in front of it
        auto ring = [&](int slot) { return net.d_bitbuf + (size_t)(slot % 3) * (net.bitbuf_bytes / sizeof(uint64_t)); };
        if (l.pool_bits_mode == 1 && l.bits_pooled_by_producer)
            ;                                  // the convolution in front of this layer wrote slot i + 1 itself (K1m)
        else if (l.pool_bits_mode == 1)
            YL_LAUNCH(launch_bit_maxpool(ring((int)i), ring((int)i + 1), B, (l.c + 63) / 64, l.h, l.w, l.out_h, l.out_w,
                                         l.size, l.stride, l.pad, s), "bit_maxpool");
        else if (l.pool_bits_mode == 2)
            YL_LAUNCH(launch_maxpool_sign_pack(input, ring((int)i + 1), B, l.c, (l.c + 63) / 64, l.h, l.w, l.out_h, l.out_w,
                                               l.size, l.stride, l.pad, s), "maxpool_sign_pack");
        if (!l.skip_f32_out)
            YL_LAUNCH(launch_maxpool(input, l.d_output, B, l.c, l.h, l.w, l.out_h, l.out_w, l.size, l.stride, l.pad, s), "maxpool");
        break;
    }
    case YL_ROUTE: {
        if (l.d_output_alias || l.skip_f32_out) break;          // alias, or quantised straight from its sources
        l.two_src_skipped = l.two_src_conv >= 0 && two_source_now(net, l.two_src_conv);
        if (l.two_src_skipped) break;                           // the convolution behind it reads the two sources itself
        size_t offset = 0;
        for (int k = 0; k < l.n; ++k) {
            const Layer &src = net.layers[l.input_layers[k]];
            YL_LAUNCH(launch_copy_rows(src.d_output, l.d_output + offset, B, l.input_sizes[k],
                                       (size_t)l.input_sizes[k], (size_t)l.outputs, s), "route");
            offset += l.input_sizes[k];
        }
        break;
    }
    case YL_SHORTCUT:
        if (l.fused_into_conv) break;          // written by the preceding conv's epilogue
        YL_LAUNCH(launch_shortcut(input, net.layers[l.index].d_output, l.d_output, B, l.w, l.h, l.c,
                                  l.out_w, l.out_h, l.out_c, l.activation, s), "shortcut");
        break;
    case YL_UPSAMPLE:
        if (l.skip_f32_out) break;             // its only reader quantises straight from this layer's input
        l.two_src_skipped = l.two_src_conv >= 0 && two_source_now(net, l.two_src_conv);
        if (l.two_src_skipped) break;          // its only reader's reader indexes this layer's input itself
        YL_LAUNCH(launch_upsample(input, l.d_output, B, l.c, l.h, l.w, l.stride, l.scale, s), "upsample");
        break;
    case YL_YOLO:
        if (l.fused_into_conv) break;          // written by the head convolution's epilogue
        YL_LAUNCH(launch_yolo(input, l.d_output, B, l.n, l.classes, l.w * l.h, s), "yolo");
        break;
    case YL_REGION:
        YL_LAUNCH(launch_region(input, l.d_output, B, l.n, l.classes, l.coords, l.w * l.h, l.softmax, s,
                                l.d_tree ? l.d_tree + l.classes : nullptr, (int)l.tree_group_size.size()), "region");
        break;
    case YL_REORG:
        YL_LAUNCH(launch_reorg(input, l.d_output, B, l.out_c, l.out_h, l.out_w, l.stride, s), "reorg");
        break;
    default:
        set_error("layer type has no device kernel");
        return YL_ERR_UNSUPPORTED;
    }
    return YL_OK;
}

// slot < 0: untimed; otherwise HIP events of timing slot `slot` are recorded around every layer
static int forward(Network &net, const float *input_dev, int slot)
{
    if (!net.on_device) { set_error("network not on device: call yl_network_to_device first"); return YL_ERR_STATE; }
    YL_HIP(hipSetDevice(net.device));
    const float *input = input_dev;
    const size_t nl = net.layers.size();
    ++net.forward_seq;
    void **ev = slot >= 0 ? &net.layer_events[(size_t)slot * (nl + 1)] : nullptr;
    for (size_t i = 0; i < nl; ++i) {
        if (ev) YL_HIP(hipEventRecord((hipEvent_t)ev[i], (hipStream_t)net.stream));
        int rc = forward_layer(net, i, input);
        if (rc != YL_OK) return rc;
        input = net.layers[i].d_output;
        if (net.head_event_base >= 0 && net.head_events[net.head_event_base + i])
            YL_HIP(hipEventRecord((hipEvent_t)net.head_events[net.head_event_base + i], (hipStream_t)net.stream));
    }
    if (ev) YL_HIP(hipEventRecord((hipEvent_t)ev[nl], (hipStream_t)net.stream));
    return YL_OK;
}

// false for tensors the fusion plan never writes (a conv folded into its [shortcut], an INT8 conv
// whose only reader takes the int8 side output)
static bool layer_materialised(const Layer &l)
{
    if (l.two_src_skipped) return false;
    if (l.type == YL_MAXPOOL || l.type == YL_ROUTE || l.type == YL_UPSAMPLE) return !l.skip_f32_out;
    return !(l.type == YL_CONVOLUTIONAL && (l.fused_shortcut >= 0 || l.fused_yolo >= 0 || l.skip_f32_out));
}

// D2H of the heads / last layer (what network_predict_* leaves in l.output; the reference pulls each [yolo] tensor with a blocking
// cudaMemcpy as its layer ends, src/yolov2_forward_network_gpu.cu:438).  Here: on out_stream, each head behind an event recorded on
// the compute stream (head_event_base >= 0: the one forward() recorded behind that head -- head 82 of yolov3 is complete 40 % into
// the pass and travels while the rest computes; < 0: one event at the stream's tail), in 16 MB chunks into library-pinned memory;
// where the destination is caller memory, finish_head_pull copies chunk k out on the host pool while chunk k+1 is on the bus.
// Images [b0, b0 + nb) of the full batch; l.d_output is the UNSHIFTED tensor (call outside a BatchWindow).
struct HeadPiece { size_t ev; char *dst; const char *pinned; size_t bytes; };

static int enqueue_head_pull(Network &net, bool also_last, int b0, int nb, int head_event_base, std::vector<HeadPiece> &pieces, size_t &n_ev)
{
    hipStream_t s = (hipStream_t)net.stream, cs = (hipStream_t)net.out_stream;
    const size_t CHUNK = (size_t)16 << 20;
    const size_t nl = net.layers.size();
    hipEvent_t tail = nullptr;
    if (head_event_base < 0) {            // one event at the tail of the compute stream (the last layer's slot of sub-batch 0 serves)
        tail = (hipEvent_t)net.head_events[nl - 1];
        YL_HIP(hipEventRecord(tail, s));
    }
    for (size_t i = 0; i < nl; ++i) {
        Layer &l = net.layers[i];
        const bool is_head = (l.type == YL_YOLO || l.type == YL_REGION);
        if (!(is_head || (also_last && i + 1 == nl))) continue;
        if (!l.host_output) continue;
        // the DMA always lands in library-pinned memory: the destination itself, or this layer's region of h_heads
        const size_t img = sizeof(float) * (size_t)l.outputs;
        char *pinned = (char *)(l.host_kind == HOST_PINNED ? l.host_output : net.h_heads + l.h_head_off) + (size_t)b0 * img;
        const bool bounce = l.host_kind != HOST_PINNED;
        const size_t bytes = img * (size_t)nb;
        const char *src = (const char *)l.d_output + (size_t)b0 * img;
        YL_HIP(hipStreamWaitEvent(cs, head_event_base >= 0 ? (hipEvent_t)net.head_events[head_event_base + i] : tail, 0));
        for (size_t off = 0; off < bytes; off += CHUNK) {
            const size_t len = bytes - off < CHUNK ? bytes - off : CHUNK;
            YL_HIP(hipMemcpyAsync(pinned + off, src + off, len, hipMemcpyDeviceToHost, cs));
            if (n_ev == net.chunk_events.size()) {
                hipEvent_t e;
                YL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                net.chunk_events.push_back(e);
            }
            YL_HIP(hipEventRecord((hipEvent_t)net.chunk_events[n_ev], cs));
            if (bounce) pieces.push_back(HeadPiece{n_ev, (char *)l.host_output + (size_t)b0 * img + off, pinned + off, len});
            ++n_ev;
        }
    }
    return YL_OK;
}

static int finish_head_pull(Network &net, const std::vector<HeadPiece> &pieces)
{
    HostCopyJob job;
    int rc = YL_OK;
    for (const HeadPiece &pc : pieces) {
        if (hipEventSynchronize((hipEvent_t)net.chunk_events[pc.ev]) != hipSuccess) { set_error("D2H of a head chunk failed"); rc = YL_ERR_DEVICE; break; }
        host_copy_async(job, pc.dst, pc.pinned, pc.bytes);
    }
    host_copy_wait(job);
    if (rc != YL_OK) return rc;
    YL_HIP(hipStreamSynchronize((hipStream_t)net.out_stream));
    YL_HIP(hipStreamSynchronize((hipStream_t)net.stream));
    return YL_OK;
}

static int pull_heads(Network &net, bool also_last)
{
    std::vector<HeadPiece> pieces;
    size_t n_ev = 0;
    int rc = enqueue_head_pull(net, also_last, 0, net.batch, -1, pieces, n_ev);
    if (rc != YL_OK) return rc;
    return finish_head_pull(net, pieces);
}

// Images [b0, b0 + nb) of the batch as a batch of their own: every kernel takes its image count and tensor bases per launch
// (forward_layer), images are independent and a batch-B pass equals B batch-1 passes bit for bit
// (tests/test_gpu_headline.py), so a pass over a window of the tensors IS that part of the full pass.  The scratch rings
// (d_qbuf, d_bitbuf, d_binbuf) are consumed inside a pass and are reused from their start by every window.
struct BatchWindow {
    Network &net;
    int full, b0;
    BatchWindow(Network &n, int b0_, int nb) : net(n), full(n.batch), b0(b0_)
    {
        shift(+1);
        net.batch = nb;
    }
    ~BatchWindow()
    {
        net.batch = full;
        shift(-1);
    }
    void shift(int sign)
    {
        for (Layer &l : net.layers)
            if (l.d_output) l.d_output += (ptrdiff_t)sign * (ptrdiff_t)b0 * (ptrdiff_t)l.outputs;
        // the library's own input tensor too: forward_layer tells it (front pad, finite) from a caller's device pointer by address
        if (net.d_input) net.d_input += (ptrdiff_t)sign * (ptrdiff_t)b0 * (ptrdiff_t)net.c * net.h * net.w;
    }
};

}  // namespace yl

// ====================================================================== C-ABI
using namespace yl;

extern "C" {

const char *yl_last_error(void) { return g_err.c_str(); }

int yl_abi_version(void) { return YL_ABI_VERSION; }

int yl_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int yl_device_synchronize(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device visible (libyolo2hip has no CPU fallback)"); return YL_ERR_DEVICE; }
    for (int d = (device < 0 ? 0 : device); d < (device < 0 ? n : device + 1); ++d) {
        if (d >= n) { set_error("device index out of range"); return YL_ERR_ARG; }
        YL_HIP(hipSetDevice(d));
        YL_HIP(hipDeviceSynchronize());
    }
    return YL_OK;
}

int yl_network_create_from_cfg(const char *cfg_path, int batch, int quantized, yl_network **out)
{
    if (!cfg_path || !out) { set_error("null argument"); return YL_ERR_ARG; }
    yl_network *n = new yl_network();
    int rc = parse_cfg_file(cfg_path, batch, quantized, n->net);
    if (rc != YL_OK) { delete n; return rc; }
    *out = n;
    return YL_OK;
}

int yl_network_create_from_desc(const yl_layer_desc *layers, int n_layers, int batch, int w, int h, int c,
                                int quantized, const float *input_calibration, int input_calibration_size,
                                yl_network **out)
{
    if (!layers || n_layers <= 0 || !out || batch <= 0) { set_error("bad argument"); return YL_ERR_ARG; }
    yl_network *n = new yl_network();
    Network &net = n->net;
    net.batch = batch; net.w = w; net.h = h; net.c = c; net.quantized = quantized;
    if (input_calibration && input_calibration_size > 0)
        net.input_calibration.assign(input_calibration, input_calibration + input_calibration_size);
    for (int i = 0; i < n_layers; ++i) {
        const yl_layer_desc &d = layers[i];
        Layer l;
        l.type = d.type; l.activation = d.activation;
        l.batch = batch; l.w = d.w; l.h = d.h; l.c = d.c; l.n = d.n;
        l.size = d.size; l.stride = d.stride; l.pad = d.pad;
        l.out_w = d.out_w; l.out_h = d.out_h; l.out_c = d.out_c;
        l.outputs = d.outputs; l.inputs = d.inputs;
        l.batch_normalize = d.batch_normalize; l.xnor = d.xnor; l.index = d.index;
        l.gpu_quantized = d.quantized;
        l.classes = d.classes; l.coords = d.coords ? d.coords : 4; l.total = d.total; l.softmax = d.softmax;
        l.scale = d.scale;
        l.host_output = d.output;
        l.host_kind = d.output ? HOST_CALLER : HOST_NONE;
        switch (d.type) {
        case YL_CONVOLUTIONAL: {
            if (!d.weights || !d.biases) { delete n; set_error("conv layer without weights/biases"); return YL_ERR_ARG; }
            const size_t nw = (size_t)d.n * d.c * d.size * d.size;
            l.weights.assign(d.weights, d.weights + nw);
            l.biases.assign(d.biases, d.biases + d.n);
            if (d.batch_normalize) {
                if (!d.scales || !d.rolling_mean || !d.rolling_variance) { delete n; set_error("BN layer without statistics"); return YL_ERR_ARG; }
                l.scales.assign(d.scales, d.scales + d.n);
                l.rolling_mean.assign(d.rolling_mean, d.rolling_mean + d.n);
                l.rolling_variance.assign(d.rolling_variance, d.rolling_variance + d.n);
            }
            if (d.weights_int8) {
                l.weights_int8.assign(d.weights_int8, d.weights_int8 + nw);
                l.input_quant_multipler = d.input_quant_multipler;
                l.weights_quant_multipler = d.weights_quant_multipler;
                l.quant_ready = true;
            }
            if (d.mean_arr) { l.mean_arr.assign(d.mean_arr, d.mean_arr + d.n); l.xnor_ready = true; }
            break;
        }
        case YL_ROUTE:
            if (!d.input_layers || !d.input_sizes || d.n <= 0) { delete n; set_error("route without inputs"); return YL_ERR_ARG; }
            l.input_layers.assign(d.input_layers, d.input_layers + d.n);
            l.input_sizes.assign(d.input_sizes, d.input_sizes + d.n);
            for (int k = 0; k < d.n; ++k)
                if (d.input_layers[k] < 0 || d.input_layers[k] >= i) { delete n; set_error("route index out of range"); return YL_ERR_ARG; }
            break;
        case YL_SHORTCUT:
            if (d.index < 0 || d.index >= i) { delete n; set_error("shortcut index out of range"); return YL_ERR_ARG; }
            break;
        case YL_YOLO:
            if (!d.mask || !d.anchors) { delete n; set_error("yolo layer without mask/anchors"); return YL_ERR_ARG; }
            l.mask.assign(d.mask, d.mask + d.n);
            l.anchors.assign(d.anchors, d.anchors + (size_t)2 * d.total);
            break;
        case YL_REGION:
            if (!d.anchors) { delete n; set_error("region layer without anchors"); return YL_ERR_ARG; }
            l.anchors.assign(d.anchors, d.anchors + (size_t)2 * d.n);
            l.total = d.n;
            if (d.tree_n > 0) {
                if (!d.tree_parent || !d.tree_group_size || d.tree_groups <= 0 || d.tree_n != d.classes) {
                    delete n; set_error("region softmax tree: bad description"); return YL_ERR_ARG;
                }
                l.tree_parent.assign(d.tree_parent, d.tree_parent + d.tree_n);
                l.tree_group_size.assign(d.tree_group_size, d.tree_group_size + d.tree_groups);
            }
            break;
        case YL_MAXPOOL: case YL_UPSAMPLE: case YL_REORG:
            break;
        default:
            delete n;
            set_error("layer type is not on the hot path");
            return YL_ERR_UNSUPPORTED;
        }
        net.layers.push_back(std::move(l));
    }
    net.weights_loaded = true;
    select_conv_modes(net);
    *out = n;
    return YL_OK;
}

int yl_network_load_weights(yl_network *net, const char *weights_path)
{
    if (!net || !weights_path) { set_error("null argument"); return YL_ERR_ARG; }
    return load_weights_file(net->net, weights_path);
}

int yl_network_load_weights_upto(yl_network *net, const char *weights_path, int cutoff)
{
    if (!net || !weights_path || cutoff < 0) { set_error("bad argument"); return YL_ERR_ARG; }
    return load_weights_file(net->net, weights_path, cutoff);
}

int yl_network_fuse_conv_batchnorm(yl_network *net)
{
    if (!net) { set_error("null argument"); return YL_ERR_ARG; }
    fuse_conv_batchnorm(net->net);
    return YL_OK;
}

int yl_network_calculate_binary_weights(yl_network *net)
{
    if (!net) { set_error("null argument"); return YL_ERR_ARG; }
    calculate_binary_weights(net->net);
    return YL_OK;
}

int yl_network_quantize(yl_network *net)
{
    if (!net) { set_error("null argument"); return YL_ERR_ARG; }
    quantize_network(net->net);
    return YL_OK;
}

int yl_network_prepare_on_device(yl_network *net, int device)
{
    if (!net) { set_error("null argument"); return YL_ERR_ARG; }
    if (!net->net.weights_loaded) { set_error("prepare_on_device: no weights loaded"); return YL_ERR_STATE; }
    return prepare_on_device(net->net, device);
}

void yl_network_destroy(yl_network *net)
{
    if (!net) return;
    if (net->net.on_device || net->net.stream) free_device(net->net);
    delete net;
}

int yl_network_num_layers(const yl_network *net) { return net ? (int)net->net.layers.size() : YL_ERR_ARG; }
int yl_network_batch(const yl_network *net) { return net ? net->net.batch : YL_ERR_ARG; }

int yl_network_input_dims(const yl_network *net, int *dims)
{
    if (!net || !dims) { set_error("null argument"); return YL_ERR_ARG; }
    dims[0] = net->net.w; dims[1] = net->net.h; dims[2] = net->net.c;
    return YL_OK;
}

#define YL_LAYER_OR(ret)                                                                \
    if (!net || i < 0 || i >= (int)net->net.layers.size()) { set_error("bad layer index"); return ret; } \
    const Layer &l = net->net.layers[i];

int yl_network_layer_info(const yl_network *net, int i, int *info)
{
    YL_LAYER_OR(YL_ERR_ARG)
    if (!info) { set_error("null argument"); return YL_ERR_ARG; }
    info[0] = l.type; info[1] = l.batch; info[2] = l.w; info[3] = l.h; info[4] = l.c; info[5] = l.n;
    info[6] = l.size; info[7] = l.stride; info[8] = l.pad; info[9] = l.out_w; info[10] = l.out_h;
    info[11] = l.out_c; info[12] = l.outputs; info[13] = l.inputs; info[14] = l.activation;
    info[15] = l.xnor; info[16] = (l.type == YL_CONVOLUTIONAL && l.conv_mode == CONV_INT8) ? 1 : ((l.type == YL_CONVOLUTIONAL && l.conv_mode == CONV_BF16) ? 2 : 0);
    info[17] = l.index; info[18] = l.classes; info[19] = l.coords; info[20] = l.total;
    info[21] = l.softmax; info[22] = (l.type == YL_CONVOLUTIONAL) ? l.conv_mode : 0; info[23] = l.batch_normalize;
    return YL_OK;
}

const float *yl_network_layer_weights(const yl_network *net, int i) { YL_LAYER_OR(nullptr) return l.weights.empty() ? nullptr : l.weights.data(); }
const float *yl_network_layer_biases(const yl_network *net, int i) { YL_LAYER_OR(nullptr) return l.biases.empty() ? nullptr : l.biases.data(); }
const int8_t *yl_network_layer_weights_int8(const yl_network *net, int i) { YL_LAYER_OR(nullptr) return l.weights_int8.empty() ? nullptr : l.weights_int8.data(); }
const float *yl_network_layer_mean_arr(const yl_network *net, int i) { YL_LAYER_OR(nullptr) return l.mean_arr.empty() ? nullptr : l.mean_arr.data(); }

int yl_network_layer_quant_multipliers(const yl_network *net, int i, float *mult)
{
    YL_LAYER_OR(YL_ERR_ARG)
    if (!mult) { set_error("null argument"); return YL_ERR_ARG; }
    mult[0] = l.input_quant_multipler; mult[1] = l.weights_quant_multipler;
    return YL_OK;
}

double yl_network_flops_per_image(const yl_network *net)
{
    if (!net) return 0.0;
    double f = 0.0;
    for (const Layer &l : net->net.layers)
        if (l.type == YL_CONVOLUTIONAL) f += 2.0 * l.n * l.size * l.size * l.c * (double)l.out_h * l.out_w;
    return f;
}

int yl_network_layer_traffic(const yl_network *net, int i, double *bytes)
{
    YL_LAYER_OR(YL_ERR_ARG)
    if (!bytes) { set_error("null argument"); return YL_ERR_ARG; }
    const Network &n = net->net;
    const double B = n.batch;
    const double in_el = B * l.inputs, out_el = B * l.outputs;
    double rd = 0, wr = 0;
    switch (l.type) {
    case YL_CONVOLUTIONAL: {
        const double wel = (double)l.n * l.c * l.size * l.size;
        const bool fused = l.fused_shortcut >= 0;
        if (l.conv_mode == CONV_INT8) {
            const double q_in = B * (double)l.h * l.w * (l.Cpad ? l.Cpad : l.c);
            if (!l.q_from_producer) { rd += 4 * in_el; wr += q_in; }      // stand-alone quantise pass
            rd += q_in + wel;
        } else if (l.conv_mode == CONV_BF16) {
            const double h_in = 2.0 * B * (double)l.h * l.w * (l.Cpad ? l.Cpad : l.c);
            if (!l.q_from_producer) { rd += 4 * in_el; wr += h_in; }      // stand-alone pack pass
            rd += h_in + 2 * wel;
        } else if (l.conv_mode == CONV_XNOR) {
            const double bits = B * (double)l.h * l.w * 8.0 * ((l.c + 63) / 64);
            if (!l.bits_from_producer) { rd += 4 * in_el; wr += bits; }   // stand-alone sign-pack pass
            rd += bits + wel / 8;
            if (l.bits_out_slot >= 0) wr += B * (double)l.out_h * l.out_w * 8.0 * ((l.n + 63) / 64);
        } else {
            // the weight image the layer's last launch read: K1x 6 B per weight (three bf16 pieces), K1r 24 B per (filter, channel,
            // filter row) = 8 B per weight (four planes x three pieces), the FP32 kernels 4 B (Winograd U: 16 floats per 9 weights)
            const bool k1x = strncmp(l.kernel_name, "conv_f32_x3", 11) == 0, k1r = strncmp(l.kernel_name, "conv_f32_row3", 13) == 0;
            const bool k1w = strncmp(l.kernel_name, "conv_f32_wino", 13) == 0;
            rd += 4 * in_el + (k1x ? 6.0 : (k1r ? 8.0 : (k1w ? 4.0 * 16.0 / 9.0 : 4.0))) * wel;
            if (l.two_src_up >= 0 && n.layers[l.two_src_up].two_src_skipped) {     // the upsampled source is read at its own resolution
                const Layer &up = n.layers[l.two_src_up];
                rd -= 4.0 * B * up.out_c * ((double)up.out_h * up.out_w - (double)up.h * up.w);
            }
            if (l.binarize_input) { rd += 4 * in_el; wr += 4 * in_el; }
            if (l.bits_out_slot >= 0) wr += B * (double)l.out_h * l.out_w * 8.0 * ((l.n + 63) / 64);
        }
        if (fused) { rd += 4 * out_el; wr += 4 * out_el; }                // [shortcut] operand in, sum out
        else if (!l.skip_f32_out) wr += 4 * out_el;
        if (l.fused_pool >= 0) wr += out_el;                              // the pooled tensor of the fused [maxpool]: a quarter of the elements
        if (l.q_out_layer >= 0) wr += (l.conv_mode == CONV_BF16 ? 2 : 1) * out_el;   // int8 / bf16 side output
        break;
    }
    case YL_SHORTCUT:
        if (!l.fused_into_conv) { rd += 8 * out_el; wr += 4 * out_el; }
        break;
    case YL_YOLO:
        if (!l.fused_into_conv) { rd += 4 * in_el; wr += 4 * out_el; }
        break;
    case YL_MAXPOOL: {
        const double wi = B * (double)l.h * l.w * 8.0 * ((l.c + 63) / 64), wo = B * (double)l.out_h * l.out_w * 8.0 * ((l.c + 63) / 64);
        if (l.pool_bits_mode == 1) { rd += wi; wr += wo; }
        else if (l.pool_bits_mode == 2) { rd += 4 * in_el; wr += wo; }
        if (!l.skip_f32_out && !l.fused_into_conv) { rd += 4 * in_el; wr += 4 * out_el; }
        break;
    }
    case YL_ROUTE:
        if (!l.d_output_alias && !(l.n == 1) && !l.skip_f32_out && !l.two_src_skipped) { rd += 4 * out_el; wr += 4 * out_el; }
        break;
    case YL_UPSAMPLE:
        if (!l.skip_f32_out && !l.two_src_skipped) { rd += 4 * in_el; wr += 4 * out_el; }
        break;
    default:
        rd += 4 * in_el; wr += 4 * out_el;
        break;
    }
    bytes[0] = rd; bytes[1] = wr;
    return YL_OK;
}

int yl_network_set_debug(yl_network *net, int on)
{
    if (!net) { set_error("null argument"); return YL_ERR_ARG; }
    if (net->net.on_device) { set_error("set_debug must precede to_device"); return YL_ERR_STATE; }
    net->net.debug = on != 0;
    return YL_OK;
}

int yl_network_set_fusion(yl_network *net, int on)
{
    if (!net) { set_error("null argument"); return YL_ERR_ARG; }
    if (net->net.on_device) { set_error("set_fusion must precede to_device"); return YL_ERR_STATE; }
    net->net.fuse = on != 0;
    return YL_OK;
}

int yl_network_to_device(yl_network *net, int device)
{
    if (!net) { set_error("null argument"); return YL_ERR_ARG; }
    int rc = to_device(net->net, device);
    if (rc != YL_OK) free_device(net->net);
    return rc;
}

int yl_network_set_stream(yl_network *net, void *hip_stream)
{
    if (!net) { set_error("null argument"); return YL_ERR_ARG; }
    Network &n = net->net;
    if (hip_stream == nullptr) {
        if (!n.own_stream) {
            if (n.device >= 0) YL_HIP(hipSetDevice(n.device));
            hipStream_t s;
            YL_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            n.stream = s; n.own_stream = true;
        }
        return YL_OK;
    }
    if (n.own_stream && n.stream) { (void)hipStreamSynchronize((hipStream_t)n.stream); (void)hipStreamDestroy((hipStream_t)n.stream); }
    n.stream = hip_stream; n.own_stream = false;
    return YL_OK;
}

int yl_network_synchronize(yl_network *net)
{
    if (!net || !net->net.on_device) { set_error("network not on device"); return YL_ERR_STATE; }
    YL_HIP(hipSetDevice(net->net.device));
    YL_HIP(hipStreamSynchronize((hipStream_t)net->net.stream));
    return YL_OK;
}

int yl_network_forward(yl_network *net, const float *input_dev)
{
    if (!net || !input_dev) { set_error("null argument"); return YL_ERR_ARG; }
    return forward(net->net, input_dev, -1);
}

// Host float images [b0, b0 + nb) -> pinned staging -> device on in_stream, in 16 MB chunks: the staging copy of a chunk is
// spread over the host pool (one core moves ~8 GB/s; the 284 MB of a 608x608 batch of 64 took 35 ms on one) and the H2D of
// chunk k runs while chunk k+1 is being staged.  h_pinned holds the whole batch, so no chunk is reused inside a call.
static int stage_input_h2d(Network &n, const float *input, int b0, int nb, hipEvent_t landed)
{
    const size_t img = (size_t)n.c * n.h * n.w * sizeof(float);
    const size_t base = (size_t)b0 * img, total = (size_t)nb * img;
    const size_t CHUNK = (size_t)16 << 20;
    const char *src = reinterpret_cast<const char *>(input) + base;
    char *pin = reinterpret_cast<char *>(n.h_pinned) + base;
    char *dev = reinterpret_cast<char *>(n.d_input) + base;
    hipStream_t cs = (hipStream_t)n.in_stream;
    for (size_t off = 0; off < total; off += CHUNK) {
        const size_t len = (total - off < CHUNK) ? total - off : CHUNK;
        HostCopyJob job;
        host_copy_async(job, pin + off, src + off, len);
        host_copy_wait(job);
        if (hipMemcpyAsync(dev + off, pin + off, len, hipMemcpyHostToDevice, cs) != hipSuccess) {
            set_error("H2D input copy failed");
            return YL_ERR_DEVICE;
        }
    }
    YL_HIP(hipEventRecord(landed, cs));
    return YL_OK;
}

// network_predict_cpu's contract (src/yolov2_forward_network.c:632-646: host floats in, every head's l.output filled, the last
// layer's returned) as a three-stage pipeline instead of the reference GPU path's stage-all -> forward -> pull-all
// (src/yolov2_forward_network_gpu.cu:547-573, :438): the batch runs as `split` sub-batches; while sub-batch k computes, the
// input of k+1 is staged and sent (in_stream) and the heads of k-1 come back (out_stream) and are copied out to the caller.
// Same bits as one pass over the whole batch (BatchWindow).  split: 2 from 32 images (a 32-image pass runs at 0.96 of the
// 64-image rate per image), 1 below; YL_PREDICT_SPLIT overrides (1 .. PREDICT_MAX_SPLIT).
static int predict_split(const Network &n)
{
    int split = n.batch >= 32 ? 2 : 1;
    if (const char *e = getenv("YL_PREDICT_SPLIT")) {
        const int v = atoi(e);
        if (v >= 1 && v <= PREDICT_MAX_SPLIT) split = v;
    }
    if (n.debug) split = 1;              // the debug tensors (d_debug) are whole-batch
    if (split > n.batch) split = n.batch;
    return split;
}

float *yl_network_predict(yl_network *net, const float *input)
{
    if (!net || !input) { set_error("null argument"); return nullptr; }
    Network &n = net->net;
    if (!n.on_device) { set_error("network not on device: call yl_network_to_device first"); return nullptr; }
    if (hipSetDevice(n.device) != hipSuccess) { set_error("hipSetDevice failed"); return nullptr; }
    const int B = n.batch, split = predict_split(n);
    const size_t nl = n.layers.size();
    std::vector<HeadPiece> pieces;
    size_t n_ev = 0;
    const bool timing = getenv("YL_PREDICT_TIMING") != nullptr;       // host-side stage times on stderr (lab)
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now();
    double t_stage = 0.0, t_enqueue = 0.0;
    // (h_pinned / h_heads of the previous call are free: every predict ends with its streams drained)
    for (int k = 0; k < split; ++k) {
        const int b0 = (int)((long long)B * k / split), nb = (int)((long long)B * (k + 1) / split) - b0;
        const double t0 = now();
        if (stage_input_h2d(n, input, b0, nb, (hipEvent_t)n.in_events[k]) != YL_OK) return nullptr;
        t_stage += now() - t0;
        if (hipStreamWaitEvent((hipStream_t)n.stream, (hipEvent_t)n.in_events[k], 0) != hipSuccess) { set_error("hipStreamWaitEvent failed"); return nullptr; }
        int frc;
        {
            BatchWindow win(n, b0, nb);
            n.head_event_base = (int)(k * nl);
            frc = forward(n, n.d_input, -1);
            n.head_event_base = -1;
        }
        if (frc != YL_OK) return nullptr;
        if (enqueue_head_pull(n, true, b0, nb, (int)(k * nl), pieces, n_ev) != YL_OK) return nullptr;
        t_enqueue += now() - t0;
    }
    const double t_pull = now();
    if (finish_head_pull(n, pieces) != YL_OK) return nullptr;
    if (timing)
        fprintf(stderr, "yl_network_predict: batch %d split %d: input staging %.2f ms (host), staging + enqueue %.2f ms, wait + copy-out %.2f ms, total %.2f ms; "
                        "%u pool threads, %zu bounce pieces\n", B, split, t_stage, t_enqueue, now() - t_pull, now() - t_start, host_copy_threads(), pieces.size());
    // last non-COST layer (src/yolov2_forward_network.c:644-645); COST never parses here
    return n.layers.back().host_output;
}

int yl_network_pull_heads(yl_network *net)
{
    if (!net || !net->net.on_device) { set_error("network not on device"); return YL_ERR_STATE; }
    YL_HIP(hipSetDevice(net->net.device));
    return pull_heads(net->net, true);
}

int yl_network_layer_output(yl_network *net, int i, float *dst_host)
{
    YL_LAYER_OR(YL_ERR_ARG)
    if (!dst_host || !net->net.on_device) { set_error("bad argument / not on device"); return YL_ERR_STATE; }
    if (!layer_materialised(l)) {
        set_error("this layer's FP32 tensor is not materialised under yl_network_set_fusion (folded into its consumer)");
        return YL_ERR_STATE;
    }
    YL_HIP(hipSetDevice(net->net.device));
    YL_HIP(hipStreamSynchronize((hipStream_t)net->net.stream));
    YL_STAGE(stage_d2h(net->net.device, dst_host, l.d_output, sizeof(float) * (size_t)net->net.batch * l.outputs));
    return YL_OK;
}

int yl_network_layer_output_image(yl_network *net, int i, int image, float *dst_host)
{
    YL_LAYER_OR(YL_ERR_ARG)
    if (!dst_host || !net->net.on_device) { set_error("bad argument / not on device"); return YL_ERR_STATE; }
    if (image < 0 || image >= net->net.batch) { set_error("image index out of range"); return YL_ERR_ARG; }
    if (!layer_materialised(l)) {
        set_error("this layer's FP32 tensor is not materialised under yl_network_set_fusion (folded into its consumer)");
        return YL_ERR_STATE;
    }
    YL_HIP(hipSetDevice(net->net.device));
    YL_HIP(hipStreamSynchronize((hipStream_t)net->net.stream));
    YL_STAGE(stage_d2h(net->net.device, dst_host, l.d_output + (size_t)image * l.outputs, sizeof(float) * (size_t)l.outputs));
    return YL_OK;
}

const float *yl_network_layer_output_dev(const yl_network *net, int i)
{
    YL_LAYER_OR(nullptr)
    if (!layer_materialised(l)) { set_error("layer output not materialised under fusion"); return nullptr; }
    return l.d_output;
}

float *yl_network_input_dev(yl_network *net) { return net ? net->net.d_input : nullptr; }

static int pull_debug(yl_network *net, int i, int32_t *dst, int want_mode)
{
    YL_LAYER_OR(YL_ERR_ARG)
    if (!dst || !net->net.on_device) { set_error("bad argument / not on device"); return YL_ERR_STATE; }
    if (l.type != YL_CONVOLUTIONAL || l.conv_mode != want_mode || !l.d_debug) {
        set_error("layer has no debug tensor of this kind (set_debug before to_device?)");
        return YL_ERR_STATE;
    }
    YL_HIP(hipSetDevice(net->net.device));
    YL_HIP(hipStreamSynchronize((hipStream_t)net->net.stream));
    YL_STAGE(stage_d2h(net->net.device, dst, l.d_debug, sizeof(int32_t) * (size_t)net->net.batch * l.outputs));
    return YL_OK;
}

int yl_network_layer_xnor_counts(yl_network *net, int i, int32_t *dst_host) { return pull_debug(net, i, dst_host, CONV_XNOR); }
int yl_network_layer_int8_acc(yl_network *net, int i, int32_t *dst_host) { return pull_debug(net, i, dst_host, CONV_INT8); }

#define YL_MAX_TIMING_SLOTS 64

static int ensure_layer_events(Network &n, int slots)
{
    const size_t nl = n.layers.size();
    while (n.layer_events.size() < (size_t)slots * (nl + 1)) {
        hipEvent_t e;
        YL_HIP(hipEventCreate(&e));
        n.layer_events.push_back(e);
    }
    return YL_OK;
}

int yl_network_forward_timed(yl_network *net, const float *input_dev, int slot)
{
    if (!net || !input_dev || slot < 0 || slot >= YL_MAX_TIMING_SLOTS) { set_error("bad argument"); return YL_ERR_ARG; }
    Network &n = net->net;
    if (!n.on_device) { set_error("network not on device"); return YL_ERR_STATE; }
    YL_HIP(hipSetDevice(n.device));
    int rc = ensure_layer_events(n, slot + 1);
    if (rc != YL_OK) return rc;
    return forward(n, input_dev, slot);
}

int yl_network_layer_times(yl_network *net, int slot, float *ms_per_layer, float *total_ms)
{
    if (!net || slot < 0) { set_error("bad argument"); return YL_ERR_ARG; }
    Network &n = net->net;
    const size_t nl = n.layers.size();
    if (!n.on_device || n.layer_events.size() < (size_t)(slot + 1) * (nl + 1)) { set_error("no timed forward recorded in this slot"); return YL_ERR_STATE; }
    YL_HIP(hipSetDevice(n.device));
    void **ev = &n.layer_events[(size_t)slot * (nl + 1)];
    YL_HIP(hipEventSynchronize((hipEvent_t)ev[nl]));
    for (size_t i = 0; i < nl && ms_per_layer; ++i)
        YL_HIP(hipEventElapsedTime(&ms_per_layer[i], (hipEvent_t)ev[i], (hipEvent_t)ev[i + 1]));
    if (total_ms) YL_HIP(hipEventElapsedTime(total_ms, (hipEvent_t)ev[0], (hipEvent_t)ev[nl]));
    return YL_OK;
}

const char *yl_network_layer_kernel(const yl_network *net, int i)
{
    YL_LAYER_OR(nullptr)
    return l.kernel_name;
}

int yl_network_profile(yl_network *net, const float *input_dev, int iters, float *ms_per_layer, float *total_ms)
{
    if (!net || !input_dev || iters <= 0) { set_error("bad argument"); return YL_ERR_ARG; }
    Network &n = net->net;
    if (!n.on_device) { set_error("network not on device"); return YL_ERR_STATE; }
    const size_t nl = n.layers.size();
    std::vector<double> acc(nl, 0.0);
    std::vector<float> ms(nl, 0.f);
    double tot = 0.0;
    for (int it = 0; it < iters; ++it) {
        int rc = yl_network_forward_timed(net, input_dev, 0);
        if (rc != YL_OK) return rc;
        float t = 0.f;
        rc = yl_network_layer_times(net, 0, ms.data(), &t);
        if (rc != YL_OK) return rc;
        for (size_t i = 0; i < nl; ++i) acc[i] += ms[i];
        tot += t;
    }
    if (ms_per_layer) for (size_t i = 0; i < nl; ++i) ms_per_layer[i] = (float)(acc[i] / iters);
    if (total_ms) *total_ms = (float)(tot / iters);
    return YL_OK;
}


// decode + NMS of the whole batch on the device, then only the FILLED rows travel: counts first, rows packed densely
// into the network's pinned block (grown on demand from the counts -- a fixed [batch][cap][6+classes] host image is
// 90 MB at batch 64 / 80 classes and 1.2 GB for a YOLO9000 head at batch 8).  Leaves n.det_counts / n.det_row_off.
static int detect_to_pinned(yl_network *net, const int *img_w, const int *img_h, float thresh, int relative, int letter,
                            float nms, int cap)
{
    Network &n = net->net;
    YL_HIP(hipSetDevice(n.device));
    const int classes = n.layers.back().classes;
    const int B = n.batch;
    const size_t row = (size_t)(6 + classes);
    const size_t need = sizeof(float) * (size_t)B * cap * row;
    if (n.det_out_bytes < need) {
        YL_HIP(hipStreamSynchronize((hipStream_t)n.stream));
        if (n.d_det_out) (void)hipFree(n.d_det_out);
        n.d_det_out = nullptr; n.det_out_bytes = 0;
        YL_HIP(hipMalloc((void **)&n.d_det_out, need));
        n.det_out_bytes = need;
    }
    if (!n.d_det_counts) YL_HIP(hipMalloc((void **)&n.d_det_counts, sizeof(int) * 2 * (size_t)B));
    if (!n.h_det_counts) YL_HIP(hipHostMalloc((void **)&n.h_det_counts, sizeof(int) * (size_t)B, hipHostMallocDefault));
    const int rc = yl_network_detect_batch(net, img_w, img_h, thresh, relative, letter, nms, cap, n.d_det_out,
                                           n.d_det_counts + B);
    if (rc != YL_OK) return rc;
    hipStream_t s = (hipStream_t)n.stream;
    YL_HIP(hipMemcpyAsync(n.h_det_counts, n.d_det_counts + B, sizeof(int) * (size_t)B, hipMemcpyDeviceToHost, s));
    YL_HIP(hipStreamSynchronize(s));
    n.det_counts.assign(n.h_det_counts, n.h_det_counts + B);
    n.det_row_off.assign((size_t)B + 1, 0);
    for (int b = 0; b < B; ++b) {
        const int c = n.det_counts[b] < cap ? n.det_counts[b] : cap;
        n.det_row_off[(size_t)b + 1] = n.det_row_off[b] + (size_t)(c > 0 ? c : 0) * row;
    }
    const size_t host_need = sizeof(float) * n.det_row_off[B];
    if (n.h_det_bytes < host_need) {
        if (n.h_det_rows) (void)hipHostFree(n.h_det_rows);
        n.h_det_rows = nullptr; n.h_det_bytes = 0;
        const size_t grow = host_need + host_need / 2 + 4096;
        YL_HIP(hipHostMalloc((void **)&n.h_det_rows, grow, hipHostMallocDefault));
        n.h_det_bytes = grow;
    }
    for (int b = 0; b < B; ++b) {
        const size_t fl = n.det_row_off[(size_t)b + 1] - n.det_row_off[b];
        if (fl)
            YL_HIP(hipMemcpyAsync(n.h_det_rows + n.det_row_off[b], n.d_det_out + (size_t)b * cap * row, sizeof(float) * fl,
                                  hipMemcpyDeviceToHost, s));
    }
    YL_HIP(hipStreamSynchronize(s));
    return YL_OK;
}

int yl_network_get_boxes(yl_network *net, int image, int w, int h, float thresh, int relative, int letter,
                         float nms, float *rows, int max_rows, int *classes_out)
{
    if (!net) { set_error("null argument"); return YL_ERR_ARG; }
    Network &n = net->net;
    if (!n.on_device) { set_error("network not on device"); return YL_ERR_STATE; }
    if (image < 0 || image >= n.batch) { set_error("image index out of range"); return YL_ERR_ARG; }
    if (w <= 0 || h <= 0) { set_error("image size must be positive"); return YL_ERR_ARG; }
    const int classes = n.layers.back().classes;
    if (classes_out) *classes_out = classes;
    const int cap = NMS_MAX_CAP;
    const size_t row = (size_t)(6 + classes);
    // the decode + NMS of the whole batch is one pass on the device: keep it while the caller walks
    // the images of one forward with the same arguments
    DetKey key{n.forward_seq, w, h, thresh, relative, letter, nms};
    if (!n.det_cache_valid || memcmp(&key, &n.det_cache_key, sizeof(key)) != 0) {
        std::vector<int> ws((size_t)n.batch, w), hs((size_t)n.batch, h);
        n.det_cache_valid = false;
        const int rc = detect_to_pinned(net, ws.data(), hs.data(), thresh, relative, letter, nms, cap);
        if (rc != YL_OK) return rc;
        n.det_cache_key = key;
        n.det_cache_valid = true;
    }
    const int count = n.det_counts[image];
    int nrows = count < cap ? count : cap;
    if (nrows > max_rows) nrows = max_rows;
    if (rows && nrows > 0) memcpy(rows, n.h_det_rows + n.det_row_off[image], sizeof(float) * row * nrows);
    return count;
}

int yl_network_set_conv_tile(yl_network *net, int cfg)
{
    if (!net) { set_error("null argument"); return YL_ERR_ARG; }
    if (!(cfg == 0 || (cfg >= 11 && cfg <= 22) || cfg == 31 || cfg == 41 || (cfg >= 51 && cfg <= 55) || (cfg >= 61 && cfg <= 70))) { set_error("unknown tile id"); return YL_ERR_ARG; }
    net->net.conv_opts.force_tile = cfg;
    return YL_OK;
}

int yl_network_set_variant(yl_network *net, int bits)
{
    if (!net) { set_error("null argument"); return YL_ERR_ARG; }
    if (bits < -1 || bits > 32767) { set_error("unknown variant bits"); return YL_ERR_ARG; }
    net->net.conv_opts.variant = bits < 0 ? YL_VARIANT_DEFAULT : bits;
    return YL_OK;
}

int yl_network_set_precision(yl_network *net, int precision)
{
    if (!net || (precision != YL_PRECISION_FP32 && precision != YL_PRECISION_BF16 && precision != YL_PRECISION_FP32_STRICT)) {
        set_error("bad argument");
        return YL_ERR_ARG;
    }
    if (net->net.on_device) { set_error("set_precision must precede to_device"); return YL_ERR_STATE; }
    if (precision == YL_PRECISION_FP32_STRICT) {
        // FP32-MFMA direct kernels only: no Winograd weight images, no three-piece (bits 10 / 11) kernels
        net->net.conv_opts.winograd = false;
        net->net.conv_opts.variant = 2 | 4 | 8 | 16 | 32;
        precision = YL_PRECISION_FP32;
    }
    net->net.precision = precision;
    select_conv_modes(net->net);
    return YL_OK;
}

int yl_network_set_int8_tile(yl_network *net, int cfg)
{
    if (!net) { set_error("null argument"); return YL_ERR_ARG; }
    if (cfg < 0 || cfg > 7) { set_error("unknown tile id"); return YL_ERR_ARG; }
    net->net.i8_tile = cfg;
    return YL_OK;
}

int yl_network_set_winograd(yl_network *net, int on)
{
    if (!net) { set_error("null argument"); return YL_ERR_ARG; }
    if (net->net.on_device) { set_error("set_winograd must precede to_device"); return YL_ERR_STATE; }
    net->net.conv_opts.winograd = on != 0;
    return YL_OK;
}

int yl_network_set_device_pack(yl_network *net, int on)
{
    if (!net) { set_error("null argument"); return YL_ERR_ARG; }
    if (net->net.on_device) { set_error("set_device_pack must precede to_device"); return YL_ERR_STATE; }
    net->net.device_pack = on != 0;
    return YL_OK;
}

long long yl_debug_layer_packed(yl_network *net, int i, int which, void *dst_host, long long dst_bytes)
{
    YL_LAYER_OR(YL_ERR_ARG)
    if (!net->net.on_device || which < 0 || which > 8) { set_error("bad argument / not on device"); return YL_ERR_STATE; }
    const void *src = nullptr;
    long long need = 0;
    if (which <= 3) {
        src = which == 0 ? (const void *)l.d_weights_t : which == 1 ? (const void *)l.d_wino32_u
            : which == 2 ? (const void *)l.d_weights_i8 : (const void *)l.d_weights_bits;
        need = src ? (long long)l.packed_bytes[which] : 0;
    } else if (which == 7) {                                  // K1x: the weights as three bf16 pieces
        src = l.d_weights_x3;
        need = src ? (long long)l.packed_bytes[4] : 0;
    } else if (which == 8) {                                  // K1r: the row-transformed weights as three bf16 pieces
        src = l.d_weights_r3;
        need = src ? (long long)l.packed_bytes[5] : 0;
    } else if (l.type == YL_CONVOLUTIONAL && l.d_thr) {       // XNOR layers: thresholds (+ the not-a-step count), mean, bias
        src = which == 4 ? (const void *)l.d_thr : which == 5 ? (const void *)l.d_mean : (const void *)l.d_biases;
        need = which == 4 ? (long long)sizeof(int) * (l.Mpad + 1) : (long long)sizeof(float) * l.n;
    }
    if (!dst_host || need == 0) return need;
    if (dst_bytes < need) { set_error("dst too small"); return YL_ERR_ARG; }
    YL_HIP(hipSetDevice(net->net.device));
    YL_HIP(hipStreamSynchronize((hipStream_t)net->net.stream));
    YL_STAGE(stage_d2h(net->net.device, dst_host, src, (size_t)need));
    return need;
}

int yl_network_set_split_k(yl_network *net, int on)
{
    if (!net) { set_error("null argument"); return YL_ERR_ARG; }
    if (net->net.on_device) { set_error("set_split_k must precede to_device"); return YL_ERR_STATE; }
    net->net.split_k = on != 0;
    return YL_OK;
}

int yl_network_set_nms_mode(yl_network *net, int mode)
{
    if (!net) { set_error("null argument"); return YL_ERR_ARG; }
    net->net.nms_mode = mode != 0;
    net->net.det_cache_valid = false;
    return YL_OK;
}

int yl_network_set_quant_rule(yl_network *net, int rule)
{
    if (!net || (rule != YL_QUANT_RULE_CPU && rule != YL_QUANT_RULE_GPU)) { set_error("bad argument"); return YL_ERR_ARG; }
    if (net->net.on_device) { set_error("set_quant_rule must precede to_device"); return YL_ERR_STATE; }
    net->net.quant_rule = rule;
    select_conv_modes(net->net);
    return YL_OK;
}

int yl_network_layer_head(const yl_network *net, int i, int *mask, int mask_cap, float *anchors, int anchors_cap,
                          int *anchors_len_out)
{
    YL_LAYER_OR(YL_ERR_ARG)
    if (l.type != YL_YOLO && l.type != YL_REGION) { set_error("not a detection head"); return YL_ERR_ARG; }
    if (anchors_len_out) *anchors_len_out = (int)l.anchors.size();
    if ((mask && mask_cap < l.n) || (anchors && (size_t)(anchors_cap < 0 ? 0 : anchors_cap) < l.anchors.size())) {
        set_error("yl_network_layer_head: caller array too small");
        return YL_ERR_ARG;
    }
    for (int k = 0; k < l.n && mask; ++k) mask[k] = (l.type == YL_YOLO) ? l.mask[k] : k;
    for (size_t k = 0; k < l.anchors.size() && anchors; ++k) anchors[k] = l.anchors[k];
    return l.n;
}

int yl_network_layer_tree(const yl_network *net, int i, int *parent, int *group_size)
{
    YL_LAYER_OR(YL_ERR_ARG)
    if (l.type != YL_REGION) { set_error("not a region layer"); return YL_ERR_ARG; }
    if (parent) for (size_t k = 0; k < l.tree_parent.size(); ++k) parent[k] = l.tree_parent[k];
    if (group_size) for (size_t k = 0; k < l.tree_group_size.size(); ++k) group_size[k] = l.tree_group_size[k];
    return l.tree_parent.empty() ? 0 : (int)l.tree_group_size.size();
}

long long yl_debug_wino_pack(const float *weights, int c, int m, int tiling, float *dst, long long dst_floats)
{
    if (!weights || c <= 0 || m <= 0 || c % 8 != 0 || tiling != 32) { set_error("bad argument"); return YL_ERR_ARG; }
    const size_t need = wino32_packed_floats(c, m);
    if (!dst) return (long long)need;
    if (dst_floats < (long long)need) { set_error("dst too small"); return YL_ERR_ARG; }
    wino32_pack_weights(weights, c, m, dst);
    return (long long)need;
}

long long yl_debug_x3_pack(const float *weights, int c, int m, int size, void *dst, long long dst_bytes)
{
    if (!weights || m <= 0 || !x3_applicable(c, m, size, 1, 0)) { set_error("bad argument"); return YL_ERR_ARG; }
    const size_t need = x3_packed_bytes(c, m, size);
    if (!dst) return (long long)need;
    if (dst_bytes < (long long)need) { set_error("dst too small"); return YL_ERR_ARG; }
    x3_pack_weights(weights, c, m, size, dst);
    return (long long)need;
}

long long yl_debug_row3_pack(const float *weights, int c, int m, void *dst, long long dst_bytes)
{
    if (!weights || m <= 0 || !row3_applicable(c, m, 3, 1, 1)) { set_error("bad argument"); return YL_ERR_ARG; }
    const size_t need = row3_packed_bytes(c, m);
    if (!dst) return (long long)need;
    if (dst_bytes < (long long)need) { set_error("dst too small"); return YL_ERR_ARG; }
    row3_pack_weights(weights, c, m, dst);
    return (long long)need;
}

// ------------------------------------------------------------------ INT8 calibration tool
float yl_entropy_from_histogram(const uint32_t *counts, int max_bin, float bin_width)
{
    if (!counts || max_bin < 129 || max_bin > (1 << 20) || !(bin_width > 0.f)) { set_error("bad argument"); return -1.f; }
    return entropy_from_counts(counts, max_bin, bin_width);
}

int yl_network_calibrate(yl_network *net, const float *images_host, int n_images, float *multipliers, int max_out)
{
    if (!net || !images_host || !multipliers || n_images <= 0) { set_error("bad argument"); return YL_ERR_ARG; }
    Network &n = net->net;
    if (!n.on_device) { set_error("network not on device"); return YL_ERR_STATE; }
    if (n.quantized) { set_error("calibration runs the FP32 network: load it with quantized = 0"); return YL_ERR_STATE; }
    if (n_images % n.batch != 0) { set_error("n_images must be a multiple of the network batch"); return YL_ERR_ARG; }
    YL_HIP(hipSetDevice(n.device));
    hipStream_t s = (hipStream_t)n.stream;
    const int B = n.batch;
    const int MAX_BIN = 4096;                 // entropy_calibration(state.input, l.inputs, 1.0 / 16, 4096)
    const float BIN_W = 1.0f / 16;
    const size_t nl = n.layers.size();
    std::vector<int> conv_ids;
    for (size_t i = 0; i < nl; ++i)
        if (n.layers[i].type == YL_CONVOLUTIONAL) conv_ids.push_back((int)i);
    if ((int)conv_ids.size() > max_out) { set_error("multipliers[] too small"); return YL_ERR_ARG; }
    // mult[layer][image], image index 0-based here (the reference's `counter` is 1-based)
    std::vector<std::vector<float>> mult(nl, std::vector<float>());
    for (int i : conv_ids) mult[i].assign((size_t)n_images, 0.f);
    unsigned *d_hist = nullptr;
    YL_HIP(hipMalloc((void **)&d_hist, sizeof(unsigned) * (size_t)B * MAX_BIN));
    unsigned *h_hist = nullptr;                // pinned: the per-layer histogram download never touches pageable memory
    const size_t h_hist_n = (size_t)B * MAX_BIN;
    if (hipHostMalloc((void **)&h_hist, sizeof(unsigned) * h_hist_n, hipHostMallocDefault) != hipSuccess) {
        (void)hipFree(d_hist); set_error("hipHostMalloc of the calibration histogram failed"); return YL_ERR_DEVICE;
    }
    int rc = YL_OK;
    for (int img0 = 0; img0 < n_images && rc == YL_OK; img0 += B) {
        memcpy(n.h_pinned, images_host + (size_t)img0 * n.c * n.h * n.w, n.pinned_bytes);
        if (hipMemcpyAsync(n.d_input, n.h_pinned, n.pinned_bytes, hipMemcpyHostToDevice, s) != hipSuccess) { rc = YL_ERR_DEVICE; break; }
        const float *input = n.d_input;
        for (size_t i = 0; i < nl && rc == YL_OK; ++i) {
            Layer &l = n.layers[i];
            if (l.type == YL_CONVOLUTIONAL) {
                // the layer's input as the forward pass sees it (network_calibrate_cpu: state.input, l.inputs)
                if (launch_hist_abs(input, (size_t)l.inputs, B, MAX_BIN, BIN_W, d_hist, s) != 0 ||
                    hipMemcpyAsync(h_hist, d_hist, sizeof(unsigned) * h_hist_n, hipMemcpyDeviceToHost, s) != hipSuccess ||
                    hipStreamSynchronize(s) != hipSuccess) { set_error("calibration histogram failed"); rc = YL_ERR_DEVICE; break; }
                // the KL scans of the B images of this layer are independent: one host thread each
                // (the reference spends most of its calibration time here, single-threaded)
                {
                    unsigned hw = std::thread::hardware_concurrency();
                    const int nt = (int)(hw == 0 ? 1 : (hw > 16 ? 16 : hw));
                    std::vector<std::thread> th;
                    float *dst = mult[i].data() + img0;
                    const unsigned *hh = h_hist;
                    for (int t0 = 1; t0 < nt && t0 < B; ++t0)
                        th.emplace_back([=] {
                            for (int b = t0; b < B; b += nt) dst[b] = entropy_from_counts(hh + (size_t)b * MAX_BIN, MAX_BIN, BIN_W);
                        });
                    for (int b = 0; b < B; b += nt) dst[b] = entropy_from_counts(hh + (size_t)b * MAX_BIN, MAX_BIN, BIN_W);
                    for (auto &x : th) x.join();
                }
            }
            rc = forward_layer(n, i, input);
            input = l.d_output;
        }
    }
    (void)hipStreamSynchronize(s);
    (void)hipFree(d_hist);
    (void)hipHostFree(h_hist);
    if (rc != YL_OK) return rc;
    // The reference's averaging, slot for slot (src/yolov2_forward_network.c:786-797): multipliers go to
    // input_mult_array[counter + i*max_num] with a 1-based image counter, the mean is taken over slots
    // 0..max_num-1.  Slot 0 of layer i is slot max_num of layer i-1, i.e. the LAST image's multiplier of
    // the previous layer if that layer is a convolution and 0 otherwise, and the last image's own
    // multiplier (slot max_num) is left out.  Reproduced as is so the numbers agree with the tool.
    for (size_t k = 0; k < conv_ids.size(); ++k) {
        const int i = conv_ids[k];
        float res = 0;
        const bool prev_conv = i > 0 && n.layers[(size_t)i - 1].type == YL_CONVOLUTIONAL;
        res += prev_conv ? mult[(size_t)i - 1][(size_t)n_images - 1] : 0.f;
        for (int j = 1; j < n_images; ++j) res += mult[i][(size_t)j - 1];
        multipliers[k] = res / n_images;
    }
    return (int)conv_ids.size();
}

static int check_image_args(yl_network *net, int image, const void *pixels, int w, int h, int c)
{
    if (!net || !pixels) { set_error("null argument"); return YL_ERR_ARG; }
    Network &n = net->net;
    if (!n.on_device) { set_error("network not on device"); return YL_ERR_STATE; }
    if (image < 0 || image >= n.batch) { set_error("image index out of range"); return YL_ERR_ARG; }
    if (w <= 0 || h <= 0) { set_error("image size must be positive"); return YL_ERR_ARG; }
    if (c != n.c) { set_error("image channels differ from the network's"); return YL_ERR_ARG; }
    if (n.w < 2 || n.h < 2) { set_error("resize_image needs a network input of at least 2x2"); return YL_ERR_UNSUPPORTED; }
    return YL_OK;
}

int yl_network_set_input_u8_dev(yl_network *net, int image, const uint8_t *pixels_dev, int w, int h, int c)
{
    const int rc = check_image_args(net, image, pixels_dev, w, h, c);
    if (rc != YL_OK) return rc;
    Network &n = net->net;
    YL_HIP(hipSetDevice(n.device));
    float *dst = n.d_input + (size_t)image * n.c * n.h * n.w;
    YL_LAUNCH(launch_load_resize_u8(pixels_dev, w, h, c, n.w, n.h, dst, n.stream), "load_resize_u8");
    return YL_OK;
}

// pinned + device staging slots of at least `bytes` per batch slot, and the per-slot "H2D done" events
static int ensure_u8_slots(Network &n, size_t bytes)
{
    hipStream_t s = (hipStream_t)n.stream;
    if (bytes > n.u8_stride) {
        // grow every slot: earlier stagings must have been consumed first
        YL_HIP(hipStreamSynchronize(s));
        if (n.in_stream) YL_HIP(hipStreamSynchronize((hipStream_t)n.in_stream));
        size_t stride = n.u8_stride ? n.u8_stride : (size_t)n.w * n.h * n.c;
        while (stride < bytes) stride += stride / 2;
        stride = (stride + 4095) & ~(size_t)4095;
        if (n.h_u8) (void)hipHostFree(n.h_u8);
        if (n.d_u8) (void)hipFree(n.d_u8);
        n.h_u8 = nullptr; n.d_u8 = nullptr; n.u8_stride = 0;
        YL_HIP(hipHostMalloc((void **)&n.h_u8, stride * n.batch, hipHostMallocDefault));
        YL_HIP(hipMalloc((void **)&n.d_u8, stride * n.batch));
        n.u8_stride = stride;
    }
    if (n.u8_events.size() != (size_t)n.batch) {
        n.u8_events.assign((size_t)n.batch, nullptr);
        for (auto &e : n.u8_events) {
            hipEvent_t ev;
            YL_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            e = ev;
            YL_HIP(hipEventRecord(ev, s));
        }
    }
    return YL_OK;
}

int yl_network_set_input_u8_batch(yl_network *net, int first, int count, const uint8_t *const *pixels_host,
                                  const int *w, const int *h, int c)
{
    if (!net || !pixels_host || !w || !h) { set_error("null argument"); return YL_ERR_ARG; }
    Network &n = net->net;
    if (count < 0 || first < 0 || (n.on_device && first + count > n.batch)) { set_error("frame range outside the batch"); return YL_ERR_ARG; }
    size_t mx = 0;
    for (int i = 0; i < count; ++i) {
        const int rc = check_image_args(net, first + i, pixels_host[i], w[i], h[i], c);
        if (rc != YL_OK) return rc;
        mx = std::max(mx, (size_t)w[i] * h[i] * c);
    }
    if (count == 0) return YL_OK;
    YL_HIP(hipSetDevice(n.device));
    int rc = ensure_u8_slots(n, mx);
    if (rc != YL_OK) return rc;
    hipStream_t s = (hipStream_t)n.stream, cs = (hipStream_t)n.in_stream;
    if (!n.u8_resized) {
        hipEvent_t ev;
        YL_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        n.u8_resized = ev;
        YL_HIP(hipEventRecord(ev, s));
    }
    // pageable -> pinned: every frame's copy on the host pool at once (a slot's previous upload must have left its pinned region)
    HostCopyJob job;
    for (int i = 0; i < count; ++i) {
        YL_HIP(hipEventSynchronize((hipEvent_t)n.u8_events[first + i]));
        host_copy_async(job, n.h_u8 + (size_t)(first + i) * n.u8_stride, pixels_host[i], (size_t)w[i] * h[i] * c);
    }
    host_copy_wait(job);
    // pinned -> device on the copy stream, behind the resize kernels that last read the device slots
    YL_HIP(hipStreamWaitEvent(cs, (hipEvent_t)n.u8_resized, 0));
    for (int i = 0; i < count; ++i) {
        const size_t off = (size_t)(first + i) * n.u8_stride;
        YL_HIP(hipMemcpyAsync(n.d_u8 + off, n.h_u8 + off, (size_t)w[i] * h[i] * c, hipMemcpyHostToDevice, cs));
        YL_HIP(hipEventRecord((hipEvent_t)n.u8_events[first + i], cs));
    }
    // conversion + resize on the compute stream, behind the last upload (events are ordered on the copy stream)
    YL_HIP(hipStreamWaitEvent(s, (hipEvent_t)n.u8_events[first + count - 1], 0));
    for (int i = 0; i < count; ++i) {
        float *dst = n.d_input + (size_t)(first + i) * n.c * n.h * n.w;
        YL_LAUNCH(launch_load_resize_u8(n.d_u8 + (size_t)(first + i) * n.u8_stride, w[i], h[i], c, n.w, n.h, dst, n.stream), "load_resize_u8");
    }
    YL_HIP(hipEventRecord((hipEvent_t)n.u8_resized, s));
    return YL_OK;
}

int yl_network_set_input_u8(yl_network *net, int image, const uint8_t *pixels_host, int w, int h, int c)
{
    int rc = check_image_args(net, image, pixels_host, w, h, c);
    if (rc != YL_OK) return rc;
    Network &n = net->net;
    YL_HIP(hipSetDevice(n.device));
    hipStream_t s = (hipStream_t)n.stream;
    const size_t bytes = (size_t)w * h * c;
    rc = ensure_u8_slots(n, bytes);
    if (rc != YL_OK) return rc;
    // the slot's staging region may still be the source of the previous frame's copy
    YL_HIP(hipEventSynchronize((hipEvent_t)n.u8_events[image]));
    uint8_t *hs = n.h_u8 + (size_t)image * n.u8_stride;
    uint8_t *ds = n.d_u8 + (size_t)image * n.u8_stride;
    memcpy(hs, pixels_host, bytes);
    YL_HIP(hipMemcpyAsync(ds, hs, bytes, hipMemcpyHostToDevice, s));
    YL_HIP(hipEventRecord((hipEvent_t)n.u8_events[image], s));
    float *dst = n.d_input + (size_t)image * n.c * n.h * n.w;
    YL_LAUNCH(launch_load_resize_u8(ds, w, h, c, n.w, n.h, dst, n.stream), "load_resize_u8");
    return YL_OK;
}

int yl_network_input_download(yl_network *net, float *dst_host)
{
    if (!net || !dst_host) { set_error("null argument"); return YL_ERR_ARG; }
    Network &n = net->net;
    if (!n.on_device) { set_error("network not on device"); return YL_ERR_STATE; }
    YL_HIP(hipSetDevice(n.device));
    YL_HIP(hipStreamSynchronize((hipStream_t)n.stream));
    YL_STAGE(stage_d2h(n.device, dst_host, n.d_input, sizeof(float) * (size_t)n.batch * n.c * n.h * n.w));
    return YL_OK;
}

static int collect_heads(Network &n, HeadDesc *heads, int *nh_out, int *classes_out)
{
    int nh = 0;
    const int classes = n.layers.back().classes;
    for (const Layer &l : n.layers) {
        if (l.type != YL_YOLO && l.type != YL_REGION) continue;
        if (nh == 4 || l.n > 16) { set_error("too many heads/anchors for compaction"); return YL_ERR_UNSUPPORTED; }
        if (l.classes != classes) { set_error("heads disagree on classes"); return YL_ERR_UNSUPPORTED; }
        HeadDesc &h = heads[nh++];
        h.out = l.d_output; h.type = l.type; h.w = l.w; h.h = l.h; h.n = l.n; h.classes = l.classes; h.outputs = l.outputs;
        h.tree_parent = l.d_tree;
        for (int k = 0; k < l.n; ++k) {
            const int an = (l.type == YL_YOLO) ? l.mask[k] : k;
            h.anchors_w[k] = l.anchors[2 * an];
            h.anchors_h[k] = l.anchors[2 * an + 1];
        }
    }
    if (nh == 0) { set_error("network has no detection head"); return YL_ERR_STATE; }
    *nh_out = nh;
    *classes_out = classes;
    return YL_OK;
}

int yl_network_compact_detections(yl_network *net, float thresh, int cap, float *records_dev, int *counts_dev)
{
    if (!net || !records_dev || !counts_dev || cap <= 0) { set_error("bad argument"); return YL_ERR_ARG; }
    Network &n = net->net;
    if (!n.on_device) { set_error("network not on device"); return YL_ERR_STATE; }
    YL_HIP(hipSetDevice(n.device));
    HeadDesc heads[4];
    int nh = 0, classes = 0;
    const int rc = collect_heads(n, heads, &nh, &classes);
    if (rc != YL_OK) return rc;
    YL_LAUNCH(launch_compact(heads, nh, n.batch, n.w, n.h, thresh, cap, 6 + classes, records_dev, counts_dev, n.stream),
              "compact");
    return YL_OK;
}

int yl_network_detect_batch(yl_network *net, const int *img_w, const int *img_h, float thresh, int relative,
                            int letter, float nms, int cap, float *records_dev, int *counts_dev)
{
    if (!net || !records_dev || !counts_dev || cap <= 0) { set_error("bad argument"); return YL_ERR_ARG; }
    if (cap > NMS_MAX_CAP) { set_error("cap exceeds YL_DETECT_MAX_CAP"); return YL_ERR_ARG; }
    if ((img_w == nullptr) != (img_h == nullptr)) { set_error("img_w and img_h must both be given or both be NULL"); return YL_ERR_ARG; }
    if (!img_w && (!relative || letter)) { set_error("absolute or letterboxed boxes need the source image sizes"); return YL_ERR_ARG; }
    Network &n = net->net;
    if (!n.on_device) { set_error("network not on device"); return YL_ERR_STATE; }
    YL_HIP(hipSetDevice(n.device));
    HeadDesc heads[4];
    int nh = 0, classes = 0;
    const int rc = collect_heads(n, heads, &nh, &classes);
    if (rc != YL_OK) return rc;
    const int B = n.batch;
    const size_t need = sizeof(float) * (size_t)B * cap * (6 + classes);
    if (n.det_scratch_bytes < need) {
        YL_HIP(hipStreamSynchronize((hipStream_t)n.stream));
        if (n.d_det_scratch) (void)hipFree(n.d_det_scratch);
        n.d_det_scratch = nullptr; n.det_scratch_bytes = 0;
        YL_HIP(hipMalloc((void **)&n.d_det_scratch, need));
        n.det_scratch_bytes = need;
    }
    if (!n.d_det_counts) YL_HIP(hipMalloc((void **)&n.d_det_counts, sizeof(int) * 2 * (size_t)B));
    const size_t meta_need = sizeof(unsigned) * (size_t)B * (1 + (classes + 31) / 32);
    if (n.det_meta_bytes < meta_need) {
        YL_HIP(hipStreamSynchronize((hipStream_t)n.stream));
        if (n.d_det_meta) (void)hipFree(n.d_det_meta);
        n.d_det_meta = nullptr; n.det_meta_bytes = 0;
        YL_HIP(hipMalloc((void **)&n.d_det_meta, meta_need));
        n.det_meta_bytes = meta_need;
    }
    ImgDims dims;
    dims.mode = 0;
    dims.wh[0] = 0;
    if (img_w) {
        bool uniform = true;
        for (int b = 0; b < B; ++b) {
            if (img_w[b] <= 0 || img_h[b] <= 0 || img_w[b] > 65535 || img_h[b] > 65535) {
                set_error("image sizes must be in 1..65535"); return YL_ERR_ARG;
            }
            uniform = uniform && img_w[b] == img_w[0] && img_h[b] == img_h[0];
        }
        if (!uniform && B > NMS_MAX_DIMS) { set_error("per-image sizes are limited to batch <= 256"); return YL_ERR_UNSUPPORTED; }
        dims.mode = uniform ? 1 : 2;
        for (int b = 0; b < (uniform ? 1 : B); ++b) dims.wh[b] = (uint32_t)img_w[b] | ((uint32_t)img_h[b] << 16);
    }
    YL_LAUNCH(launch_compact(heads, nh, B, n.w, n.h, thresh, cap, 6 + classes, n.d_det_scratch, n.d_det_counts, n.stream),
              "compact");
    YL_LAUNCH(launch_nms(n.d_det_scratch, n.d_det_counts, B, cap, classes, nms, n.w, n.h, dims, relative, letter,
                         records_dev, counts_dev, n.d_det_meta, n.nms_mode, n.stream), "nms");
    return YL_OK;
}

int yl_network_get_boxes_batch(yl_network *net, const int *img_w, const int *img_h, float thresh, int relative,
                               int letter, float nms, int cap, float *rows_host, int *counts_host)
{
    if (!net || !rows_host || !counts_host || cap <= 0) { set_error("bad argument"); return YL_ERR_ARG; }
    Network &n = net->net;
    if (!n.on_device) { set_error("network not on device"); return YL_ERR_STATE; }
    n.det_cache_valid = false;                       // the pinned block is about to be overwritten
    const int rc = detect_to_pinned(net, img_w, img_h, thresh, relative, letter, nms, cap);
    if (rc != YL_OK) return rc;
    const size_t row = (size_t)(6 + n.layers.back().classes);
    for (int b = 0; b < n.batch; ++b) {
        counts_host[b] = n.det_counts[b];
        const size_t fl = n.det_row_off[(size_t)b + 1] - n.det_row_off[b];
        if (fl) memcpy(rows_host + (size_t)b * cap * row, n.h_det_rows + n.det_row_off[b], sizeof(float) * fl);
    }
    return YL_OK;
}

}  // extern "C"
