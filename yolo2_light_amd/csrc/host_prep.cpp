// host_prep.cpp -- one-time model preparation passes on the host.
//
// These define what "fused batchnorm", "binary weights" and "INT8 weights +
// multipliers" mean numerically, so they reproduce the reference's scalar C
// arithmetic operation-for-operation (same types, same order):
//   load_weights_upto_cpu / load_convolutional_weights_cpu  src/additionally.c:3491 / 3459
//   yolov2_fuse_conv_batchnorm                              src/additionally.c:67-109
//   binarize_weights / get_mean_array                       src/additionally.c:113-126 / 188-194
//   quantinization_and_get_multipliers                      src/yolov2_forward_network_quantized.c:1402-1494
//   get_distribution / get_multiplier / max_abs             src/yolov2_forward_network_quantized.c:35-87 / 23-27
// The device-side layouts (k-major FP32 panels, channel-fastest INT8, 64-bit
// sign words) are built from these host arrays in runtime.hip.
#include "yl_internal.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace yl {

// cutoff < 0: strict (every conv layer must be in the file).  cutoff >= 0: load_weights_upto_cpu's contract
// (src/additionally.c:3491-3526): only layers [0, cutoff) are read and a file that ends early is not an error --
// the reference never checks fread, the remaining parameters keep their initial values (backbone-only files).
int load_weights_file(Network &net, const char *path, int cutoff) {
    FILE *fp = fopen(path, "rb");
    if (!fp) { set_error(std::string("cannot open weights file: ") + path); return YL_ERR_IO; }
    int32_t major = 0, minor = 0, revision = 0;
    bool ok = fread(&major, 4, 1, fp) == 1 && fread(&minor, 4, 1, fp) == 1 && fread(&revision, 4, 1, fp) == 1;
    if (ok) {
        if ((major * 10 + minor) >= 2) { uint64_t seen; ok = fread(&seen, 8, 1, fp) == 1; }
        else { int32_t seen; ok = fread(&seen, 4, 1, fp) == 1; }
    }
    if (!ok) { fclose(fp); set_error("weights file: truncated header"); return YL_ERR_IO; }
    for (size_t i = 0; i < net.layers.size() && (cutoff < 0 || (int)i < cutoff); ++i) {
        Layer &l = net.layers[i];
        if (l.type != YL_CONVOLUTIONAL) continue;
        const size_t num = (size_t)l.n * l.c * l.size * l.size;
        bool good = fread(l.biases.data(), 4, l.n, fp) == (size_t)l.n;
        if (good && l.batch_normalize) {
            good = fread(l.scales.data(), 4, l.n, fp) == (size_t)l.n &&
                   fread(l.rolling_mean.data(), 4, l.n, fp) == (size_t)l.n &&
                   fread(l.rolling_variance.data(), 4, l.n, fp) == (size_t)l.n;
        }
        if (good) good = fread(l.weights.data(), 4, num, fp) == num;
        if (!good && cutoff >= 0) break;          // the reference's tolerant form was asked for
        if (!good) {
            // the reference ignores short reads (it never checks fread); the strict entry point refuses:
            // a silently half-loaded model can only produce wrong detections.
            fclose(fp);
            char msg[128];
            snprintf(msg, sizeof(msg), "weights file ends inside conv layer %zu", i);
            set_error(msg);
            return YL_ERR_IO;
        }
    }
    fclose(fp);
    net.weights_loaded = true;
    return YL_OK;
}

void fuse_conv_batchnorm(Network &net) {
    for (Layer &l : net.layers) {
        if (l.type != YL_CONVOLUTIONAL || !l.batch_normalize) continue;
        const size_t filter_size = (size_t)l.size * l.size * l.c;
        for (int f = 0; f < l.n; ++f) {
            // epsilon is added OUTSIDE the sqrt
            l.biases[f] = l.biases[f] - l.scales[f] * l.rolling_mean[f] / (sqrtf(l.rolling_variance[f]) + .000001f);
            for (size_t i = 0; i < filter_size; ++i) {
                const size_t wi = f * filter_size + i;
                l.weights[wi] = l.weights[wi] * l.scales[f] / (sqrtf(l.rolling_variance[f]) + .000001f);
            }
        }
        l.batch_normalize = 0;
    }
}

void calculate_binary_weights(Network &net) {
    for (Layer &l : net.layers) {
        if (l.type != YL_CONVOLUTIONAL || !l.xnor) continue;
        const size_t k = (size_t)l.size * l.size * l.c;
        l.mean_arr.assign(l.n, 0.f);
        for (int f = 0; f < l.n; ++f) {
            // `float mean; mean += fabs(w)` : double add rounded to float each step
            float mean = 0;
            for (size_t i = 0; i < k; ++i) mean = (float)((double)mean + fabs((double)l.weights[f * k + i]));
            mean = mean / (float)k;          // `mean / size` with int size -> float division
            // mean_arr[f] = fabs(binary_weights[f*k]) = |+-mean|
            l.mean_arr[f] = (float)fabs((double)mean);
        }
        l.xnor_ready = true;
    }
}

// second half of get_multiplier (src/yolov2_forward_network_quantized.c): the window of `bits_length` power-of-two
// ranges holding the most weights decides the multiplier; count[j] = #{w : 2^(j-16) <= w < 2^(j-15)}
float multiplier_from_range_counts(const int *count, int bits_length) {
    const int number_of_ranges = 32;
    const float start_range = 1.F / 65536;
    int max_count_range = 0, index_max_count = 0;
    for (int j = 0; j < number_of_ranges; ++j) {
        int counter = 0;
        for (int i = j; i < (j + bits_length) && i < number_of_ranges; ++i) counter += count[i];
        if (max_count_range < counter) { max_count_range = counter; index_max_count = j; }
    }
    return 1 / (start_range * powf(2.f, (float)index_max_count));
}

namespace {

// get_multiplier(arr, size, bits_length) -- only strictly positive values land in a bin
float weights_multiplier(const float *arr, size_t n, int bits_length) {
    const int number_of_ranges = 32;
    const float start_range = 1.F / 65536;
    int count[32];
    memset(count, 0, sizeof(count));
    for (size_t i = 0; i < n; ++i) {
        const float w = arr[i];
        float cur_range = start_range;
        for (int j = 0; j < number_of_ranges; ++j) {
            if (fabs((double)cur_range) <= (double)w && (double)w < fabs((double)(cur_range * 2))) count[j]++;
            cur_range *= 2;
        }
    }
    return multiplier_from_range_counts(count, bits_length);
}

inline int max_abs_int(int src, int max_val) {
    if (abs(src) > abs(max_val)) src = (src > 0) ? max_val : -max_val;
    return src;
}

}  // namespace

void quantize_network(Network &net) {
    int counter = 0;
    for (Layer &l : net.layers) {
        if (l.type != YL_CONVOLUTIONAL) continue;
        const size_t weights_size = (size_t)l.size * l.size * l.c * l.n;
        l.weights_quant_multipler = weights_multiplier(l.weights.data(), weights_size, 8) / 4;
        l.weights_int8.assign(weights_size, 0);
        for (size_t i = 0; i < weights_size; ++i) {
            const float w = l.weights[i] * l.weights_quant_multipler;
            // max_abs(int src, ...) receives the float -> C truncation toward zero
            l.weights_int8[i] = (int8_t)max_abs_int((int)w, 127);
        }
        // the calibration index counts EVERY conv layer
        l.input_quant_multipler = (counter < (int)net.input_calibration.size()) ? net.input_calibration[counter] : 40.f;
        ++counter;
        l.quant_ready = true;
    }
}

void select_conv_modes(Network &net) {
    for (size_t i = 0; i < net.layers.size(); ++i) {
        Layer &l = net.layers[i];
        if (l.type != YL_CONVOLUTIONAL) continue;
        // CPU rule, yolov2_forward_network_q: `i >= 1 && l.activation != LINEAR` (quantized.c:1036);
        // GPU rule, forward_network_gpu_cudnn_quantized: the parser's `l.quantized`
        // (src/additionally.c:3557-3559, 3996-4004) -- yl_network_set_quant_rule
        const bool q = net.quant_rule == YL_QUANT_RULE_GPU ? (l.gpu_quantized != 0)
                                                           : (i >= 1 && l.activation != YL_LINEAR);
        if (net.quantized && q) l.conv_mode = CONV_INT8;
        // forward_convolutional_layer_cpu: `l.xnor && l.align_bit_weights && stride==1 && pad==1`
        // (yolov2_forward_network.c:116)
        else if (l.xnor && l.stride == 1 && l.pad == 1 && l.size == 3) l.conv_mode = CONV_XNOR;
        // opt-in (yl_network_set_precision): FP32 convolutions with whole 8-channel groups take bf16 operands
        else if (net.precision == YL_PRECISION_BF16 && !l.xnor && (l.c % 8) == 0 && l.size <= 5) l.conv_mode = CONV_BF16;
        else l.conv_mode = CONV_F32;
    }
}

}  // namespace yl
