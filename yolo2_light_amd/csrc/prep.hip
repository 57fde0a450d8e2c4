// prep.hip -- the reference's one-time model preparation passes on the GPU.
//
// main.c:167-171 runs, on one host core, yolov2_fuse_conv_batchnorm (src/additionally.c:67-110),
// calculate_binary_weights (src/additionally.c:306-318 -> binarize_weights :113-126) and, under -quantized,
// quantinization_and_get_multipliers (src/yolov2_forward_network_quantized.c:1402-1446 -> get_multiplier :1371).
// host_prep.cpp restates them (pinned against the reference library in tests/test_host_prep.py); here the same
// arithmetic runs on the device with explicitly rounded operations, so the folded weights / biases, mean_arr,
// weights_int8 and multipliers are BIT-IDENTICAL to the host passes (tests/test_gpu_prep.py).  The results come
// back to the host model because the kernel-layout packers (k-major panels, Winograd U in double, int8 / bf16
// units, sign words) run there.
#include <hip/hip_runtime.h>
#include <cmath>
#include <vector>

#include "yl_internal.h"

namespace yl {

namespace {

// weights[f][i] = weights[f][i] * scales[f] / (sqrtf(variance[f]) + .000001f)       (additionally.c:88-96)
// biases[f]     = biases[f] - scales[f] * mean[f] / (sqrtf(variance[f]) + .000001f)  (additionally.c:82-83)
__global__ __launch_bounds__(256) void fold_bn_kernel(float *__restrict__ w, float *__restrict__ bias,
                                                      const float *__restrict__ scales, const float *__restrict__ mean,
                                                      const float *__restrict__ var, int n, size_t k)
{
    const size_t total = (size_t)n * k;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(idx / k);
        const float den = __fadd_rn(sqrtf(var[f]), .000001f);      // sqrtf: correctly rounded (hipcc default); __fsqrt_rn is the NATIVE approximation in this toolchain
        w[idx] = __fdiv_rn(__fmul_rn(w[idx], scales[f]), den);
        if (idx - (size_t)f * k == 0) bias[f] = __fsub_rn(bias[f], __fdiv_rn(__fmul_rn(scales[f], mean[f]), den));
    }
}

// mean_arr[f] = fabs(mean(|w[f][:]|)) with the reference's accumulation: `float mean; mean += fabs(w)` is a double
// add rounded to float at every step, in index order (binarize_weights, additionally.c:113-126) -- sequential per
// filter by construction, one lane per filter
__global__ __launch_bounds__(64) void xnor_mean_kernel(const float *__restrict__ w, float *__restrict__ mean_arr, int n, int k)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    const float *wf = w + (size_t)f * k;
    float mean = 0.f;
    for (int i = 0; i < k; ++i) mean = (float)__dadd_rn((double)mean, fabs((double)wf[i]));
    mean = __fdiv_rn(mean, (float)k);
    mean_arr[f] = fabsf(mean);
}

// get_multiplier's histogram: count[j] = #{ w : 2^(j-16) <= w < 2^(j-15) }, j = 0..31 (only positive weights land)
__global__ __launch_bounds__(256) void range_hist_kernel(const float *__restrict__ w, size_t total, int *__restrict__ count)
{
    __shared__ int local[32];
    if (threadIdx.x < 32) local[threadIdx.x] = 0;
    __syncthreads();
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const float v = w[idx];
        float cur = 1.F / 65536;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            if (cur <= v && v < cur * 2) atomicAdd(&local[j], 1);
            cur *= 2;
        }
    }
    __syncthreads();
    if (threadIdx.x < 32 && local[threadIdx.x]) atomicAdd(&count[threadIdx.x], local[threadIdx.x]);
}

// weights_int8[i] = max_abs((int)(w[i] * mult), 127)        (quantized.c:1429-1436: float -> int truncates)
__global__ __launch_bounds__(256) void quantize_weights_kernel(const float *__restrict__ w, int8_t *__restrict__ q, size_t total, float mult)
{
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)__fmul_rn(w[idx], mult);
        q[idx] = (int8_t)(abs(v) > 127 ? (v > 0 ? 127 : -127) : v);
    }
}

inline unsigned blocks_for(size_t total) { size_t g = (total + 255) / 256; if (g > 4096) g = 4096; return (unsigned)(g ? g : 1); }

#define PREP_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(std::string("prepare_on_device: ") + hipGetErrorString(e_)); rc = YL_ERR_DEVICE; goto done; } } while (0)
// staging.hip sets the error text; its D2H wants the producer finished (these kernels run on the null stream,
// the staging stream is non-blocking, so nothing orders them implicitly)
#define PREP_RC(x) do { int rc_ = (x); if (rc_ != YL_OK) { rc = rc_; goto done; } } while (0)
#define PREP_D2H(dst, src, bytes) do { PREP_HIP(hipStreamSynchronize(0)); PREP_RC(stage_d2h(device, dst, src, bytes)); } while (0)

}  // namespace

int prepare_on_device(Network &net, int device)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { set_error("no HIP device visible (libyolo2hip has no CPU fallback)"); return YL_ERR_DEVICE; }
    if (device < 0 || device >= count) { set_error("device index out of range"); return YL_ERR_ARG; }
    if (net.on_device) { set_error("prepare_on_device must precede to_device"); return YL_ERR_STATE; }
    int rc = YL_OK;
    float *d_w = nullptr, *d_small = nullptr;
    int8_t *d_q = nullptr;
    int *d_cnt = nullptr;
    size_t cap_w = 0, cap_n = 0;
    int counter = 0;
    if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice failed"); return YL_ERR_DEVICE; }
    for (Layer &l : net.layers) {
        if (l.type != YL_CONVOLUTIONAL) continue;
        const size_t k = (size_t)l.size * l.size * l.c, total = k * l.n;
        if (total > cap_w || (size_t)l.n > cap_n) {
            if (d_w) (void)hipFree(d_w);
            if (d_small) (void)hipFree(d_small);
            if (d_q) (void)hipFree(d_q);
            d_w = nullptr; d_small = nullptr; d_q = nullptr;
            cap_w = total > cap_w ? total : cap_w;
            cap_n = (size_t)l.n > cap_n ? (size_t)l.n : cap_n;
            PREP_HIP(hipMalloc((void **)&d_w, cap_w * sizeof(float)));
            PREP_HIP(hipMalloc((void **)&d_small, 5 * cap_n * sizeof(float)));      // bias, scales, mean, variance, mean_arr
            PREP_HIP(hipMalloc((void **)&d_q, cap_w));
        }
        if (!d_cnt) PREP_HIP(hipMalloc((void **)&d_cnt, 32 * sizeof(int)));
        float *d_bias = d_small, *d_scales = d_small + cap_n, *d_mean = d_small + 2 * cap_n, *d_var = d_small + 3 * cap_n,
              *d_marr = d_small + 4 * cap_n;
        PREP_RC(stage_h2d(device, d_w, l.weights.data(), total * sizeof(float)));
        if (l.batch_normalize) {
            PREP_RC(stage_h2d(device, d_bias, l.biases.data(), l.n * sizeof(float)));
            PREP_RC(stage_h2d(device, d_scales, l.scales.data(), l.n * sizeof(float)));
            PREP_RC(stage_h2d(device, d_mean, l.rolling_mean.data(), l.n * sizeof(float)));
            PREP_RC(stage_h2d(device, d_var, l.rolling_variance.data(), l.n * sizeof(float)));
            hipLaunchKernelGGL(fold_bn_kernel, dim3(blocks_for(total)), dim3(256), 0, 0, d_w, d_bias, d_scales, d_mean, d_var, l.n, k);
            PREP_HIP(hipGetLastError());
            PREP_D2H(l.weights.data(), d_w, total * sizeof(float));
            PREP_D2H(l.biases.data(), d_bias, l.n * sizeof(float));
            l.batch_normalize = 0;
        }
        if (l.xnor) {
            hipLaunchKernelGGL(xnor_mean_kernel, dim3((l.n + 63) / 64), dim3(64), 0, 0, d_w, d_marr, l.n, (int)k);
            PREP_HIP(hipGetLastError());
            l.mean_arr.assign(l.n, 0.f);
            PREP_D2H(l.mean_arr.data(), d_marr, l.n * sizeof(float));
            l.xnor_ready = true;
        }
        if (net.quantized) {
            int h_cnt[32];
            PREP_HIP(hipMemset(d_cnt, 0, 32 * sizeof(int)));
            hipLaunchKernelGGL(range_hist_kernel, dim3(blocks_for(total)), dim3(256), 0, 0, d_w, total, d_cnt);
            PREP_HIP(hipGetLastError());
            PREP_D2H(h_cnt, d_cnt, sizeof(h_cnt));
            l.weights_quant_multipler = multiplier_from_range_counts(h_cnt, 8) / 4;
            hipLaunchKernelGGL(quantize_weights_kernel, dim3(blocks_for(total)), dim3(256), 0, 0, d_w, d_q, total, l.weights_quant_multipler);
            PREP_HIP(hipGetLastError());
            l.weights_int8.assign(total, 0);
            PREP_D2H(l.weights_int8.data(), d_q, total);
            // the calibration index counts EVERY conv layer
            l.input_quant_multipler = (counter < (int)net.input_calibration.size()) ? net.input_calibration[counter] : 40.f;
            l.quant_ready = true;
        }
        ++counter;
    }
done:
    if (d_w) (void)hipFree(d_w);
    if (d_small) (void)hipFree(d_small);
    if (d_q) (void)hipFree(d_q);
    if (d_cnt) (void)hipFree(d_cnt);
    return rc;
}

}  // namespace yl
