// layers.hip -- K4..K10: the small HBM-bound layers as simple coalesced gfx950
// kernels (one output element per lane, consecutive lanes -> consecutive
// addresses, grid-stride over a capped grid).
//
//   maxpool   forward_maxpool_layer_avx            src/additionally.c:1448-1482
//   shortcut  forward_shortcut_layer_cpu/shortcut_cpu  src/yolov2_forward_network.c:444 / 410
//   upsample  forward_upsample_layer_cpu/upsample_cpu  src/yolov2_forward_network.c:398 / 380
//   route     forward_route_layer_cpu              src/yolov2_forward_network.c:318
//   yolo      forward_yolo_layer_cpu               src/yolov2_forward_network.c:453
//   region    forward_region_layer_cpu/softmax_cpu src/yolov2_forward_network.c:511 / 476
//   reorg     forward_reorg_layer_cpu              src/yolov2_forward_network.c:337
//   compact   yolo_num_detections/get_yolo_detections/get_yolo_box  src/additionally.c:4207/4328/4317
//             get_region_boxes_cpu/get_region_box_cpu               src/yolov2_forward_network.c:664/653
#include <hip/hip_runtime.h>
#include <cfloat>

#include "kernels.h"
#include "activations.h"
#include "activations.h"
#include "../../include/yolo2_hip.h"

namespace yl {

static inline unsigned grid_for(size_t n, int block = 256)
{
    size_t g = (n + block - 1) / block;
    const size_t cap = 256 * 16;          // 256 CUs x 16 workgroups, grid-stride beyond
    if (g > cap) g = cap;
    if (g == 0) g = 1;
    return (unsigned)g;
}

// ---------------------------------------------------------------- maxpool
__global__ __launch_bounds__(256) void maxpool_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                      size_t total, int C, int H, int W, int OH, int OW,
                                                      int size, int stride, int off)
{
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx % OW);
        size_t t = idx / OW;
        const int i = (int)(t % OH);
        t /= OH;                                   // t = k + C*b
        const float *src = in + t * (size_t)H * W;
        float mx = -FLT_MAX;
        for (int n = 0; n < size; ++n) {
            const int cur_h = off + i * stride + n;
            for (int m = 0; m < size; ++m) {
                const int cur_w = off + j * stride + m;
                const bool valid = cur_h >= 0 && cur_h < H && cur_w >= 0 && cur_w < W;
                const float val = valid ? src[(size_t)cur_h * W + cur_w] : -FLT_MAX;
                mx = (val > mx) ? val : mx;
            }
        }
        out[idx] = mx;
    }
}

int launch_maxpool(const float *in, float *out, int B, int C, int H, int W, int OH, int OW,
                   int size, int stride, int pad, void *stream)
{
    const size_t total = (size_t)B * C * OH * OW;
    const int off = -pad / 2;                      // C integer division, SURVEY A4
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       in, out, total, C, H, W, OH, OW, size, stride, off);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------- shortcut
__device__ __forceinline__ float act_apply(float v, int act) { return yl_act_epilogue(v, act); }

__global__ __launch_bounds__(256) void shortcut_same_kernel(const float4 *__restrict__ in, const float4 *__restrict__ add,
                                                            float4 *__restrict__ out, size_t n4, int act)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 a = in[i], b = add[i];
        float4 r;
        r.x = act_apply(a.x + b.x, act); r.y = act_apply(a.y + b.y, act);
        r.z = act_apply(a.z + b.z, act); r.w = act_apply(a.w + b.w, act);
        out[i] = r;
    }
}

__global__ __launch_bounds__(256) void shortcut_general_kernel(const float *__restrict__ in, const float *__restrict__ add,
                                                               float *__restrict__ out, size_t total,
                                                               int w1, int h1, int c1, int w2, int h2, int c2,
                                                               int stride, int sample, int minw, int minh, int minc, int act)
{
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % w2);
        size_t t = idx / w2;
        const int y = (int)(t % h2);
        t /= h2;
        const int k = (int)(t % c2);
        const size_t b = t / c2;
        float v = in[idx];
        if (k < minc && (x % sample) == 0 && (y % sample) == 0) {
            const int i = x / sample, j = y / sample;
            if (i < minw && j < minh)
                v = v + add[(size_t)i * stride + (size_t)w1 * ((size_t)j * stride + (size_t)h1 * (k + (size_t)c1 * b))];
        }
        out[idx] = act_apply(v, act);
    }
}

int launch_shortcut(const float *in, const float *add, float *out, int B, int w1, int h1, int c1,
                    int w2, int h2, int c2, int act, void *stream)
{
    const size_t total = (size_t)B * w2 * h2 * c2;
    if (w1 == w2 && h1 == h2 && c1 == c2 && (total % 4) == 0) {
        hipLaunchKernelGGL(shortcut_same_kernel, dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream,
                           (const float4 *)in, (const float4 *)add, (float4 *)out, total / 4, act);
    } else {
        int stride = w1 / w2, sample = w2 / w1;
        if (stride < 1) stride = 1;
        if (sample < 1) sample = 1;
        const int minw = w1 < w2 ? w1 : w2, minh = h1 < h2 ? h1 : h2, minc = c1 < c2 ? c1 : c2;
        hipLaunchKernelGGL(shortcut_general_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           in, add, out, total, w1, h1, c1, w2, h2, c2, stride, sample, minw, minh, minc, act);
    }
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------- upsample
__global__ __launch_bounds__(256) void upsample_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                       size_t total, int H, int W, int stride, float scale)
{
    const int OW = W * stride, OH = H * stride;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % OW);
        size_t t = idx / OW;
        const int j = (int)(t % OH);
        t /= OH;                                   // t = k + C*b
        out[idx] = __fmul_rn(scale, in[t * (size_t)H * W + (size_t)(j / stride) * W + i / stride]);
    }
}

int launch_upsample(const float *in, float *out, int B, int C, int H, int W, int stride, float scale, void *stream)
{
    const size_t total = (size_t)B * C * H * W * stride * stride;
    hipLaunchKernelGGL(upsample_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       in, out, total, H, W, stride, scale);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------- route (row copies)
__global__ __launch_bounds__(256) void copy_rows_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                        int rows, int row_elems, size_t src_stride, size_t dst_stride)
{
    const size_t total = (size_t)rows * row_elems;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const size_t r = idx / row_elems;
        const size_t e = idx - r * row_elems;
        dst[r * dst_stride + e] = src[r * src_stride + e];
    }
}

__global__ __launch_bounds__(256) void copy_rows4_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst,
                                                         int rows, int row_elems4, size_t src_stride4, size_t dst_stride4)
{
    const size_t total = (size_t)rows * row_elems4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const size_t r = idx / row_elems4;
        const size_t e = idx - r * row_elems4;
        dst[r * dst_stride4 + e] = src[r * src_stride4 + e];
    }
}

int launch_copy_rows(const float *src, float *dst, int rows, int row_elems, size_t src_stride, size_t dst_stride, void *stream)
{
    const bool v4 = (row_elems % 4 == 0) && (src_stride % 4 == 0) && (dst_stride % 4 == 0) &&
                    (((uintptr_t)src | (uintptr_t)dst) % 16 == 0);
    if (v4) {
        const size_t total = (size_t)rows * (row_elems / 4);
        hipLaunchKernelGGL(copy_rows4_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const float4 *)src, (float4 *)dst, rows, row_elems / 4, src_stride / 4, dst_stride / 4);
    } else {
        const size_t total = (size_t)rows * row_elems;
        hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           src, dst, rows, row_elems, src_stride, dst_stride);
    }
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------- yolo
// copy + logistic on entries {0,1} (x,y) and {4..4+classes} (obj, classes) of every anchor;
// layout stays [B][n*(5+classes)][h][w]
__global__ __launch_bounds__(256) void yolo_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                   size_t total, int per_anchor, int wh)
{
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int entry = (int)((idx / wh) % per_anchor);
        const float x = in[idx];
        // logistic_activate: 1./(1. + exp(-x)) in double (src/additionally.h:84)
        out[idx] = (entry == 2 || entry == 3) ? x : (float)(1. / (1. + exp((double)(-x))));
    }
}

int launch_yolo(const float *in, float *out, int B, int n, int classes, int wh, void *stream)
{
    const int per_anchor = 4 + classes + 1;
    const size_t total = (size_t)B * n * per_anchor * wh;
    hipLaunchKernelGGL(yolo_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       in, out, total, per_anchor, wh);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------- region
// one lane per (b, cell, anchor): CHW -> HWC flatten, logistic(obj) in float, softmax over classes
__global__ __launch_bounds__(256) void region_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                     size_t total, int n, int classes, int coords, int wh, int softmax,
                                                     const int *__restrict__ tree_group_size, int tree_groups)
{
    const int size = coords + classes + 1;
    const int layers = size * n;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int a = (int)(idx % n);                 // anchor
        size_t t = idx / n;
        const int cell = (int)(t % wh);
        const size_t b = t / wh;
        const float *src = in + b * (size_t)layers * wh + (size_t)(a * size) * wh + cell;   // channel c at src[c*wh]
        float *dst = out + b * (size_t)layers * wh + (size_t)cell * layers + (size_t)a * size;
        for (int c = 0; c < coords; ++c) dst[c] = src[(size_t)c * wh];
        const float o = src[(size_t)coords * wh];
        dst[coords] = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-o)));
        if (tree_group_size) {
            // softmax_tree (src/yolov2_forward_network.c:494-507, :556-562): softmax_cpu over every group of the
            // class vector; takes precedence over `softmax=1` like the reference's if / else-if
            int c0 = 0;
            for (int g = 0; g < tree_groups; ++g) {
                const int gs = tree_group_size[g];
                float largest = -FLT_MAX;
                for (int c = c0; c < c0 + gs; ++c) {
                    const float v = src[(size_t)(coords + 1 + c) * wh];
                    if (v > largest) largest = v;
                }
                float sum = 0.f;
                for (int c = c0; c < c0 + gs; ++c) {
                    const float e = expf(__fsub_rn(src[(size_t)(coords + 1 + c) * wh], largest));
                    sum = __fadd_rn(sum, e);
                    dst[coords + 1 + c] = e;
                }
                for (int c = c0; c < c0 + gs; ++c) dst[coords + 1 + c] = __fdiv_rn(dst[coords + 1 + c], sum);
                c0 += gs;
            }
        } else if (softmax) {
            float largest = -FLT_MAX;
            for (int c = 0; c < classes; ++c) {
                const float v = src[(size_t)(coords + 1 + c) * wh];
                if (v > largest) largest = v;
            }
            float sum = 0.f;
            for (int c = 0; c < classes; ++c) {
                // expf(input[i]/temp - largest/temp), temp = 1
                const float e = expf(__fsub_rn(src[(size_t)(coords + 1 + c) * wh], largest));
                sum = __fadd_rn(sum, e);
                dst[coords + 1 + c] = e;
            }
            for (int c = 0; c < classes; ++c) dst[coords + 1 + c] = __fdiv_rn(dst[coords + 1 + c], sum);
        } else {
            for (int c = 0; c < classes; ++c) dst[coords + 1 + c] = src[(size_t)(coords + 1 + c) * wh];
        }
    }
}

int launch_region(const float *in, float *out, int B, int n, int classes, int coords, int wh, int softmax, void *stream,
                  const int *tree_group_size, int tree_groups)
{
    const size_t total = (size_t)B * wh * n;
    hipLaunchKernelGGL(region_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       in, out, total, n, classes, coords, wh, softmax, tree_group_size, tree_groups);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------- reorg
__global__ __launch_bounds__(256) void reorg_kernel(const float *__restrict__ x, float *__restrict__ out,
                                                    size_t total, int out_c, int out_h, int out_w, int stride)
{
    const int in_c = out_c / (stride * stride);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % out_w);
        size_t t = idx / out_w;
        const int j = (int)(t % out_h);
        t /= out_h;
        const int k = (int)(t % out_c);
        const size_t b = t / out_c;
        const int c2 = k % in_c;
        const int offset = k / in_c;
        const int w2 = i * stride + offset % stride;
        const int h2 = j * stride + offset / stride;
        out[idx] = x[w2 + (size_t)out_w * stride * (h2 + (size_t)out_h * stride * (c2 + (size_t)in_c * b))];
    }
}

int launch_reorg(const float *in, float *out, int B, int out_c, int out_h, int out_w, int stride, void *stream)
{
    const size_t total = (size_t)B * out_c * out_h * out_w;
    hipLaunchKernelGGL(reorg_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       in, out, total, out_c, out_h, out_w, stride);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------- binarize (xnor FP32 fallback)
__global__ __launch_bounds__(256) void binarize_kernel(const float *__restrict__ in, float *__restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (in[i] > 0.f) ? 1.f : -1.f;
}

int launch_binarize(const float *in, float *out, size_t n, void *stream)
{
    hipLaunchKernelGGL(binarize_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, in, out, n);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------- activate_array (the rare activations)
// activate_array_cpu_custom (src/additionally.c:1436-1445) as its own pass, like in the reference: a convolution whose
// activation is neither LINEAR nor LEAKY runs its kernel with a linear epilogue and this kernel finishes the tensor
// in place (the same float goes into activate(), so the result is what a fused epilogue would give).  Keeping the
// 13-way switch with its double exp() out of the MFMA kernels keeps their register allocation where it is.
__global__ __launch_bounds__(256) void activate_kernel(float *__restrict__ x, size_t n, int act)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        x[i] = yl_activate(x[i], act);
}

int launch_activate(float *x, size_t n, int act, void *stream)
{
    hipLaunchKernelGGL(activate_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, n, act);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------- K13: calibration histogram
// H[b][bin] = #{ x in image b : lround(fabs(x) / bin_width) == bin }, saturated at max_bin-1: the
// data-parallel half of entropy_calibration (src/yolov2_forward_network_quantized.c:1306-1313).
// fabs and the division are done in double like the reference (`fabs(float)` promotes); |x|/bin_width
// is exact for the power-of-two bin widths the tool uses, and floor(v + 0.5) == lround(v) for v >= 0.
// One workgroup owns a private LDS histogram (max_bin <= 4096 -> 16 KB) and flushes it with one
// global atomic per non-empty bin: integer counts, order-independent, exact.
__global__ __launch_bounds__(256) void hist_abs_kernel(const float *__restrict__ x, size_t per_image, int max_bin,
                                                       double bin_width, unsigned *__restrict__ hist)
{
    __shared__ unsigned lh[4096];
    for (int i = threadIdx.x; i < max_bin; i += 256) lh[i] = 0;
    __syncthreads();
    const int b = blockIdx.y;
    const float *src = x + (size_t)b * per_image;
    const int last_bin = max_bin - 1;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per_image; i += (size_t)gridDim.x * 256) {
        const double v = __ddiv_rn(fabs((double)src[i]), bin_width);
        // NaN compares false and lands in the last bin like +inf; the reference's lround(NaN) is
        // unspecified, a calibration input with NaNs has no defined answer
        int bin = (v < (double)last_bin) ? (int)floor(v + 0.5) : last_bin;
        if (bin > last_bin) bin = last_bin;
        atomicAdd(&lh[bin], 1u);
    }
    __syncthreads();
    unsigned *dst = hist + (size_t)b * max_bin;
    for (int i = threadIdx.x; i < max_bin; i += 256)
        if (lh[i]) atomicAdd(&dst[i], lh[i]);
}

int launch_hist_abs(const float *x, size_t per_image, int batch, int max_bin, float bin_width, unsigned *hist, void *stream)
{
    if (max_bin > 4096 || max_bin < 129) return (int)hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(hist, 0, sizeof(unsigned) * (size_t)batch * max_bin, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    size_t blocks = (per_image + 256 * 16 - 1) / (256 * 16);
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(hist_abs_kernel, dim3((unsigned)blocks, (unsigned)batch), dim3(256), 0, (hipStream_t)stream,
                       x, per_image, max_bin, (double)bin_width, hist);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------- K10: detection compaction
// expf as the reference's host computes it.  get_region_box_cpu (src/yolov2_forward_network.c:653-661)
// calls libm's expf; glibc >= 2.27 evaluates it in double -- k = round(x*32/ln2), a 32-entry table
// of 2^(i/32), a cubic in the reduced argument, one rounding to float at the end -- and on an
// FMA-capable x86 host the build it dispatches to contracts every a*b+c.  Restated with explicit
// f64 fma's; verified against glibc 2.35's expf over every float in (-87, 88) (2 237 530 112
// inputs, 0 mismatches).  Device expf (<= 1 ulp) differed from it in ~6 % of the boxes.
__device__ __constant__ unsigned long long kExp2fTab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};

__device__ __forceinline__ float expf_host_libm(float x)
{
    if (!(x > -87.0f && x < 88.0f)) return expf(x);          // overflow/underflow/NaN tails: not boxes anyone keeps
    const double xd = (double)x;
    const double inv_ln2_n = 0x1.71547652b82fep+5;           // 32 / ln 2
    const double shift = 0x1.8p52;
    const double z = __dmul_rn(inv_ln2_n, xd);
    double kd = __dadd_rn(z, shift);
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd = __dsub_rn(kd, shift);
    const double r = __fma_rn(inv_ln2_n, xd, -kd);
    const unsigned long long t = kExp2fTab[ki & 31] + (ki << 47);
    const double sc = __longlong_as_double((long long)t);
    const double q = __fma_rn(0x1.c6af84b912394p-20, r, 0x1.ebfce50fac4f3p-13);
    const double r2 = __dmul_rn(r, r);
    double y = __fma_rn(0x1.62e42ff0c52d6p-6, r, 1.0);
    y = __fma_rn(q, r2, y);
    return (float)__dmul_rn(y, sc);
}

struct HeadsDev {
    HeadDesc h[4];
    int n_heads;
};

// Wave-aggregated slot allocation: all lanes of a wave that pass the threshold for the same
// image get consecutive slots from ONE atomicAdd on counts[image] (lanes of a wave almost
// always belong to one image; the loop handles the image-boundary wave).
__device__ __forceinline__ int wave_alloc_slot(int *counts, int b, bool pass)
{
    const unsigned long long lane_lt = (1ull << (threadIdx.x & 63)) - 1ull;
    unsigned long long todo = __ballot(pass);
    int slot = -1;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int bl = __shfl(b, leader);
        const unsigned long long grp = __ballot(pass && b == bl);
        int base = 0;
        if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(&counts[bl], __popcll(grp));
        base = __shfl(base, leader);
        if (pass && b == bl) slot = base + __popcll(grp & lane_lt);
        todo &= ~grp;
    }
    return slot;
}

// one lane per (image, head, cell, anchor).  Record row (stride = 6 + classes):
//   x y w h objectness scan_key prob[classes]
// boxes are relative to the network input (== get_network_boxes(net, 1, 1, thresh, ., 0, relative=1, ., 0)).
// Order inside an image is by atomic slot, i.e. NOT the reference's scan order; scan_key (exact in
// f32) = position of the (head, cell, anchor) triple in the reference's scan
// (get_yolo_detections: heads in layer order, cells row-major, anchors fastest), so a consumer can
// restore that order (nms_kernel in detect.hip does).
__global__ __launch_bounds__(256) void compact_kernel(HeadsDev hd, int B, int netw, int neth, float thresh,
                                                      int cap, int row_stride, float *__restrict__ records,
                                                      int *__restrict__ counts)
{
    int key_base = 0;
    for (int hi = 0; hi < hd.n_heads; key_base += hd.h[hi].w * hd.h[hi].h * hd.h[hi].n, ++hi) {
        const HeadDesc &h = hd.h[hi];
        const int wh = h.w * h.h;
        const size_t total = (size_t)B * wh * h.n;
        const size_t stride_all = (size_t)gridDim.x * blockDim.x;
        // every lane of a wave runs the same number of iterations (wave_alloc_slot uses ballots)
        const size_t iters = (total + stride_all - 1) / stride_all;
        for (size_t it = 0; it < iters; ++it) {
            const size_t idx = it * stride_all + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
            const bool in_range = idx < total;
            // anchor index is the SLOWEST of (n, cell) inside an image so lanes walk cells (coalesced)
            const int cell = in_range ? (int)(idx % wh) : 0;
            const size_t t = in_range ? idx / wh : 0;
            const int n = (int)(t % h.n);
            const int b = (int)(t / h.n);
            const int row = cell / h.w, col = cell % h.w;
            const float *p = h.out + (size_t)b * h.outputs;
            if (h.type == YL_YOLO) {
                const float *e = p + (size_t)n * wh * (5 + h.classes) + cell;    // entry k at e[k*wh]
                const float objectness = in_range ? e[4 * (size_t)wh] : 0.f;
                const bool pass = in_range && (objectness > thresh);
                const int slot = wave_alloc_slot(counts, b, pass);
                if (pass && slot < cap) {
                    float *r = records + ((size_t)b * cap + slot) * row_stride;
                    r[0] = __fdiv_rn(__fadd_rn((float)col, e[0]), (float)h.w);
                    r[1] = __fdiv_rn(__fadd_rn((float)row, e[(size_t)wh]), (float)h.h);
                    r[2] = (float)(exp((double)e[2 * (size_t)wh]) * (double)h.anchors_w[n] / (double)netw);
                    r[3] = (float)(exp((double)e[3 * (size_t)wh]) * (double)h.anchors_h[n] / (double)neth);
                    r[4] = objectness;
                    r[5] = (float)(key_base + cell * h.n + n);
                    for (int j = 0; j < h.classes; ++j) {
                        const float prob = __fmul_rn(objectness, e[(size_t)(5 + j) * wh]);
                        r[6 + j] = (prob > thresh) ? prob : 0.f;
                    }
                }
            } else {    // REGION: flattened HWC rows, every (cell, anchor) is a detection
                const int index = cell * h.n + n;
                const float *e = p + (size_t)index * (h.classes + 5);
                const int slot = wave_alloc_slot(counts, b, in_range);
                if (in_range && slot < cap) {
                    const float scale = e[4];
                    float *r = records + ((size_t)b * cap + slot) * row_stride;
                    const float lx = (float)(1. / (1. + exp((double)(-e[0]))));
                    const float ly = (float)(1. / (1. + exp((double)(-e[1]))));
                    r[0] = __fdiv_rn(__fadd_rn((float)col, lx), (float)h.w);
                    r[1] = __fdiv_rn(__fadd_rn((float)row, ly), (float)h.h);
                    r[2] = __fdiv_rn(__fmul_rn(expf_host_libm(e[2]), h.anchors_w[n]), (float)h.w);
                    r[3] = __fdiv_rn(__fmul_rn(expf_host_libm(e[3]), h.anchors_h[n]), (float)h.h);
                    r[4] = 1.f;
                    r[5] = (float)(key_base + cell * h.n + n);
                    if (h.tree_parent) {
                        // YOLO9000 (get_region_boxes_cpu, src/yolov2_forward_network.c:690-712): hierarchy_predictions
                        // (src/additionally.c:1878: p[j] *= p[parent[j]], parents first), then from the last class down
                        // the first one above .5 keeps its probability, all others become 0; the box passes when
                        // scale > thresh.  The record row is the scratch (the reference scribbles over l.output).
                        for (int j = 0; j < h.classes; ++j) {
                            const int par = h.tree_parent[j];
                            float v = e[5 + j];
                            if (par >= 0) v = __fmul_rn(v, r[6 + par]);
                            r[6 + j] = v;
                        }
                        bool found = false;
                        for (int j = h.classes - 1; j >= 0; --j) {
                            float v = r[6 + j];
                            if (!found && v > .5f) found = true;
                            else v = 0.f;
                            r[6 + j] = (scale > thresh) ? v : 0.f;
                        }
                    } else {
                        for (int j = 0; j < h.classes; ++j) {
                            const float prob = __fmul_rn(scale, e[5 + j]);
                            r[6 + j] = (prob > thresh) ? prob : 0.f;
                        }
                    }
                }
            }
        }
    }
}

int launch_compact(const HeadDesc *heads, int n_heads, int B, int netw, int neth, float thresh,
                   int cap, int row_stride, float *records, int *counts, void *stream)
{
    if (n_heads > 4) return (int)hipErrorInvalidValue;
    HeadsDev hd;
    hd.n_heads = n_heads;
    size_t mx = 1;
    for (int i = 0; i < n_heads; ++i) {
        hd.h[i] = heads[i];
        const size_t t = (size_t)B * heads[i].w * heads[i].h * heads[i].n;
        if (t > mx) mx = t;
    }
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(int) * B, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(compact_kernel, dim3(grid_for(mx)), dim3(256), 0, (hipStream_t)stream,
                       hd, B, netw, neth, thresh, cap, row_stride, records, counts);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ split-K second stage
// partial sums of `parts` channel ranges (conv_f32_x3.hip / conv_f32_row3.hip with ConvF32Args::ksplit > 1) -> the layer's tensor.
// The partials are added in range order, then the bias, then the activation with the convolution kernels' arithmetic
// (leaky as (float)(.1 * (double)x)), then the fused [shortcut] operand: a fixed order, the same bits on every run.
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float *__restrict__ ws, int parts, size_t part_stride,
                                                            const float *__restrict__ bias, int M, int OHW, size_t total, int act,
                                                            const float *__restrict__ add, float *__restrict__ out,
                                                            float *__restrict__ out_add)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float v = ws[i];
        for (int s = 1; s < parts; ++s) v = __fadd_rn(v, ws[(size_t)s * part_stride + i]);
        const int m = (int)((i / (size_t)OHW) % (size_t)M);
        v = __fadd_rn(v, bias[m]);
        if (act == YL_LEAKY) {
            const float t = (float)(.1 * (double)v);
            v = (v > 0.f) ? v : t;
        }
        if (out) out[i] = v;
        if (add) out_add[i] = __fadd_rn(v, add[i]);
    }
}

int launch_splitk_finish(const float *ws, int parts, size_t part_stride, const float *bias, int B, int M, int OHW, int act,
                         const float *add, float *out, float *out_add, void *stream)
{
    if (!ws || parts < 2 || !bias || (!out && !add) || (add && !out_add) || (act != YL_LEAKY && act != YL_LINEAR)) return (int)hipErrorInvalidValue;
    const size_t total = (size_t)B * M * OHW;
    size_t g = (total + 255) / 256;
    if (g > 256 * 8) g = 256 * 8;
    if (g == 0) g = 1;
    hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, ws, parts, part_stride, bias, M, OHW,
                       total, act, add, out, out_add);
    return (int)hipGetLastError();
}

}  // namespace yl
