// pack.hip -- the kernel-layout weight packers on the device ("one-time prep on device").
//
// yl_network_to_device used to build every packed weight image in host loops (k-major FP32 panels, Winograd U in
// double, int8 / bf16 16-byte units, XNOR sign words -- the device-side counterparts of what the reference does on
// one host core in binary_align_weights, src/additionally.c:196-302, and of the cudnnTransformTensor repack of its
// GPU path, src/yolov2_forward_network_quantized.c:1489-1492) and then uploaded the 1.8-4x inflated results.  Here
// the prepared weights travel once, as they are, and these kernels write the packed images; every element is
// produced by the SAME arithmetic as the host packers (pure moves, comparisons, the round-to-nearest-even bf16
// conversion, and G g G^T in IEEE double without contraction), so the images are bit-identical
// (tests/test_gpu_prep.py compares them word for word).  One-time work: plain grid-stride kernels, coalesced on
// the source side.
#include <hip/hip_runtime.h>

#include "yl_internal.h"

namespace yl {

namespace {

inline unsigned pack_blocks(size_t total) { size_t g = (total + 255) / 256; if (g > 8192) g = 8192; return (unsigned)(g ? g : 1); }

// dst[k_dev][m] (k-major, [Kpad][Mpad], pre-zeroed) = w[m][c][t]; tap-major inside 16-channel blocks when asked;
// mean != nullptr: the xnor FP32 fallback's +-mean weights (binarize_weights, src/additionally.c:113-126)
__global__ __launch_bounds__(256) void pack_kmajor_kernel(const float *__restrict__ w, const float *__restrict__ mean,
                                                          float *__restrict__ dst, int M, int C, int taps, int Mpad, int tapmajor)
{
    const size_t K = (size_t)C * taps;
    const size_t total = (size_t)M * K;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / K);
        const int k_ref = (int)(idx - (size_t)m * K);
        const int c = k_ref / taps, t = k_ref - c * taps;
        const int k_dev = tapmajor ? (((c / 16) * taps + t) * 16 + (c % 16)) : k_ref;
        float v = w[idx];
        if (mean) v = (v > 0.f) ? mean[m] : -mean[m];
        dst[(size_t)k_dev * Mpad + m] = v;
    }
}

// U = G g G^T in double, rounded once to float (wino32_pack_weights): one lane per (tile_m,
// m in tile, channel); filters beyond M give zeros
__global__ __launch_bounds__(256) void pack_wino_kernel(const float *__restrict__ w, float *__restrict__ dst, int C, int M, int tiles_m)
{
    constexpr int BMT = 32;          // filters per workgroup tile
    const double G[4][3] = {{1., 0., 0.}, {.5, .5, .5}, {.5, -.5, .5}, {0., 0., 1.}};
    const int nkb = C / 4;
    const size_t total = (size_t)tiles_m * BMT * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const int ml = (int)((idx / C) % BMT);
        const int tm = (int)(idx / ((size_t)C * BMT));
        const int m = tm * BMT + ml;
        const int kb = c >> 2, kl = c & 3;
        double u[4][4];
        if (m < M) {
            const float *g = w + ((size_t)m * C + c) * 9;
            double t[4][3];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int b = 0; b < 3; ++b)
                    t[i][b] = __dadd_rn(__dadd_rn(__dmul_rn(G[i][0], (double)g[0 * 3 + b]), __dmul_rn(G[i][1], (double)g[1 * 3 + b])),
                                        __dmul_rn(G[i][2], (double)g[2 * 3 + b]));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    u[i][j] = __dadd_rn(__dadd_rn(__dmul_rn(t[i][0], G[j][0]), __dmul_rn(t[i][1], G[j][1])), __dmul_rn(t[i][2], G[j][2]));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) u[i][j] = 0.;
        }
        float *panel = dst + ((size_t)tm * nkb + kb) * (64 * BMT);
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            const float v = (float)u[xi >> 2][xi & 3];
            panel[xi * 128 + (kl & 1) * 64 + ml * 2 + (kl >> 1)] = v;                       // [xi][half][m 32][kk]
        }
    }
}

// int8 16-byte units [K16pad][Mpad][16] (pre-zeroed), K16 index = tap * G + c / 16
__global__ __launch_bounds__(256) void pack_i8_units_kernel(const int8_t *__restrict__ wq, int8_t *__restrict__ dst, int M, int C, int taps,
                                                            int G, int Mpad)
{
    const size_t total = (size_t)M * C * taps;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(idx % taps);
        const int c = (int)((idx / taps) % C);
        const int m = (int)(idx / ((size_t)taps * C));
        const int g = t * G + c / 16;
        dst[((size_t)g * Mpad + m) * 16 + (c % 16)] = wq[idx];
    }
}

// bf16 16-byte units [K8pad][Mpad][8] (pre-zeroed), round to nearest even as runtime.hip's f32_to_bf16_rne
__global__ __launch_bounds__(256) void pack_bf16_units_kernel(const float *__restrict__ w, uint16_t *__restrict__ dst, int M, int C, int taps,
                                                              int G, int Mpad)
{
    const size_t total = (size_t)M * C * taps;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(idx % taps);
        const int c = (int)((idx / taps) % C);
        const int m = (int)(idx / ((size_t)taps * C));
        const int g = t * G + c / 8;
        unsigned u = __float_as_uint(w[idx]);
        uint16_t h;
        if ((u & 0x7fffffffu) > 0x7f800000u) h = (uint16_t)((u >> 16) | 0x40);      // NaN stays NaN
        else { u += 0x7fffu + ((u >> 16) & 1u); h = (uint16_t)(u >> 16); }
        dst[((size_t)g * Mpad + m) * 8 + (c % 8)] = h;
    }
}

// K1x weights (conv_f32_x3.hip): every weight as three bf16 pieces, w3[panel][piece 3][k-octet 2][Mpad][8] (pre-zeroed), panel =
// (c / 16) * taps + tap; the same integer round-to-nearest-even split as x3_pack_weights on the host (the subtractions are exact)
__global__ __launch_bounds__(256) void pack_x3_kernel(const float *__restrict__ w, uint16_t *__restrict__ dst, int M, int C, int taps, int Mpad)
{
    const size_t total = (size_t)M * C * taps;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(idx % taps);
        const int c = (int)((idx / taps) % C);
        const int m = (int)(idx / ((size_t)taps * C));
        float r = w[idx];
        uint16_t h[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
            unsigned u = __float_as_uint(r);
            u += 0x7fffu + ((u >> 16) & 1u);
            h[pc] = (uint16_t)(u >> 16);
            r = __fsub_rn(r, __uint_as_float((unsigned)h[pc] << 16));
        }
        const size_t panel = (size_t)(c / 16) * taps + t;
        const int oct = (c % 16) / 8, e = c % 8;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) dst[(((panel * 3 + pc) * 2 + oct) * Mpad + m) * 8 + e] = h[pc];
    }
}

// K1r weights (conv_f32_row3.hip): the row transform U = G g of every filter row (U0 = g0, U1 = (g0 + g1 + g2) / 2,
// U2 = (g0 - g1 + g2) / 2, U3 = g2; formed in double, rounded once) as three bf16 pieces,
// wr[group = (c / 16) * 3 + ky][plane 4][piece 3][k-octet 2][Mpad][8] (pre-zeroed); the arithmetic of row3_pack_weights on the host
__global__ __launch_bounds__(256) void pack_row3_kernel(const float *__restrict__ w, uint16_t *__restrict__ dst, int M, int C, int Mpad)
{
    const size_t total = (size_t)M * C * 3;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int ky = (int)(idx % 3);
        const int c = (int)((idx / 3) % C);
        const int m = (int)(idx / ((size_t)3 * C));
        const float *g = w + ((size_t)m * C + c) * 9 + ky * 3;
        const double g0 = g[0], g1 = g[1], g2 = g[2];
        float u[4];
        u[0] = g[0];
        u[1] = (float)__dadd_rn(__dadd_rn(__dmul_rn(.5, g0), __dmul_rn(.5, g1)), __dmul_rn(.5, g2));
        u[2] = (float)__dadd_rn(__dsub_rn(__dmul_rn(.5, g0), __dmul_rn(.5, g1)), __dmul_rn(.5, g2));
        u[3] = g[2];
        const size_t group = (size_t)(c / 16) * 3 + ky;
        const int oct = (c % 16) / 8, e = c % 8;
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            float r = u[xi];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                unsigned b = __float_as_uint(r);
                b += 0x7fffu + ((b >> 16) & 1u);
                const uint16_t h = (uint16_t)(b >> 16);
                r = __fsub_rn(r, __uint_as_float((unsigned)h << 16));
                dst[((((group * 4 + xi) * 3 + pc) * 2 + oct) * Mpad + m) * 8 + e] = h;
            }
        }
    }
}

// XNOR sign words [Mpad/2][Cw][2][9] (pre-set to all ones: channel-pad bits and pad filters never match);
// bit = (w > 0) (src/additionally.c:123,1544); one lane per (m, tap, channel word)
__global__ __launch_bounds__(256) void pack_xnor_words_kernel(const float *__restrict__ w, uint64_t *__restrict__ dst, int M, int C, int Cw)
{
    const size_t total = (size_t)M * 9 * Cw;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int cw = (int)(idx % Cw);
        const int t = (int)((idx / Cw) % 9);
        const int m = (int)(idx / ((size_t)Cw * 9));
        uint64_t word = 0;
        for (int b = 0; b < 64; ++b) {
            const int c = cw * 64 + b;
            const bool bit = (c < C) ? (w[((size_t)m * C + c) * 9 + t] > 0.f) : true;
            if (bit) word |= (1ull << b);
        }
        dst[((size_t)(m / 2) * Cw + cw) * 18 + (m % 2) * 9 + t] = word;
    }
}

}  // namespace

int dev_pack_kmajor(const float *d_w, const float *d_mean, float *d_dst, int M, int C, int taps, int Mpad, int tapmajor, void *stream)
{
    hipLaunchKernelGGL(pack_kmajor_kernel, dim3(pack_blocks((size_t)M * C * taps)), dim3(256), 0, (hipStream_t)stream,
                       d_w, d_mean, d_dst, M, C, taps, Mpad, tapmajor);
    return (int)hipGetLastError();
}

int dev_pack_wino(const float *d_w, float *d_dst, int C, int M, void *stream)
{
    const int bm = 32;
    const int tiles_m = (M + bm - 1) / bm;
    const unsigned g = pack_blocks((size_t)tiles_m * bm * C);
    hipLaunchKernelGGL(pack_wino_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, d_w, d_dst, C, M, tiles_m);
    return (int)hipGetLastError();
}

int dev_pack_i8_units(const int8_t *d_wq, int8_t *d_dst, int M, int C, int taps, int G, int Mpad, void *stream)
{
    hipLaunchKernelGGL(pack_i8_units_kernel, dim3(pack_blocks((size_t)M * C * taps)), dim3(256), 0, (hipStream_t)stream,
                       d_wq, d_dst, M, C, taps, G, Mpad);
    return (int)hipGetLastError();
}

int dev_pack_bf16_units(const float *d_w, uint16_t *d_dst, int M, int C, int taps, int G, int Mpad, void *stream)
{
    hipLaunchKernelGGL(pack_bf16_units_kernel, dim3(pack_blocks((size_t)M * C * taps)), dim3(256), 0, (hipStream_t)stream,
                       d_w, d_dst, M, C, taps, G, Mpad);
    return (int)hipGetLastError();
}

int dev_pack_x3(const float *d_w, void *d_dst, int M, int C, int taps, int Mpad, void *stream)
{
    hipLaunchKernelGGL(pack_x3_kernel, dim3(pack_blocks((size_t)M * C * taps)), dim3(256), 0, (hipStream_t)stream,
                       d_w, (uint16_t *)d_dst, M, C, taps, Mpad);
    return (int)hipGetLastError();
}

int dev_pack_row3(const float *d_w, void *d_dst, int M, int C, int Mpad, void *stream)
{
    hipLaunchKernelGGL(pack_row3_kernel, dim3(pack_blocks((size_t)M * C * 3)), dim3(256), 0, (hipStream_t)stream,
                       d_w, (uint16_t *)d_dst, M, C, Mpad);
    return (int)hipGetLastError();
}

int dev_pack_xnor_words(const float *d_w, uint64_t *d_dst, int M, int C, int Cw, void *stream)
{
    hipLaunchKernelGGL(pack_xnor_words_kernel, dim3(pack_blocks((size_t)M * 9 * Cw)), dim3(256), 0, (hipStream_t)stream,
                       d_w, d_dst, M, C, Cw);
    return (int)hipGetLastError();
}

}  // namespace yl
