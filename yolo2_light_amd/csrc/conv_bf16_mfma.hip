// conv_bf16_mfma.hip -- K1b: opt-in BF16 variant of the FP32 convolution on v_mfma_f32_32x32x16_bf16
// (north_star (a) "FP32/BF16"; yl_network_set_precision(net, YL_PRECISION_BF16)).
//
// Same layer as conv_f32_mfma.hip computes (forward_convolutional_layer_cpu's FP32 branch,
// src/yolov2_forward_network.c:204-261) with the two GEMM operands rounded to bf16 (round-to-nearest-even) and
// FP32 accumulation, bias and activation.  Products of two bf16 numbers are exact in FP32, so against an FP32
// convolution of the ROUNDED operands the result differs only by summation order (tests compare it that way,
// tight); against the FP32 reference the operand rounding costs ~2^-9 relative per element, which is outside the
// 1e-4 contract -- hence opt-in and reported separately.
//
// The data path is the INT8 kernel's (conv_i8_mfma.hip), a 16-byte unit now holding 8 bf16 channels:
//   activations  act_h[B][G][H][W][8] bf16, G = Cpad/8 channel groups (power of two)
//   weights      w_h[K8pad][Mpad][8]  bf16, K8 index = tap*G + cg, zero padded
//   one lane feeds one unit per MFMA operand (lanes 0-31 k 0-7, lanes 32-63 k 8-15 of the 32x32x16 step)
// and the epilogue works in the MFMA C/D layout: FP32 NCHW rows through buffer stores, the fused [shortcut] operand
// prefetched before the last K panel, and the next BF16 layer's input written as bf16 units straight from the
// accumulator registers (a lane's 4 consecutive rows are 8 contiguous bytes of a unit: no cross-lane exchange).
#include <hip/hip_runtime.h>
#include <cstdio>

#include "kernels.h"
#include "../../include/yolo2_hip.h"

namespace yl {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------ FP32 NCHW -> bf16 units
// one lane per (b, cg, pixel): 8 coalesced plane reads, one 16-byte coalesced store; (g_off, G_total) as in
// launch_quantize_nhwc (a multi-input [route] is converted source by source)
__global__ __launch_bounds__(256) void pack_bf16_nc8_kernel(const float *__restrict__ in, uint4 *__restrict__ out,
                                                            size_t total, int C, int HW, int G, int G_total, int g_off)
{
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int pix = (int)(idx % HW);
        size_t t = idx / HW;
        const int cg = (int)(t % G);
        const size_t b = t / G;
        const float *src = in + (b * C + (size_t)cg * 8) * HW + pix;
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = cg * 8 + 2 * j;
            const f32x2 v = {c < C ? src[(size_t)(2 * j) * HW] : 0.f, c + 1 < C ? src[(size_t)(2 * j + 1) * HW] : 0.f};
            w[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
        }
        out[((size_t)b * G_total + g_off + cg) * HW + pix] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

int launch_pack_bf16(const float *in, void *out, int B, int C, int H, int W, int Cpad, void *stream, int g_off, int G_total)
{
    const int G = Cpad / 8;
    if (G_total <= 0) G_total = G;
    const size_t total = (size_t)B * G * H * W;
    size_t g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g == 0) g = 1;
    hipLaunchKernelGGL(pack_bf16_nc8_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                       in, (uint4 *)out, total, C, H * W, G, G_total, g_off);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ BF16 MFMA implicit GEMM (main loop = conv_i8_mfma.hip's)
constexpr int BK16 = 8;          // 16-byte k-units per LDS panel (= 4 MFMA k-steps of 32)

struct ConvBf16Dev {
    const void *in_q;
    const void *w_q;
    const float *bias;
    const float *add;
    float *out_add;
    float *out;
    void *q_out;            // bf16 side output for the next BF16 conv (nullptr = none)
    int q_G;
    int B, G, Gshift, H, W, M, Mpad, OH, OW;
    int size, stride, pad, act;
    int K16, K16pad;
    int Ntotal, OHW, tiles_m;
};

template <int BM, int BN, int WM, int WN, bool TAPPANEL, bool MFULL>
__global__ __launch_bounds__(WM * WN * 64) void conv_bf16_mfma_kernel(ConvBf16Dev p)
{
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr int KSTEPS = BK16 / 2;
    static_assert(BN % 64 == 0 && BN <= NT && NT % BN == 0, "B panel mapping");
    constexpr int A_UNITS = BK16 * BM;                 // 16-byte units per A panel
    constexpr int APT = (A_UNITS + NT - 1) / NT;
    constexpr bool A_FULL = (A_UNITS % NT) == 0;
    constexpr int BPT = BK16 * BN / NT;
    constexpr int G_STEP = NT / BN;
    static_assert(BPT >= 1 && BK16 % G_STEP == 0, "B panel mapping");

    __shared__ __attribute__((aligned(16))) uint4 smem[2 * BK16 * BM + 2 * BK16 * BN + BM / 4];
    uint4 *As = smem;
    uint4 *Bs = smem + 2 * BK16 * BM;
    float *bias_s = reinterpret_cast<float *>(smem + 2 * BK16 * BM + 2 * BK16 * BN);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int half = lane >> 5;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tile_m = logical % p.tiles_m;
    const int tile_n = logical / p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    if (tid < BM) bias_s[tid] = (m0 + tid < p.M) ? p.bias[m0 + tid] : 0.f;

    const int n_local = tid % BN;
    const int g0 = __builtin_amdgcn_readfirstlane(tid / BN);
    const int HW = p.H * p.W;
    const int n_g = n0 + n_local;
    const bool n_ok = n_g < p.Ntotal;
    const int bimg = n_g / p.OHW;
    const int pix = n_g - bimg * p.OHW;
    const int oy = pix / p.OW;
    const int ox = pix - oy * p.OW;
    const int iy0 = oy * p.stride - p.pad;
    const int ix0 = ox * p.stride - p.pad;

    // buffer descriptor over act_q, based at the first image of this tile, shifted back by
    // pad*(W+1) units so lane offsets are non-negative; invalid taps -> voffset 0xFFFFFFFF -> 0
    const int b_first = n0 / p.OHW;
    const size_t img_units = (size_t)p.G * HW;
    const char *tile_base = (const char *)p.in_q + ((size_t)b_first * img_units) * 16 - (ptrdiff_t)p.pad * (p.W + 1) * 16;
    size_t rec = ((size_t)p.B - b_first) * img_units * 16 + (size_t)p.pad * (p.W + 1) * 16;
    if (rec > 0xFFFFFFFEull) rec = 0xFFFFFFFEull;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)tile_base, 0, (int)(unsigned)rec, 0x00020000);
    const int voff = (int)(((unsigned)(bimg - b_first) * (unsigned)img_units +
                            (unsigned)(oy * p.stride) * (unsigned)p.W + (unsigned)(ox * p.stride)) * 16u);

    // inverted tap validity, bit t = tap index ky*size+kx (size <= 5 -> 25 bits)
    unsigned ntapmask = 0xFFFFFFFFu;
    if (n_ok) {
        unsigned m = 0;
        for (int ky = 0; ky < p.size; ++ky)
            for (int kx = 0; kx < p.size; ++kx) {
                const int iy = iy0 + ky, ix = ix0 + kx;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) m |= 1u << (ky * p.size + kx);
            }
        ntapmask = ~m;
    }

    v4i a_reg[APT];
    v4i b_reg[BPT];
    // TAPPANEL: per-panel state of the NEXT panel to load (all its units share one tap)
    int pn_soff = 0, pn_tinv = 0;

#define YL_PANEL_SETUP(KB)                                                                          \
    if (TAPPANEL) {                                                                                 \
        const int g = (KB) * BK16;                                                                  \
        const int tap = g >> p.Gshift;                                                              \
        const int cg = g & (p.G - 1);                                                               \
        const int ky = (p.size == 3) ? ((tap * 11) >> 5) : ((p.size == 1) ? 0 : tap / p.size);      \
        const int kx = tap - ky * p.size;                                                           \
        pn_soff = (cg * HW + ky * p.W + kx) * 16;                                                   \
        pn_tinv = __builtin_amdgcn_sbfe((int)ntapmask, tap, 1);                                     \
    }
#define YL_LOAD_A(KB, E)                                                                            \
    {                                                                                               \
        const int idx = tid + (E) * NT;                                                             \
        if (A_FULL || idx < A_UNITS) {                                                              \
            const int gr = idx / BM;                                                                \
            const int mm = idx - gr * BM;                                                           \
            a_reg[E] = *reinterpret_cast<const v4i *>(                                              \
                (const char *)p.w_q + ((size_t)((KB) * BK16 + gr) * p.Mpad + m0 + mm) * 16);                      \
        }                                                                                           \
    }
#define YL_LOAD_B(KB, E)                                                                            \
    {                                                                                               \
        int soff, tinv;                                                                             \
        if (TAPPANEL) {       /* K16 % BK16 == 0 here: no K tail */                                 \
            soff = pn_soff + (g0 + (E) * G_STEP) * HW * 16;                                         \
            tinv = pn_tinv;                                                                         \
        } else {                                                                                    \
            const int g = (KB) * BK16 + g0 + (E) * G_STEP;        /* wave-uniform */                \
            const int tap = g >> p.Gshift;                                                          \
            const int cg = g & (p.G - 1);                                                           \
            const int ky = (p.size == 3) ? ((tap * 11) >> 5) : ((p.size == 1) ? 0 : tap / p.size);  \
            const int kx = tap - ky * p.size;                                                       \
            const int kinv = (g >= p.K16) ? -1 : 0;                                                 \
            soff = kinv ? 0 : (cg * HW + ky * p.W + kx) * 16;                                       \
            tinv = __builtin_amdgcn_sbfe((int)ntapmask, kinv ? 31 : tap, 1) | kinv;                 \
        }                                                                                           \
        b_reg[E] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff | tinv, soff, 0)); \
    }
#define YL_STORE_A(BUF, E)                                                                          \
    {                                                                                               \
        const int idx = tid + (E) * NT;                                                             \
        if (A_FULL || idx < A_UNITS) As[(BUF) * BK16 * BM + idx] = __builtin_bit_cast(uint4, a_reg[E]); \
    }
#define YL_STORE_B(BUF, E)                                                                          \
    { Bs[(BUF) * BK16 * BN + (g0 + (E) * G_STEP) * BN + n_local] = __builtin_bit_cast(uint4, b_reg[E]); }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int wm = wave / WN, wn = wave - wm * WN;
    const int wm0 = wm * TM * 32, wn0 = wn * TN * 32;
    const int nkb = p.K16pad / BK16;

    // ---- prologue: panel 0 -> LDS[0]; panel 1 -> registers ----
    YL_PANEL_SETUP(0)
#pragma unroll
    for (int e = 0; e < APT; ++e) YL_LOAD_A(0, e)
#pragma unroll
    for (int e = 0; e < BPT; ++e) YL_LOAD_B(0, e)
#pragma unroll
    for (int e = 0; e < APT; ++e) YL_STORE_A(0, e)
#pragma unroll
    for (int e = 0; e < BPT; ++e) YL_STORE_B(0, e)
    if (nkb > 1) {
        YL_PANEL_SETUP(1)
#pragma unroll
        for (int e = 0; e < APT; ++e) YL_LOAD_A(1, e)
#pragma unroll
        for (int e = 0; e < BPT; ++e) YL_LOAD_B(1, e)
    }
    __syncthreads();

    // one k-block; DO_STORE: registers (panel kb+1) -> LDS[buf^1]; DO_LOAD: panel kb+2 -> registers
#define YL_ITER(KB, DO_STORE, DO_LOAD)                                                             \
    {                                                                                              \
        const int buf = (KB) & 1;                                                                  \
        if (DO_LOAD) { YL_PANEL_SETUP((KB) + 2) }                                                  \
        const uint4 *Ab = As + buf * BK16 * BM + wm0 + l31;                                        \
        const uint4 *Bb = Bs + buf * BK16 * BN + wn0 + l31;                                        \
        v4i av[2][TM], bv[2][TN];                                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) av[0][i] = __builtin_bit_cast(v4i, Ab[half * BM + i * 32]); \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) bv[0][j] = __builtin_bit_cast(v4i, Bb[half * BN + j * 32]); \
        _Pragma("unroll") for (int ks = 0; ks < KSTEPS; ++ks) {                                    \
            const int cur = ks & 1, nxt = cur ^ 1;                                                 \
            _Pragma("unroll") for (int e = ks * APT / KSTEPS; e < (ks + 1) * APT / KSTEPS; ++e) {  \
                if (DO_STORE) YL_STORE_A(buf ^ 1, e)                                               \
                if (DO_LOAD) YL_LOAD_A((KB) + 2, e)                                                \
            }                                                                                      \
            _Pragma("unroll") for (int e = ks * BPT / KSTEPS; e < (ks + 1) * BPT / KSTEPS; ++e) {  \
                if (DO_STORE) YL_STORE_B(buf ^ 1, e)                                               \
                if (DO_LOAD) YL_LOAD_B((KB) + 2, e)                                                \
            }                                                                                      \
            if (ks + 1 < KSTEPS) {                                                                 \
                _Pragma("unroll") for (int i = 0; i < TM; ++i)                                     \
                    av[nxt][i] = __builtin_bit_cast(v4i, Ab[(2 * (ks + 1) + half) * BM + i * 32]); \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                     \
                    bv[nxt][j] = __builtin_bit_cast(v4i, Bb[(2 * (ks + 1) + half) * BN + j * 32]); \
            }                                                                                      \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                         \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                     \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[cur][i]), __builtin_bit_cast(bf16x8, bv[cur][j]), acc[i][j], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
    }

    int kb = 0;
    for (; kb + 2 < nkb; ++kb) { YL_ITER(kb, true, true) __syncthreads(); }
    if (kb + 1 < nkb) { YL_ITER(kb, true, false) __syncthreads(); ++kb; }

    // ---- epilogue addressing (C/D layout): this lane's pixel columns and row offsets ----
    const int OHW = p.OHW;
    const int ob_first = (n0 + wn0) / OHW;                       // wave-uniform
    int voff_o[TN], voff_q[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + l31;
        const int ob = n / OHW;
        const int opix = n - ob * OHW;
        const bool ok = n < p.Ntotal;
        voff_o[j] = ok ? (int)(((unsigned)(ob - ob_first) * (unsigned)p.M * (unsigned)OHW + (unsigned)opix +
                                4u * (unsigned)half * (unsigned)OHW) * 4u) : -1;
        voff_q[j] = ok ? (int)((((unsigned)(ob - ob_first) * (unsigned)p.q_G * (unsigned)OHW + (unsigned)opix) * 16u) +
                               8u * (unsigned)half) : -1;
    }
    const size_t img_out = (size_t)p.M * OHW;
    size_t orec = ((size_t)p.B - ob_first) * img_out * 4;
    if (orec > 0xFFFFFFFEull) orec = 0xFFFFFFFEull;
    const bool has_add = p.add != nullptr;
    const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc(
        (void *)((has_add ? p.add : p.bias) + (has_add ? (size_t)ob_first * img_out : 0)), 0, has_add ? (int)(unsigned)orec : 0, 0x00020000);
    const int row_bytes = OHW * 4;
    // row validity of rows that can lie beyond M (only when the filter count is not a multiple of the tile)
#define YL_ROW_OFF(I, E) ((m0 + wm0 + (I) * 32 + ((E) & 3) + 8 * ((E) >> 2)) * row_bytes)
#define YL_ROW_OK(I, E) (MFULL || (m0 + wm0 + (I) * 32 + ((E) & 3) + 8 * ((E) >> 2) + 4 * half) < p.M)

    // [shortcut] operand: fetched now, lands under the MFMAs of the last panel
    float addv[TM][TN][16];
    if (has_add) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    addv[i][j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        rs_add, YL_ROW_OK(i, e) ? voff_o[j] : -1, YL_ROW_OFF(i, e), 0));
    }
    YL_ITER(kb, false, false)
#undef YL_ITER
#undef YL_PANEL_SETUP
#undef YL_LOAD_A
#undef YL_LOAD_B
#undef YL_STORE_A
#undef YL_STORE_B

    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.out ? p.out + (size_t)ob_first * img_out : (float *)p.bias), 0, p.out ? (int)(unsigned)orec : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_oadd = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(has_add ? p.out_add + (size_t)ob_first * img_out : (float *)p.bias), 0, has_add ? (int)(unsigned)orec : 0, 0x00020000);
    const size_t img_q = (size_t)p.q_G * OHW * 16;
    size_t qrec = ((size_t)p.B - ob_first) * img_q;
    if (qrec > 0xFFFFFFFEull) qrec = 0xFFFFFFFEull;
    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.q_out ? (char *)p.q_out + (size_t)ob_first * img_q : (char *)p.bias), 0, p.q_out ? (int)(unsigned)qrec : 0, 0x00020000);
    const bool leaky = p.act == YL_LEAKY;

#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float bias_r[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) bias_r[e] = bias_s[wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float y[16];
            // the FP32 path's epilogue arithmetic (conv_f32_mfma.hip): + bias, leaky as (float)(.1 * (double)x)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[i][j][e] + bias_r[e];
                if (leaky) v = (v > 0.f) ? v : (float)(.1 * (double)v);
                y[e] = v;
            }
            if (p.out) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[e]), rs_out,
                                                          YL_ROW_OK(i, e) ? voff_o[j] : -1, YL_ROW_OFF(i, e), 0);
            }
            if (has_add) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    y[e] = __fadd_rn(y[e], addv[i][j][e]);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[e]), rs_oadd,
                                                          YL_ROW_OK(i, e) ? voff_o[j] : -1, YL_ROW_OFF(i, e), 0);
                }
            }
            // ---- bf16 side output for the next BF16 convolution: rows 8u + 4*half + 0..3 of this lane are 4
            //      consecutive channels of the 8-channel unit u: lanes 0-31 write bytes 0-7, lanes 32-63 bytes 8-15
            //      of 32 consecutive units -- 512 contiguous bytes per store instruction, no cross-lane exchange ----
            if (p.q_out) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x2 lo = {y[4 * u + 0], y[4 * u + 1]}, hi = {y[4 * u + 2], y[4 * u + 3]};
                    v2u d;
                    d[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2));      // v_cvt_pk_bf16_f32: RNE
                    d[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2));
                    const int cg = (m0 + wm0 + i * 32 + 8 * u) >> 3;
                    const bool unit_ok = MFULL || (m0 + wm0 + i * 32 + 8 * u) < p.M;
                    __builtin_amdgcn_raw_buffer_store_b64(d, rs_q, unit_ok ? voff_q[j] : -1, cg * OHW * 16, 0);
                }
            }
        }
    }
#undef YL_ROW_OFF
#undef YL_ROW_OK
}

template <int BM, int BN, int WM, int WN>
static int launch_bf16_tile(ConvBf16Dev p, hipStream_t s)
{
    p.tiles_m = (p.M + BM - 1) / BM;
    const long long blocks = (long long)p.tiles_m * ((p.Ntotal + BN - 1) / BN);
    if (blocks <= 0 || blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    constexpr int NT = WM * WN * 64;
    const bool mfull = (p.M % BM) == 0;
    // a panel of BK16 units lies inside one tap when G is a multiple of BK16 (C >= 128)
    const bool tap = p.G >= BK16;
    dim3 grid((unsigned)blocks), block(NT);
    if (tap && mfull) hipLaunchKernelGGL((conv_bf16_mfma_kernel<BM, BN, WM, WN, true, true>), grid, block, 0, s, p);
    else if (tap) hipLaunchKernelGGL((conv_bf16_mfma_kernel<BM, BN, WM, WN, true, false>), grid, block, 0, s, p);
    else if (mfull) hipLaunchKernelGGL((conv_bf16_mfma_kernel<BM, BN, WM, WN, false, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv_bf16_mfma_kernel<BM, BN, WM, WN, false, false>), grid, block, 0, s, p);
    return (int)hipGetLastError();
}

// tile: 0 = heuristic, 1 = 64x128 (4 waves), 2 = 32x256, 3 = 128x128 (wave tile 64x64), 4 = 128x256 (8 waves), 5 = 64x256
int launch_conv_bf16(const ConvBf16Args &a, int tile, void *stream, char *name, size_t name_len)
{
    ConvBf16Dev d;
    d.in_q = a.in_h; d.w_q = a.w_h; d.bias = a.bias; d.out = a.out;
    d.add = a.add; d.out_add = a.out_add;
    d.q_out = a.h_out; d.q_G = a.h_G;
    d.B = a.B; d.G = a.Cpad / 8; d.H = a.H; d.W = a.W; d.M = a.M; d.Mpad = a.Mpad; d.OH = a.OH; d.OW = a.OW;
    d.size = a.size; d.stride = a.stride; d.pad = a.pad; d.act = a.act;
    if (d.G <= 0 || (d.G & (d.G - 1)) != 0 || a.size > 5) return (int)hipErrorInvalidValue;
    if (a.h_out && (a.M % 8) != 0) return (int)hipErrorInvalidValue;
    d.Gshift = 0;
    while ((1 << d.Gshift) < d.G) ++d.Gshift;
    d.K16 = a.size * a.size * d.G;
    d.K16pad = (d.K16 + BK16 - 1) / BK16 * BK16;
    d.OHW = a.OH * a.OW;
    const long long nt = (long long)a.B * d.OHW;
    if (nt > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    d.Ntotal = (int)nt;
    d.tiles_m = 0;
    hipStream_t s = (hipStream_t)stream;
    if (tile == 0) {
        auto nblocks = [&](int bm, int bn) { return (long long)((a.M + bm - 1) / bm) * ((nt + bn - 1) / bn); };
        if (a.M <= 32) tile = 2;
        else if (a.M <= 64) tile = 1;
        else if (nblocks(128, 128) >= 512) tile = 3;
        else tile = 1;
    }
    if ((tile == 3 || tile == 4) && a.Mpad % 128 != 0) tile = 1;
    const char *t = "?";
    int rc;
    switch (tile) {
    case 1: t = "64x128"; rc = launch_bf16_tile<64, 128, 2, 2>(d, s); break;
    case 2: t = "32x256"; rc = launch_bf16_tile<32, 256, 1, 4>(d, s); break;
    case 3: t = "128x128"; rc = launch_bf16_tile<128, 128, 2, 2>(d, s); break;
    case 4: t = "128x256w8"; rc = launch_bf16_tile<128, 256, 2, 4>(d, s); break;
    case 5: t = "64x256"; rc = launch_bf16_tile<64, 256, 1, 4>(d, s); break;
    default: return (int)hipErrorInvalidValue;
    }
    if (name) snprintf(name, name_len, "conv_bf16_mfma<%s>", t);
    return rc;
}

}  // namespace yl
