// conv_i8_mfma.hip -- K2: the `-quantized` INT8 convolution on gfx950 INT8 MFMA.
//
// Replaces forward_convolutional_layer_q (src/yolov2_forward_network_quantized.c:527-631:
// input quantisation :554-560, im2col_cpu_int8 :186-209, gemm_nn_int8_int16 :469-491,
// dequant/bias/leaky :596-627) and the reference GPU's cudnnTransformTensor +
// INT8x4 cudnnConvolutionBiasActivationForward (src/yolov2_forward_network_gpu.cu:189-229).
//
// Data layout in HBM (chosen for 16-byte coalescing, not inherited from the reference):
//   activations  act_q[B][G][H][W][16]  int8, G = Cpad/16 channel groups ("NC/16HW16"):
//                for a fixed channel group consecutive pixels are consecutive 16-byte units,
//                so both the quantise stores and the conv's im2col gathers are coalesced dwordx4.
//   weights      w_q[K16pad][Mpad][16]  int8, K16 index = tap*G + cg (tap = ky*size+kx),
//                i.e. k-major panels of 16-byte units, zero padded.
// Math: v_mfma_i32_32x32x32_i8.  Each lane feeds 16 consecutive k-bytes of one filter (A)
// and of one output pixel (B); integer accumulation is exact, so any k order that A and B
// share gives the reference's acc32 bit for bit.  The epilogue replays the reference's
// scalar arithmetic exactly:
//   o16 = clamp_abs(acc32 / 32 [C truncating division], 32767)
//   y   = o16 * ALPHA1 ; y += bias ; leaky: y > 0 ? y : y / 10
//
// What bounds it: activations stay FP32 between layers where a [shortcut]/route/head
// reads them (reference semantics), so a residual block moves 8 bytes of FP32 residual stream per
// output next to 1-2 bytes of int8 -- HBM-bound, the MFMA work is 5-30 % of a layer.  Round 1's
// kernel spent more issue slots in its epilogue than in its K loop (per output ~40 VALU + an LDS
// transpose + per-element branches) and more LDS cycles staging 64x128 tiles than MFMA cycles.
// This version:
//   * the epilogue works in the MFMA C/D register layout -- no LDS, no wave barriers:
//       - a lane owns pixel column n and rows (e&3)+8*(e>>2)+4*half: FP32 NCHW loads/stores are one
//         dword per lane, 32 consecutive pixels = one 128-byte line per row and half-wave, addressed
//         through buffer descriptors (row offset in an SGPR, pixel offset in a VGPR, columns beyond
//         the tensor dropped by the range check);
//       - the [shortcut] operand is fetched BEFORE the last K panel and lands under its MFMAs;
//       - the int8 side output for the next layer packs the lane's 4 consecutive channels into a
//         dword and one v_permlane32_swap gives each lane 8 contiguous bytes of the 16-byte unit
//         (lanes 0-31 bytes 0-7, lanes 32-63 bytes 8-15): 512 contiguous bytes per store instruction;
//   * exact arithmetic in ~16 VALU per output: the int16 clamp is applied first (|acc| < 2^20 is
//     exact in float), acc/32 = trunc(acc * 2^-5), y/10 by Markstein's fma sequence, leaky as
//     max(y, y/10); the two data-dependent corners (|y| < 1e-30 where the fma sequence could meet
//     subnormals, |x*mult| >= 32768 where `int16_t = float` wraps) are detected per 32x32 block with
//     v_min3/v_max3 and recomputed by the plain formulas on a cold path;
//   * tile choice per layer: 128x128 (wave tile 64x64: half the LDS operand traffic per MFMA) for
//     K-heavy layers, 64x128 / 32x256 for narrow-M layers.
#include <hip/hip_runtime.h>
#include <climits>
#include <cstdlib>

#include "kernels.h"
#include "epilogue.h"
#include "../../include/yolo2_hip.h"

namespace yl {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------ K2a: quantise + repack
// one lane per (b, cg, pixel): 16 coalesced plane reads, one 16-byte coalesced store
// G = channel groups of THIS source, written at groups [g_off, g_off + G) of a tensor with G_total groups (a
// multi-input [route] is quantised source by source into one int8 tensor: the FP32 concatenation is never built)
// up > 1: `in` is the INPUT of a nearest-neighbour [upsample] (scale 1) whose output this pass would otherwise read -- quantising is
// pointwise, so quantise(upsample(x)) = upsample(quantise(x)): pixel (y, x) of the H x W output takes (y / up, x / up) of the
// (H / up) x (W / up) tensor, and the upsampled FP32 tensor is never written (forward_upsample_layer_cpu, src/additionally.c)
__global__ __launch_bounds__(256) void quantize_nc16_kernel(const float *__restrict__ in, int8_t *__restrict__ out,
                                                            size_t total, int C, int HW, int G, int G_total, int g_off, float mult,
                                                            int W, int up)
{
    const int HWin = up > 1 ? HW / (up * up) : HW;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int pix = (int)(idx % HW);
        size_t t = idx / HW;
        const int cg = (int)(t % G);
        const size_t b = t / G;
        int spix = pix;
        if (up > 1) {
            const int y = pix / W, x = pix - y * W;
            spix = (y / up) * (W / up) + x / up;
        }
        const float *src = in + (b * C + (size_t)cg * 16) * HWin + spix;
        unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int q = 0;
            if (cg * 16 + j < C) {
                q = quantize_input_i8(src[(size_t)j * HWin], mult);
            }
            w[j >> 2] |= ((unsigned)(q & 0xFF)) << ((j & 3) * 8);
        }
        uint4 v = make_uint4(w[0], w[1], w[2], w[3]);
        *reinterpret_cast<uint4 *>(out + (((size_t)b * G_total + g_off + cg) * HW + pix) * 16) = v;
    }
}

int launch_quantize_nhwc(const float *in, int8_t *out, int B, int C, int H, int W, int Cpad, float mult, void *stream,
                         int g_off, int G_total, int up)
{
    if (up < 1 || (up > 1 && (H % up || W % up))) return (int)hipErrorInvalidValue;
    const int G = Cpad / 16;
    if (G_total <= 0) G_total = G;
    const size_t total = (size_t)B * G * H * W;
    size_t g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g == 0) g = 1;
    hipLaunchKernelGGL(quantize_nc16_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                       in, out, total, C, H * W, G, G_total, g_off, mult, W, up);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ K2b: INT8 MFMA implicit GEMM
//   - one LDS panel = BK16 16-byte k-units (128 bytes of K), double-buffered; registers hold
//     panel kb+1 and are refilled with panel kb+2 in slices interleaved with the MFMAs (as K1)
//   - when G >= BK16 (C >= 128) a panel lies inside one tap: tap decode once per panel
//   - BK16 = 16-byte k-units per LDS panel: 8 (= 4 MFMA k-steps of 32; 64 KB of LDS at 128x128: 2 workgroups
//     per CU) or 4 (half the LDS, a barrier every 2 k-steps, more workgroups per CU to overlap epilogues)

struct ConvI8Dev {
    const int8_t *in_q;
    const int8_t *w_q;
    const float *bias;
    const float *add;
    float *out_add;
    float *out;
    int32_t *dbg;
    int8_t *q_out;          // quantised side output for the next INT8 conv (nullptr = none)
    float q_mult;
    int q_G;
    int B, G, Gshift, H, W, M, Mpad, OH, OW;
    int size, stride, pad, act;
    int K16, K16pad;
    float alpha1;
    int Ntotal, OHW, tiles_m;
    int no_corner;
};

// y / 10 correctly rounded (the reference's leaky on this path is `y / 10`, quantized.c:625):
// Markstein's sequence q = RN(y*r), e = fma(-10, q, y), q' = fma(e, r, q) with r = RN(1/10) returns
// RN(y/10) for normal operands; tiny |y| (possible subnormal intermediates) take the IEEE divide.
__device__ __forceinline__ float div10_markstein(float y)
{
    const float r = 0.1f;
    const float q = __fmul_rn(y, r);
    const float e = __fmaf_rn(-10.f, q, y);
    return __fmaf_rn(e, r, q);
}
__device__ __forceinline__ float div10_exact(float y)
{
    if (fabsf(y) < 1e-30f) return __fdiv_rn(y, 10.f);
    return div10_markstein(y);
}

// NOC bit 0 / bit 1: the host proved that the leaky / quantise corner cannot occur in this layer (ConvI8Args::no_corner):
// their per-output tracking (v_min/v_max over |.|, 2 VALU each) and cold paths are compiled out.
template <int BM, int BN, int WM, int WN, bool TAPPANEL, bool MFULL, int BK16 = 8, int NOC = 0>
__global__ __launch_bounds__(WM * WN * 64) void conv_i8_mfma_kernel(ConvI8Dev p)
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
    const int8_t *tile_base = p.in_q + ((size_t)b_first * img_units) * 16 - (ptrdiff_t)p.pad * (p.W + 1) * 16;
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
                p.w_q + ((size_t)((KB) * BK16 + gr) * p.Mpad + m0 + mm) * 16);                      \
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

    v16i acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;

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
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0); \
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

    if (p.dbg) {          // parity hook: int16-clamped accumulators, straight from the C/D layout
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn0 + j * 32 + l31;
            if (n >= p.Ntotal) continue;
            const int ob = n / p.OHW;
            const size_t obase = (size_t)ob * p.M * p.OHW + (n - ob * p.OHW);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                    if (m < p.M) {
                        const int a = acc[i][j][e];
                        int o = (a + ((a >> 31) & 31)) >> 5;
                        o = o > 32767 ? 32767 : (o < -32767 ? -32767 : o);
                        p.dbg[obase + (size_t)m * p.OHW] = o;
                    }
                }
        }
    }

    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.out ? p.out + (size_t)ob_first * img_out : (float *)p.bias), 0, p.out ? (int)(unsigned)orec : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_oadd = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(has_add ? p.out_add + (size_t)ob_first * img_out : (float *)p.bias), 0, has_add ? (int)(unsigned)orec : 0, 0x00020000);
    const size_t img_q = (size_t)p.q_G * OHW * 16;
    size_t qrec = ((size_t)p.B - ob_first) * img_q;
    if (qrec > 0xFFFFFFFEull) qrec = 0xFFFFFFFEull;
    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.q_out ? p.q_out + (size_t)ob_first * img_q : (int8_t *)p.bias), 0, p.q_out ? (int)(unsigned)qrec : 0, 0x00020000);
    const bool leaky = p.act == YL_LEAKY;
    const float alpha1 = p.alpha1, q_mult = p.q_mult;

#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float bias_r[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) bias_r[e] = bias_s[wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float y[16];
            float tmin = 1.f;
            // ---- o16 -> y (fast forms; the corners are caught by tmin and redone below).  o16 =
            //      max_abs(acc / 32, 32767): clamping acc to +-(32767*32+31) FIRST gives the same o16 (the
            //      clamp only acts where |acc/32| >= 32768) and makes acc exact in float, so the C
            //      truncating division is trunc(acc * 2^-5) ----
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int a = acc[i][j][e];
                const int c = a < -1048575 ? -1048575 : (a > 1048575 ? 1048575 : a);
                const float o = truncf(__fmul_rn((float)c, 0.03125f));
                float v = __fadd_rn(__fmul_rn(o, alpha1), bias_r[e]);
                if (leaky) {
                    if constexpr (!(NOC & 1)) tmin = fminf(tmin, fabsf(v));
                    v = fmaxf(v, div10_markstein(v));
                }
                y[e] = v;
            }
            if (!(NOC & 1) && leaky && __builtin_amdgcn_ballot_w64(tmin < 1e-30f) != 0ull) {      // cold: zeros / near-subnormal outputs
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int a = acc[i][j][e];
                    int o = (a + ((a >> 31) & 31)) >> 5;
                    o = o > 32767 ? 32767 : (o < -32767 ? -32767 : o);
                    float v = __fadd_rn(__fmul_rn((float)o, alpha1), bias_r[e]);
                    y[e] = (v > 0.f) ? v : div10_exact(v);
                }
            }
            // ---- FP32 outputs in the C/D layout: one 128-byte line per row and half-wave ----
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
            // ---- int8 side output for the next INT8 convolution (its input multiplier) ----
            if (p.q_out) {
                unsigned pk[4];
                float tmax = 0.f;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    int c[4];
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const float t = __fmul_rn(y[g4 * 4 + r4], q_mult);
                        if constexpr (!(NOC & 2)) tmax = fmaxf(tmax, fabsf(t));
                        const int ci = (int)t;                          // v_cvt_i32_f32: truncation
                        c[r4] = ci < -127 ? -127 : (ci > 127 ? 127 : ci);
                    }
                    pk[g4] = __builtin_amdgcn_perm((unsigned)c[1], (unsigned)c[0], 0x0C0C0400u) |
                             __builtin_amdgcn_perm((unsigned)c[3], (unsigned)c[2], 0x04000C0Cu);
                }
                if (!(NOC & 2) && __builtin_amdgcn_ballot_w64(!(tmax < 32768.f)) != 0ull) {    // cold: the int16 wrap-around corner / NaN
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        unsigned w = 0;
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4)
                            w |= ((unsigned)(quantize_input_i8(y[g4 * 4 + r4], q_mult) & 0xFF)) << (8 * r4);
                        pk[g4] = w;
                    }
                }
                // rows 0-3 | 8-11 (lanes 0-31) and 4-7 | 12-15 (lanes 32-63) of each 16-channel unit:
                // after the half exchange lanes 0-31 hold bytes 0-7, lanes 32-63 bytes 8-15
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(pk[2 * u], pk[2 * u + 1], false, false);
                    v2u d;
                    d[0] = sw[0]; d[1] = sw[1];
                    const int cg = (m0 + wm0 + i * 32 + 16 * u) >> 4;
                    const bool unit_ok = MFULL || (m0 + wm0 + i * 32 + 16 * u) < p.M;
                    __builtin_amdgcn_raw_buffer_store_b64(d, rs_q, unit_ok ? voff_q[j] : -1, cg * OHW * 16, 0);
                }
            }
        }
    }
#undef YL_ROW_OFF
#undef YL_ROW_OK
}

template <int BM, int BN, int WM, int WN, int BK16 = 8>
static int launch_i8_tile(ConvI8Dev p, hipStream_t s)
{
    p.K16pad = (p.K16 + BK16 - 1) / BK16 * BK16;          // <= the host's padding to 8 units
    p.tiles_m = (p.M + BM - 1) / BM;
    const long long blocks = (long long)p.tiles_m * ((p.Ntotal + BN - 1) / BN);
    if (blocks <= 0 || blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    constexpr int NT = WM * WN * 64;
    const bool mfull = (p.M % BM) == 0;
    // a panel of BK16 units lies inside one tap when G is a multiple of BK16 (C >= 128)
    const bool tap = p.G >= BK16;
    dim3 grid((unsigned)blocks), block(NT);
    // the corner-free instances exist for the common case (whole tap panels, M a multiple of the tile) of the two
    // tiles the heuristic picks
    constexpr bool NOC_TILE = BK16 == 8 && ((BM == 128 && BN == 128) || (BM == 64 && BN == 128));
    if constexpr (NOC_TILE) {
        if (tap && mfull && p.no_corner) {
            if (p.no_corner == 3) hipLaunchKernelGGL((conv_i8_mfma_kernel<BM, BN, WM, WN, true, true, BK16, 3>), grid, block, 0, s, p);
            else if (p.no_corner == 1) hipLaunchKernelGGL((conv_i8_mfma_kernel<BM, BN, WM, WN, true, true, BK16, 1>), grid, block, 0, s, p);
            else hipLaunchKernelGGL((conv_i8_mfma_kernel<BM, BN, WM, WN, true, true, BK16, 2>), grid, block, 0, s, p);
            return (int)hipGetLastError();
        }
    }
    if (tap && mfull) hipLaunchKernelGGL((conv_i8_mfma_kernel<BM, BN, WM, WN, true, true, BK16>), grid, block, 0, s, p);
    else if (tap) hipLaunchKernelGGL((conv_i8_mfma_kernel<BM, BN, WM, WN, true, false, BK16>), grid, block, 0, s, p);
    else if (mfull) hipLaunchKernelGGL((conv_i8_mfma_kernel<BM, BN, WM, WN, false, true, BK16>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv_i8_mfma_kernel<BM, BN, WM, WN, false, false, BK16>), grid, block, 0, s, p);
    return (int)hipGetLastError();
}

// tile: 0 = heuristic, 1 = 64x128 (4 waves, wave tile 32x64), 2 = 32x256, 3 = 128x128 (wave tile 64x64),
//       4 = 128x256 (8 waves, wave tile 64x64), 5 = 64x256 (wave tile 64x64), 6 / 7 = 128x128 / 64x128 with half-depth
//       LDS panels (BK16 = 4)
int launch_conv_i8(const ConvI8Args &a, int tile, void *stream, char *name, size_t name_len)
{
    ConvI8Dev d;
    d.in_q = a.in_q; d.w_q = a.w_q; d.bias = a.bias; d.out = a.out; d.dbg = a.dbg;
    d.add = a.add; d.out_add = a.out_add;
    d.q_out = a.q_out; d.q_mult = a.q_mult; d.q_G = a.q_G;
    d.no_corner = a.no_corner & 3;
    if (a.add) d.no_corner &= ~2;            // the side output quantises conv + [shortcut]: its range is data
    if (a.dbg) d.no_corner = 0;
    d.B = a.B; d.G = a.Cpad / 16; d.H = a.H; d.W = a.W; d.M = a.M; d.Mpad = a.Mpad; d.OH = a.OH; d.OW = a.OW;
    d.size = a.size; d.stride = a.stride; d.pad = a.pad; d.act = a.act; d.alpha1 = a.alpha1;
    if (d.G <= 0 || (d.G & (d.G - 1)) != 0 || a.size > 5) return (int)hipErrorInvalidValue;
    if (a.q_out && (a.M % 16) != 0) return (int)hipErrorInvalidValue;
    d.Gshift = 0;
    while ((1 << d.Gshift) < d.G) ++d.Gshift;
    d.K16 = a.size * a.size * d.G;
    d.K16pad = (d.K16 + 7) / 8 * 8;
    d.OHW = a.OH * a.OW;
    const long long nt = (long long)a.B * d.OHW;
    if (nt > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    d.Ntotal = (int)nt;
    d.tiles_m = 0;
    hipStream_t s = (hipStream_t)stream;
    if (tile == 0) {
        auto nblocks = [&](int bm, int bn) { return (long long)((a.M + bm - 1) / bm) * ((nt + bn - 1) / bn); };
        // measured per layer on yolov3-608 b64 (profiles/r2_i8_tile_sweep.txt): 32x256 for M = 32; 64x128 up
        // to 64 filters; from 128 filters the 64x64 wave tile (half the LDS operand traffic per MFMA) wins on
        // every layer, 1x1 included, while there are enough tiles to fill the chip twice
        if (a.M <= 32) tile = 2;
        else if (a.M <= 64) tile = 1;
        else if (nblocks(128, 128) >= 512) tile = 3;
        else tile = 1;
    }
    if ((tile == 3 || tile == 4 || tile == 6) && a.Mpad % 128 != 0) tile = 1;
    const char *t = "?";
    int rc;
    switch (tile) {
    case 1: t = "64x128"; rc = launch_i8_tile<64, 128, 2, 2>(d, s); break;
    case 2: t = "32x256"; rc = launch_i8_tile<32, 256, 1, 4>(d, s); break;
    case 3: t = "128x128"; rc = launch_i8_tile<128, 128, 2, 2>(d, s); break;
    case 4: t = "128x256w8"; rc = launch_i8_tile<128, 256, 2, 4>(d, s); break;
    case 5: t = "64x256"; rc = launch_i8_tile<64, 256, 1, 4>(d, s); break;
    case 6: t = "128x128k4"; rc = launch_i8_tile<128, 128, 2, 2, 4>(d, s); break;
    case 7: t = "64x128k4"; rc = launch_i8_tile<64, 128, 2, 2, 4>(d, s); break;
    default: return (int)hipErrorInvalidValue;
    }
    if (name) snprintf(name, name_len, "conv_i8_mfma<%s>", t);
    return rc;
}

}  // namespace yl
