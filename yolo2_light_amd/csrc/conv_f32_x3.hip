// conv_f32_x3.hip -- K1x: the FP32 convolution of conv_f32_mfma.hip on the BF16 matrix pipe, FP32-exact operands.
//
// Same layer (forward_convolutional_layer_cpu's FP32 branch, src/yolov2_forward_network.c:204-261: im2col + gemm_nn +
// bias + leaky), FP32 tensors in and out.  On gfx950 v_mfma_f32_32x32x2_f32 runs on the FP32 vector datapath: 64 cycles
// for 4 K MACs per lane-pair, and every VALU instruction of a neighbour wave takes ~4.5 of those cycles away
// (tools/mfma_valu_contention.hip).  v_mfma_f32_32x32x16_bf16 has its own pipe: 34 cycles for EIGHT times the MACs, VALU
// work beside it is free.  An FP32 number is the exact sum of three bf16 numbers,
//
//     a = a1 + a2 + a3,   a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)      (8 + 8 + 8 significand bits, RNE),
//
// a product of two bf16 numbers is exact in FP32, and the matrix pipe accumulates in FP32.  So
//
//     a * b  =  a1 b1 + (a1 b2 + a2 b1) + (a2 b2 + a1 b3 + a3 b1)  +  [a2 b3 + a3 b2 + a3 b3]
//
// six MFMAs per 16 k instead of eight FP32 MFMAs: 204 cycles instead of 520.  The dropped bracket: round-to-nearest leaves
// |a - a1| <= 2^-8 |a| and |a - a1 - a2| <= 2^-16 |a| (a3 holds that residual exactly), so in the worst case a2 b3 + a3 b2 <=
// 2 * 2^-24 |a b| and a3 b3 <= 2^-32 |a b| -- two half-ulps of the FP32 product; for operands spread over a binade the
// residuals are uniform, |a2| ~ 2^-9.4 |a| and |a3| ~ 2^-17.4 |a| rms, and the bracket is ~2^-26 |a b| rms, a quarter of the
// half-ulp (2^-24 |a b|) FP32 arithmetic itself loses wherever it rounds a product.  What differs
// from the FP32-MFMA kernel in practice is the summation inside the matrix pipe: measured against a float64 convolution a layer
// is up to 1.4x as far from the truth as conv_f32_mfma.hip's result (and at most 1.5x, asserted per shape in
// tests/test_gpu_parity.py::test_conv_x3_vs_oracle; every layer of yolov3 with this kernel on stays inside the bound of
// test_fp32_error_vs_float64_truth, measured 1.0 of it with K1x on all 75 layers, 0.72 in the shipped mix).  Inputs whose first
// piece overflows bf16 (|x| >= 3.3961e38: beyond bf16's largest finite value 3.3895e38 by half its spacing) become Inf and the result NaN where FP32 arithmetic may
// still give a finite product, and pieces below 2^-126 flush to zero (residuals of |x| < 2^-108): outside any activation this
// path produces, stated in DESIGN.md and pinned by tests/test_gpu_row3.py::test_three_piece_kernels_at_the_edges_of_the_format.  This is NOT the opt-in BF16 variant
// (conv_bf16_mfma.hip rounds the operands to ONE bf16 piece, 2^-9 relative).
//
// Weights are split once (x3_pack_weights); activations are split by the staging threads between their global load and
// their LDS store (12 VALU per pair of elements -- free beside this MFMA).
//
//   K order      panels of 16 channels: panel p = (channel block p / taps, tap p % taps)   [conv_f32_mfma.hip's tap-major order]
//   weights      w3[panel][piece 3][k-octet 2][Mpad][8] bf16   (one 16-byte unit = 8 consecutive k of one filter)
//   LDS stage    A[piece][k-octet][BM] units, B[piece][k-octet][BN] units; one ds_read_b128 per MFMA operand
//   lanes        lane (l31, half) feeds row / column l31 and k-octet `half` of the 32x32x16 step
//   epilogue     MFMA C/D layout: one dword per lane and accumulator row, 32 consecutive pixels per row = 128-byte lines
//
// Applicability: C % 16 == 0, any size <= 5 through the tap decode, optional fused [shortcut]; the layers with INT8 / sign /
// pooled side outputs stay on conv_f32_mfma.hip; the [yolo] layer behind a linear 1x1 head is folded into the epilogue.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "kernels.h"
#include "../../include/yolo2_hip.h"

// Lab builds only (tools/ab_builds.sh, ABFILE=conv_f32_x3: -DX_DBG=<bits>; results are garbage by design): 1 no global loads in the
// K loop, 2 no split / LDS stores, 4 no MFMAs, 8 no epilogue stores.  The shipped library is built without -DYL_LAB, which forces X_DBG to 0.
#if !defined(YL_LAB)
#undef X_DBG
#endif
#ifndef X_DBG
#define X_DBG 0
#endif

namespace yl {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

// round-to-nearest-even bf16 of a finite float, as its 16 high bits (the host half of the split; v_cvt_pk_bf16_f32 on the device)
inline uint16_t bf16_rne_host(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf16_to_float_host(uint16_t h)
{
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

struct ConvX3Dev {
    const float *in;
    const void *w3;
    const float *bias;
    const float *add;       // fused [shortcut]: out_add = act(conv) + add (nullptr = none)
    float *out_add;
    float *out;             // may be nullptr when only out_add is wanted
    int B, C, H, W, M, Mpad, OH, OW;
    int size, stride, pad, act;
    int taps, nkb;
    int Ntotal, OHW, tiles_m;
    int yolo_entries;       // > 0: the [yolo] layer behind a linear 1x1 head folded into the epilogue (as in conv_f32_mfma.hip)
    // SRC2 instances (1x1 layers): the input is the channel concatenation of TWO tensors that is never built -- channels 0 .. C1 - 1
    // = `in`, a tensor of (H / up) x (W / up) read through a nearest-neighbour [upsample] by `up` (pixel (y, x) takes (y / up, x / up)),
    // channels C1 .. C - 1 = `in2` at H x W: [upsample] -> [route] -> conv of yolov3 (forward_upsample_layer_cpu / forward_route_layer_cpu)
    const float *in2;
    int C1, up;
    // split K (gridDim.y = ranges): range r reads channels from r * ks_in_off floats on, weights from r * ks_w_off bytes on, and writes
    // its partial sums r * ks_out_off floats into the workspace `out` points at; nkb = the panels of ONE range, C stays the tensor's
    int ksplit;
    size_t ks_in_off, ks_w_off, ks_out_off;
};

// two FP32 values -> their three bf16 pieces, packed (low half = x)
__device__ __forceinline__ void split3_pair(float x, float y, unsigned &p1, unsigned &p2, unsigned &p3)
{
    const f32x2 v = {x, y};
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));          // v_cvt_pk_bf16_f32: RNE
    const float x1 = __uint_as_float(p1 << 16), y1 = __uint_as_float(p1 & 0xFFFF0000u);
    const f32x2 r = {x - x1, y - y1};                                               // exact
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    const float x2 = __uint_as_float(p2 << 16), y2 = __uint_as_float(p2 & 0xFFFF0000u);
    const f32x2 q = {r[0] - x2, r[1] - y2};                                         // exact
    p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2));
}

// KS: 1 / 3 = the layer's filter size (compile-time tap decode), 0 = any size <= 5
// YOLO: rows m with m % yolo_entries not in {2, 3} get logistic_activate (forward_yolo_layer_cpu; the expression of
// yolo_kernel in layers.hip, so the tensor is bit-identical to the unfused pair of layers)
// SRC2 (KS = 1): two-source input, see ConvX3Dev -- a panel of 16 channels lies in one source (C1 % 16 == 0), which one is a scalar
// decision per panel: descriptor, lane offset and channel stride are selected, the loads are the same
template <int BM, int BN, int WM, int WN, int KS, bool MFULL, bool YOLO = false, bool PIPE = false, bool SRC2 = false>
// (occupancy as before the straight-line epilogue: 148 VGPRs = three waves per SIMD at 128 x 128, four at 64 x 128 -- hipcc otherwise
// computes all 64 outputs of a block at once and takes 192-244 registers)
__global__ __launch_bounds__(WM * WN * 64, (BM == 128 && BN == 128) ? 3 : (BM == 64 ? (BN == 64 ? 3 : 4) : 2)) void conv_f32_x3_kernel(ConvX3Dev p)
{
    if (p.ksplit > 1) {                     // this workgroup's channel range (uniform: scalar arithmetic on the kernel arguments)
        const size_t r = blockIdx.y;
        p.in += r * p.ks_in_off;
        p.w3 = reinterpret_cast<const char *>(p.w3) + r * p.ks_w_off;
        p.out += r * p.ks_out_off;
    }
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr int A_UNITS = 6 * BM;                      // 16-byte units of one A panel
    constexpr int APT = (A_UNITS + NT - 1) / NT;
    constexpr bool A_FULL = (A_UNITS % NT) == 0;
    constexpr int OPT = (2 * BN) / NT;                   // k-octets (of one pixel) per thread: 1 or 2
    static_assert(OPT == 1 || OPT == 2, "B panel mapping");
    static_assert(NT % BM == 0, "A panel mapping");
    constexpr int A_STEP = NT / BM;                      // (piece, k-octet) rows of the A panel one pass of the threads covers

    // two LDS stages of 24 KB (128 x 128 tile): three workgroups per CU.  (A three-stage form with the fragments of panel
    // kb+1 prefetched under the MFMAs of panel kb -- 74 KB, 172 VGPRs, two workgroups per CU -- was 8 % slower.)
    __shared__ __attribute__((aligned(16))) uint4 smem[2 * 6 * BM + 2 * 6 * BN + BM / 4];
    uint4 *As = smem;
    uint4 *Bs = smem + 2 * 6 * BM;
    float *bias_s = reinterpret_cast<float *>(smem + 2 * 6 * BM + 2 * 6 * BN);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int half = lane >> 5;

    // XCD-aware tile order (blocks b, b+8, ... share an L2): consecutive logical tiles = the filter tiles of one pixel tile
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    // (uniform values the compiler computes on the vector unit -- the division -- must be pinned to SGPRs: a buffer
    // descriptor in VGPRs turns every load into a waterfall loop)
    const int tile_n = __builtin_amdgcn_readfirstlane(logical / p.tiles_m);
    const int tile_m = logical - tile_n * p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int size = KS > 0 ? KS : p.size;

    if (tid < BM) bias_s[tid] = (m0 + tid < p.M) ? p.bias[m0 + tid] : 0.f;

    // ---- staging role: pixel n_local, k-octet(s) oct0 .. oct0 + OPT - 1 of every panel ----
    const int n_local = tid % BN;
    const int oct0 = __builtin_amdgcn_readfirstlane((tid / BN) * OPT);      // wave-uniform (BN is a multiple of 64)
    const int HW = p.H * p.W;
    const int n_g = n0 + n_local;
    const bool n_ok = n_g < p.Ntotal;
    const int bimg = n_ok ? n_g / p.OHW : 0;
    const int pix = n_g - bimg * p.OHW;
    const int oy = pix / p.OW;
    const int ox = pix - oy * p.OW;
    const int iy0 = oy * p.stride - p.pad;
    const int ix0 = ox * p.stride - p.pad;

    // buffer descriptor over the input, based at the first image of this tile and shifted back by pad*(W+1) elements so
    // that lane offsets are non-negative; out-of-image taps -> voffset 0xFFFFFFFF -> the range check returns 0.0
    const int b_first = __builtin_amdgcn_readfirstlane(n0 / p.OHW);
    const size_t img_floats = (size_t)p.C * HW;
    const float *tile_base = p.in + (size_t)b_first * img_floats - (ptrdiff_t)p.pad * (p.W + 1);
    size_t rec = (((size_t)p.B - b_first) * img_floats + (size_t)p.pad * (p.W + 1)) * sizeof(float);
    if (rec > 0xFFFFFFFEull) rec = 0xFFFFFFFEull;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)tile_base, 0, (int)(unsigned)rec, 0x00020000);
    const int voff = (int)(((unsigned)(bimg - b_first) * (unsigned)img_floats + (unsigned)(oy * p.stride) * (unsigned)p.W +
                            (unsigned)(ox * p.stride)) * 4u);
    // SRC2: the descriptor / lane offset above are not used; source 1 (upsampled on the fly) and source 2 get their own
    static_assert(!SRC2 || KS == 1, "two-source input: 1x1 layers");
    const int W1 = SRC2 ? p.W / p.up : 0, HW1 = SRC2 ? (p.H / p.up) * W1 : 0, C2 = SRC2 ? p.C - p.C1 : 0;
    const size_t img1 = (size_t)(SRC2 ? p.C1 : 0) * HW1, img2 = (size_t)C2 * HW;
    size_t rec1 = ((size_t)p.B - b_first) * img1 * sizeof(float), rec2 = ((size_t)p.B - b_first) * img2 * sizeof(float);
    if (rec1 > 0xFFFFFFFEull) rec1 = 0xFFFFFFFEull;
    if (rec2 > 0xFFFFFFFEull) rec2 = 0xFFFFFFFEull;
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void *)(SRC2 ? p.in + (size_t)b_first * img1 : p.in), 0, SRC2 ? (int)(unsigned)rec1 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void *)(SRC2 ? p.in2 + (size_t)b_first * img2 : p.in), 0, SRC2 ? (int)(unsigned)rec2 : 0, 0x00020000);
    const int voff1 = !SRC2 ? 0 : (n_ok ? (int)(((unsigned)(bimg - b_first) * (unsigned)img1 + (unsigned)(oy / p.up) * (unsigned)W1 + (unsigned)(ox / p.up)) * 4u) : -1);
    const int voff2 = !SRC2 ? 0 : (n_ok ? (int)(((unsigned)(bimg - b_first) * (unsigned)img2 + (unsigned)oy * (unsigned)p.W + (unsigned)ox) * 4u) : -1);
    unsigned ntapmask = 0xFFFFFFFFu;          // inverted tap validity, bit t = ky * size + kx
    if (n_ok) {
        unsigned m = 0;
        for (int ky = 0; ky < size; ++ky)
            for (int kx = 0; kx < size; ++kx) {
                const int iy = iy0 + ky, ix = ix0 + kx;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) m |= 1u << (ky * size + kx);
            }
        ntapmask = ~m;
    }

    // the packed weights through a buffer descriptor: lane offset = (row of the first pass, filter), the panel / pass / tile
    // offsets in the scalar soffset
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.w3, 0, (int)((unsigned)p.nkb * 6u * (unsigned)p.Mpad * 16u), 0x00020000);
    const int a_voff = ((tid / BM) * p.Mpad + (tid % BM)) * 16;
    v4i a_reg[APT];
    float b_raw[OPT][8];
    int pn_tap = 0, pn_c0 = 0;                // tap / first channel of the NEXT panel to be loaded
    int pn_soff = 0, pn_tinv = 0;
    bool pn_first = true;                     // SRC2: the next panel's channels lie in source 1
#define X3_PANEL_SETUP()                                                                           \
    if constexpr (SRC2) {                                                                          \
        pn_first = pn_c0 < p.C1;                                                                   \
        pn_soff = pn_first ? pn_c0 * HW1 * 4 : (pn_c0 - p.C1) * HW * 4;                            \
        pn_tinv = pn_c0 < p.C ? 0 : -1;                                                            \
    } else {                                                                                       \
        const int ky = (KS == 3) ? ((pn_tap * 11) >> 5) : ((KS == 1) ? 0 : pn_tap / size);         \
        const int kx = pn_tap - ky * size;                                                         \
        pn_soff = (pn_c0 * HW + ky * p.W + kx) * 4;                                                \
        pn_tinv = __builtin_amdgcn_sbfe((int)ntapmask, pn_tap, 1);                                 \
    }
#define X3_PANEL_ADVANCE()                                                                         \
    {                                                                                              \
        ++pn_tap;                                                                                  \
        if (pn_tap >= p.taps) { pn_tap = 0; pn_c0 += 16; }                                         \
    }
#define X3_LOAD_A(KB, E)                                                                           \
    {                                                                                              \
        /* row pk = piece * 2 + k-octet = tid / BM + E * A_STEP; rows >= 6 (partial last pass) are skipped wave by wave */ \
        if (A_FULL || tid + (E) * NT < A_UNITS)                                                    \
            a_reg[E] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(              \
                rs_w, a_voff, (((KB) * 6 + (E) * A_STEP) * p.Mpad + m0) * 16, 0));                 \
    }
#define X3_STORE_A(BUF, E)                                                                         \
    {                                                                                              \
        const int idx = tid + (E) * NT;                                                            \
        if (A_FULL || idx < A_UNITS) As[(BUF) * 6 * BM + idx] = __builtin_bit_cast(uint4, a_reg[E]); \
    }
#define X3_LOAD_B()                                                                                \
    if constexpr (SRC2) {                                                                          \
        /* (two straight-line forms behind one scalar branch: a select of the 128-bit descriptor would cost four s_cselect per load) */ \
        if (pn_first) {                                                                            \
            _Pragma("unroll") for (int o = 0; o < OPT; ++o)                                        \
                _Pragma("unroll") for (int e = 0; e < 8; ++e)                                      \
                    b_raw[o][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(  \
                        rs1, voff1 | pn_tinv, pn_soff + ((oct0 + o) * 8 + e) * HW1 * 4, 0));       \
        } else {                                                                                   \
            _Pragma("unroll") for (int o = 0; o < OPT; ++o)                                        \
                _Pragma("unroll") for (int e = 0; e < 8; ++e)                                      \
                    b_raw[o][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(  \
                        rs2, voff2 | pn_tinv, pn_soff + ((oct0 + o) * 8 + e) * HW * 4, 0));        \
        }                                                                                          \
    } else {                                                                                       \
        _Pragma("unroll") for (int o = 0; o < OPT; ++o)                                            \
            _Pragma("unroll") for (int e = 0; e < 8; ++e)                                          \
                b_raw[o][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(      \
                    rsrc, voff | pn_tinv, pn_soff + ((oct0 + o) * 8 + e) * HW * 4, 0));            \
    }
#define X3_STORE_B(BUF)                                                                            \
    {                                                                                              \
        _Pragma("unroll") for (int o = 0; o < OPT; ++o) {                                          \
            unsigned u1[4], u2[4], u3[4];                                                          \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                          \
                split3_pair(b_raw[o][2 * t], b_raw[o][2 * t + 1], u1[t], u2[t], u3[t]);            \
            uint4 *dst = Bs + (BUF) * 6 * BN + (oct0 + o) * BN + n_local;                          \
            dst[0 * 2 * BN] = make_uint4(u1[0], u1[1], u1[2], u1[3]);                              \
            dst[1 * 2 * BN] = make_uint4(u2[0], u2[1], u2[2], u2[3]);                              \
            dst[2 * 2 * BN] = make_uint4(u3[0], u3[1], u3[2], u3[3]);                              \
        }                                                                                          \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int wm = wave / WN, wn = wave - wm * WN;
    const int wm0 = wm * TM * 32, wn0 = wn * TN * 32;
    const int nkb = p.nkb;

    // ---- prologue: panel 0 -> LDS[0]; panel 1 -> registers ----
    X3_PANEL_SETUP()
#pragma unroll
    for (int e = 0; e < APT; ++e) X3_LOAD_A(0, e)
    X3_LOAD_B()
    X3_PANEL_ADVANCE()
#pragma unroll
    for (int e = 0; e < APT; ++e) X3_STORE_A(0, e)
    X3_STORE_B(0)
    if (nkb > 1) {
        X3_PANEL_SETUP()
#pragma unroll
        for (int e = 0; e < APT; ++e) X3_LOAD_A(1, e)
        X3_LOAD_B()
        X3_PANEL_ADVANCE()
    }
    __syncthreads();

    // one panel: fragments of panel kb from LDS[buf]; DO_STORE: registers (panel kb+1) -> split -> LDS[buf^1];
    // DO_LOAD: panel kb+2 -> registers; 6 * TM * TN MFMAs, smallest terms first
#define X3_ITER(KB, DO_STORE, DO_LOAD)                                                             \
    {                                                                                              \
        const int buf = (KB) & 1;                                                                  \
        int a_off = buf * 6 * BM + half * BM + wm0 + l31;                                          \
        int b_off = buf * 6 * BN + half * BN + wn0 + l31;                                          \
        asm volatile("" : "+v"(a_off), "+v"(b_off));       /* one base register each, immediate offsets below */ \
        const uint4 *Ab = As + a_off;                                                              \
        const uint4 *Bb = Bs + b_off;                                                              \
        v4i av[3][TM], bv[3][TN];                                                                  \
        _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) {                                         \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) av[pc][i] = __builtin_bit_cast(v4i, Ab[pc * 2 * BM + i * 32]); \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) bv[pc][j] = __builtin_bit_cast(v4i, Bb[pc * 2 * BN + j * 32]); \
        }                                                                                          \
        if (DO_STORE && !(X_DBG & 2)) {                                                            \
            _Pragma("unroll") for (int e = 0; e < APT; ++e) X3_STORE_A(buf ^ 1, e)                 \
            X3_STORE_B(buf ^ 1)                                                                    \
        }                                                                                          \
        if (DO_LOAD && !(X_DBG & 1)) {                                                             \
            X3_PANEL_SETUP()                                                                       \
            _Pragma("unroll") for (int e = 0; e < APT; ++e) X3_LOAD_A((KB) + 2, e)                 \
            X3_LOAD_B()                                                                            \
            X3_PANEL_ADVANCE()                                                                     \
        }                                                                                          \
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0};                                                  \
        constexpr int TB[6] = {0, 2, 1, 0, 1, 0};                                                  \
        if (!(X_DBG & 4))                                                                          \
        _Pragma("unroll") for (int t = 0; t < 6; ++t)                                              \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                         \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                     \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[TA[t]][i]), \
                                                                        __builtin_bit_cast(bf16x8, bv[TB[t]][j]), acc[i][j], 0, 0, 0); \
        if (PIPE && (DO_STORE) && X_DBG == 0) {                                                    \
            /* PIPE: the fragment reads first, then the split / LDS stores / requests pinned between the MFMAs (conv_f32_row3.hip's */ \
            /* schedule; masks 0x008 MFMA, 0x002 VALU, 0x020 VMEM read, 0x100 DS read, 0x200 DS write) */ \
            __builtin_amdgcn_sched_group_barrier(0x100, 3 * (TM + TN), 0);                         \
            _Pragma("unroll") for (int i_ = 0; i_ < 6 * TM * TN; ++i_) {                           \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                 \
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);                                 \
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                 \
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                 \
            }                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
    }

    int kb = 0;
    for (; kb + 2 < nkb; ++kb) { X3_ITER(kb, true, true) __syncthreads(); }
    if (kb + 1 < nkb) { X3_ITER(kb, true, false) __syncthreads(); ++kb; }
    X3_ITER(kb, false, false)
#undef X3_ITER
#undef X3_PANEL_SETUP
#undef X3_PANEL_ADVANCE
#undef X3_LOAD_A
#undef X3_STORE_A
#undef X3_LOAD_B
#undef X3_STORE_B

    // ---- epilogue (C/D layout): + bias, activation with conv_f32_mfma.hip's arithmetic, FP32 NCHW rows ----
    const int OHW = p.OHW;
    const int ob_first = __builtin_amdgcn_readfirstlane((n0 + wn0) / OHW);
    int voff_o[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + l31;
        const int ob = n / OHW;
        const int opix = n - ob * OHW;
        voff_o[j] = n < p.Ntotal ? (int)(((unsigned)(ob - ob_first) * (unsigned)p.M * (unsigned)OHW + (unsigned)opix +
                                          4u * (unsigned)half * (unsigned)OHW) * 4u) : -1;
    }
    const size_t img_out = (size_t)p.M * OHW;
    size_t orec = ((size_t)p.B - ob_first) * img_out * 4;
    if (orec > 0xFFFFFFFEull) orec = 0xFFFFFFFEull;
    const bool has_out = p.out != nullptr && (X_DBG & 8) == 0, has_add = p.add != nullptr && (X_DBG & 8) == 0;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(has_out ? p.out + (size_t)ob_first * img_out : (float *)p.bias), 0, has_out ? (int)(unsigned)orec : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(has_add ? p.add + (size_t)ob_first * img_out : p.bias), 0, has_add ? (int)(unsigned)orec : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_oadd = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(has_add ? p.out_add + (size_t)ob_first * img_out : (float *)p.bias), 0, has_add ? (int)(unsigned)orec : 0, 0x00020000);
    const int row_bytes = OHW * 4;
    const bool leaky = p.act == YL_LEAKY;
    // One wave-uniform branch picks the output form (MODE 0: out, 1: out_add only, 2: both) and the activation; inside it the code is
    // straight-line: the [shortcut] operands of eight accumulator rows are requested before their arithmetic, and leaky is
    // three conversions / multiplies and a select (left as a ternary hipcc branches around the double-precision path for every
    // output, and a run-time `has_add` / `leaky` inside the loops became a branch per store).  Same arithmetic as before.
    auto epilogue = [&](auto mode_tag, auto leaky_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool LEAKY = decltype(leaky_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float bias_r[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) bias_r[e] = bias_s[wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int e0 = 0; e0 < 16; e0 += 8) {
                    int vo[8];
                    float addv[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int e = e0 + k;
                        const int mrow = m0 + wm0 + i * 32 + (e & 3) + 8 * (e >> 2);
                        const bool ok = MFULL || (mrow + 4 * half) < p.M;
                        vo[k] = ok ? voff_o[j] : -1;
                        if constexpr (MODE >= 1)
                            addv[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_add, vo[k], mrow * row_bytes, 0));
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int e = e0 + k;
                        float v = acc[i][j][e] + bias_r[e];
                        if constexpr (LEAKY) {
                            float t = (float)(.1 * (double)v);
                            asm volatile("" : "+v"(t));
                            v = (v > 0.f) ? v : t;
                        }
                        const int mrow = m0 + wm0 + i * 32 + (e & 3) + 8 * (e >> 2);
                        if constexpr (YOLO) {
                            const int entry = (mrow + 4 * half) % p.yolo_entries;
                            if (entry != 2 && entry != 3) v = (float)(1. / (1. + exp((double)(-v))));
                        }
                        if constexpr (MODE != 1)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, vo[k], mrow * row_bytes, 0);
                        if constexpr (MODE >= 1)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, __fadd_rn(v, addv[k])), rs_oadd, vo[k], mrow * row_bytes, 0);
                    }
                }
            }
        }
    };
    if (leaky) {
        if (!has_add) epilogue(std::integral_constant<int, 0>{}, std::true_type{});
        else if (!has_out) epilogue(std::integral_constant<int, 1>{}, std::true_type{});
        else epilogue(std::integral_constant<int, 2>{}, std::true_type{});
    } else {
        if (!has_add) epilogue(std::integral_constant<int, 0>{}, std::false_type{});
        else if (!has_out) epilogue(std::integral_constant<int, 1>{}, std::false_type{});
        else epilogue(std::integral_constant<int, 2>{}, std::false_type{});
    }
}

template <int BM, int BN, int WM, int WN, bool PIPE = false>
int launch_x3_tile(ConvX3Dev p, hipStream_t s)
{
    p.tiles_m = (p.M + BM - 1) / BM;
    const long long blocks = (long long)p.tiles_m * ((p.Ntotal + BN - 1) / BN);
    if (blocks <= 0 || blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks, (unsigned)(p.ksplit > 1 ? p.ksplit : 1)), block(WM * WN * 64);
    const bool mfull = (p.M % BM) == 0;
#define X3_GO(KS)                                                                                  \
    {                                                                                              \
        if (mfull) hipLaunchKernelGGL((conv_f32_x3_kernel<BM, BN, WM, WN, KS, true, false, PIPE>), grid, block, 0, s, p); \
        else hipLaunchKernelGGL((conv_f32_x3_kernel<BM, BN, WM, WN, KS, false, false, PIPE>), grid, block, 0, s, p); \
    }
    if (p.in2) {            // two-source 1x1 (x3_two_source_ok): the 128 x 128 and 64 x 64 tiles have the instance
        if constexpr ((BM == 128 && BN == 128 && PIPE) || (BM == 64 && BN == 64)) {
            if (mfull) hipLaunchKernelGGL((conv_f32_x3_kernel<BM, BN, WM, WN, 1, true, false, PIPE, true>), grid, block, 0, s, p);
            else hipLaunchKernelGGL((conv_f32_x3_kernel<BM, BN, WM, WN, 1, false, false, PIPE, true>), grid, block, 0, s, p);
        } else return (int)hipErrorInvalidValue;
    } else if (p.size == 1 && p.yolo_entries > 0) {
        if (mfull) hipLaunchKernelGGL((conv_f32_x3_kernel<BM, BN, WM, WN, 1, true, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((conv_f32_x3_kernel<BM, BN, WM, WN, 1, false, true>), grid, block, 0, s, p);
    } else if (p.yolo_entries > 0) return (int)hipErrorInvalidValue;
    else if (p.size == 1) X3_GO(1)
    else if (p.size == 3) X3_GO(3)
    else X3_GO(0)
#undef X3_GO
    return (int)hipGetLastError();
}

constexpr int X3_MPAD = 128;          // the widest filter tile

}  // namespace

bool x3_applicable(int C, int M, int size, int stride, int pad)
{
    (void)stride; (void)pad;
    return C >= 16 && (C % 16) == 0 && M >= 1 && size >= 1 && size <= 5;
}

size_t x3_packed_bytes(int C, int M, int size)
{
    const size_t mpad = (size_t)(M + X3_MPAD - 1) / X3_MPAD * X3_MPAD;
    return (size_t)(C / 16) * size * size * 6 * mpad * 16;
}

// w: [M][C][size][size] (the reference's l.weights).  dst: [panel][piece 3][k-octet 2][Mpad][8] bf16, panel = (c / 16) * taps + tap
void x3_pack_weights(const float *w, int C, int M, int size, void *dst)
{
    const int taps = size * size;
    const size_t mpad = (size_t)(M + X3_MPAD - 1) / X3_MPAD * X3_MPAD;
    uint16_t *d = static_cast<uint16_t *>(dst);
    memset(d, 0, x3_packed_bytes(C, M, size));
    for (int m = 0; m < M; ++m)
        for (int c = 0; c < C; ++c)
            for (int t = 0; t < taps; ++t) {
                const float a = w[((size_t)m * C + c) * taps + t];
                const uint16_t h1 = bf16_rne_host(a);
                const float r1 = a - bf16_to_float_host(h1);
                const uint16_t h2 = bf16_rne_host(r1);
                const float r2 = r1 - bf16_to_float_host(h2);
                const uint16_t h3 = bf16_rne_host(r2);
                const size_t panel = (size_t)(c / 16) * taps + t;
                const int oct = (c % 16) / 8, e = c % 8;
                const uint16_t hs[3] = {h1, h2, h3};
                for (int pc = 0; pc < 3; ++pc)
                    d[(((panel * 3 + pc) * 2 + oct) * mpad + m) * 8 + e] = hs[pc];
            }
}

// a 1x1 layer whose input is [route]([upsample](a.in), a.in2), read from the two tensors directly (ConvX3Dev::in2)
bool x3_two_source_ok(const ConvF32Args &a)
{
    return a.x3_w && a.in2 && a.size == 1 && a.stride == 1 && a.pad == 0 && a.in2_C1 > 0 && a.in2_C1 < a.C && (a.in2_C1 % 16) == 0 &&
           ((a.C - a.in2_C1) % 16) == 0 && a.in2_up >= 1 && (a.H % a.in2_up) == 0 && (a.W % a.in2_up) == 0 && a.M > 64 &&
           a.OH == a.H && a.OW == a.W && a.yolo_entries == 0 && !a.q_out && !a.bits_out && !a.pool_out;
}

// tile: 0 = heuristic, 1 = 128x128 (wave tile 64x64), 2 = 64x128 (32x64), 3 = 32x256 (32x64), 4 = 64x64 (two waves of 32x64: small grids),
// 5 = 128x128 without the pinned schedule (A/B)
int launch_conv_f32_x3(const ConvF32Args &a, int tile, void *stream, char *name, size_t name_len, bool plain)
{
    if (!a.x3_w || !x3_applicable(a.C, a.M, a.size, a.stride, a.pad) || (!a.out && !a.add) || (a.add && !a.out_add) || a.q_out ||
        a.bits_out || a.pool_out || (a.yolo_entries > 0 && (a.size != 1 || a.add)))
        return (int)hipErrorInvalidValue;
    if (a.in2 && (!x3_two_source_ok(a) || (tile != 0 && tile != 1 && tile != 4) || plain)) return (int)hipErrorInvalidValue;
    ConvX3Dev d;
    d.in = a.in; d.w3 = a.x3_w; d.bias = a.bias; d.out = a.out; d.add = a.add; d.out_add = a.out_add;
    d.in2 = a.in2; d.C1 = a.in2_C1; d.up = a.in2_up;
    d.B = a.B; d.C = a.C; d.H = a.H; d.W = a.W; d.M = a.M; d.OH = a.OH; d.OW = a.OW;
    d.Mpad = (a.M + X3_MPAD - 1) / X3_MPAD * X3_MPAD;
    d.size = a.size; d.stride = a.stride; d.pad = a.pad; d.act = a.act;
    d.taps = a.size * a.size;
    d.yolo_entries = a.yolo_entries;
    d.nkb = (a.C / 16) * d.taps;
    d.OHW = a.OH * a.OW;
    const long long nt = (long long)a.B * d.OHW;
    // lane offsets are 32-bit byte offsets from the first image of a tile: a tile of <= 256 pixels spans 256 / OHW + 2 images
    if (nt > 0x7fffffffLL || (long long)a.C * a.H * a.W * 4 * (256 / d.OHW + 2) >= 0xFFFFFFF0LL ||
        (long long)a.M * d.OHW * 4 * (256 / d.OHW + 2) >= 0xFFFFFFF0LL)
        return (int)hipErrorInvalidValue;
    d.Ntotal = (int)nt;
    d.tiles_m = 0;
    hipStream_t s = (hipStream_t)stream;
    // split K: whole channel blocks per range; the partial passes run bias-free and linear into the workspace, the second stage
    // (launch_splitk_finish) applies bias, activation and the fused [shortcut]
    const int cblocks = a.C / 16;
    int ksplit = (a.ksplit > 1 && a.ks_ws && a.ks_zeros && !a.in2 && a.yolo_entries == 0) ? a.ksplit : 1;
    while (ksplit > 1 && cblocks % ksplit != 0) --ksplit;
    d.ksplit = ksplit; d.ks_in_off = 0; d.ks_w_off = 0; d.ks_out_off = 0;
    if (ksplit > 1) {
        const int cb_part = cblocks / ksplit;
        d.nkb = cb_part * d.taps;
        d.ks_in_off = (size_t)cb_part * 16 * a.H * a.W;
        d.ks_w_off = (size_t)d.nkb * 6 * d.Mpad * 16;
        d.ks_out_off = (size_t)a.B * a.M * d.OHW;
        d.out = a.ks_ws; d.add = nullptr; d.out_add = nullptr; d.bias = a.ks_zeros; d.act = YL_LINEAR;
    }
    if (tile == 0) {
        tile = a.M <= 32 ? 3 : (a.M <= 64 ? 2 : 1);
        // grids far below the chip (8 images per GPU at 19 x 19: 92 workgroups of 128 x 128 for 256 CUs): 64 x 64 tiles
        const int n_cu = device_cu_count();
        if (tile == 1 && (long long)((a.M + 127) / 128) * ((nt + 127) / 128) * ksplit < (long long)n_cu) tile = 4;
    }
    const char *t = "?";
    int rc;
    // every tile runs the pinned schedule (round 5: -8 ... -11 % per launch stand-alone on all of yolov3's direct shapes,
    // profiles/r5_ab_x3_pinned_schedule.txt; same bits); `plain` (variant bit 12, or tile 5) keeps hipcc's own order for A/B runs
    if (plain && tile == 1) tile = 5;
    switch (tile) {
    case 1: t = "128x128"; rc = launch_x3_tile<128, 128, 2, 2, true>(d, s); break;
    case 2: t = "64x128"; rc = plain ? launch_x3_tile<64, 128, 2, 2>(d, s) : launch_x3_tile<64, 128, 2, 2, true>(d, s); break;
    case 3: t = "32x256"; rc = launch_x3_tile<32, 256, 1, 4>(d, s); break;
    case 4: t = "64x64"; rc = launch_x3_tile<64, 64, 2, 1>(d, s); break;
    case 5: t = "128x128,plain"; rc = launch_x3_tile<128, 128, 2, 2>(d, s); break;
    default: return (int)hipErrorInvalidValue;
    }
    char sp[16] = "";
    if (ksplit > 1) snprintf(sp, sizeof(sp), ",split%d", ksplit);
    if (name) snprintf(name, name_len, "conv_f32_x3<%s,ks%d%s%s>", t, a.size, a.yolo_entries > 0 ? ",yolo" : (a.in2 ? ",up+route" : ""), sp);
    if (rc == 0 && ksplit > 1)
        rc = launch_splitk_finish(a.ks_ws, ksplit, d.ks_out_off, a.bias, a.B, a.M, d.OHW, a.act, a.add, a.out, a.out_add, stream);
    return rc;
}

}  // namespace yl
