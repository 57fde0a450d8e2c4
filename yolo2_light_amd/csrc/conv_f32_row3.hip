// conv_f32_row3.hip -- K1r: the 3x3 / stride-1 / pad-1 FP32 convolution as ROW-WISE Winograd F(2,3) on the BF16 matrix
// pipe, every operand the exact sum of three bf16 pieces.
//
// Same layer as conv_f32_wino32.hip and conv_f32_x3.hip (forward_convolutional_layer_cpu's FP32 branch,
// src/yolov2_forward_network.c:204-261: im2col + gemm_nn + bias + leaky; gemm_nn src/additionally.c:1272-1286), FP32
// tensors in and out.  conv_f32_wino32.hip runs F(2x2,3x3) on v_mfma_f32_32x32x2_f32, which shares the FP32 vector
// datapath with every VALU instruction on the SIMD and stalls near 0.64 of its peak; conv_f32_x3.hip moved the direct
// layers to v_mfma_f32_32x32x16_bf16 with three-piece operands (6 MFMAs per 16 k, FP32-class accuracy), but six BF16
// MFMAs per 16 k and tap do not beat 8 / 2.25 FP32 ones, and F(2x2,3x3) with three-piece operands does not fit the LDS
// (16 planes x 3 pieces).  The ONE-dimensional transform does: for an output row oy and a pair of output columns
// (2t, 2t+1), with d0..d3 = the input row iy = oy + ky - 1 at columns 2t-1 .. 2t+2 and g0..g2 = the filter row ky,
//
//     V0 = d0 - d2    V1 = d1 + d2          V2 = d2 - d1          V3 = d1 - d3          (FP32, one rounding each)
//     U0 = g0         U1 = (g0+g1+g2)/2     U2 = (g0-g1+g2)/2     U3 = g2               (double, rounded once)
//     M[xi] = sum over (c, ky) of U[xi] * V[xi]                                         (four GEMMs with K = 3 C)
//     Y(2t) = M0 + M1 + M2        Y(2t+1) = M1 - M2 - M3
//
// 4 multiplies per 2 outputs and (c, ky) instead of 6: with U and V split into three bf16 pieces each that is
// 6 / 1.5 = 4 BF16 MFMAs per 16 k and tap triple -- 136 matrix-pipe cycles where FP32 F(2x2,3x3) needs 228 and K1x 204.
// Accuracy: the transforms only add and subtract (U's halves are formed in double), the split is exact, the dropped
// cross terms a2 b3 + a3 b2 + a3 b3 are ~2^-26 |a b| rms, <= 2^-23 |a b| worst case (conv_f32_x3.hip); checked against the oracle and against a
// float64 convolution like the other FP32 kernels (tests/test_gpu_row3.py, tests/test_gpu_parity.py::test_fp32_error_vs_float64_truth).
//
//   tiles        n = (b * H + oy) * TW + tx, TW = ceil(W / 2); a workgroup owns BM filters x BT consecutive tiles x 4 planes
//   K order      groups g = (channel block c / 16, ky); a panel = PP planes of one group (PP = 2: two panels per group)
//   weights      wr[group][plane 4][piece 3][k-octet 2][Mpad][8] bf16 (row3_pack_weights / pack_row3_kernel)
//   LDS stage    A[plane][piece][k-octet][BM], B[plane][piece][k-octet][BT] 16-byte units, two stages
//   staging      a thread owns one tile and FOUR channels of every group: four 16-byte row loads (columns 2t-1 .. 2t+2;
//                the row above / below the image through the buffer range check, the columns left / right of it by
//                selects), V for the four planes, three-piece split (conv_f32_x3.hip's), one ds_write_b64 per plane and
//                piece; lane pairs write the two halves of a unit, a wave 512 contiguous bytes
//   waves        wave (wm, wn) owns TM x TN blocks of 32 filters x 32 tiles for ALL four planes: the output transform is
//                lane-local (no LDS exchange)
//   epilogue     MFMA C/D layout: Y pairs as 8-byte stores (32 lanes = 256 contiguous bytes per filter row); + bias, leaky
//                with conv_f32_mfma.hip's arithmetic, fused [shortcut]
//
// Applicability: size 3, stride 1, pad 1, C % 16 == 0, the input tensor library-owned (front pad: the left-edge tile of
// the first row reads one float in front of the tensor, selected away; the last tile of the last row up to two floats
// behind it: yl_internal.h ACT_FRONT_PAD / ACT_TAIL_PAD).
//
// Measured on MI355X, yolov3-608 batch 64: 0.83-0.87 ms per launch in the network where the FP32 Winograd kernel took
// 1.06-1.08; 0.42 of the BF16 matrix peak issued (0.52 at the 1.94 GHz the part grants); the kernel is power-bound, and operand
// traffic is what the power buys (profiles/r5_clock_power_row3_loads.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "kernels.h"
#include "../../include/yolo2_hip.h"

// Lab builds only (tools/ab_builds.sh, ABFILE=conv_f32_row3: -DYL_LAB -DX_DBG=<bits>; results are garbage by design): 1 no global loads in
// the K loop, 2 no split / B stores, 4 no A stores, 8 no MFMAs, 16 no fragment reads, 32 no epilogue stores, 64 no barriers in
// the K loop, 128 no A (weight) loads, 256 no input-row loads (+512: their transform + split stay in the loop).  The shipped
// library is built without -DYL_LAB, which forces X_DBG to 0: every guard below folds away.
#if !defined(YL_LAB)
#undef X_DBG
#endif
#ifndef X_DBG
#define X_DBG 0
#endif

namespace yl {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

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

struct ConvRow3Dev {
    const float *in;
    const void *wr;
    const float *bias;
    const float *add;       // fused [shortcut]: out_add = act(conv) + add (nullptr = none)
    float *out_add;
    float *out;             // may be nullptr when only out_add is wanted
    int B, C, H, W, M, Mpad;
    int TW;                 // tiles per image row
    int HTW;                // tiles per image
    int Ntiles;             // B * HTW
    int G;                  // groups = 3 * C / 16 (split K: of ONE channel range)
    int act;
    int tiles_m;
    // split K (gridDim.y = ranges), as in conv_f32_x3.hip; Cpart = the channels of one range (C stays the tensor's: image stride)
    int ksplit, Cpart;
    size_t ks_in_off, ks_w_off, ks_out_off;
};

// two FP32 values -> their three bf16 pieces, packed (low half = x); conv_f32_x3.hip's split
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

// The epilogue every K1r kernel shares (MFMA C/D layout): Y(2t) = (M0 + M1) + M2, Y(2t+1) = (M1 - M2) - M3, + bias, activation, FP32 NCHW,
// fused [shortcut].  acc[plane][i][j]: blocks of 32 filters x 32 tiles at (m0 + wm0 + 32 i, n0 + wn0 + 32 j).
template <int TM, int TN, bool MFULL, bool WEVEN>
__device__ __forceinline__ void row3_epilogue(const ConvRow3Dev &p, const f32x16 (&acc)[4][TM][TN], const float *bias_s, int m0, int n0, int wm0, int wn0)
{
    const int HW = p.H * p.W;
    // (the lane's coordinates are derived again from threadIdx.x: kept alive across the K loop they are what the ping-pong form,
    //  at 256 registers, spilled)
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int l31_e = tid_e & 31, half_e = (tid_e >> 5) & 1;
    const int ob_first = __builtin_amdgcn_readfirstlane((n0 + wn0) / p.HTW);
    int voff_o[TN];
    bool px1[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + l31_e;
        const int ob = n / p.HTW;
        const int orem = n - ob * p.HTW;
        const int ooy = orem / p.TW;
        const int otx = orem - ooy * p.TW;
        voff_o[j] = n < p.Ntiles ? (int)(((unsigned)(ob - ob_first) * (unsigned)p.M * (unsigned)HW + (unsigned)ooy * (unsigned)p.W +
                                          (unsigned)(2 * otx) + 4u * (unsigned)half_e * (unsigned)HW) * 4u) : -1;
        px1[j] = n < p.Ntiles && (WEVEN || (2 * otx + 1 < p.W));      // (a tile past the end has voff_o = -1: -1 + 4 would be a valid offset)
    }
    const size_t img_out = (size_t)p.M * HW;
    size_t orec = ((size_t)p.B - ob_first) * img_out * 4;
    if (orec > 0xFFFFFFFEull) orec = 0xFFFFFFFEull;
    const bool has_out = p.out != nullptr && (X_DBG & 32) == 0, has_add = p.add != nullptr && (X_DBG & 32) == 0;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(has_out ? p.out + (size_t)ob_first * img_out : (float *)p.bias), 0, has_out ? (int)(unsigned)orec : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(has_add ? p.add + (size_t)ob_first * img_out : p.bias), 0, has_add ? (int)(unsigned)orec : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_oadd = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(has_add ? p.out_add + (size_t)ob_first * img_out : (float *)p.bias), 0, has_add ? (int)(unsigned)orec : 0, 0x00020000);
    const int row_bytes = HW * 4;
    const bool leaky = p.act == YL_LEAKY;
    // one wave-uniform branch picks the output form; inside it the [shortcut] operand of a whole 32 x 32 block is requested
    // before the block's arithmetic (MODE 0: out, 1: out_add only, 2: both)
    auto epilogue = [&](auto mode_tag, auto leaky_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool LEAKY = decltype(leaky_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float bias_r[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) bias_r[e] = bias_s[wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half_e];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int e0 = 0; e0 < 16; e0 += 8) {         // eight accumulator rows at a time: their [shortcut] operands first
                    int vo[8], vo1[8];
                    f32x2 addv[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int e = e0 + k;
                        const int mrow = m0 + wm0 + i * 32 + (e & 3) + 8 * (e >> 2);
                        const bool ok = MFULL || (mrow + 4 * half_e) < p.M;
                        vo[k] = ok ? voff_o[j] : -1;
                        vo1[k] = (ok && px1[j]) ? voff_o[j] + 4 : -1;
                        if constexpr (MODE >= 1) {
                            if constexpr (WEVEN)
                                addv[k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_add, vo[k], mrow * row_bytes, 0));
                            else {
                                addv[k][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_add, vo[k], mrow * row_bytes, 0));
                                addv[k][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_add, vo1[k], mrow * row_bytes, 0));
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int e = e0 + k;
                        const float q0 = acc[0][i][j][e], q1 = acc[1][i][j][e], q2 = acc[2][i][j][e], q3 = acc[3][i][j][e];
                        float y0 = ((q0 + q1) + q2) + bias_r[e];
                        float y1 = ((q1 - q2) - q3) + bias_r[e];
                        if constexpr (LEAKY) {
                            // (float)(.1 * (double)x), conv_f32_mfma.hip's arithmetic, as three conversions / multiplies and a select:
                            // left as a ternary hipcc branches around the double-precision path for every output
                            float t0 = (float)(.1 * (double)y0), t1 = (float)(.1 * (double)y1);
                            asm volatile("" : "+v"(t0), "+v"(t1));
                            y0 = (y0 > 0.f) ? y0 : t0;
                            y1 = (y1 > 0.f) ? y1 : t1;
                        }
                        const int mrow = m0 + wm0 + i * 32 + (e & 3) + 8 * (e >> 2);
                        if constexpr (MODE != 1) {
                            if constexpr (WEVEN) {
                                const f32x2 y = {y0, y1};
                                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, y), rs_out, vo[k], mrow * row_bytes, 0);
                            } else {
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y0), rs_out, vo[k], mrow * row_bytes, 0);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y1), rs_out, vo1[k], mrow * row_bytes, 0);
                            }
                        }
                        if constexpr (MODE >= 1) {
                            const float s0 = __fadd_rn(y0, addv[k][0]), s1 = __fadd_rn(y1, addv[k][1]);
                            if constexpr (WEVEN) {
                                const f32x2 y = {s0, s1};
                                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, y), rs_oadd, vo[k], mrow * row_bytes, 0);
                            } else {
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s0), rs_oadd, vo[k], mrow * row_bytes, 0);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s1), rs_oadd, vo1[k], mrow * row_bytes, 0);
                            }
                        }
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

// BM x BT: filters x tiles of a workgroup; WM x WN waves, each TM x TN blocks of 32 x 32 for all four planes;
// PP: planes per LDS panel (1 or 2); MFULL: M % BM == 0; WEVEN: W even (Y pairs are 8-byte aligned and always whole)
// SCHED 0: one barrier at the end of a panel, the fragments of a panel are read after it (conv_f32_x3.hip's loop)
// SCHED 1 (PP = 2): the barrier sits BETWEEN the two planes of a panel -- first plane: its MFMAs run from fragments that are
//         already in registers while the next panel is split into the other stage and the second plane's fragments are
//         read; barrier; second plane: its MFMAs cover the reads of the NEXT panel's first-plane fragments.  No LDS
//         latency stands in front of an MFMA block, and no wave waits at the barrier with an idle matrix pipe behind it.
// SCHED 2: SCHED 1 with the staging work pinned into the shadows of the wave's own MFMAs (sched_group_barrier): left to itself hipcc
//         issues the ~60 VALU + 9 LDS stores of a panel as one block in front of the MFMAs, and the two waves of a SIMD -- same
//         workgroup, same barrier -- stage at the same time while the matrix pipe waits (shipped default of the 128 x 128 tile)
// LOADX2: the input rows as ONE 8-byte load per (tile, channel) -- lane = tile, a wave reads 512 contiguous bytes of a channel row --
//         and the two outer columns (2t - 1, 2t + 2) from the neighbour lanes by DPP wave_shr:1 / wave_shl:1 (lanes 0 / 63: a
//         two-lane dword load); false: one 16-byte load per (tile, channel) at an 8-byte lane stride (the first version: every
//         element fetched twice, two channels per instruction)
template <int BM, int BT, int WM, int WN, int PP, bool MFULL, bool WEVEN, int SCHED, bool LOADX2>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4 && BM * BT == 64 * 64) ? 3 : 2) void conv_f32_row3_kernel(ConvRow3Dev p)
{
    if (p.ksplit > 1) {                     // this workgroup's channel range
        const size_t r = blockIdx.y;
        p.in += r * p.ks_in_off;
        p.wr = reinterpret_cast<const char *>(p.wr) + r * p.ks_w_off;
        p.out += r * p.ks_out_off;
    }
    static_assert(SCHED == 0 || PP == 2, "the mid-panel barrier needs two planes per panel");
    // (A ping-pong form -- the two waves of a SIMD alternating between a pure-MFMA slot and a staging slot -- was built and measured in
    //  round 5: bit-identical, 30 % slower, because without LDS-DMA the staging half's requests sit on every slot's critical path;
    //  profiles/r5_ab_row3_pingpong.txt.)
    constexpr int NT = WM * WN * 64;
    static_assert(NT == 4 * BT, "one thread per (tile, channel quad)");
    static_assert(!LOADX2 || BT % 64 == 0, "a wave stages 64 consecutive tiles");
    static_assert(PP == 1 || PP == 2, "planes per panel");
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BT / (WN * 32);
    constexpr int NIT = 4 / PP;                          // panels (loop iterations) per group
    constexpr int STAGE_A = PP * 6 * BM;                 // 16-byte units
    constexpr int STAGE_B = PP * 6 * BT;
    constexpr int APT = (STAGE_A + NT - 1) / NT;
    constexpr bool A_FULL = (STAGE_A % NT) == 0;
    static_assert(NT % BM == 0, "A panel mapping");
    constexpr int A_STEP = NT / BM;                      // (plane, piece, k-octet) rows one pass of the threads covers

    __shared__ __attribute__((aligned(16))) uint4 smem[2 * STAGE_A + 2 * STAGE_B + BM / 4];
    uint4 *As = smem;
    uint4 *Bs = smem + 2 * STAGE_A;
    float *bias_s = reinterpret_cast<float *>(smem + 2 * STAGE_A + 2 * STAGE_B);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int half = lane >> 5;

    // XCD-aware tile order (blocks b, b+8, ... share an L2): consecutive logical tiles = the filter tiles of one tile range
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int qq = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
    const int logical = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    const int tile_n = __builtin_amdgcn_readfirstlane(logical / p.tiles_m);
    const int tile_m = logical - tile_n * p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BT;

    if (tid < BM) bias_s[tid] = (m0 + tid < p.M) ? p.bias[m0 + tid] : 0.f;

    // ---- staging role: tile s_tile, channels 4 q .. 4 q + 3 of every group ----
    // LOADX2: lane = tile (64 consecutive tiles per wave), the channel quad is wave-uniform; otherwise lane pairs = the two
    // channel quads of a k-octet (the two halves of a 16-byte LDS unit)
    constexpr int TWAVES = LOADX2 ? BT / 64 : BT / 32;   // waves along the tiles
    const int s_tile = LOADX2 ? (wave % TWAVES) * 64 + lane : (wave % TWAVES) * 32 + (lane >> 1);
    const int s_q = LOADX2 ? wave / TWAVES : (wave / TWAVES) * 2 + (lane & 1);
    const int s_oct = s_q >> 1;                          // k-octet
    const int s_qlo = s_q & 1;
    const int HW = p.H * p.W;
    const int n_g = n0 + s_tile;
    const bool n_ok = n_g < p.Ntiles;
    const int bimg = n_ok ? n_g / p.HTW : 0;
    const int trem = n_g - bimg * p.HTW;
    const int oy = trem / p.TW;
    const int tx = trem - oy * p.TW;
    const bool cm0 = tx > 0, cm2 = 2 * tx + 1 < p.W, cm3 = 2 * tx + 2 < p.W;

    // buffer descriptor over the input, based at the first image of this workgroup's tiles and shifted back by one row (LOADX2)
    // or one row and one column: lane offset (oy * W + 2 tx) = row oy - 1, column 2 tx (LOADX2) / 2 tx - 1; rows outside the
    // image -> voffset 0xFFFFFFFF -> 0.0
    constexpr int COL0 = LOADX2 ? 0 : 1;
    const int b_first = __builtin_amdgcn_readfirstlane(n0 / p.HTW);
    const size_t img_floats = (size_t)p.C * HW;
    const float *tile_base = p.in + (size_t)b_first * img_floats - (ptrdiff_t)(p.W + COL0);
    size_t rec = (((size_t)p.B - b_first) * img_floats + (size_t)(p.W + COL0)) * sizeof(float);
    if (rec > 0xFFFFFFFEull) rec = 0xFFFFFFFEull;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)tile_base, 0, (int)(unsigned)rec, 0x00020000);
    const int voff = (int)(((unsigned)(bimg - b_first) * (unsigned)img_floats + (unsigned)(s_q * 4) * (unsigned)HW +
                            (unsigned)oy * (unsigned)p.W + (unsigned)(2 * tx)) * 4u);
    // LOADX2: the outer column only the first / last lane of a wave cannot get from a neighbour (column 2 tx - 1 / 2 tx + 2);
    // every other lane, and a column outside the image, asks for nothing (offset -1)
    const int hvoff = !LOADX2 ? -1 : (lane == 0 ? (cm0 ? voff - 4 : -1) : (lane == 63 ? (cm3 ? voff + 8 : -1) : -1));
    unsigned nrowmask = 7u;                   // inverted row validity, bit ky
    if (n_ok) {
        unsigned m = 0;
        for (int ky = 0; ky < 3; ++ky)
            if (oy + ky - 1 >= 0 && oy + ky - 1 < p.H) m |= 1u << ky;
        nrowmask = ~m;
    }

    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.wr, 0, (int)((unsigned)p.G * 24u * (unsigned)p.Mpad * 16u), 0x00020000);
    const int a_voff = ((tid / BM) * p.Mpad + (tid % BM)) * 16;
    const int b_lds = s_oct * BT * 16 + s_tile * 16 + s_qlo * 8;        // byte offset inside a (plane, piece) slab of a B stage

    v4i a_reg[APT];
    f32x4 raw[LOADX2 ? 1 : 4];                // x4 form: columns 2t-1 .. 2t+2 of the quad's four channels
    f32x2 rw2[LOADX2 ? 4 : 1];                // x2 form: columns 2t, 2t+1
    float rwh[LOADX2 ? 4 : 1];                //          + the outer column of lanes 0 / 63
    float V[4][4];                            // [plane][channel of the quad]
    if constexpr (X_DBG != 0) {               // lab builds skip producers: give every consumer a defined value
#pragma unroll
        for (int e = 0; e < APT; ++e) a_reg[e] = v4i{0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};
#pragma unroll
        for (int i = 0; i < (LOADX2 ? 1 : 4); ++i) raw[i] = f32x4{1.f, 2.f, 3.f, 4.f};
#pragma unroll
        for (int i = 0; i < (LOADX2 ? 4 : 1); ++i) { rw2[i] = f32x2{1.f, 2.f}; rwh[i] = 3.f; }
    }
    int ld_ky = 0, ld_c0 = 0;                 // group of the NEXT raw load

    // (always issued: past the last group the lane offsets are all-ones and the range check answers without touching memory)
    auto load_raw = [&]() {
        if constexpr ((X_DBG & (1 | 256)) != 0) {
            if constexpr ((X_DBG & 512) != 0) {        // 512: keep the transform + split in the loop (the skipped loads' registers become opaque)
#pragma unroll
                for (int i = 0; i < (LOADX2 ? 1 : 4); ++i) asm volatile("" : "+v"(raw[i]));
#pragma unroll
                for (int i = 0; i < (LOADX2 ? 4 : 1); ++i) asm volatile("" : "+v"(rw2[i]), "+v"(rwh[i]));
            }
            return;
        }
        const int soff = (ld_c0 * HW + ld_ky * p.W) * 4;
        const int tinv = __builtin_amdgcn_sbfe((int)nrowmask, ld_ky, 1) | (ld_c0 < p.Cpart ? 0 : -1);
        if constexpr (LOADX2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rw2[i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff | tinv, soff + i * HW * 4, 0));
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rwh[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, hvoff | tinv, soff + i * HW * 4, 0));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                raw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff | tinv, soff + i * HW * 4, 0));
        }
        ++ld_ky;
        if (ld_ky == 3) { ld_ky = 0; ld_c0 += 16; }
    };
    auto transform = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float d0, d1, d2, d3;
            if constexpr (LOADX2) {
                const float x = rw2[i][0], y = rw2[i][1], h = rwh[i];
                // wave_shr:1 -- lane l takes lane l-1's y (column 2t-1), lane 0 keeps h; wave_shl:1 -- lane l takes lane l+1's x
                // (column 2t+2), lane 63 keeps h.  A neighbour in another row or image delivers a value the masks drop.
                const float l = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, h), __builtin_bit_cast(int, y), 0x138, 0xf, 0xf, false));
                const float r = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, h), __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, false));
                d0 = cm0 ? l : 0.f; d1 = x; d2 = cm2 ? y : 0.f; d3 = cm3 ? r : 0.f;
            } else {
                d0 = cm0 ? raw[i][0] : 0.f; d1 = raw[i][1]; d2 = cm2 ? raw[i][2] : 0.f; d3 = cm3 ? raw[i][3] : 0.f;
            }
            V[0][i] = d0 - d2;
            V[1][i] = d1 + d2;
            V[2][i] = d2 - d1;
            V[3][i] = d1 - d3;
        }
    };
    // planes xi0 .. xi0 + PP - 1 of V -> three pieces -> B stage `buf`
    auto store_b = [&](int buf, int xi0) {
        if constexpr ((X_DBG & 2) != 0) return;
#pragma unroll
        for (int pl = 0; pl < PP; ++pl) {
            unsigned a1, a2, a3, c1, c2, c3;
            split3_pair(V[xi0 + pl][0], V[xi0 + pl][1], a1, a2, a3);
            split3_pair(V[xi0 + pl][2], V[xi0 + pl][3], c1, c2, c3);
            char *dst = reinterpret_cast<char *>(Bs + buf * STAGE_B + pl * 6 * BT) + b_lds;
            *reinterpret_cast<uint2 *>(dst + 0 * 2 * BT * 16) = make_uint2(a1, c1);
            *reinterpret_cast<uint2 *>(dst + 1 * 2 * BT * 16) = make_uint2(a2, c2);
            *reinterpret_cast<uint2 *>(dst + 2 * 2 * BT * 16) = make_uint2(a3, c3);
        }
    };
    // A panel `it` (= group * NIT + panel of the group): rows (plane, piece, k-octet) x BM filters
    auto load_a = [&](int it) {
        if constexpr ((X_DBG & (1 | 128)) != 0) return;
#pragma unroll
        for (int e = 0; e < APT; ++e)
            if (A_FULL || tid + e * NT < STAGE_A)
                a_reg[e] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(
                    rs_w, a_voff, ((it * PP * 6 + e * A_STEP) * p.Mpad + m0) * 16, 0));
    };
    auto store_a = [&](int buf) {
        if constexpr ((X_DBG & 4) != 0) return;
#pragma unroll
        for (int e = 0; e < APT; ++e) {
            const int idx = tid + e * NT;
            if (A_FULL || idx < STAGE_A) As[buf * STAGE_A + idx] = __builtin_bit_cast(uint4, a_reg[e]);
        }
    };

    f32x16 acc[4][TM][TN];
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[xi][i][j][e] = 0.f;

    const int wm = wave / WN, wn = wave - wm * WN;
    const int wm0 = wm * TM * 32, wn0 = wn * TN * 32;
    const int G = p.G;

    constexpr int TA[6] = {2, 0, 1, 1, 0, 0};          // pieces of the six products, smallest terms first
    constexpr int TB[6] = {0, 2, 1, 0, 1, 0};
    struct Frags { v4i a[3][TM], b[3][TN]; };
    // fragments of plane `pl` (of the panel) from stage `buf`
    auto read_frags = [&](Frags &f, int buf, int pl) {
        if constexpr ((X_DBG & 16) != 0) return;
        int a_off = buf * STAGE_A + half * BM + wm0 + l31;
        int b_off = buf * STAGE_B + half * BT + wn0 + l31;
        asm volatile("" : "+v"(a_off), "+v"(b_off));           // one base register each, immediate offsets below
        const uint4 *Ab = As + a_off;
        const uint4 *Bb = Bs + b_off;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
            for (int i = 0; i < TM; ++i) f.a[pc][i] = __builtin_bit_cast(v4i, Ab[(pl * 3 + pc) * 2 * BM + i * 32]);
#pragma unroll
            for (int j = 0; j < TN; ++j) f.b[pc][j] = __builtin_bit_cast(v4i, Bb[(pl * 3 + pc) * 2 * BT + j * 32]);
        }
    };
    auto mfma_plane = [&](const Frags &f, int xi) {
        if constexpr ((X_DBG & 8) != 0) return;
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[xi][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, f.a[TA[t]][i]), __builtin_bit_cast(bf16x8, f.b[TB[t]][j]), acc[xi][i][j], 0, 0, 0);
    };

    int g = 0;
    // ---- prologue: group 0 -> V, its first panel -> LDS[0]; group 1 requested; A panel 1 -> registers ----
    load_a(0);
    load_raw();
    transform();
    load_raw();                               // group 1 (G >= 3)
    store_a(0);
    store_b(0, 0);
    load_a(1);
    __syncthreads();

    if constexpr (SCHED == 0) {
        // One group = NIT panels.  Panel h of group g (iteration it = g * NIT + h) computes from LDS[h & 1] while the NEXT panel
        // (h + 1 of g, or panel 0 of g + 1 -- then V is re-formed from the raw rows requested one group earlier, and the rows of
        // group g + 2 are requested) is split into LDS[(h + 1) & 1] and the A panel after that travels to registers.
#define ROW3_GROUP(LAST)                                                                           \
        _Pragma("unroll") for (int h = 0; h < NIT; ++h) {                                          \
            const int it = g * NIT + h;                                                            \
            if (h + 1 < NIT) {                                                                     \
                store_a((h + 1) & 1);                                                              \
                store_b((h + 1) & 1, (h + 1) * PP);                                                \
                if (!(LAST) || h + 2 < NIT) load_a(it + 2);                                        \
            } else if (!(LAST)) {                                                                  \
                /* the A panel is requested BEFORE the input rows: vmcnt retires in order, so the next panel's wait for */ \
                /* its A units leaves the four row loads in flight for a whole group */                \
                store_a(0);                                                                        \
                load_a(it + 2);                                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                 \
                transform();                                                                       \
                load_raw();                                                                        \
                store_b(0, 0);                                                                     \
            }                                                                                      \
            _Pragma("unroll") for (int pl = 0; pl < PP; ++pl) {                                    \
                Frags f;                                                                           \
                if constexpr ((X_DBG & 16) != 0) memset(&f, 0x3f, sizeof(f));                      \
                read_frags(f, h & 1, pl);                                                          \
                mfma_plane(f, h * PP + pl);                                                        \
            }                                                                                      \
            if ((X_DBG & 64) == 0 && (!(LAST) || h + 1 < NIT)) __syncthreads();                    \
        }
        for (; g + 1 < G; ++g) { ROW3_GROUP(false) }
        { ROW3_GROUP(true) }
#undef ROW3_GROUP
    } else {
        // SCHED 1 / 2, PP = 2: panel h of group g = planes 2 h, 2 h + 1, stage h.  f0 always holds the first-plane fragments of the panel
        // about to be computed (read behind the previous panel's barrier, under its second plane's MFMAs).
        Frags f0, f1;
        if constexpr ((X_DBG & 16) != 0) { memset(&f0, 0x3f, sizeof(f0)); memset(&f1, 0x3f, sizeof(f1)); }
        read_frags(f0, 0, 0);
#define ROW3_PANEL(H, LASTP, LOADA)                                                                \
        {                                                                                          \
            if (!(LASTP)) {                                                                        \
                store_a(((H) + 1) & 1);                                                            \
                if (LOADA) load_a(g * 2 + (H) + 2);                                                \
                if ((H) == 1) transform();                                                         \
                store_b(((H) + 1) & 1, (((H) + 1) & 1) * 2);                                       \
            }                                                                                      \
            read_frags(f1, (H), 1);                                                                \
            mfma_plane(f0, (H) * 2);                                                               \
            if (SCHED == 2 && X_DBG == 0) {                                                        \
                /* masks: 0x008 MFMA, 0x002 VALU, 0x020 VMEM read, 0x100 DS read, 0x200 DS write */      \
                _Pragma("unroll") for (int i_ = 0; i_ < 6 * TM * TN; ++i_) {                       \
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                             \
                    __builtin_amdgcn_sched_group_barrier(0x002, (H) == 1 ? 8 : 5, 0);              \
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                             \
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                             \
                    if (i_ < APT) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);               \
                }                                                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                 \
            }                                                                                      \
            if (!(LASTP)) {                                                                        \
                if ((X_DBG & 64) == 0) __syncthreads();                                            \
                read_frags(f0, ((H) + 1) & 1, 0);                                                  \
                if ((H) == 1) load_raw();                                                          \
            }                                                                                      \
            mfma_plane(f1, (H) * 2 + 1);                                                           \
            if (SCHED == 2 && X_DBG == 0) {                                                        \
                _Pragma("unroll") for (int i_ = 0; i_ < 6 * TM * TN; ++i_) {                       \
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                             \
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                             \
                    if (i_ < 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                 \
                }                                                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                 \
            }                                                                                      \
        }
        for (; g + 1 < G; ++g) {
            ROW3_PANEL(0, false, true)
            ROW3_PANEL(1, false, true)
        }
        ROW3_PANEL(0, false, false)
        ROW3_PANEL(1, true, false)
#undef ROW3_PANEL
    }

    row3_epilogue<TM, TN, MFULL, WEVEN>(p, acc, bias_s, m0, n0, wm0, wn0);
}

// ---- the "view" form of the 128 x 128 tile (TW <= 63) ----
// The B panel of group (c, ky) at tile n = (b, oy, tx) is V of input row oy + ky - 1: the SAME transformed, split row serves the tiles
// of three output rows.  The kernel above forms it three times (once per ky).  Here a workgroup stages V ONCE per 16-channel block,
// for the rows its 128 consecutive tiles touch -- entries e = 0 .. 127 + 2 TW, entry e = the row-tile n0 - TW + e of the
// (b, iy, tx) order -- and the panel of (c, ky) is a VIEW of that buffer: tile t reads entry t + ky TW (a row above / below the
// image: an all-zero entry, picked per lane).  Row loads, transforms, splits and LDS stores drop by 3 * 128 / (128 + 2 TW): 1.9 x at
// 76 x 76, 2.3 x at 38 x 38, 2.6 x at 19 x 19.  All four planes of 256 entries do not fit the LDS twice, so the K loop runs in plane
// pairs: slots (c, h, ky) -- planes 2h, 2h + 1 of channel block c against filter row ky -- with the pair's half of the buffer
// (X: planes 0, 1; Y: planes 2, 3) rewritten while the other half is read: X(c + 1) during (c, 1, *), Y(c) during (c, 0, *), V2 / V3
// kept in registers in between.  Per plane the products accumulate in the order of the other tiles: bit-identical results.
//   staging roles   round 0: entries 0 .. 127, round 1: entries 128 .. 255; wave w stages entries 32 (w & 3) + lane / 2 of a round for
//                   the channel quads of k-octet w >> 2 (lane pairs = the two halves of a 16-byte unit); a wave whose round-1
//                   entries lie past 127 + 2 TW skips the round (wave-uniform)
//   LDS             A[2 stages][plane 2][piece 3][k-octet 2][128] + V[2 halves][plane 2][piece 3][k-octet 2][256] 16-byte units = 144 KB
template <bool MFULL, bool WEVEN>
__global__ __launch_bounds__(512, 2) void conv_f32_row3v_kernel(ConvRow3Dev p)
{
    constexpr int BM = 128, BT = 128, WN = 4, TM = 2, TN = 1, NT = 512;
    constexpr int NE = 256, ZE = NE - 1;                // entries of a V slab; the all-zero entry
    constexpr int STAGE_A = 2 * 6 * BM;                 // 16-byte units
    constexpr int HALF_V = 2 * 6 * NE;
    constexpr int APT = STAGE_A / NT, A_STEP = NT / BM;
    static_assert(STAGE_A % NT == 0 && NT % BM == 0, "A panel mapping");

    __shared__ __attribute__((aligned(16))) uint4 smem[2 * STAGE_A + 2 * HALF_V + BM / 4];
    uint4 *As = smem;
    uint4 *Vs = smem + 2 * STAGE_A;
    float *bias_s = reinterpret_cast<float *>(smem + 2 * STAGE_A + 2 * HALF_V);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int half = lane >> 5;

    // XCD-aware tile order, as above
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int qq = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
    const int logical = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    const int tile_n = __builtin_amdgcn_readfirstlane(logical / p.tiles_m);
    const int tile_m = logical - tile_n * p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BT;

    if (tid < BM) bias_s[tid] = (m0 + tid < p.M) ? p.bias[m0 + tid] : 0.f;
    if (tid < 24)                                       // the zero entry of every (half, plane, piece, k-octet) slab
        Vs[(tid >> 1) * 2 * NE + (tid & 1) * NE + ZE] = make_uint4(0u, 0u, 0u, 0u);

    // ---- staging role ----
    const int HW = p.H * p.W;
    const int wcol = wave & 3;
    const int s_oct = wave >> 2, s_qlo = lane & 1, s_q = s_oct * 2 + s_qlo;
    const bool act1 = 128 + wcol * 32 < BT + 2 * p.TW;   // wave-uniform: does round 1 hold an entry a view reads?
    const int e_first = n0 - p.TW;                      // row-tile index of entry 0 (may be negative)
    const int b_first = __builtin_amdgcn_readfirstlane((e_first > 0 ? e_first : 0) / p.HTW);
    const size_t img_floats = (size_t)p.C * HW;
    // descriptor based one column in front of the first image any entry lies in: lane offset (iy * W + 2 tx) = column 2 tx - 1
    const float *tile_base = p.in + (size_t)b_first * img_floats - 1;
    size_t rec = (((size_t)p.B - b_first) * img_floats + 1) * sizeof(float);
    if (rec > 0xFFFFFFFEull) rec = 0xFFFFFFFEull;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)tile_base, 0, (int)(unsigned)rec, 0x00020000);
    int voff[2];
    unsigned cmask = 0;                                 // column validity: bits 3 r + {0: 2tx-1, 1: 2tx+1, 2: 2tx+2}
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int e = r * 128 + wcol * 32 + (lane >> 1);
        const int eg = e_first + e;
        const bool ok = eg >= 0 && eg < p.Ntiles && e != ZE;
        const int egc = ok ? eg : 0;
        const int b = egc / p.HTW;
        const int rem = egc - b * p.HTW;
        const int iy = rem / p.TW;
        const int tx = rem - iy * p.TW;
        voff[r] = ok ? (int)(((unsigned)(b - b_first) * (unsigned)img_floats + (unsigned)(s_q * 4) * (unsigned)HW +
                              (unsigned)iy * (unsigned)p.W + (unsigned)(2 * tx)) * 4u) : -1;
        cmask |= ((tx > 0 ? 1u : 0u) | (2 * tx + 1 < p.W ? 2u : 0u) | (2 * tx + 2 < p.W ? 4u : 0u)) << (3 * r);
    }
    const int v_lds = (s_oct * NE + wcol * 32 + (lane >> 1)) * 16 + s_qlo * 8;       // byte offset of the round-0 entry in a (plane, piece) slab

    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.wr, 0, (int)((unsigned)p.G * 24u * (unsigned)p.Mpad * 16u), 0x00020000);
    const int a_voff = ((tid / BM) * p.Mpad + (tid % BM)) * 16;

    v4i a_reg[APT];
    f32x4 raw[2][4];                          // [round][channel of the quad]: columns 2tx-1 .. 2tx+2
    float V[2][4][4];                         // [round][plane][channel of the quad]
    int ld_c0 = 0;                            // channel block of the NEXT row loads

    auto load_raw = [&]() {
        const int soff = ld_c0 * HW * 4;
        const int tinv = ld_c0 < p.C ? 0 : -1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            raw[0][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[0] | tinv, soff + i * HW * 4, 0));
        if (act1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                raw[1][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[1] | tinv, soff + i * HW * 4, 0));
        }
        ld_c0 += 16;
    };
    auto transform = [&](int r) {
        const bool cm0 = (cmask >> (3 * r)) & 1u, cm2 = (cmask >> (3 * r + 1)) & 1u, cm3 = (cmask >> (3 * r + 2)) & 1u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d0 = cm0 ? raw[r][i][0] : 0.f, d1 = raw[r][i][1], d2 = cm2 ? raw[r][i][2] : 0.f, d3 = cm3 ? raw[r][i][3] : 0.f;
            V[r][0][i] = d0 - d2;
            V[r][1][i] = d1 + d2;
            V[r][2][i] = d2 - d1;
            V[r][3][i] = d1 - d3;
        }
    };
    // planes 2 hh, 2 hh + 1 of round r -> three pieces -> half hh of the V buffer
    auto store_v = [&](int hh, int r) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            unsigned a1, a2, a3, c1, c2, c3;
            split3_pair(V[r][2 * hh + pl][0], V[r][2 * hh + pl][1], a1, a2, a3);
            split3_pair(V[r][2 * hh + pl][2], V[r][2 * hh + pl][3], c1, c2, c3);
            char *dst = reinterpret_cast<char *>(Vs + hh * HALF_V + pl * 6 * NE) + v_lds + r * 128 * 16;
            *reinterpret_cast<uint2 *>(dst + 0 * 2 * NE * 16) = make_uint2(a1, c1);
            *reinterpret_cast<uint2 *>(dst + 1 * 2 * NE * 16) = make_uint2(a2, c2);
            *reinterpret_cast<uint2 *>(dst + 2 * 2 * NE * 16) = make_uint2(a3, c3);
        }
    };
    // A panel pa = (c * 3 + ky) * 2 + h: rows (plane of the pair, piece, k-octet) x 128 filters
    auto load_a = [&](int pa) {
#pragma unroll
        for (int e = 0; e < APT; ++e)
            a_reg[e] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(
                rs_w, a_voff, ((pa * 12 + e * A_STEP) * p.Mpad + m0) * 16, 0));
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int e = 0; e < APT; ++e) As[buf * STAGE_A + tid + e * NT] = __builtin_bit_cast(uint4, a_reg[e]);
    };

    f32x16 acc[4][TM][TN];
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[xi][i][0][e] = 0.f;

    const int wm = wave / WN, wn = wave - wm * WN;
    const int wm0 = wm * TM * 32, wn0 = wn * TN * 32;

    // the three views of this lane's tile: entry t + ky TW, or the zero entry for a row outside the image
    int b_view[3];
    {
        const int n = n0 + wn0 + l31;
        const int rem = n % p.HTW;
        const int oy = rem / p.TW;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const bool ok = oy + ky - 1 >= 0 && oy + ky - 1 < p.H;
            b_view[ky] = half * NE + (ok ? wn0 + l31 + ky * p.TW : ZE);
        }
    }

    constexpr int TA[6] = {2, 0, 1, 1, 0, 0};          // pieces of the six products, smallest terms first
    constexpr int TB[6] = {0, 2, 1, 0, 1, 0};
    struct Frags { v4i a[3][TM], b[3]; };
    auto read_frags = [&](Frags &f, int abuf, int hh, int ky, int pl) {
        int a_off = abuf * STAGE_A + half * BM + wm0 + l31;
        int b_off = hh * HALF_V + b_view[ky];
        asm volatile("" : "+v"(a_off), "+v"(b_off));           // one base register each, immediate offsets below
        const uint4 *Ab = As + a_off;
        const uint4 *Bb = Vs + b_off;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
            for (int i = 0; i < TM; ++i) f.a[pc][i] = __builtin_bit_cast(v4i, Ab[(pl * 3 + pc) * 2 * BM + i * 32]);
            f.b[pc] = __builtin_bit_cast(v4i, Bb[(pl * 3 + pc) * 2 * NE]);
        }
    };
    auto mfma_plane = [&](const Frags &f, int xi) {
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                acc[xi][i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    __builtin_bit_cast(bf16x8, f.a[TA[t]][i]), __builtin_bit_cast(bf16x8, f.b[TB[t]]), acc[xi][i][0], 0, 0, 0);
    };

    const int CB = p.C / 16;
    int c = 0;
    // slot (c, H, KY): stage (H * 3 + KY) & 1 of A.  It stores the A panel of the next slot, requests the one after, does its share
    // of the V staging, then computes its two planes; one barrier at its end.
    //   (c, 0, 0) / (c, 0, 1): Y(c) <- the V2, V3 kept in registers, round 0 / round 1; after (c, 0, 1) the rows of block c + 1 are requested
    //   (c, 1, 0): rows of block c + 1 -> V; X(c + 1) <- V0, V1 of round 0        (c, 1, 1): ... of round 1
    auto slot = [&](auto h_tag, auto ky_tag, auto last_tag) {
        constexpr int H = decltype(h_tag)::value, KY = decltype(ky_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr int S = H * 3 + KY;
        constexpr bool HAS_NEXT = !(LAST && S == 5), HAS_NEXT2 = !(LAST && S >= 4);
        if constexpr (HAS_NEXT) store_a((S + 1) & 1);
        if constexpr (HAS_NEXT2) {
            constexpr int S2 = (S + 2) % 6;
            load_a(((c + (S + 2) / 6) * 3 + S2 % 3) * 2 + S2 / 3);
        }
        if constexpr (H == 0 && KY == 0) store_v(1, 0);
        if constexpr (H == 0 && KY == 1) {
            if (act1) store_v(1, 1);
            if constexpr (!LAST) load_raw();
        }
        if constexpr (H == 1 && KY == 0 && !LAST) {
            transform(0);
            if (act1) transform(1);
            store_v(0, 0);
        }
        if constexpr (H == 1 && KY == 1 && !LAST) {
            if (act1) store_v(0, 1);
        }
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            Frags f;
            read_frags(f, S & 1, H, KY, pl);
            mfma_plane(f, H * 2 + pl);
        }
        if constexpr (HAS_NEXT) __syncthreads();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    // ---- prologue: rows of block 0 -> V; X(0); A panel of slot 0 -> LDS, of slot 1 -> registers ----
    load_a(0);
    load_raw();
    transform(0);
    if (act1) transform(1);
    store_a(0);
    store_v(0, 0);
    if (act1) store_v(0, 1);
    load_a(2);                                 // slot (0, 0, 1): panel (0 * 3 + 1) * 2 + 0
    __syncthreads();

    for (; c + 1 < CB; ++c) {
        slot(I0{}, I0{}, std::false_type{});
        slot(I0{}, I1{}, std::false_type{});
        slot(I0{}, I2{}, std::false_type{});
        slot(I1{}, I0{}, std::false_type{});
        slot(I1{}, I1{}, std::false_type{});
        slot(I1{}, I2{}, std::false_type{});
    }
    slot(I0{}, I0{}, std::true_type{});
    slot(I0{}, I1{}, std::true_type{});
    slot(I0{}, I2{}, std::true_type{});
    slot(I1{}, I0{}, std::true_type{});
    slot(I1{}, I1{}, std::true_type{});
    slot(I1{}, I2{}, std::true_type{});

    row3_epilogue<TM, TN, MFULL, WEVEN>(p, acc, bias_s, m0, n0, wm0, wn0);
}

int launch_row3_view(ConvRow3Dev p, hipStream_t s)
{
    p.tiles_m = (p.M + 127) / 128;
    const long long blocks = (long long)p.tiles_m * ((p.Ntiles + 127) / 128);
    if (blocks <= 0 || blocks > 0x7fffffffLL || p.TW > 63) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks), block(512);
    const bool mfull = (p.M % 128) == 0, weven = (p.W & 1) == 0;
    if (mfull && weven) hipLaunchKernelGGL((conv_f32_row3v_kernel<true, true>), grid, block, 0, s, p);
    else if (mfull) hipLaunchKernelGGL((conv_f32_row3v_kernel<true, false>), grid, block, 0, s, p);
    else if (weven) hipLaunchKernelGGL((conv_f32_row3v_kernel<false, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv_f32_row3v_kernel<false, false>), grid, block, 0, s, p);
    return (int)hipGetLastError();
}

template <int BM, int BT, int WM, int WN, int PP, int SCHED, bool LOADX2 = false>
int launch_row3_tile(ConvRow3Dev p, hipStream_t s)
{
    p.tiles_m = (p.M + BM - 1) / BM;
    const long long blocks = (long long)p.tiles_m * ((p.Ntiles + BT - 1) / BT);
    if (blocks <= 0 || blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks, (unsigned)(p.ksplit > 1 ? p.ksplit : 1)), block(WM * WN * 64);
    const bool mfull = (p.M % BM) == 0, weven = (p.W & 1) == 0;
    if (mfull && weven) hipLaunchKernelGGL((conv_f32_row3_kernel<BM, BT, WM, WN, PP, true, true, SCHED, LOADX2>), grid, block, 0, s, p);
    else if (mfull) hipLaunchKernelGGL((conv_f32_row3_kernel<BM, BT, WM, WN, PP, true, false, SCHED, LOADX2>), grid, block, 0, s, p);
    else if (weven) hipLaunchKernelGGL((conv_f32_row3_kernel<BM, BT, WM, WN, PP, false, true, SCHED, LOADX2>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv_f32_row3_kernel<BM, BT, WM, WN, PP, false, false, SCHED, LOADX2>), grid, block, 0, s, p);
    return (int)hipGetLastError();
}

constexpr int ROW3_MPAD = 128;        // the widest filter tile

}  // namespace

bool row3_applicable(int C, int M, int size, int stride, int pad)
{
    return C >= 16 && (C % 16) == 0 && M >= 1 && size == 3 && stride == 1 && pad == 1;
}

// lane offsets are 32-bit byte offsets from the first image of a workgroup's tiles: 128 tiles span 128 / HTW + 2 images
bool row3_fits(int B, int C, int M, int H, int W)
{
    const long long htw = (long long)H * ((W + 1) / 2);
    const long long span = 128 / (htw > 0 ? htw : 1) + 2;
    return htw > 0 && (long long)B * htw <= 0x7fffffffLL && (long long)C * H * W * 4 * span < 0xFFFFFFF0LL && (long long)M * H * W * 4 * span < 0xFFFFFFF0LL;
}

size_t row3_packed_bytes(int C, int M)
{
    const size_t mpad = (size_t)(M + ROW3_MPAD - 1) / ROW3_MPAD * ROW3_MPAD;
    return (size_t)(C / 16) * 3 * 24 * mpad * 16;
}

// the four row-transform values of one filter row, formed in double and rounded once (pack_row3_kernel does the same)
static inline void row3_u(const float *g, float u[4])
{
    const double g0 = g[0], g1 = g[1], g2 = g[2];
    u[0] = g[0];
    u[1] = (float)((.5 * g0 + .5 * g1) + .5 * g2);
    u[2] = (float)((.5 * g0 - .5 * g1) + .5 * g2);
    u[3] = g[2];
}

// w: [M][C][3][3] (the reference's l.weights).  dst: [group = (c / 16) * 3 + ky][plane 4][piece 3][k-octet 2][Mpad][8] bf16
void row3_pack_weights(const float *w, int C, int M, void *dst)
{
    const size_t mpad = (size_t)(M + ROW3_MPAD - 1) / ROW3_MPAD * ROW3_MPAD;
    uint16_t *d = static_cast<uint16_t *>(dst);
    memset(d, 0, row3_packed_bytes(C, M));
    for (int m = 0; m < M; ++m)
        for (int c = 0; c < C; ++c)
            for (int ky = 0; ky < 3; ++ky) {
                float u[4];
                row3_u(w + ((size_t)m * C + c) * 9 + ky * 3, u);
                const size_t group = (size_t)(c / 16) * 3 + ky;
                const int oct = (c % 16) / 8, e = c % 8;
                for (int xi = 0; xi < 4; ++xi) {
                    const float a = u[xi];
                    const uint16_t h1 = bf16_rne_host(a);
                    const float r1 = a - bf16_to_float_host(h1);
                    const uint16_t h2 = bf16_rne_host(r1);
                    const float r2 = r1 - bf16_to_float_host(h2);
                    const uint16_t h3 = bf16_rne_host(r2);
                    const uint16_t hs[3] = {h1, h2, h3};
                    for (int pc = 0; pc < 3; ++pc)
                        d[((((group * 4 + xi) * 3 + pc) * 2 + oct) * mpad + m) * 8 + e] = hs[pc];
                }
            }
}

// tile: 0 = heuristic; 128x128 tiles, 8 waves: 1 = mid-panel barrier with the staging work pinned between the MFMAs, 2 = barrier at
// the panel's end, 3 = 1 with the 8-byte + DPP row loads (A/B); 128x64, 4 waves (two workgroups per CU): 4 = mid-panel barrier,
// 5 = one plane per panel; 64x64, 4 waves (three workgroups per CU): 7 = pinned, 9 = end barrier, 6 = 7 with the 8-byte + DPP row
// loads; 8 = 64x128, 8 waves, pinned; 10 = 128x128 in the view form (tile 1 on maps wider than 126).  view: the heuristic
// picks 10 wherever it would pick 1 (variant bit 13)
int launch_conv_f32_row3(const ConvF32Args &a, int tile, void *stream, char *name, size_t name_len, bool view)
{
    if (!a.row3_w || !row3_applicable(a.C, a.M, a.size, a.stride, a.pad) || a.OH != a.H || a.OW != a.W || !a.in_front_pad ||
        (!a.out && !a.add) || (a.add && !a.out_add) || a.q_out || a.bits_out || a.pool_out || a.yolo_entries > 0)
        return (int)hipErrorInvalidValue;
    ConvRow3Dev d;
    d.in = a.in; d.wr = a.row3_w; d.bias = a.bias; d.out = a.out; d.add = a.add; d.out_add = a.out_add;
    d.B = a.B; d.C = a.C; d.H = a.H; d.W = a.W; d.M = a.M;
    d.Mpad = (a.M + ROW3_MPAD - 1) / ROW3_MPAD * ROW3_MPAD;
    d.TW = (a.W + 1) / 2;
    d.HTW = a.H * d.TW;
    d.G = (a.C / 16) * 3;
    d.act = a.act;
    const long long nt = (long long)a.B * d.HTW;
    if (!row3_fits(a.B, a.C, a.M, a.H, a.W)) return (int)hipErrorInvalidValue;
    d.Ntiles = (int)nt;
    d.tiles_m = 0;
    hipStream_t s = (hipStream_t)stream;
    // split K (conv_f32_x3.hip has the same): whole channel blocks per range, partial passes bias-free and linear into the workspace
    const int cblocks = a.C / 16;
    int ksplit = (a.ksplit > 1 && a.ks_ws && a.ks_zeros) ? a.ksplit : 1;
    while (ksplit > 1 && (cblocks % ksplit != 0 || cblocks / ksplit < 2)) --ksplit;
    d.ksplit = ksplit; d.Cpart = a.C; d.ks_in_off = 0; d.ks_w_off = 0; d.ks_out_off = 0;
    if (ksplit > 1) {
        const int cb_part = cblocks / ksplit;
        d.Cpart = cb_part * 16;
        d.G = cb_part * 3;
        d.ks_in_off = (size_t)d.Cpart * a.H * a.W;
        d.ks_w_off = (size_t)d.G * 24 * d.Mpad * 16;
        d.ks_out_off = (size_t)a.B * a.M * a.H * a.W;
        d.out = a.ks_ws; d.add = nullptr; d.out_add = nullptr; d.bias = a.ks_zeros; d.act = YL_LINEAR;
        if (tile == 10) tile = 1;                      // the view form has no range loop
    }
    if (tile == 0) {
        // measured on MI355X at batch 64 (profiles/r5_sweep_row3_tiles_b64.txt, ms for [256,128,76^2] | [512,256,38^2] | [1024,512,19^2]):
        // 128x128 end-barrier 0.797 | 0.752 | 0.771, mid-barrier 0.837 | 0.773 | 0.809, one plane per panel 0.883 | 0.811 | 0.835,
        // 128x64 0.850 | 0.831 | 0.859, 64x64 0.954 | 0.928 | 1.009; with 64 filters ([64,32,304^2]) 64x64 1.178, 64x128 1.287.
        // On grids that do not fill the chip the work of the busiest CU decides: workgroups per CU x tile area x the tile's
        // relative cost above.
        const int n_cu = device_cu_count();
        // cost of a tile in units of one 128 x 128 workgroup's life: W = a workgroup of that tile alone on a CU, f[k-1] = its slowdown
        // with k of them resident (128 x 64: two fit a CU, 64 x 64: three); beyond that, rounds.  Fitted on both regimes -- yolov3-608 at
        // batch 64 (3 ... 91 workgroups per CU: 128x128 wins everywhere) and grids below the chip (yolov3-tiny 416 at batch 32,
        // yolov3 at 8 images: profiles/r5_sweep_row3_tiles_b64.txt, r5_small_grid_tiles_b8.txt, r5_sweep_tiny_416_b32_tiles.txt), where a
        // layer's time is ONE workgroup's K loop and the question is only how many of them share a CU.
        auto cost = [&](int bm, int bt, double W, int resident, const double *f) {
            const long long nwg = (long long)((a.M + bm - 1) / bm) * ((nt + bt - 1) / bt) * ksplit;     // (every range a workgroup set of 1 / ksplit the K loop: the same total)
            const long long per = (nwg + n_cu - 1) / n_cu;                  // workgroups on the busiest CU
            if (per <= resident) return W * f[per - 1];
            // more than fit at once: rounds -- whole ones for the busiest CU, fractional ones on average (the dispatcher refills a CU as
            // soon as a workgroup leaves, and the tail runs with fewer neighbours): the mean of the two matches the sweeps
            const double whole = (double)((per + resident - 1) / resident), mean = (double)nwg / ((double)n_cu * resident);
            return (resident == 1 ? whole : 0.5 * (whole + mean)) * W * f[resident - 1];
        };
        static const double f128[1] = {1.0}, f128x64[2] = {1.0, 1.46}, f64[3] = {1.0, 1.4, 2.1};
        if (a.M <= 64) tile = 7;
        else {
            const double c128 = cost(128, 128, 1.0, 1, f128), c64t = cost(128, 64, 0.79, 2, f128x64), c64 = cost(64, 64, 0.53, 3, f64);
            tile = (c128 <= c64t && c128 <= c64) ? 1 : (c64t <= c64 ? 4 : 7);
            if (tile == 1 && view && d.TW <= 63 && ksplit == 1) tile = 10;
            // (a finer 64x32 tile was measured at 8 images per GPU and gains nothing: below one workgroup per CU a layer's time
            //  is one workgroup's K loop -- 96 groups at 19 x 19 -- whatever the tile)
        }
    }
    const char *t = "?";
    int rc;
    switch (tile) {
    case 1: t = "128x128t,pipe"; rc = launch_row3_tile<128, 128, 2, 4, 2, 2>(d, s); break;
    case 2: t = "128x128t,end"; rc = launch_row3_tile<128, 128, 2, 4, 2, 0>(d, s); break;
    case 3: t = "128x128t,pipe,x2"; rc = launch_row3_tile<128, 128, 2, 4, 2, 2, true>(d, s); break;
    case 4: t = "128x64t,mid"; rc = launch_row3_tile<128, 64, 2, 2, 2, 1>(d, s); break;
    case 5: t = "128x64t,pp1"; rc = launch_row3_tile<128, 64, 2, 2, 1, 0>(d, s); break;
    case 6: t = "64x64t,pipe,x2"; rc = launch_row3_tile<64, 64, 2, 2, 2, 2, true>(d, s); break;
    case 7: t = "64x64t,pipe"; rc = launch_row3_tile<64, 64, 2, 2, 2, 2>(d, s); break;
    case 8: t = "64x128t,pipe"; rc = launch_row3_tile<64, 128, 2, 4, 2, 2>(d, s); break;
    case 9: t = "64x64t,end"; rc = launch_row3_tile<64, 64, 2, 2, 2, 0>(d, s); break;
    case 10:
        if (d.TW > 63) { t = "128x128t,pipe"; rc = launch_row3_tile<128, 128, 2, 4, 2, 2>(d, s); }       // a view spans 128 + 2 TW <= 254 entries
        else { t = "128x128t,view"; rc = launch_row3_view(d, s); }
        break;
    default: return (int)hipErrorInvalidValue;
    }
    char sp[16] = "";
    if (ksplit > 1) snprintf(sp, sizeof(sp), ",split%d", ksplit);
    if (name) snprintf(name, name_len, "conv_f32_row3<%s%s>", t, sp);
    if (rc == 0 && ksplit > 1)
        rc = launch_splitk_finish(a.ks_ws, ksplit, d.ks_out_off, a.bias, a.B, a.M, a.H * a.W, a.act, a.add, a.out, a.out_add, stream);
    return rc;
}

}  // namespace yl
