// conv_f32_first.hip -- K1f: FP32 3x3 / stride 1 / pad 1 convolution of an RGB (C <= 3) image into AT MOST 32 filters
// on the VALU: the first layer of yolov3 (3 -> 32), yolov3-tiny and tiny-yolo-obj_xnor (3 -> 16); K = 27.
//
// Same function and the same arithmetic order as conv_f32_smallk.hip / conv_f32_mfma.hip (forward_convolutional_layer_cpu
// FP32 branch, src/yolov2_forward_network.c:204-261): per output an fma chain over k = (c, ky, kx) ascending, then
// + bias, then the activation -- v_mfma_f32_32x32x2_f32 is that fmaf chain bit for bit, so this kernel returns the
// bits of the MFMA kernels (tests/test_gpu_parity.py::test_conv_first_layer_kernel_bit_identical).
//
// Why not the MFMA: the work per pixel is only 27 x M fmas, while the MFMA form (K1s) pays ~14 VALU per pixel for
// decode, 27 dword gathers and a cross-lane epilogue next to the matrix cycles (PMC, DESIGN.md K1s: VALU-bound), and
// with 16 filters half of every 32-row MFMA is empty.  Here a lane owns FOUR consecutive output pixels of a row and
// every filter: passes of 8 filters x 4 pixels = 32 accumulators, a 3 x 6 input window per channel (three 16-byte
// loads + DPP neighbour exchange), weights out of LDS as broadcast reads (one k-row of 8 weights = 2 ds_read_b128 feeds
// 32 v_fmac).  Outputs leave without any cross-lane traffic, whichever the next layer wants:
//   * FP32 rows: one 16-byte store per filter (1 KB contiguous per wave instruction),
//   * the sign words of the filters for an XNOR convolution behind the layer (bit m = filter m, built in registers),
//   * the int8 units act_q[B][M/16][H][W][16] for an INT8 convolution behind the layer (64 contiguous bytes per lane).
// Measured (MI355X, profiles/r3_bench_layers_*): tiny-yolo-xnor 128 x 416 x 416 -> sign words 0.75 ms (K1s) -> 0.30 ms;
// yolov3 64 x 608 x 608 -> int8 only 0.88 -> 0.76 ms; -> FP32 rows 1.07 -> 0.98 ms (3 GB of stores: HBM-write-bound).
//
// Three things hipcc had to be talked out of (each cost 0.5-1.2 KB of scratch per lane until found; ISA-checked by
// tests/test_isa_lint.py): LICM hoisting the window loads out of the pass loop, the scheduler lifting all 27 LDS row
// reads to the top of the unrolled body, and -- the real one -- code SINKING of every fma chain into the conditional
// store block of its consumer.  See the comments at the asm("") statements below.
#include <hip/hip_runtime.h>
#include <cstdio>

#include "kernels.h"
#include "epilogue.h"
#include "../../include/yolo2_hip.h"

namespace yl {

typedef float s4f __attribute__((ext_vector_type(4)));

namespace {

struct ConvFirstDev {
    const float *in;
    const float *wt;       // k-major packed [Kpad][Mpad], K order (c, ky, kx); columns >= M are zero
    const float *bias;
    float *out;            // [B][M][H][W] or nullptr
    uint64_t *bits_out;    // [B][1][H][W] sign words or nullptr
    float *pool_out;       // fused [maxpool] 2x2 / stride 2 behind the layer: [B][M][H/2][W/2] or nullptr (pool kernel only)
    int8_t *q_out;         // act_q[B][q_G][H][W][16] int8 or nullptr
    float q_mult;
    int q_G;
    int B, H, W, M, Mpad, act;
    int Wq;                // groups of 4 output pixels per row
    long long total;       // B * H * Wq lanes
    unsigned rec;          // bytes of the input tensor
};

}  // namespace

// one channel's window of a lane: R rows x 6 columns around its four pixels (R = 3; 4 in the pooling kernel, whose
// lanes own two output rows)
template <int R>
__device__ __forceinline__ void first_load_window(float (&win)[R][6], __amdgpu_buffer_rsrc_t rsrc, const int (&voff)[R],
                                                  const int (&eoff)[R], int soff, bool has_l, bool has_r)
{
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const s4f v = __builtin_bit_cast(s4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[r], soff, 0));
        const float e = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, eoff[r], soff, 0));
        // wave_shr:1 -- lane i takes lane i-1's .w, lane 0 keeps `e`; wave_shl:1 -- lane i takes lane i+1's .x
        // (element copies first: __builtin_bit_cast applied directly to `v.w` read element 0 with this hipcc)
        const float vx = v.x, vw = v.w;
        const float l = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, e), __builtin_bit_cast(int, vw), 0x138, 0xf, 0xf, false));
        const float rr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, e), __builtin_bit_cast(int, vx), 0x130, 0xf, 0xf, false));
        win[r][0] = has_l ? l : 0.f;
        win[r][1] = v.x; win[r][2] = v.y; win[r][3] = v.z; win[r][4] = v.w;
        win[r][5] = has_r ? rr : 0.f;
    }
}

// SIGNS: only the sign words are wanted (FP32 first layer -> [maxpool] -> XNOR convolution, tiny-yolo-obj_xnor.cfg): the
// activation is not evaluated -- linear and leaky keep the sign, (x > 0) == (leaky(x) > 0) including +-0 -- and no FP32
// row is stored; the bit of filter m is (acc + bias > 0), the same float sum the full epilogue activates.
template <int C, int MP, bool Q, bool SIGNS = false>
__global__ __launch_bounds__(256, Q ? 3 : 4) void conv_f32_first_kernel(ConvFirstDev p)
{
    static_assert(!(Q && SIGNS), "sign words or int8 units");
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = idx < p.total;
    const int q = (int)(idx % p.Wq);
    const long long t = idx / p.Wq;
    const int oy = (int)(t % p.H);
    const int b = live ? (int)(t / p.H) : 0;
    const int ox0 = 4 * q;
    const int HW = p.H * p.W;

    // Window addressing.  A lane needs columns ox0-1 .. ox0+4 of three rows: ONE 16-byte load per row brings
    // ox0 .. ox0+3 (64 lanes x 16 B contiguous), the two outer columns are the neighbour lanes' .w / .x (DPP wave
    // shifts), zero at the ends of an image row, and a 2-lane load for lanes 0 and 63 whose neighbour is in another
    // wave.  (First version: 18 dword loads per row triple at a 16-byte lane stride -- 108 quarter-rate texture
    // instructions per wave, which bounded the kernel: 0.585 ms on 128 x 416 x 416.)  Rows outside the image (and
    // dead lanes) get voffset -1: the range check of the descriptor returns 0.0 = the zero padding.
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, (int)p.rec, 0x00020000);
    const int lane = threadIdx.x & 63;
    const bool edge = lane == 0 || lane == 63;
    const bool has_l = q > 0, has_r = q < p.Wq - 1;
    int voff[3], eoff[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int iy = oy - 1 + r;
        const bool rok = live && iy >= 0 && iy < p.H;
        const unsigned base = (((unsigned)b * C) * (unsigned)HW + (unsigned)(iy * p.W + ox0)) * 4u;
        voff[r] = rok ? (int)base : -1;
        // lane 0 fetches its left neighbour, lane 63 its right one (one wave owns >= 2 lanes, so never both)
        // (every other lane: voffset -1, no memory access.  Unconditional on purpose -- a branch around the load
        // splits the unrolled body into blocks and hipcc's register allocation falls apart again)
        eoff[r] = !rok || !edge ? -1 : lane == 0 ? (has_l ? (int)(base - 4u) : -1) : (has_r ? (int)(base + 16u) : -1);
    }

    // The 27 x 16 weights sit in LDS (1.7 KB per workgroup, filled once) and reach the fmas as VGPR operands through
    // broadcast ds_read_b128 (every lane reads the same address: no bank conflict, in-order returns).  Scalar
    // operands (s_load_dwordx16 per k-row) were the first plan: hipcc hoisted all 27 row loads to the top of the
    // unrolled body -- 432 scalar registers, spilled to VGPR lanes, 7 000-22 000 v_readlane per kernel.
    __shared__ __attribute__((aligned(16))) float wl[9 * C * MP];
    for (int i = threadIdx.x; i < 9 * C * MP; i += 256) wl[i] = p.wt[(size_t)(i / MP) * p.Mpad + (i % MP)];
    __syncthreads();

    unsigned word[4] = {0u, 0u, 0u, 0u};
    const size_t pix = (size_t)oy * p.W + ox0;
    // The input windows of the lane's four pixels: 3 rows x 6 columns per channel.  RES: loaded ONCE and kept in
    // registers through all passes (54 values; one exposed memory latency per lane).  The int8-output kernels do not
    // have the registers for that next to their quantiser (168 + scratch at three waves per SIMD) and re-read 18
    // values per pass and channel (L1 hits after the first pass).
    constexpr bool RES = !Q;
    float wina[RES ? C : 1][3][6];
    if (RES) {
#pragma unroll
        for (int c = 0; c < C; ++c) first_load_window<3>(wina[RES ? c : 0], rsrc, voff, eoff, c * HW * 4, has_l, has_r);
    }
    unsigned qheld[4][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}, {0u, 0u}};   // bytes 0-7 of the open 16-channel int8 unit
    // Passes of 8 filters: 32 accumulators + 18 window values + two k-rows of 8 weights stay far below the register
    // budget of four waves per SIMD.  The loop is NOT unrolled: one body, MP / 8 trips.
    const int passes = (p.M + 7) >> 3;
#pragma unroll 1
    for (int mh = 0; mh < passes; ++mh) {
        // (v_pk_fma_f32 over filter pairs -- 432 packed instead of 864 scalar fmas, weight pair x broadcast pixel -- was
        // built and measured: bit-identical and SLOWER, 0.365 vs 0.304 ms on 128 x 416 x 416; the scalar chain stays)
        float acc[8][4];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int px = 0; px < 4; ++px) acc[m][px] = 0.f;
        const float *wh = wl + mh * 8;
        // row k+1 is read ahead of the 32 fmas of row k, and sched_barriers fence each step: without them the scheduler
        // lifts all 27 row reads (216 registers) to the top of the unrolled body and the allocator spills them
        float4 n0 = *reinterpret_cast<const float4 *>(wh), n1 = *reinterpret_cast<const float4 *>(wh + 4);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (!RES) {
                // the channel offset is laundered through an empty asm: otherwise LICM proves the loads invariant in
                // `mh`, hoists them in front of the pass loop and the kernel is back at 54 resident values
                int soff = c * HW * 4;
                asm volatile("" : "+s"(soff));
                first_load_window<3>(wina[0], rsrc, voff, eoff, soff, has_l, has_r);
            }
            const float (&win)[3][6] = wina[RES ? c : 0];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int k = c * 9 + ky * 3 + kx;
                    const int kn = k + 1 < 9 * C ? k + 1 : k;
                    const float w[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
                    {
                        int wo = kn * MP;                       // laundered as well: keeps row k+1's read HERE
                        asm volatile("" : "+v"(wo));
                        n0 = *reinterpret_cast<const float4 *>(wh + wo);
                        n1 = *reinterpret_cast<const float4 *>(wh + wo + 4);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < 8; ++m)
#pragma unroll
                        for (int px = 0; px < 4; ++px) acc[m][px] = __fmaf_rn(w[m], win[ky][kx + px], acc[m][px]);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        // Pin the accumulators HERE.  Their only consumers sit under `if (live)` / `if (m < p.M)`, and LLVM's code
        // sinking moved every fma chain down into the per-filter store block of its consumer -- leaving all 27 weight
        // rows and 54 window values live across the whole body (0.5-1.2 KB of scratch in every earlier build).
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int px = 0; px < 4; ++px) asm volatile("" : "+v"(acc[m][px]));
        // finish the 32 values in place: + bias, activation (wave-uniform guards)
#pragma unroll
        for (int ml = 0; ml < 8; ++ml) {
            const int m = mh * 8 + ml;
            const float bv = m < p.M ? p.bias[m] : 0.f;
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                float v = acc[ml][px] + bv;
                if (!SIGNS && p.act == YL_LEAKY) v = (v > 0.f) ? v : (float)(.1 * (double)v);
                acc[ml][px] = v;
                if (m < p.M) word[px] |= (v > 0.f ? 1u : 0u) << m;
            }
        }
        if (!SIGNS && live && p.out) {
#pragma unroll
            for (int ml = 0; ml < 8; ++ml) {
                const int m = mh * 8 + ml;
                if (m < p.M)
                    *reinterpret_cast<float4 *>(p.out + ((size_t)b * p.M + m) * HW + pix) =
                        make_float4(acc[ml][0], acc[ml][1], acc[ml][2], acc[ml][3]);
            }
        }
        if (Q) {
            // int8 side output act_q[B][q_G][H][W][16] for an INT8 convolution behind the layer: a lane holds 8
            // consecutive channels of 4 consecutive pixels per pass = half a unit each; every second pass writes the
            // four complete units (64 contiguous bytes per lane).  Fast quantisation = trunc + clamp; the
            // `int16_t = float` wrap corner is detected per wave and redone exactly (epilogue.h, store_q_from_cd).
            unsigned qb[4][2];
            float tmax = 0.f;
#pragma unroll
            for (int px = 0; px < 4; ++px)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    int c[4];
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const float t = __fmul_rn(acc[h * 4 + r4][px], p.q_mult);
                        tmax = fmaxf(tmax, fabsf(t));
                        const int ci = (int)t;
                        c[r4] = ci < -127 ? -127 : (ci > 127 ? 127 : ci);
                    }
                    qb[px][h] = __builtin_amdgcn_perm((unsigned)c[1], (unsigned)c[0], 0x0C0C0400u) |
                                __builtin_amdgcn_perm((unsigned)c[3], (unsigned)c[2], 0x04000C0Cu);
                }
            if (__builtin_amdgcn_ballot_w64(!(tmax < 32768.f)) != 0ull) {
#pragma unroll
                for (int px = 0; px < 4; ++px)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        unsigned w = 0;
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4)
                            w |= ((unsigned)(quantize_input_i8(acc[h * 4 + r4][px], p.q_mult) & 0xFF)) << (8 * r4);
                        qb[px][h] = w;
                    }
            }
            if (mh & 1) {
                if (live) {
                    int8_t *o = p.q_out + (((size_t)b * p.q_G + (mh >> 1)) * HW + pix) * 16;
#pragma unroll
                    for (int px = 0; px < 4; ++px)
                        *reinterpret_cast<uint4 *>(o + px * 16) = make_uint4(qheld[px][0], qheld[px][1], qb[px][0], qb[px][1]);
                }
            } else {
#pragma unroll
                for (int px = 0; px < 4; ++px) { qheld[px][0] = qb[px][0]; qheld[px][1] = qb[px][1]; }
            }
        }
    }
    if (live && p.bits_out) {
        uint64_t *o = p.bits_out + (size_t)b * HW + pix;
        *reinterpret_cast<ulonglong2 *>(o) = make_ulonglong2((uint64_t)word[0], (uint64_t)word[1]);
        *reinterpret_cast<ulonglong2 *>(o + 2) = make_ulonglong2((uint64_t)word[2], (uint64_t)word[3]);
    }
}


// K1f with the 2x2 / stride-2 [maxpool] behind the layer folded in (forward_maxpool_layer_cpu,
// src/additionally.c:1448-1482; H, W even: window origin 0 and no out-of-range taps).  yolov3-tiny's first layer
// wrote 354 MB of FP32 per batch of 32 that the pooling kernel read straight back (0.14 of the 1.84 ms step).  Here a
// lane owns a 2-row x 4-column patch of the convolution output = two pooling windows: the two rows are computed one
// after the other from a resident 4-row window (the same fma chains in the same order as the kernel above => the same
// bits), the pooled pair leaves as one 8-byte store per filter (64 lanes x 8 B contiguous), and the full-resolution
// rows are written only when something else reads them (p.out != nullptr).
template <int C, int MP>
__global__ __launch_bounds__(256, 3) void conv_f32_first_pool_kernel(ConvFirstDev p)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = idx < p.total;
    const int PH = p.H >> 1, PW = p.W >> 1;
    const int q = (int)(idx % p.Wq);
    const long long t = idx / p.Wq;
    const int oyp = (int)(t % PH);
    const int b = live ? (int)(t / PH) : 0;
    const int oy0 = 2 * oyp, ox0 = 4 * q;
    const int HW = p.H * p.W;

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, (int)p.rec, 0x00020000);
    const int lane = threadIdx.x & 63;
    const bool edge = lane == 0 || lane == 63;
    const bool has_l = q > 0, has_r = q < p.Wq - 1;
    int voff[4], eoff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int iy = oy0 - 1 + r;
        const bool rok = live && iy >= 0 && iy < p.H;
        const unsigned base = (((unsigned)b * C) * (unsigned)HW + (unsigned)(iy * p.W + ox0)) * 4u;
        voff[r] = rok ? (int)base : -1;
        eoff[r] = !rok || !edge ? -1 : lane == 0 ? (has_l ? (int)(base - 4u) : -1) : (has_r ? (int)(base + 16u) : -1);
    }

    __shared__ __attribute__((aligned(16))) float wl[9 * C * MP];
    for (int i = threadIdx.x; i < 9 * C * MP; i += 256) wl[i] = p.wt[(size_t)(i / MP) * p.Mpad + (i % MP)];
    __syncthreads();

    float wina[C][4][6];
#pragma unroll
    for (int c = 0; c < C; ++c) first_load_window<4>(wina[c], rsrc, voff, eoff, c * HW * 4, has_l, has_r);

    const size_t pix = (size_t)oy0 * p.W + ox0;
    const size_t ppix = (size_t)oyp * PW + 2 * q;
    // passes of FOUR filters (the kernel above: eight): 72 resident window values + 16 accumulators + 8 running maxima
    // stay below the 168 registers of three waves per SIMD
    constexpr int FP = 4;
    const int passes = (p.M + FP - 1) / FP;
#pragma unroll 1
    for (int mh = 0; mh < passes; ++mh) {
        float hp[FP][2];                                  // running max of the two windows, per filter of the pass
        const float *wh = wl + mh * FP;
#pragma unroll
        for (int ry = 0; ry < 2; ++ry) {
            float acc[FP][4];
#pragma unroll
            for (int m = 0; m < FP; ++m)
#pragma unroll
                for (int px = 0; px < 4; ++px) acc[m][px] = 0.f;
            int w0 = 0;
            asm volatile("" : "+v"(w0));                  // keeps the first row's read inside this ry block
            float4 n0 = *reinterpret_cast<const float4 *>(wh + w0);
#pragma unroll
            for (int c = 0; c < C; ++c) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int k = c * 9 + ky * 3 + kx;
                        const int kn = k + 1 < 9 * C ? k + 1 : k;
                        const float w[FP] = {n0.x, n0.y, n0.z, n0.w};
                        {
                            int wo = kn * MP;
                            asm volatile("" : "+v"(wo));
                            n0 = *reinterpret_cast<const float4 *>(wh + wo);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int m = 0; m < FP; ++m)
#pragma unroll
                            for (int px = 0; px < 4; ++px) acc[m][px] = __fmaf_rn(w[m], wina[c][ry + ky][kx + px], acc[m][px]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
#pragma unroll
            for (int m = 0; m < FP; ++m)
#pragma unroll
                for (int px = 0; px < 4; ++px) asm volatile("" : "+v"(acc[m][px]));      // no sinking of the chains (see above)
#pragma unroll
            for (int ml = 0; ml < FP; ++ml) {
                const int m = mh * FP + ml;
                const float bv = m < p.M ? p.bias[m] : 0.f;
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    float v = acc[ml][px] + bv;
                    if (p.act == YL_LEAKY) v = (v > 0.f) ? v : (float)(.1 * (double)v);
                    acc[ml][px] = v;
                }
                // the reference's scan order: rows, then columns, `if (val > max) max = val` from -FLT_MAX
#pragma unroll
                for (int w2 = 0; w2 < 2; ++w2) {
                    float mx = ry == 0 ? -3.402823466e+38f : hp[ml][w2];
                    mx = (acc[ml][2 * w2] > mx) ? acc[ml][2 * w2] : mx;
                    mx = (acc[ml][2 * w2 + 1] > mx) ? acc[ml][2 * w2 + 1] : mx;
                    hp[ml][w2] = mx;
                }
            }
            if (live && p.out) {
#pragma unroll
                for (int ml = 0; ml < FP; ++ml) {
                    const int m = mh * FP + ml;
                    if (m < p.M)
                        *reinterpret_cast<float4 *>(p.out + ((size_t)b * p.M + m) * HW + pix + (size_t)ry * p.W) =
                            make_float4(acc[ml][0], acc[ml][1], acc[ml][2], acc[ml][3]);
                }
            }
        }
        if (live) {
#pragma unroll
            for (int ml = 0; ml < FP; ++ml) {
                const int m = mh * FP + ml;
                if (m < p.M)
                    *reinterpret_cast<float2 *>(p.pool_out + ((size_t)b * p.M + m) * (size_t)(PH * PW) + ppix) = make_float2(hp[ml][0], hp[ml][1]);
            }
        }
    }
}

bool first_layer_valu_applicable(const ConvF32Args &a)
{
    const long long in_bytes = (long long)a.B * a.C * a.H * a.W * 4;
    return (a.W & 3) == 0 && a.W >= 8 && !a.tapmajor && a.size == 3 && a.stride == 1 && a.pad == 1 && a.C >= 1 && a.C <= 3 && a.M >= 1 && a.M <= 32 &&
           a.OH == a.H && a.OW == a.W && a.Mpad >= (a.M <= 16 ? 16 : 32) && a.Kpad >= 9 * a.C && (!a.q_out || a.M % 16 == 0) && !a.add && a.yolo_entries == 0 &&
           in_bytes < 0xFFFFFFFELL && (a.act == YL_LINEAR || a.act == YL_LEAKY) &&
           (!a.pool_out || (!a.q_out && !a.bits_out && (a.H & 1) == 0));          // fused [maxpool]: FP32 outputs, whole windows
}

int launch_conv_f32_first(const ConvF32Args &a, void *stream, char *name, size_t name_len)
{
    if (!first_layer_valu_applicable(a)) return (int)hipErrorInvalidValue;
    ConvFirstDev d;
    d.in = a.in; d.wt = a.wt; d.bias = a.bias; d.out = a.out; d.bits_out = a.bits_out; d.pool_out = a.pool_out;
    d.q_out = a.q_out; d.q_mult = a.q_mult; d.q_G = a.q_G;
    d.B = a.B; d.H = a.H; d.W = a.W; d.M = a.M; d.Mpad = a.Mpad; d.act = a.act;
    d.Wq = (a.W + 3) / 4;
    d.total = (long long)a.B * (a.pool_out ? a.H / 2 : a.H) * d.Wq;          // pool kernel: a lane owns two output rows
    d.rec = (unsigned)((long long)a.B * a.C * a.H * a.W * 4);
    const long long blocks = (d.total + 255) / 256;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t s = (hipStream_t)stream;
    const bool wide = a.M > 16;
    if (a.pool_out) {
#define YL_FIRST_POOL(CC)                                                                                   \
    do {                                                                                                    \
        if (wide) hipLaunchKernelGGL((conv_f32_first_pool_kernel<CC, 32>), grid, block, 0, s, d);           \
        else hipLaunchKernelGGL((conv_f32_first_pool_kernel<CC, 16>), grid, block, 0, s, d);                \
    } while (0)
        switch (a.C) {
        case 1: YL_FIRST_POOL(1); break;
        case 2: YL_FIRST_POOL(2); break;
        default: YL_FIRST_POOL(3); break;
        }
#undef YL_FIRST_POOL
        if (name) snprintf(name, name_len, "conv_f32_first<valu,2x4px,m%d,%s>", wide ? 32 : 16, a.out ? "pool+" : "pool");
        return (int)hipGetLastError();
    }
#define YL_FIRST_LAUNCH(CC, MPP, QQ) hipLaunchKernelGGL((conv_f32_first_kernel<CC, MPP, QQ>), grid, block, 0, s, d)
    const bool signs = a.bits_out && !a.out && !a.q_out;
#define YL_FIRST_C(CC)                                                                             \
    do {                                                                                           \
        if (a.q_out) { if (wide) YL_FIRST_LAUNCH(CC, 32, true); else YL_FIRST_LAUNCH(CC, 16, true); }   \
        else if (signs) {                                                                          \
            if (wide) hipLaunchKernelGGL((conv_f32_first_kernel<CC, 32, false, true>), grid, block, 0, s, d);   \
            else hipLaunchKernelGGL((conv_f32_first_kernel<CC, 16, false, true>), grid, block, 0, s, d);        \
        } else { if (wide) YL_FIRST_LAUNCH(CC, 32, false); else YL_FIRST_LAUNCH(CC, 16, false); }      \
    } while (0)
    switch (a.C) {
    case 1: YL_FIRST_C(1); break;
    case 2: YL_FIRST_C(2); break;
    default: YL_FIRST_C(3); break;
    }
#undef YL_FIRST_C
#undef YL_FIRST_LAUNCH
    if (name) snprintf(name, name_len, "conv_f32_first<valu,4px,m%d%s>", wide ? 32 : 16, a.q_out ? (a.out ? ",q" : ",qonly") : (signs ? ",signs" : ""));
    return (int)hipGetLastError();
}

}  // namespace yl
