// conv_f32_firstm.hip -- K1m: the RGB first layer (3x3 / stride 1 / pad 1, C = 3, K = 27) on the FP32 MATRIX pipe, for the two
// outputs that are not bound by their stores: the sign words of an XNOR network's first layer (<= 16 filters) and the int8 units
// of an INT8 network's first layer (32 filters).
//
// Same layer and the same bits as conv_f32_first.hip (K1f; forward_convolutional_layer_cpu's FP32 branch,
// src/yolov2_forward_network.c:204-261): per output an fmaf chain over k = (c, ky, kx) ascending, + bias, activation.
// v_mfma_f32_16x16x4_f32 and v_mfma_f32_32x32x2_f32 ARE that chain, bit for bit (tools/mfma_chain_probe.hip: 0 of 256 000
// outputs differ from fmaf chains on MI355X), so K1m returns K1f's bits (tests/test_gpu_parity.py::test_conv_first_layer_kernel_*).
//
// Why now: K1f runs at 0.89 of the VALU's FP32 rate (64 fmas per SIMD and 4 cycles; v_pk_fma_f32 issues at half rate and buys
// nothing: profiles/r5_ab_first_layer_pk_fma.txt), and the matrix pipe does twice that.  Round 3's MFMA first-layer kernel (K1s) lost
// to K1f because it paid ~14 VALU per pixel for address decode, gathered its 27 operands with dword loads from global memory and
// wasted half of every 32-row MFMA on 16-filter layers.  Here:
//   * a workgroup (4 waves) walks 8 patches of 16 rows x 32 columns; a patch's 3 x 18 x 34 input window goes through LDS once
//     (9 dword loads per thread, the zero padding from the buffer range check), requested one patch ahead into registers and stored
//     into the second LDS buffer behind the current patch's MFMAs;
//   * the B operand of MFMA step t is ONE ds_read_b32 per lane at a loop-invariant per-lane address (pixel + the offset of
//     k = 4 t + lane / 16, resp. 2 t + lane / 32) plus an immediate for the block: no address arithmetic in the loop;
//   * the A operand (the weights of the lane's filter for its k's) sits in 7 resp. 14 registers for the whole kernel;
//   * 16 filters run on v_mfma_f32_16x16x4_f32 (7 per 16 pixels, no empty rows), 32 filters on v_mfma_f32_32x32x2_f32 (14 per 32
//     pixels, the filters permuted over the rows so that a lane ends up with 8 consecutive channels of each 16-channel unit);
//   * sign words: 4 bits per lane, OR over the four lane groups by two ds_bpermute, 16 lanes x 8 bytes per store; int8 units: K1f's
//     quantiser (trunc + clamp, the `int16_t = float` corner redone exactly), 8 bytes per lane and unit, 512 contiguous bytes per
//     instruction.
#include <hip/hip_runtime.h>
#include <cstdio>

#include "kernels.h"
#include "epilogue.h"
#include "../../include/yolo2_hip.h"

namespace yl {

typedef float m4f __attribute__((ext_vector_type(4)));
typedef float m16f __attribute__((ext_vector_type(16)));

namespace {

constexpr int FM_TY = 16, FM_TX = 32;               // output patch of a workgroup
constexpr int FM_ROWS = FM_TY + 2, FM_LROW = FM_TX + 2;
constexpr int FM_PLANE = FM_ROWS * FM_LROW;         // floats per channel of the LDS window
constexpr int FM_ZERO = 3 * FM_PLANE;               // a run of 1.0f: the operand of k = 27, whose "weight" is the bias (below), at any block offset
constexpr int FM_NZERO = 128;
constexpr int FM_BUF = FM_ZERO + FM_NZERO;          // floats per LDS buffer (two of them)

struct ConvFirstMDev {
    const float *in;
    const float *wt;       // k-major packed [Kpad][Mpad], K order (c, ky, kx); columns >= M are zero
    const float *bias;
    uint64_t *bits_out;    // [B][1][H][W] sign words (16-filter kernel)
    int8_t *q_out;         // act_q[B][q_G][H][W][16] int8 (32-filter kernel)
    float q_mult;
    int q_G;
    int B, H, W, M, Mpad, Kpad, act;
    int tiles_x, tiles_y;
    long long tiles;       // B * tiles_y * tiles_x
    unsigned rec;          // bytes of the input tensor
};

// A workgroup walks FM_TPW consecutive patches (b, ty, tx order): the window of patch i + 1 is requested before patch i is computed
// and stored into the other LDS buffer after it -- no load latency between patches, weights and addresses set up once.
constexpr int FM_TPW = 8;

struct FirstmWin { float v[3][3]; };

// the input window of patch `tile` -> registers: rows y0 - 1 .. y0 + 16, columns x0 - 1 .. x0 + 32 of the three channels; a thread owns
// one column of the window and rows r0, r0 + 7, r0 + 14 (threads 238 .. 255 idle; a patch past the last one loads nothing)
__device__ __forceinline__ void firstm_load(const ConvFirstMDev &p, FirstmWin &w, long long tile)
{
    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, (int)p.rec, 0x00020000);
    const bool live = tile < p.tiles;
    const int tx = (int)(tile % p.tiles_x);
    const long long t2 = tile / p.tiles_x;
    const int ty = (int)(t2 % p.tiles_y), b = (int)(t2 / p.tiles_y);
    const int y0 = ty * FM_TY, x0 = tx * FM_TX;
    const int col = tid % FM_LROW, r0 = tid / FM_LROW;
    const int ix = x0 - 1 + col;
    const bool cok = live && tid < 7 * FM_LROW && ix >= 0 && ix < p.W;
    const int HW = p.H * p.W;
#pragma unroll
    for (int ps = 0; ps < 3; ++ps) {
        const int r = r0 + 7 * ps;
        const int iy = y0 - 1 + r;
        const bool ok = cok && r < FM_ROWS && iy >= 0 && iy < p.H;
        const int voff = ok ? (int)((((unsigned)b * 3u) * (unsigned)HW + (unsigned)(iy * p.W + ix)) * 4u) : -1;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            w.v[ps][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, c * HW * 4, 0));
    }
}
__device__ __forceinline__ void firstm_store(float *win, const FirstmWin &w)
{
    const int tid = threadIdx.x;
    const int col = tid % FM_LROW, r0 = tid / FM_LROW;
    if (tid < 7 * FM_LROW) {
#pragma unroll
        for (int ps = 0; ps < 3; ++ps) {
            const int r = r0 + 7 * ps;
            if (r < FM_ROWS) {
#pragma unroll
                for (int c = 0; c < 3; ++c) win[c * FM_PLANE + r * FM_LROW + col] = w.v[ps][c];
            }
        }
    }
}

// byte offset of operand k inside the window, relative to the output pixel's own (row, column) entry
__device__ __forceinline__ int firstm_koff(int k)
{
    const int c = k / 9, rem = k - 9 * c, ky = rem / 3, kx = rem - 3 * ky;
    return (c * FM_PLANE + ky * FM_LROW + kx) * 4;
}

// ---- up to 16 filters -> sign words: bit m of word (b, y, x) = (conv + bias > 0) ----
// (the activation is not evaluated: linear and leaky keep the sign, as in K1f's SIGNS instances)
// POOL: the 2x2 / stride-2 [maxpool] behind the layer folded in -- the sign of a window's maximum is the OR of the window's signs
// (bit_maxpool_kernel's rule, layers.hip) -- the words of [B][1][H/2][W/2] are written instead (H, W even: whole windows)
template <bool POOL>
__global__ __launch_bounds__(256) void conv_f32_firstm_signs_kernel(ConvFirstMDev p)
{
    __shared__ float win2[2 * FM_BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const long long tile0 = (long long)blockIdx.x * FM_TPW;

    FirstmWin wreg;
    firstm_load(p, wreg, tile0);
    if (tid < FM_NZERO) { win2[FM_ZERO + tid] = 1.f; win2[FM_BUF + FM_ZERO + tid] = 1.f; }

    // A: the weights of filter n for k = 4 t + g; B addresses: this lane's pixel column + the window offset of that k.
    // K = 27 leaves one k of the seventh step free: it carries the BIAS -- A = bias[n], B = 1.0 -- so the chain ends with
    // fmaf(bias, 1, acc) = acc + bias, K1f's float add, inside the matrix pipe
    float a[7];
    int baddr[7];
#pragma unroll
    for (int t = 0; t < 7; ++t) {
        const int k = 4 * t + g;
        a[t] = (k < 27) ? p.wt[(size_t)k * p.Mpad + n] : (n < p.M ? p.bias[n] : 0.f);
        baddr[t] = (k < 27) ? firstm_koff(k) + (wave * 4 * FM_LROW + n) * 4 : FM_ZERO * 4;
    }
    firstm_store(win2, wreg);
    __syncthreads();

#pragma unroll 1
    for (int it = 0; it < FM_TPW; ++it) {
        const long long tile = tile0 + it;
        if (tile >= p.tiles) break;                  // (workgroup-uniform)
        firstm_load(p, wreg, tile + 1 < tile0 + FM_TPW ? tile + 1 : p.tiles);
        const int tx = (int)(tile % p.tiles_x);
        const long long t2 = tile / p.tiles_x;
        const int ty = (int)(t2 % p.tiles_y), b = (int)(t2 / p.tiles_y);
        const int y0 = ty * FM_TY, x0 = tx * FM_TX;
        const char *wb = reinterpret_cast<const char *>(win2 + (it & 1) * FM_BUF);
        unsigned held[2] = {0u, 0u};                 // POOL: the words of the window's upper row
        // a wave owns rows 4 w .. 4 w + 3 of the patch: 8 blocks of 16 pixels, two at a time
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            m4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 7; ++t) {
                const int off0 = i * FM_LROW * 4, off1 = off0 + 64;
                const float b0 = *reinterpret_cast<const float *>(wb + baddr[t] + off0);
                const float b1 = *reinterpret_cast<const float *>(wb + baddr[t] + off1);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b1, acc1, 0, 0, 0);
            }
            const int y = y0 + wave * 4 + i;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const m4f &acc = hf ? acc1 : acc0;
                unsigned word = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    word |= ((acc[e] > 0.f && 4 * g + e < p.M) ? 1u : 0u) << (4 * g + e);
                word |= (unsigned)__shfl_xor((int)word, 16);
                word |= (unsigned)__shfl_xor((int)word, 32);
                const int x = x0 + hf * 16 + n;
                if constexpr (!POOL) {
                    if (g == 0 && y < p.H && x < p.W) p.bits_out[((size_t)b * p.H + y) * p.W + x] = (uint64_t)word;
                } else {
                    if ((i & 1) == 0) held[hf] = word;
                    else {
                        word |= held[hf];
                        word |= (unsigned)__shfl_xor((int)word, 1);
                        if (g == 0 && (n & 1) == 0 && y < p.H && x < p.W)
                            p.bits_out[((size_t)b * (p.H >> 1) + (y >> 1)) * (size_t)(p.W >> 1) + (x >> 1)] = (uint64_t)word;
                    }
                }
            }
        }
        firstm_store(win2 + ((it + 1) & 1) * FM_BUF, wreg);
        __syncthreads();
    }
}

// ---- 32 filters -> int8 units act_q[B][q_G][H][W][16] ----
// MFMA row r carries filter f(r) = r with bits 2 and 3 swapped: the lane of half h then holds, in accumulator e, channel
// 8 h + (e & 7) of unit e >> 3 -- 8 consecutive bytes of each unit.
__global__ __launch_bounds__(256) void conv_f32_firstm_q_kernel(ConvFirstMDev p)
{
    __shared__ float win2[2 * FM_BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    const long long tile0 = (long long)blockIdx.x * FM_TPW;

    FirstmWin wreg;
    firstm_load(p, wreg, tile0);
    if (tid < FM_NZERO) { win2[FM_ZERO + tid] = 1.f; win2[FM_BUF + FM_ZERO + tid] = 1.f; }

    const int fr = (n & 19) | ((n & 4) << 1) | ((n & 8) >> 1);          // filter of MFMA row n
    float a[14];
    int baddr[14];
#pragma unroll
    for (int t = 0; t < 14; ++t) {
        const int k = 2 * t + h;
        a[t] = fr < p.M ? (k < 27 ? p.wt[(size_t)k * p.Mpad + fr] : p.bias[fr]) : 0.f;      // k = 27: the bias, against B = 1.0 (see above)
        baddr[t] = (k < 27) ? firstm_koff(k) + (wave * 4 * FM_LROW + n) * 4 : FM_ZERO * 4;
    }
    firstm_store(win2, wreg);
    __syncthreads();

    const int HW = p.H * p.W;
    const bool leaky = p.act == YL_LEAKY;
#pragma unroll 1
    for (int it = 0; it < FM_TPW; ++it) {
        const long long tile = tile0 + it;
        if (tile >= p.tiles) break;                  // (workgroup-uniform)
        firstm_load(p, wreg, tile + 1 < tile0 + FM_TPW ? tile + 1 : p.tiles);
        const int tx = (int)(tile % p.tiles_x);
        const long long t2 = tile / p.tiles_x;
        const int ty = (int)(t2 % p.tiles_y), b = (int)(t2 / p.tiles_y);
        const int y0 = ty * FM_TY, x0 = tx * FM_TX;
        const char *wb = reinterpret_cast<const char *>(win2 + (it & 1) * FM_BUF);
        // a wave owns rows 4 w .. 4 w + 3 of the patch: one block of 32 pixels each, two at a time
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            m16f acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
            for (int t = 0; t < 14; ++t) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float bv = *reinterpret_cast<const float *>(wb + baddr[t] + (2 * i2 + j) * FM_LROW * 4);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bv, acc[j], 0, 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int y = y0 + wave * 4 + 2 * i2 + j, x = x0 + n;
                float v[16];
                float tmax = 0.f;
                unsigned pk[4];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float y1 = acc[j][e];
                    if (leaky) {
                        float t1 = (float)(.1 * (double)y1);
                        asm volatile("" : "+v"(t1));
                        y1 = (y1 > 0.f) ? y1 : t1;
                    }
                    v[e] = y1;
                }
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    int c[4];
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const float t = __fmul_rn(v[g4 * 4 + r4], p.q_mult);
                        tmax = fmaxf(tmax, fabsf(t));
                        const int ci = (int)t;
                        c[r4] = ci < -127 ? -127 : (ci > 127 ? 127 : ci);
                    }
                    pk[g4] = __builtin_amdgcn_perm((unsigned)c[1], (unsigned)c[0], 0x0C0C0400u) |
                             __builtin_amdgcn_perm((unsigned)c[3], (unsigned)c[2], 0x04000C0Cu);
                }
                if (__builtin_amdgcn_ballot_w64(!(tmax < 32768.f)) != 0ull) {      // the `int16_t = float` wrap corner: redo exactly
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        unsigned w = 0;
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) w |= ((unsigned)(quantize_input_i8(v[g4 * 4 + r4], p.q_mult) & 0xFF)) << (8 * r4);
                        pk[g4] = w;
                    }
                }
                if (y < p.H && x < p.W) {
                    const size_t pix = (size_t)y * p.W + x;
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        if (16 * u < p.M)
                            *reinterpret_cast<uint2 *>(p.q_out + (((size_t)b * p.q_G + u) * HW + pix) * 16 + 8 * h) = make_uint2(pk[2 * u], pk[2 * u + 1]);
                }
            }
        }
        firstm_store(win2 + ((it + 1) & 1) * FM_BUF, wreg);
        __syncthreads();
    }
}

}  // namespace

// sign words only (<= 16 filters; bits_pooled: the words of the 2x2 / stride-2 [maxpool] behind the layer) or int8 units only (32 filters)
// of an RGB first layer
bool first_layer_mfma_applicable(const ConvF32Args &a)
{
    if (!first_layer_valu_applicable(a) || a.C != 3 || a.pool_out || a.out) return false;
    if (a.bits_pooled && ((a.H | a.W) & 1)) return false;
    const bool signs = a.bits_out && !a.q_out && a.M <= 16;
    const bool qonly = a.q_out && !a.bits_out && a.M == 32 && a.q_G >= 2;
    return signs || qonly;
}

int launch_conv_f32_firstm(const ConvF32Args &a, void *stream, char *name, size_t name_len)
{
    if (!first_layer_mfma_applicable(a)) return (int)hipErrorInvalidValue;
    ConvFirstMDev d;
    d.in = a.in; d.wt = a.wt; d.bias = a.bias; d.bits_out = a.bits_out; d.q_out = a.q_out; d.q_mult = a.q_mult; d.q_G = a.q_G;
    d.B = a.B; d.H = a.H; d.W = a.W; d.M = a.M; d.Mpad = a.Mpad; d.Kpad = a.Kpad; d.act = a.act;
    d.tiles_x = (a.W + FM_TX - 1) / FM_TX;
    d.tiles_y = (a.H + FM_TY - 1) / FM_TY;
    d.rec = (unsigned)((long long)a.B * a.C * a.H * a.W * 4);
    d.tiles = (long long)a.B * d.tiles_x * d.tiles_y;
    const long long blocks = (d.tiles + FM_TPW - 1) / FM_TPW;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (a.q_out) hipLaunchKernelGGL(conv_f32_firstm_q_kernel, grid, block, 0, s, d);
    else if (a.bits_pooled) hipLaunchKernelGGL(conv_f32_firstm_signs_kernel<true>, grid, block, 0, s, d);
    else hipLaunchKernelGGL(conv_f32_firstm_signs_kernel<false>, grid, block, 0, s, d);
    if (name) snprintf(name, name_len, a.q_out ? "conv_f32_first<mfma32x32x2,m32,qonly>" :
                       (a.bits_pooled ? "conv_f32_first<mfma16x16x4,m16,signs,pool>" : "conv_f32_first<mfma16x16x4,m16,signs>"));
    return (int)hipGetLastError();
}

}  // namespace yl
