// conv_xnor.hip -- K3: the BIT1-XNOR convolution as 64-bit packed XNOR + popcount.
//
// Replaces the XNOR branch of forward_convolutional_layer_cpu
// (src/yolov2_forward_network.c:116-203: repack_input/float_to_bit/im2col/transpose_uint32 or
// im2col_cpu_custom_bin/transpose_bin, then gemm_nn_custom_bin_mean_transposed,
// src/additionally.c:1504-1534) and the reference GPU pipeline (float_to_bit_gpu,
// repack_input_kernel_bin, transpose_*, gemm_nn_custom_bin_mean_transposed_*,
// src/gpu.cu:1028-2046).  CDNA4 has no 1-bit MFMA: this is VALU work
// (v_xnor_b32 + v_bcnt_u32_b32 with a free accumulate) behind HBM-bound FP32 I/O.
//
// Layout in HBM:
//   activations  bits[B][Cw][H][W] uint64, bit (c & 63) of word plane (c >> 6) = (x[b][c][y][x] > 0)
//                (src/additionally.c:132,1544); pad channels = 0.  Consecutive pixels are
//                consecutive 8-byte words -> coalesced pack stores and tap loads.
//   weights      wbits[Mpad/2][Cw][2][9] uint64 (filter pairs interleaved per channel word: one kernel step =
//                18 contiguous words), bit = (w_fused > 0); pad channels = 1, so a pad bit
//                never matches (activation 0 vs weight 1) and out-of-image taps -- loaded as 0
//                through the buffer descriptor's range check -- count as -1 on every REAL
//                channel exactly like the reference's zero-padded bit im2col.
// Mapping: one lane = one output pixel; the filter loop is wave-uniform, so the compiler
// fetches weight words with scalar loads (SMEM) and they enter v_xnor as SGPR operands: no
// LDS, no vector traffic for weights.  Input words of a channel chunk stay in VGPRs while FT
// filters are accumulated; count = #matching bits is integer-exact.
//   out = act( (2*count - K) * mean[f] + bias[f] ),  K = 9*C   (src/additionally.c:1531)
#include <hip/hip_runtime.h>
#include <cstdio>

#include "kernels.h"
#include "../../include/yolo2_hip.h"

namespace yl {

// ------------------------------------------------------------------ K3a: sign-bit packing
__global__ __launch_bounds__(256) void pack_sign_bits_kernel(const float *__restrict__ in, uint64_t *__restrict__ out,
                                                             size_t total, int C, int HW, int Cw)
{
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int pix = (int)(idx % HW);
        size_t t = idx / HW;
        const int cw = (int)(t % Cw);
        const size_t b = t / Cw;
        const float *src = in + (b * C + (size_t)cw * 64) * HW + pix;
        const int nc = (C - cw * 64) < 64 ? (C - cw * 64) : 64;
        unsigned lo = 0, hi = 0;
        for (int j = 0; j < nc && j < 32; ++j) lo |= (src[(size_t)j * HW] > 0.f ? 1u : 0u) << j;
        for (int j = 32; j < nc; ++j) hi |= (src[(size_t)j * HW] > 0.f ? 1u : 0u) << (j - 32);
        out[idx] = ((uint64_t)hi << 32) | lo;
    }
}

int launch_pack_sign_bits(const float *in, uint64_t *out, int B, int C, int H, int W, int Cw, void *stream)
{
    const size_t total = (size_t)B * Cw * H * W;
    size_t g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g == 0) g = 1;
    hipLaunchKernelGGL(pack_sign_bits_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                       in, out, total, C, H * W, Cw);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ K3c: pooling on sign words
// maxpool followed by an XNOR convolution only needs sign(max) = OR of the window's sign bits (the
// reference pads max-pool windows with -FLT_MAX, src/yolov2_forward_network.c:268-300: out-of-image taps
// never set a bit): bits_out[b][cw][oy][ox] = OR over the in-image taps of bits_in[b][cw][.][.]
__global__ __launch_bounds__(256) void bit_maxpool_kernel(const uint64_t *__restrict__ in, uint64_t *__restrict__ out,
                                                          size_t total, int H, int W, int OH, int OW,
                                                          int size, int stride, int off)
{
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx % OW);
        size_t t = idx / OW;
        const int i = (int)(t % OH);
        t /= OH;                                   // t = cw + Cw*b
        const uint64_t *src = in + t * (size_t)H * W;
        uint64_t acc = 0;
        for (int n = 0; n < size; ++n) {
            const int cur_h = off + i * stride + n;
            for (int m = 0; m < size; ++m) {
                const int cur_w = off + j * stride + m;
                if (cur_h >= 0 && cur_h < H && cur_w >= 0 && cur_w < W) acc |= src[(size_t)cur_h * W + cur_w];
            }
        }
        out[idx] = acc;
    }
}

int launch_bit_maxpool(const uint64_t *in, uint64_t *out, int B, int Cw, int H, int W, int OH, int OW,
                       int size, int stride, int pad, void *stream)
{
    const size_t total = (size_t)B * Cw * OH * OW;
    size_t g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g == 0) g = 1;
    hipLaunchKernelGGL(bit_maxpool_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                       in, out, total, H, W, OH, OW, size, stride, -pad / 2);
    return (int)hipGetLastError();
}

// FP32 producer -> maxpool -> XNOR convolution: the pooled FP32 tensor is never needed, one lane takes one
// (image, 64-channel word, output pixel) and ORs (x > 0) over its window for up to 64 channels
__global__ __launch_bounds__(256) void maxpool_sign_pack_kernel(const float *__restrict__ in, uint64_t *__restrict__ out,
                                                                size_t total, int C, int Cw, int H, int W, int OH, int OW,
                                                                int size, int stride, int off)
{
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx % OW);
        size_t t = idx / OW;
        const int i = (int)(t % OH);
        t /= OH;
        const int cw = (int)(t % Cw);
        const size_t b = t / Cw;
        const int nc = (C - cw * 64) < 64 ? (C - cw * 64) : 64;
        const float *src = in + (b * C + (size_t)cw * 64) * (size_t)H * W;
        unsigned lo = 0, hi = 0;
        for (int c = 0; c < nc; ++c) {
            bool any = false;
            for (int n = 0; n < size; ++n) {
                const int cur_h = off + i * stride + n;
                if (cur_h < 0 || cur_h >= H) continue;
                for (int m = 0; m < size; ++m) {
                    const int cur_w = off + j * stride + m;
                    if (cur_w >= 0 && cur_w < W) any = any || (src[(size_t)c * H * W + (size_t)cur_h * W + cur_w] > 0.f);
                }
            }
            if (c < 32) lo |= (any ? 1u : 0u) << c;
            else hi |= (any ? 1u : 0u) << (c - 32);
        }
        out[idx] = ((uint64_t)hi << 32) | lo;
    }
}

int launch_maxpool_sign_pack(const float *in, uint64_t *out, int B, int C, int Cw, int H, int W, int OH, int OW,
                             int size, int stride, int pad, void *stream)
{
    const size_t total = (size_t)B * Cw * OH * OW;
    size_t g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g == 0) g = 1;
    hipLaunchKernelGGL(maxpool_sign_pack_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                       in, out, total, C, Cw, H, W, OH, OW, size, stride, -pad / 2);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ K3b: XNOR + popcount conv (3x3, stride 1, pad 1)
struct ConvXnorDev {
    const uint64_t *in_bits;
    const uint64_t *w_bits;
    const float *mean;
    const float *bias;
    float *out;               // FP32 output, or nullptr when only the sign words / the fused sum are wanted
    const float *add;         // fused [shortcut] operand (same shape as the output) or nullptr
    float *out_add;
    uint64_t *out_bits;       // sign words of the activation for a following XNOR layer, or nullptr
    int32_t *dbg;
    const int *thr;           // count thresholds (sign-only epilogue) or nullptr
    int B, C, Cw, H, W, M, act;
    int out_Cw;               // words per pixel of out_bits
    int Ntotal, HW;
};

// One lane = one output pixel; the filter loop is wave-uniform, so weight words arrive by scalar loads and
// enter v_xnor / v_bcnt as SGPR operands.  FT filters per lane (64 where the layer has them: the 9*CWC input
// words of a channel chunk are fetched once per 64 filters, and the 64 sign bits of the result are exactly one
// word of the next layer's input).  W32: C <= 32 -- the upper halves of the single word are padding (activation
// bits 0, weight bits 1: never a match), so only the lower 32 bits are counted.
//
// Weight stream.  SMEM returns out of order, so the only wait is lgkmcnt(0): a prefetch can be one STEP ahead
// and no more.  PMC on the first version (profiles/r2_pmc_xnor_first_version.txt): 69 % of the wave cycles in
// s_waitcnt -- hipcc's scheduler sank each scalar load to ~4 instructions before its wait.  Here a step is 18
// weight words = 36 SGPRs (two filters x one channel word: a two-word chunk needed more scalar state than 102
// SGPRs hold and was spilled to VGPR lanes, one v_readlane per two useful instructions); the next step's three
// loads (x16 + x16 + x4) are issued, a sched_barrier pins them there, and the ~90 VALU instructions (~180 cycles)
// of the current step run before hipcc's own wait in front of the first use.  (An inline-asm form of the same
// loads was WRONG on some instances: hipcc copied the destination SGPRs before the data had landed.)
typedef int s16 __attribute__((ext_vector_type(16)));
typedef int s4 __attribute__((ext_vector_type(4)));
struct WSet { s16 a, b; s4 c; };          // 36 dwords = 18 sign words

__device__ __forceinline__ void wset_load(WSet &w, const uint64_t *ptr)
{
    const int *q = reinterpret_cast<const int *>(ptr);      // wave-uniform address -> s_load_dwordx16 / x4
    w.a = *reinterpret_cast<const s16 *>(q);
    w.b = *reinterpret_cast<const s16 *>(q + 16);
    w.c = *reinterpret_cast<const s4 *>(q + 32);
}
template <int K> __device__ __forceinline__ unsigned wset_dword(const WSet &w)
{
    if constexpr (K < 16) return (unsigned)w.a[K];
    else if constexpr (K < 32) return (unsigned)w.b[K - 16];
    else return (unsigned)w.c[K - 32];
}

// c += popcount(x) as ONE instruction: v_bcnt_u32_b32 adds its third operand.  Left to itself hipcc emits
// v_bcnt(x, 0) and merges pairs with v_add3_u32 -- 5 VALU per 64 bit-MACs instead of the 4 the popcount roof is
// priced with (1152 v_xnor + 1152 v_bcnt + 576 v_add3 per channel word and 64 filters, round-2 ISA).  The chain
// through c is 4 instructions long (two filters alternate, each v_bcnt sits behind its own v_xnor) and four waves
// per SIMD cover it.  Pure register arithmetic: no memory clobber, the compiler still tracks the scalar loads
// feeding x.
__device__ __forceinline__ int popc_acc(unsigned x, int c)
{
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(c) : "v"(x));
    return c;
}
template <int CWC, bool W32, int W0, int T, int I, int N>
__device__ __forceinline__ void xor_taps(const WSet &w, const unsigned (&lo)[9][CWC], const unsigned (&hi)[9][CWC], unsigned (&xl)[N], unsigned (&xh)[N])
{
    if constexpr (I < N) {
        xl[I] = lo[T + I][0] ^ wset_dword<2 * (W0 + T + I)>(w);
        xh[I] = W32 ? 0u : (hi[T + I][0] ^ wset_dword<2 * (W0 + T + I) + 1>(w));
        xor_taps<CWC, W32, W0, T, I + 1, N>(w, lo, hi, xl, xh);
    }
}

// accumulate taps [T, T1) of one filter whose 9 words start at word W0 of the set.
// Round 4: the accumulators count MISMATCHES, popcount(x ^ w), and the kernel turns them into match counts once per
// filter (count = counted bits - mismatches).  v_xnor_b32 is a half-rate instruction on gfx950 (4.3 clk per wave64
// instruction), v_xor_b32 a full-rate one (2.4), v_bcnt_u32_b32 half-rate either way (tools/valu_issue_bench.hip,
// profiles/r4_valu_issue_bench.txt); alternating the two kinds costs the slower rate for both, so the xors of a group
// of taps are issued together in front of their popcounts (measured on the microbenchmark: 4.29 -> 3.91 clk per
// instruction).  Padding and halo bits behave as before: a channel-pad bit is 0 in the input and 1 in the weights
// (always a mismatch), an out-of-image word reads as 0.
template <int CWC, bool W32, int W0, int T, int T1>
__device__ __forceinline__ int xnor_acc(const WSet &w, const unsigned (&lo)[9][CWC], const unsigned (&hi)[9][CWC], int c)
{
    constexpr int NT_ = T1 - T;
    if constexpr (NT_ > 0) {
        unsigned xl[NT_], xh[NT_];
        xor_taps<CWC, W32, W0, T, 0, NT_>(w, lo, hi, xl, xh);
#pragma unroll
        for (int i = 0; i < NT_; ++i) {
            c = popc_acc(xl[i], c);
            if (!W32) c = popc_acc(xh[i], c);
        }
    }
    return c;
}

// taps [T, T + N) of BOTH filters of a step: the xors of the two filters first, then their popcounts with the two
// accumulator chains alternating
template <int CWC, bool W32, int T, int N>
__device__ __forceinline__ void xnor_acc2(const WSet &w, const unsigned (&lo)[9][CWC], const unsigned (&hi)[9][CWC], int &ca, int &cb)
{
    unsigned al[N], ah[N], bl[N], bh[N];
    xor_taps<CWC, W32, 0, T, 0, N>(w, lo, hi, al, ah);
    xor_taps<CWC, W32, 9, T, 0, N>(w, lo, hi, bl, bh);
    // the xors (plain C: the weight dwords stay SGPR operands) are pinned in front of their popcounts; hipcc's scheduler
    // interleaves the two kinds one by one otherwise
#pragma unroll
    for (int i = 0; i < N; ++i) {
        asm("" : "+v"(al[i]), "+v"(bl[i]));
        if (!W32) asm("" : "+v"(ah[i]), "+v"(bh[i]));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        ca = popc_acc(al[i], ca);
        cb = popc_acc(bl[i], cb);
        if (!W32) { ca = popc_acc(ah[i], ca); cb = popc_acc(bh[i], cb); }
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int CWC, int FT, bool W32>
__global__ __launch_bounds__(256) void conv_xnor_kernel(ConvXnorDev p)
{
    static_assert(CWC == 1, "a step is two filters x one channel word");
    constexpr int FS = 2;                       // filters per step
    const int tid = threadIdx.x;
    const int n = blockIdx.x * 256 + tid;
    const int f0 = blockIdx.y * FT;
    const bool n_ok = n < p.Ntotal;
    const int n_first = blockIdx.x * 256;
    const int b_first = n_first / p.HW;                          // uniform
    const int bimg = n / p.HW;
    const int pix = n - bimg * p.HW;
    const int y = pix / p.W;
    const int x = pix - y * p.W;

    // descriptor based at the first image of this block, shifted back one row + one pixel
    const size_t img_words = (size_t)p.Cw * p.HW;
    const uint64_t *base = p.in_bits + (size_t)b_first * img_words - (p.W + 1);
    size_t rec = (((size_t)p.B - b_first) * img_words + (size_t)(p.W + 1)) * 8;
    if (rec > 0xFFFFFFFEull) rec = 0xFFFFFFFEull;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)(unsigned)rec, 0x00020000);
    const int voff = (int)(((unsigned)(bimg - b_first) * (unsigned)img_words + (unsigned)y * (unsigned)p.W + (unsigned)x) * 8u);

    // inverted validity of the 9 taps (bit t set <=> tap outside the image -> word reads as 0)
    unsigned ntap = 0xFFFFFFFFu;
    if (n_ok) {
        unsigned m = 0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = y + ky - 1, ix = x + kx - 1;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) m |= 1u << (ky * 3 + kx);
            }
        ntap = ~m;
    }

    int cnt[FT];
#pragma unroll
    for (int f = 0; f < FT; ++f) cnt[f] = 0;

    const int nchunk = p.Cw;
    // weights of step (filter pair q, word cw): wbits[q][cw][2][9]
    WSet ws[2];
    wset_load(ws[0], p.w_bits + (size_t)(f0 / 2) * nchunk * 18);
    for (int ch = 0; ch < nchunk; ++ch) {
        const int cw0 = ch * CWC;
        unsigned in_lo[9][CWC], in_hi[9][CWC];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int tinv = __builtin_amdgcn_sbfe((int)ntap, t, 1);
            const int ky = t / 3, kx = t - ky * 3;
#pragma unroll
            for (int w = 0; w < CWC; ++w) {
                const int soff = ((cw0 + w) * p.HW + ky * p.W + kx) * 8;
                if (W32) {
                    in_lo[t][w] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff | tinv, soff, 0);
                    in_hi[t][w] = 0;
                } else {
                    typedef unsigned v2u __attribute__((ext_vector_type(2)));
                    const v2u v = __builtin_bit_cast(v2u, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff | tinv, soff, 0));
                    in_lo[t][w] = v[0];
                    in_hi[t][w] = v[1];
                }
            }
        }
#pragma unroll
        for (int st = 0; st < FT / FS; ++st) {
            WSet &cur = ws[st & 1];
            WSet &nxt = ws[(st + 1) & 1];
            // SMEM has one counter and returns out of order: hipcc waits lgkmcnt(0) at the first use of `cur`.  Use
            // `cur` HERE (the first tap of the first filter), before the prefetch is issued, so that wait covers only
            // loads issued a step ago.  (An asm statement as the "use" makes hipcc treat memory as clobbered and fall
            // back to vector loads for the weights.)
            cnt[2 * st] = xnor_acc<CWC, W32, 0, 0, 1>(cur, in_lo, in_hi, cnt[2 * st]);
            __builtin_amdgcn_sched_barrier(0);
            // next step: the following filter(s) of this chunk, or the first of the next chunk (the set loaded
            // past the last step of the last chunk is never used: the weights are padded by one step)
            {
                const int qn = f0 / 2 + ((st + 1 < FT / FS) ? st + 1 : 0);
                const int chn = (st + 1 < FT / FS) ? ch : ch + 1;
                wset_load(nxt, p.w_bits + ((size_t)qn * nchunk + chn) * 18);
            }
            __builtin_amdgcn_sched_barrier(0);          // the prefetch stays above this step's arithmetic
            // (tap 0 of the second filter, then taps 1-8 of both in two groups of four: 16 xors, 16 popcounts)
            cnt[2 * st + 1] = xnor_acc<CWC, W32, 9, 0, 1>(cur, in_lo, in_hi, cnt[2 * st + 1]);
            xnor_acc2<CWC, W32, 1, 4>(cur, in_lo, in_hi, cnt[2 * st], cnt[2 * st + 1]);
            xnor_acc2<CWC, W32, 5, 4>(cur, in_lo, in_hi, cnt[2 * st], cnt[2 * st + 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // mismatches -> matches: every counted bit (9 taps x Cw words x 32 or 64 bits, pads included) either matched or not
    {
        const int counted = 9 * p.Cw * (W32 ? 32 : 64);
#pragma unroll
        for (int f = 0; f < FT; ++f) cnt[f] = counted - cnt[f];
    }
    if (!n_ok) return;
    const int K = 9 * p.C;
    const size_t obase = (size_t)bimg * p.M * p.HW + pix;
    unsigned sign_lo = 0, sign_hi = 0;
    // Sign-only epilogue: between two XNOR layers nothing but (result > 0) is consumed, and the result
    // fl(fl((2*count - K) * mean) + bias) -- leaky keeps the sign -- is a non-decreasing function of the integer
    // count (mean = mean|w| >= 0, rounding is monotone), i.e. a step at a per-filter threshold that
    // xnor_threshold_kernel found by evaluating THIS expression for every count (and verified to be a step).  One
    // compare per filter instead of cvt / mul / add / the double-precision leaky / compare: ~3 VALU instead of ~14
    // per output, which on the thin layers (K = 144 ... 576 bits) was as much work as the popcounts themselves.
    if (p.thr && !p.out && !p.add && !p.dbg) {
#pragma unroll
        for (int f = 0; f < FT; ++f) {
            const int t = p.thr[f0 + f];                          // wave-uniform: scalar load; pad filters hold INT_MAX
            if (f < 32) sign_lo |= (cnt[f] >= t ? 1u : 0u) << f;
            else sign_hi |= (cnt[f] >= t ? 1u : 0u) << (f - 32);
        }
    } else
#pragma unroll
    for (int f = 0; f < FT; ++f) {
        const int m = f0 + f;
        if (m < p.M) {
            const size_t oi = obase + (size_t)m * p.HW;
            if (p.dbg) p.dbg[oi] = cnt[f];
            float v = __fmul_rn((float)(2 * cnt[f] - K), p.mean[m]);
            v = __fadd_rn(v, p.bias[m]);
            if (p.act == YL_LEAKY) v = (v > 0.f) ? v : (float)(.1 * (double)v);
            if (p.out) p.out[oi] = v;
            if (p.add) p.out_add[oi] = __fadd_rn(v, p.add[oi]);       // shortcut_cpu: out = add + conv, linear
            // bit = (x > 0) of the value the next layer would read (src/additionally.c:132,1544)
            if (f < 32) sign_lo |= (v > 0.f ? 1u : 0u) << f;
            else sign_hi |= (v > 0.f ? 1u : 0u) << (f - 32);
        }
    }
    if (p.out_bits) {
        uint64_t *dst = p.out_bits + ((size_t)bimg * p.out_Cw + (f0 >> 6)) * p.HW + pix;
        if (FT == 64) *dst = ((uint64_t)sign_hi << 32) | sign_lo;
        else if (p.M <= 32) *dst = (uint64_t)sign_lo;       // the word's only writer: upper half 0, not what an earlier layer left in the ring slot
        else reinterpret_cast<unsigned *>(dst)[(f0 >> 5) & 1] = sign_lo;      // 32-filter tiles: half a word each
    }
}

// thresholds of the sign-only epilogue: one lane per filter evaluates the kernel's own expression for every count
__global__ __launch_bounds__(64) void xnor_threshold_kernel(const float *__restrict__ mean, const float *__restrict__ bias,
                                                            int *__restrict__ thr, int *__restrict__ bad, int M, int Mpad, int K)
{
    const int m = blockIdx.x * 64 + threadIdx.x;
    if (m >= Mpad) return;
    if (m >= M) { thr[m] = 0x7fffffff; return; }
    const float mu = mean[m], bv = bias[m];
    int t = K + 1;
    bool step = true;
    for (int c = 0; c <= K; ++c) {
        const float v = __fadd_rn(__fmul_rn((float)(2 * c - K), mu), bv);
        const bool pos = v > 0.f;
        if (pos && t == K + 1) t = c;
        if (!pos && t != K + 1) step = false;                    // positive below, not positive above: no threshold
    }
    thr[m] = t;
    if (!step) atomicAdd(bad, 1);
}

int launch_xnor_thresholds(const float *mean, const float *bias, int *thr, int *bad, int M, int Mpad, int K, void *stream)
{
    hipLaunchKernelGGL(xnor_threshold_kernel, dim3((unsigned)((Mpad + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                       mean, bias, thr, bad, M, Mpad, K);
    return (int)hipGetLastError();
}

template <int CWC, int FT, bool W32>
static int launch_xnor(const ConvXnorDev &d, hipStream_t s)
{
    // 32-filter tiles over more than one tile: always whole output words (both halves of the last sign word are
    // written -- the upper one as zeros -- whatever an earlier layer left in the ring slot; weights, thresholds
    // are padded to 64 filters)
    const int tiles = (FT == 32 && d.M > 32) ? 2 * ((d.M + 63) / 64) : (d.M + FT - 1) / FT;
    dim3 grid((unsigned)((d.Ntotal + 255) / 256), (unsigned)tiles);
    hipLaunchKernelGGL((conv_xnor_kernel<CWC, FT, W32>), grid, dim3(256), 0, s, d);
    return (int)hipGetLastError();
}

int launch_conv_xnor(const ConvXnorArgs &a, void *stream, char *name, size_t name_len)
{
    ConvXnorDev d;
    d.in_bits = a.in_bits; d.w_bits = a.w_bits; d.mean = a.mean; d.bias = a.bias; d.out = a.out; d.dbg = a.dbg;
    d.add = a.add; d.out_add = a.out_add; d.thr = a.thr;
    d.out_bits = a.out_bits; d.out_Cw = (a.M + 63) / 64;
    d.B = a.B; d.C = a.C; d.Cw = a.Cw; d.H = a.H; d.W = a.W; d.M = a.M; d.act = a.act;
    d.HW = a.H * a.W;
    const long long nt = (long long)a.B * d.HW;
    if (nt > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    d.Ntotal = (int)nt;
    hipStream_t s = (hipStream_t)stream;
    // weights are padded to a multiple of 64 filters (runtime.hip), FT must divide that.
    // Filter tile: 64 where the grid is deep, 32 where it is shallow.  A 64-filter workgroup occupies 5 wave slots per
    // SIMD (86-95 VGPRs); tiny-yolo's 13 x 13 x 1024 layers at batch 128 are 1 360 such workgroups = 5.3 per CU, so
    // 80 of the 256 CUs ran a sixth workgroup alone after the other five had finished: 0.70-0.73 of the popcount
    // roof on the two layers that hold 40 % of the bit work.  Half-size tiles (7-8 slots) balance: the input words
    // are fetched twice (9 loads per 1 152 VALU instructions instead of per 2 304), two workgroups write one half
    // of an output sign word each.
    const long long wg64 = (long long)((d.Ntotal + 255) / 256) * ((a.M + 63) / 64);
    const bool ft32 = a.M < 64 || (a.ft_mode == 0 && wg64 < 16 * 256) || a.ft_mode == 32;
    if (name) snprintf(name, name_len, "conv_xnor<ft%d,%s%s>", ft32 ? 32 : 64, a.C <= 32 ? "w32" : "w64",
                       (a.thr && !a.out && !a.add && !a.dbg) ? ",thr" : "");      // ",thr": the count-threshold epilogue really runs
    if (a.C <= 32) return !ft32 ? launch_xnor<1, 64, true>(d, s) : launch_xnor<1, 32, true>(d, s);
    return !ft32 ? launch_xnor<1, 64, false>(d, s) : launch_xnor<1, 32, false>(d, s);
}

}  // namespace yl
