// conv_f32_smallk.hip -- K1s: FP32 convolution of the FIRST layer of a detector (C*size*size <= 32, filters <= 32,
// e.g. yolov3 layer 0: 3 -> 32 channels, 3x3) on v_mfma_f32_32x32x2_f32, without any LDS staging.
//
// Same arithmetic as conv_f32_mfma.hip (forward_convolutional_layer_cpu's FP32 branch,
// src/yolov2_forward_network.c:204-261: im2col + gemm_nn + bias + leaky), same packed weights (k-major [Kpad][Mpad],
// K order (c,ky,kx) like im2col_cpu).  The general kernel is built for long K loops; with K = 27 its workgroup spends
// its life in the prologue (decode, first panel's latency) and epilogue: 32 TF / 2.4 TB/s of output on yolov3-608.
// Here the whole weight matrix lives in 14-16 VGPRs of every wave (one A fragment per k-step), a wave owns 32*TN
// consecutive output pixels per tile and gathers its B fragments straight into registers (lane = pixel, the lane's
// half picks the k of the pair; out-of-image taps get voffset -1 and read 0.0 from the buffer range check), runs
// KSTEPS*TN MFMAs and stores through the shared epilogue (NCHW rows via a wave-private LDS strip, and/or the int8
// side output of a following INT8 convolution).  A wave walks TPW consecutive tiles; 4 waves per SIMD hide each
// other's load and store phases.
#include <hip/hip_runtime.h>
#include <cstdio>

#include "kernels.h"
#include "epilogue.h"
#include "../../include/yolo2_hip.h"

namespace yl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct ConvSmallKDev {
    const float *in;
    const float *wt;
    const float *bias;
    const float *add;
    float *out_add;
    float *out;
    int8_t *q_out;
    float q_mult;
    int q_G;
    uint64_t *bits_out;   // sign words of the activated output for an XNOR convolution behind it: bits[B][1][OH][OW]
    int B, C, H, W, M, OH, OW;
    int K, Mpad;
    int size, stride, pad;
    int act;
    int Ntotal, OHW;
    int ntiles;           // tiles of 32*TN pixels
    unsigned rec;         // buffer bytes
};

constexpr int SK_TPW = 4;     // tiles a wave walks

}  // namespace

// three waves per SIMD (<= 168 registers) for the K <= 28 instance every shipped first layer takes; the K = 29..32
// instance holds two more k-steps of fragments and would spill there (tools/isa_lint.py guards both)
template <int TN, int KSTEPS>
__global__ __launch_bounds__(256, KSTEPS <= 14 ? 3 : 2) void conv_f32_smallk_kernel(ConvSmallKDev p)
{
    __shared__ __attribute__((aligned(16))) float smem[4 * 16 * TN * 32 + 32];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int half = lane >> 5;
    float *strip = smem + wave * (16 * TN * 32);

    // bijective XCD remap: consecutive logical workgroups (neighbouring image rows) share an XCD's L2
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);

    const int HW = p.H * p.W;
    const int CHW = p.C * HW;
    const int ss = p.size * p.size;

    // per-lane constants of the k pair (2s, 2s+1): this lane's k = 2s + half
    float a[KSTEPS];
    int koff[KSTEPS];
    int tbit[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        const int k = 2 * s + half;
        a[s] = p.wt[(size_t)k * p.Mpad + l31];          // rows >= K and columns >= M of the packed panel are zero
        if (k < p.K) {
            const int c = k / ss;
            const int t = k - c * ss;
            const int ky = t / p.size;
            const int kx = t - ky * p.size;
            koff[s] = (c * HW + ky * p.W + kx) * 4;
            tbit[s] = t;
        } else {
            koff[s] = 0;
            tbit[s] = 31;                                // bit 31 of the tap mask is always "invalid"
        }
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.in - (ptrdiff_t)p.pad * (p.W + 1)), 0, (int)p.rec, 0x00020000);

    // bias in C/D order per half: bias_lds[half][e] = bias[(e & 3) + 8 * (e >> 2) + 4 * half] (4 ds_read_b128 per tile)
    float *bias_lds = smem + 4 * 16 * TN * 32;
    if (tid < 32) {
        const int e = tid & 15, hf = tid >> 4;
        const int m = (e & 3) + 8 * (e >> 2) + 4 * hf;
        bias_lds[tid] = (m < p.M) ? p.bias[m] : 0.f;
    }
    __syncthreads();

    const int tile0 = (logical * 4 + wave) * SK_TPW;
    // (image, row, column) of the wave's first pixel: the only divisions; the tiles a wave walks are consecutive, so
    // the position advances by carries (PMC on the first version: 32 VALU per MFMA, a third of them pixel decode)
    int b0, oy0, ox0;
    {
        const int n0 = tile0 * (32 * TN);
        b0 = n0 / p.OHW;
        const int pix0 = n0 - b0 * p.OHW;
        oy0 = pix0 / p.OW;
        ox0 = pix0 - oy0 * p.OW;
    }
#pragma unroll 1
    for (int it = 0; it < SK_TPW; ++it) {
        const int tile = tile0 + it;
        if (tile >= p.ntiles) break;
        const int n_base = tile * (32 * TN);
        float bfr[TN][KSTEPS];
        int ob_j[TN], opix_j[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n_base + j * 32 + l31;
            unsigned ntap = 0xFFFFFFFFu;
            int pixoff = 0;
            int ox = ox0 + j * 32 + l31, oy = oy0, bimg = b0;
            while (ox >= p.OW) { ox -= p.OW; ++oy; }
            while (oy >= p.OH) { oy -= p.OH; ++bimg; }
            ob_j[j] = bimg;
            opix_j[j] = oy * p.OW + ox;
            if (n < p.Ntotal) {
                const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
                // taps inside the image = (valid rows) x (valid columns)
                unsigned rm = 0, cm = 0;
                for (int k = 0; k < p.size; ++k) {
                    if ((unsigned)(iy0 + k) < (unsigned)p.H) rm |= 1u << k;
                    if ((unsigned)(ix0 + k) < (unsigned)p.W) cm |= 1u << k;
                }
                unsigned ok = 0;
                for (int ky = 0; ky < p.size; ++ky)
                    if ((rm >> ky) & 1u) ok |= cm << (ky * p.size);
                ntap = ~ok | 0x80000000u;
                pixoff = (int)(((unsigned)bimg * (unsigned)CHW + (unsigned)(oy * p.stride) * (unsigned)p.W +
                                (unsigned)(ox * p.stride)) * 4u);
            }
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                const int inv = __builtin_amdgcn_sbfe((int)ntap, tbit[s], 1);      // -1 when the tap is outside
                bfr[j][s] = __builtin_bit_cast(float,
                    __builtin_amdgcn_raw_buffer_load_b32(rsrc, (pixoff + koff[s]) | inv, 0, 0));
            }
        }
        f32x16 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bfr[j][s], acc[j], 0, 0, 0);

        float vals[TN][16];
        float bias_v[16];
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            const float4 b4 = *reinterpret_cast<const float4 *>(bias_lds + half * 16 + e4 * 4);
            bias_v[e4 * 4 + 0] = b4.x; bias_v[e4 * 4 + 1] = b4.y; bias_v[e4 * 4 + 2] = b4.z; bias_v[e4 * 4 + 3] = b4.w;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float v = acc[j][e] + bias_v[e];
                if (p.act == YL_LEAKY) v = (v > 0.f) ? v : (float)(.1 * (double)v);
                vals[j][e] = v;
            }
        if (p.bits_out) {
            // Sign-domain hand-over (conv_xnor.hip): an XNOR convolution reads only (x > 0) of its input, so this layer
            // emits ONE 64-bit word per pixel (bit m = filter m, M <= 32; filters beyond M have zero weights and zero
            // bias: bit 0, as the consumer's channel padding wants) instead of M floats -- the word layout of
            // pack_sign_bits_kernel.  A pixel's 32 rows sit in two lanes (l31, half): one v_permlane32_swap ORs them.
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                unsigned bits = 0;
#pragma unroll
                for (int e = 0; e < 16; ++e) bits |= (vals[j][e] > 0.f ? 1u : 0u) << ((e & 3) + 8 * (e >> 2));
                bits <<= 4 * half;
                const auto sw = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);
                const unsigned word = sw[0] | sw[1];
                const int n = n_base + j * 32 + l31;
                if (half == 0 && n < p.Ntotal) p.bits_out[(size_t)ob_j[j] * p.OHW + opix_j[j]] = (uint64_t)word;
            }
        }
        if (!p.out && !p.add && !p.q_out) {
            // nothing else wants this tile (its FP32 tensor has no reader under the fusion plan)
        } else if (p.q_out && !p.out && !p.add)
            store_q_from_cd<TN>(vals, 0, p.M, n_base, p.Ntotal, p.OHW, p.q_out, p.q_mult, p.q_G, lane, ob_j, opix_j);
        else if (p.q_out)
            store_rows_via_lds_q<TN>(strip, vals, 0, p.M, n_base, p.Ntotal, p.OHW, p.out, p.add, p.out_add,
                                     p.q_out, p.q_mult, p.q_G, lane);
        else
            store_rows_via_lds<TN>(strip, vals, 0, p.M, n_base, p.Ntotal, p.OHW, p.out, p.add, p.out_add, lane);
        // next tile of this wave: 32*TN pixels further
        ox0 += 32 * TN;
        while (ox0 >= p.OW) { ox0 -= p.OW; ++oy0; }
        while (oy0 >= p.OH) { oy0 -= p.OH; ++b0; }
    }
}

bool smallk_applicable(const ConvF32Args &a)
{
    const long long in_bytes = (long long)a.B * a.C * a.H * a.W * 4 + (long long)a.pad * (a.W + 1) * 4;
    return !a.tapmajor && a.K <= 32 && a.M <= 32 && a.size >= 1 && a.size <= 5 && a.Mpad >= 32 &&
           a.Kpad >= ((a.K + 1) / 2) * 2 && in_bytes < 0xFFFFFFFELL && (!a.q_out || a.M % 16 == 0);
}

int launch_conv_f32_smallk(const ConvF32Args &a, void *stream, char *name, size_t name_len)
{
    if (!smallk_applicable(a)) return (int)hipErrorInvalidValue;
    constexpr int TN = 2;
    ConvSmallKDev d;
    d.in = a.in; d.wt = a.wt; d.bias = a.bias; d.add = a.add; d.out_add = a.out_add; d.out = a.out;
    d.q_out = a.q_out; d.q_mult = a.q_mult; d.q_G = a.q_G;
    d.bits_out = a.bits_out;
    d.B = a.B; d.C = a.C; d.H = a.H; d.W = a.W; d.M = a.M; d.OH = a.OH; d.OW = a.OW;
    d.K = a.K; d.Mpad = a.Mpad; d.size = a.size; d.stride = a.stride; d.pad = a.pad; d.act = a.act;
    d.OHW = a.OH * a.OW;
    const long long nt = (long long)a.B * d.OHW;
    if (nt > 0x7fffffffLL - 64) return (int)hipErrorInvalidValue;
    d.Ntotal = (int)nt;
    d.ntiles = (int)((nt + 32 * TN - 1) / (32 * TN));
    d.rec = (unsigned)((long long)a.B * a.C * a.H * a.W * 4 + (long long)a.pad * (a.W + 1) * 4);
    const int per_wg = 4 * SK_TPW;
    const dim3 grid((unsigned)((d.ntiles + per_wg - 1) / per_wg)), block(256);
    hipStream_t s = (hipStream_t)stream;
    const int ksteps = (a.K + 1) / 2;
    if (ksteps <= 14 && a.Kpad >= 28) hipLaunchKernelGGL((conv_f32_smallk_kernel<TN, 14>), grid, block, 0, s, d);
    else if (a.Kpad >= 32) hipLaunchKernelGGL((conv_f32_smallk_kernel<TN, 16>), grid, block, 0, s, d);
    else return (int)hipErrorInvalidValue;
    if (name) snprintf(name, name_len, "conv_f32_smallk<32x%d,k%d>", 32 * TN, ksteps <= 14 ? 28 : 32);
    return (int)hipGetLastError();
}

}  // namespace yl
