// epilogue.h -- shared store path of the MFMA convolution kernels (K1, K2).
//
// A 32x32 MFMA accumulator gives each lane ONE pixel column and 16 filter rows, so storing it
// directly costs one 4-byte store instruction per value and touches only 128 contiguous bytes
// per row.  Measured on MI355X (profiles/r1_int8_pmc.txt) that store tail -- not the MFMAs --
// bounds the INT8 kernel and the HBM-bound early FP32 layers.  Here a wave stages 8 filter rows
// x (TN*32) pixels of finished values in a wave-private LDS strip and writes them back as rows:
// every store instruction moves 16 bytes per lane, 4x fewer instructions, up to 512 contiguous
// bytes per row (NCHW: a row is one filter over consecutive pixels).  The optional fused
// [shortcut] operand is read with the same 16-byte row accesses.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace yl {

// vals[j][e]: finished outputs of one 32-row (TM) block of this wave in MFMA C/D layout:
//   column n = n_base + j*32 + (lane & 31), row m = m_base + (e & 3) + 8*(e >> 2) + 4*(lane >> 5)
// strip: wave-private LDS, >= 8 * TN * 32 floats, 16-byte aligned.
// out may be nullptr (only out_add wanted); add/out_add may be nullptr.
template <int TN>
__device__ __forceinline__ void store_rows_via_lds(float *strip, const float (&vals)[TN][16], int m_base, int M,
                                                   int n_base, int Ntotal, int OHW, float *out,
                                                   const float *add, float *out_add, int lane)
{
    constexpr int ROW = TN * 32;              // floats per staged row
    constexpr int LPR = TN * 8;               // lanes (float4) per row
    constexpr int RPI = 64 / LPR;             // rows per store instruction
    constexpr int NI = 8 / RPI;               // store instructions per 8-row chunk
    static_assert(RPI >= 1 && NI >= 1, "TN in {1,2,4}");
    const int l31 = lane & 31, half = lane >> 5;
    const int c4 = lane % LPR;                // float4 column owned by this lane when storing
    const int rsub = lane / LPR;
    const int n = n_base + c4 * 4;            // first of this lane's 4 consecutive pixels
    // (image, pixel) of the 4 pixels: contiguous in NCHW iff they share the image
    const int ob = n / OHW;
    const int opix = n - ob * OHW;
    const bool in_range = n + 3 < Ntotal;
    const bool same_img = in_range && (opix + 3 < OHW);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) strip[(r + 4 * half) * ROW + j * 32 + l31] = vals[j][4 * g + r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            const int row = t * RPI + rsub;                    // 0..7 inside the chunk
            const int m = m_base + 8 * g + row;
            const float4 v = *reinterpret_cast<const float4 *>(strip + row * ROW + c4 * 4);
            if (m < M && n < Ntotal) {
                const size_t o = ((size_t)ob * M + m) * OHW + opix;
                if (same_img && ((o & 3) == 0)) {
                    if (out) *reinterpret_cast<float4 *>(out + o) = v;
                    if (add) {
                        const float4 a = *reinterpret_cast<const float4 *>(add + o);
                        float4 s;
                        s.x = __fadd_rn(v.x, a.x); s.y = __fadd_rn(v.y, a.y);
                        s.z = __fadd_rn(v.z, a.z); s.w = __fadd_rn(v.w, a.w);
                        *reinterpret_cast<float4 *>(out_add + o) = s;
                    }
                } else {
                    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int nq = n + q;
                        if (nq < Ntotal) {
                            const int obq = nq / OHW;
                            const size_t oq = ((size_t)obq * M + m) * OHW + (nq - obq * OHW);
                            if (out) out[oq] = vv[q];
                            if (add) out_add[oq] = __fadd_rn(vv[q], add[oq]);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// x_q = max_abs((int16_t)(x * mult), 127) exactly as gcc/x86-64 compiles the reference's
// `int16_t src = state.input[z] * l.input_quant_multipler` (src/yolov2_forward_network_quantized.c:556-559):
// cvttss2si (0x80000000 when out of range / NaN), low 16 bits, then clamp_abs 127.
__device__ __forceinline__ int quantize_input_i8(float x, float mult)
{
    const float tv = __fmul_rn(x, mult);
    const int i32 = (fabsf(tv) < 2147483648.f) ? (int)tv : (int)0x80000000;
    const int s = (int)(short)(i32 & 0xFFFF);
    return (abs(s) > 127) ? ((s > 0) ? 127 : -127) : s;
}

// The int8 side output alone, straight from the MFMA C/D layout (no LDS): a lane owns pixel column n and rows
// (e&3) + 8*(e>>2) + 4*half; its 4 consecutive channels of each row group pack into one dword, and one
// v_permlane32_swap per 16-channel unit hands every lane 8 contiguous bytes of the unit (lanes 0-31 bytes 0-7,
// lanes 32-63 bytes 8-15): 512 contiguous bytes per store instruction.  Fast quantisation = trunc + clamp; the
// `int16_t = float` wrap corner (|x*mult| >= 32768, see quantize_input_i8) is detected per block and redone.
// Requires m_base % 32 == 0 and M % 16 == 0.
// ob[j] / opix[j]: image and pixel of column j of this lane when the caller already knows them (no division here),
// or nullptr
template <int TN>
__device__ __forceinline__ void store_q_from_cd(const float (&vals)[TN][16], int m_base, int M, int n_base, int Ntotal,
                                                int OHW, int8_t *q_out, float q_mult, int q_G, int lane,
                                                const int *ob_known = nullptr, const int *opix_known = nullptr)
{
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n_base + j * 32 + l31;
        unsigned pk[4];
        float tmax = 0.f;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            int c[4];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float t = __fmul_rn(vals[j][g4 * 4 + r4], q_mult);
                tmax = fmaxf(tmax, fabsf(t));
                const int ci = (int)t;
                c[r4] = ci < -127 ? -127 : (ci > 127 ? 127 : ci);
            }
            pk[g4] = __builtin_amdgcn_perm((unsigned)c[1], (unsigned)c[0], 0x0C0C0400u) |
                     __builtin_amdgcn_perm((unsigned)c[3], (unsigned)c[2], 0x04000C0Cu);
        }
        if (__builtin_amdgcn_ballot_w64(!(tmax < 32768.f)) != 0ull) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                unsigned w = 0;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    w |= ((unsigned)(quantize_input_i8(vals[j][g4 * 4 + r4], q_mult) & 0xFF)) << (8 * r4);
                pk[g4] = w;
            }
        }
        const int ob = ob_known ? ob_known[j] : n / OHW;
        const size_t unit0 = (size_t)ob * q_G * OHW + (opix_known ? opix_known[j] : n - ob * OHW);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const auto sw = __builtin_amdgcn_permlane32_swap(pk[2 * u], pk[2 * u + 1], false, false);
            const int cg = (m_base + 16 * u) >> 4;
            if (n < Ntotal && m_base + 16 * u < M) {
                uint2 d = make_uint2(sw[0], sw[1]);
                *reinterpret_cast<uint2 *>(q_out + (unit0 + (size_t)cg * OHW) * 16 + 8 * half) = d;
            }
        }
    }
}

// As store_rows_via_lds, plus a quantised side output for the NEXT INT8 convolution: the tensor
// the next layer consumes (out, or out_add when a [shortcut] is fused) is also written as
// act_q[B][q_G][OH][OW][16] int8 with the next layer's input multiplier, so that layer needs no
// separate quantise pass (and `out` may be skipped entirely when nothing else reads it).
// Works on 16-row halves (one 16-channel group each); strip >= 16 * TN * 32 floats.
// Requires m_base % 32 == 0 and M % 16 == 0.
template <int TN>
__device__ __forceinline__ void store_rows_via_lds_q(float *strip, const float (&vals)[TN][16], int m_base, int M,
                                                     int n_base, int Ntotal, int OHW, float *out,
                                                     const float *add, float *out_add,
                                                     int8_t *q_out, float q_mult, int q_G, int lane)
{
    constexpr int ROW = TN * 32;
    constexpr int LPR = TN * 8;
    constexpr int RPI = 64 / LPR;
    constexpr int NI = 16 / RPI;              // store instructions per 16-row half
    const int l31 = lane & 31, half = lane >> 5;
    const int c4 = lane % LPR;
    const int rsub = lane / LPR;
    const int n = n_base + c4 * 4;
    const int ob = n / OHW;
    const int opix = n - ob * OHW;
    const bool in_range = n + 3 < Ntotal;
    const bool same_img = in_range && (opix + 3 < OHW);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int gg = 0; gg < 2; ++gg)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    strip[(8 * gg + r + 4 * half) * ROW + j * 32 + l31] = vals[j][4 * (2 * h + gg) + r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (out || add) {
#pragma unroll
            for (int t = 0; t < NI; ++t) {
                const int row = t * RPI + rsub;                    // 0..15 inside the half
                const int m = m_base + 16 * h + row;
                float4 v = *reinterpret_cast<const float4 *>(strip + row * ROW + c4 * 4);
                if (m < M && n < Ntotal) {
                    const size_t o = ((size_t)ob * M + m) * OHW + opix;
                    if (same_img && ((o & 3) == 0)) {
                        if (out) *reinterpret_cast<float4 *>(out + o) = v;
                        if (add) {
                            const float4 a = *reinterpret_cast<const float4 *>(add + o);
                            v.x = __fadd_rn(v.x, a.x); v.y = __fadd_rn(v.y, a.y);
                            v.z = __fadd_rn(v.z, a.z); v.w = __fadd_rn(v.w, a.w);
                            *reinterpret_cast<float4 *>(out_add + o) = v;
                        }
                    } else {
                        float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int nq = n + q;
                            if (nq < Ntotal) {
                                const int obq = nq / OHW;
                                const size_t oq = ((size_t)obq * M + m) * OHW + (nq - obq * OHW);
                                if (out) out[oq] = vv[q];
                                if (add) { vv[q] = __fadd_rn(vv[q], add[oq]); out_add[oq] = vv[q]; }
                            }
                        }
                        v.x = vv[0]; v.y = vv[1]; v.z = vv[2]; v.w = vv[3];
                    }
                    // the quantised side output is taken from the tensor the next layer reads
                    if (add) *reinterpret_cast<float4 *>(strip + row * ROW + c4 * 4) = v;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // ---- quantised 16-byte units: one lane = one pixel, 16 channels down the strip column ----
        const int cg = (m_base + 16 * h) >> 4;
        if (m_base + 16 * h < M) {
#pragma unroll
            for (int pass = 0; pass < (ROW + 63) / 64; ++pass) {
                const int col = pass * 64 + lane;
                const int nq = n_base + col;
                if (col < ROW && nq < Ntotal) {
                    unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int qv = quantize_input_i8(strip[r * ROW + col], q_mult);
                        w[r >> 2] |= ((unsigned)(qv & 0xFF)) << ((r & 3) * 8);
                    }
                    const int obq = nq / OHW;
                    const size_t unit = ((size_t)obq * q_G + cg) * OHW + (nq - obq * OHW);
                    *reinterpret_cast<uint4 *>(q_out + unit * 16) = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

}  // namespace yl
