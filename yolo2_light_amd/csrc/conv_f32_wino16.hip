// conv_f32_wino16.hip -- K1w, round-3 form: 3x3 / stride 1 / pad 1 FP32 convolution as Winograd F(2x2,3x3) on
// v_mfma_f32_16x16x4_f32, every wave holding ALL 16 planes of its (filter, tile) block.
//
// Same function as conv_f32_wino32.hip (forward_convolutional_layer_cpu FP32 branch,
// src/yolov2_forward_network.c:204-261: out = act(conv3x3(in, w) + bias)), same workgroup tile (32 filters x 64
// tiles x 4-channel panels, 48 KB of LDS, two workgroups per CU) and the same staging (one 4x4 patch per thread
// and panel, transformed in registers in the shadow of the MFMAs).  What changed, and why:
//
//   * round 2's kernel gave a wave HALF of the planes of a 32x32 (filter, tile) block, so the output transform
//     Y = A^T M A needed the other half from the partner wave: 128 ds_write_b32 + 128 ds_read_b32 per lane, three
//     barriers and ~1 900 instructions per wave behind the K loop -- as many VALU issues as the whole K loop of a
//     128-channel layer (PMC: 6.8 VALU per MFMA over the kernel against 3.7 inside the loop), i.e. 15-30 % of a
//     workgroup's life on the 76^2 / 152^2 layers, spent with the matrix pipe idle on that wave.
//   * with the 16x16x4 MFMA (same FLOP rate: 32 cycles per instruction instead of 64 for a quarter of the block) a
//     wave keeps all 16 planes of a 32-filter x 16-tile block in the same 128 accumulator registers.  The output
//     transform is then 24 adds per 2x2 output tile on the lane's own registers: no LDS exchange, no barrier, ~3x
//     fewer instructions behind the loop.  One MFMA consumes one whole 4-channel panel of one plane (K = 4).
//   * fragments come out of LDS as 16-byte reads: A[xi/2][k][m16][fb][xi&1] gives a lane two planes x two filter
//     blocks per ds_read_b128 (lane-linear: conflict-free), B[xi/4][k][t][xi&3] four planes of its tile; a B
//     fragment feeds both filter blocks.  12 ds_read_b128 per wave and panel (was 16 ds_read_b64).
//   * the panel is cut into two halves of 16 MFMAs by PLANE (0-7 / 8-15) instead of by k: the fragments of the
//     other half are fetched under the running half, so no LDS latency stands in front of an MFMA block, with
//     48 instead of 64 fragment registers.
//
//   waves: w = wave -> tiles [16 w, +16) of the workgroup's 64; every wave reads the whole A panel
//   lane:  A operand  A[m = 16 fb + (l & 15)][k = l >> 4],  B operand  B[k = l >> 4][t = l & 15]
//          accumulator acc[xi][fb][r] = M[xi][m = 16 fb + 4 (l >> 4) + r][t = l & 15]
//   k of the MFMA = channel within the panel, ascending: the fma chain of one accumulator visits the channels in the
//   same order as conv_f32_wino32.hip's two K = 2 steps.
#include <hip/hip_runtime.h>
#include <cstdio>

#include "kernels.h"
#include "../../include/yolo2_hip.h"

namespace yl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
// the return type of __builtin_amdgcn_raw_buffer_load_b128 (see conv_f32_wino32.hip on __uint_as_float)
typedef unsigned int u32x4v __attribute__((__vector_size__(16)));

namespace {

constexpr int YBM = 32;                  // filters per workgroup
constexpr int YBT = 64;                  // tiles per workgroup
constexpr int YBK = 4;                   // channels per panel = K of one MFMA
constexpr int YPA = 16 * YBK * YBM;      // floats per A panel = 2048 (8 KB)
constexpr int YPB = 16 * YBK * YBT;      // floats per B panel = 4096 (16 KB)

struct ConvWino16Dev {
    const float *in;
    const float *u;        // packed U: [tile_m][panel][xi/2][k 4][m16][fb 2][xi&1]
    const float *bias;
    const float *add;
    float *out_add;
    float *out;
    int B, C, H, W, M;
    int th, tw, tpi, T;
    int tiles_m, tiles_t, nkb;
    int act;
};

// patch columns: the first tile of a row is loaded one float to the right and rotated (no load starts in front of
// the tensor); columns beyond the image are zeroed
__device__ __forceinline__ void fix_rows16(float (&d)[16], bool left, bool inv2, bool inv3)
{
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float x = d[r * 4 + 0], y = d[r * 4 + 1], z = d[r * 4 + 2], w = d[r * 4 + 3];
        d[r * 4 + 0] = left ? 0.f : x;
        d[r * 4 + 1] = left ? x : y;
        const float c2 = left ? y : z;
        const float c3 = left ? z : w;
        d[r * 4 + 2] = inv2 ? 0.f : c2;
        d[r * 4 + 3] = inv3 ? 0.f : c3;
    }
}

// V = B^T d B
__device__ __forceinline__ void input_transform16(const float (&d)[16], float (&v)[16])
{
    float w[16];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        w[0 * 4 + s] = d[0 * 4 + s] - d[2 * 4 + s];
        w[1 * 4 + s] = d[1 * 4 + s] + d[2 * 4 + s];
        w[2 * 4 + s] = d[2 * 4 + s] - d[1 * 4 + s];
        w[3 * 4 + s] = d[1 * 4 + s] - d[3 * 4 + s];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i * 4 + 0] = w[i * 4 + 0] - w[i * 4 + 2];
        v[i * 4 + 1] = w[i * 4 + 1] + w[i * 4 + 2];
        v[i * 4 + 2] = w[i * 4 + 2] - w[i * 4 + 1];
        v[i * 4 + 3] = w[i * 4 + 1] - w[i * 4 + 3];
    }
}

}  // namespace

// Epilogue of one wave: Y = A^T M A on the lane's own registers, + bias, leaky, fused [shortcut]; no LDS, no barrier.
// wt = which 16-tile quarter of the workgroup's 64 tiles this wave owns.
template <bool APF>
__device__ __forceinline__ void wino16_epilogue(const ConvWino16Dev &p, const f32x4 (&acc)[16][2], int wt, int lane, int m0, int t0)
{
    const int l15 = lane & 15;
    const int lk = lane >> 4;
    const int HW = p.H * p.W;
    // ---- epilogue: Y = A^T M A on the lane's own registers, + bias, leaky, fused [shortcut]; no LDS, no barrier ----
    const int tg_e = t0 + wt * 16 + l15;
    const bool t_ok_e = tg_e < p.T;
    const int b_e = t_ok_e ? tg_e / p.tpi : 0;
    const int r_e = tg_e - b_e * p.tpi;
    const int ti_e = r_e / p.tw;
    const int tj_e = r_e - ti_e * p.tw;
    const int oy = 2 * ti_e, ox = 2 * tj_e;
    const bool row1 = oy + 1 < p.H;
    const bool col1 = ox + 1 < p.W;
    const bool vec2 = col1 && ((p.W & 1) == 0);
    const unsigned HW4 = (unsigned)HW * 4u;
    const unsigned W4 = (unsigned)p.W * 4u;
    // byte offset of (b_e, m0 + 4 * lk, oy, ox): 32-bit (the launcher keeps Winograd to tensors below 4 GB)
    const unsigned obase = ((((unsigned)b_e * (unsigned)p.M + (unsigned)(m0 + 4 * lk)) * (unsigned)p.H + (unsigned)oy) *
                            (unsigned)p.W + (unsigned)ox) * 4u;
    const char *addb = reinterpret_cast<const char *>(p.add);
    char *outb = reinterpret_cast<char *>(p.out);
    char *oaddb = reinterpret_cast<char *>(p.out_add);
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
        float apf[4][2][2];
        if constexpr (APF) {
            if (p.add) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) apf[rr][i][0] = apf[rr][i][1] = 0.f;
                    if (m0 + 16 * fb + 4 * lk + rr < p.M && t_ok_e) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            if (i == 1 && !row1) break;
                            const unsigned o = obase + (unsigned)(16 * fb + rr) * HW4 + (unsigned)i * W4;
                            if (vec2) {
                                const float2 a = *reinterpret_cast<const float2 *>(addb + o);
                                apf[rr][i][0] = a.x; apf[rr][i][1] = a.y;
                            } else {
                                apf[rr][i][0] = *reinterpret_cast<const float *>(addb + o);
                                if (col1) apf[rr][i][1] = *reinterpret_cast<const float *>(addb + o + 4u);
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int m = m0 + 16 * fb + 4 * lk + rr;
            // rows of A^T M: tmp0 = (M0 + M1) + M2, tmp1 = M1 - (M2 + M3)   (same association as conv_f32_wino32.hip)
            float tmp[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float M0 = acc[0 + j][fb][rr], M1 = acc[4 + j][fb][rr], M2 = acc[8 + j][fb][rr], M3 = acc[12 + j][fb][rr];
                tmp[0][j] = (M0 + M1) + M2;
                tmp[1][j] = M1 - (M2 + M3);
            }
            if (m < p.M && t_ok_e) {
                const float bv = p.bias[m];
                float y[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    y[i][0] = ((tmp[i][0] + tmp[i][1]) + tmp[i][2]) + bv;
                    y[i][1] = ((tmp[i][1] - tmp[i][2]) - tmp[i][3]) + bv;
                    if (p.act == YL_LEAKY) {
                        y[i][0] = (y[i][0] > 0.f) ? y[i][0] : (float)(.1 * (double)y[i][0]);
                        y[i][1] = (y[i][1] > 0.f) ? y[i][1] : (float)(.1 * (double)y[i][1]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (i == 1 && !row1) break;
                    const unsigned o = obase + (unsigned)(16 * fb + rr) * HW4 + (unsigned)i * W4;
                    if (vec2) {
                        if (p.out) *reinterpret_cast<float2 *>(outb + o) = make_float2(y[i][0], y[i][1]);
                        if (p.add) {
                            float2 a;
                            if constexpr (APF) a = make_float2(apf[rr][i][0], apf[rr][i][1]);
                            else a = *reinterpret_cast<const float2 *>(addb + o);
                            *reinterpret_cast<float2 *>(oaddb + o) =
                                make_float2(__fadd_rn(y[i][0], a.x), __fadd_rn(y[i][1], a.y));
                        }
                    } else {
                        if (p.out) {
                            *reinterpret_cast<float *>(outb + o) = y[i][0];
                            if (col1) *reinterpret_cast<float *>(outb + o + 4u) = y[i][1];
                        }
                        if (p.add) {
                            *reinterpret_cast<float *>(oaddb + o) =
                                __fadd_rn(y[i][0], APF ? apf[rr][i][0] : *reinterpret_cast<const float *>(addb + o));
                            if (col1)
                                *reinterpret_cast<float *>(oaddb + o + 4u) =
                                    __fadd_rn(y[i][1], APF ? apf[rr][i][1] : *reinterpret_cast<const float *>(addb + o + 4u));
                        }
                    }
                }
            }
        }
    }
}

// APF: the fused [shortcut] operand of a filter block is requested before that block's output transform
template <bool APF>
__global__ __launch_bounds__(256, 2) void conv_f32_wino16_kernel(ConvWino16Dev p)
{
    __shared__ __attribute__((aligned(16))) float smem[2 * YPA + 2 * YPB];      // 48 KB
    float *As = smem;
    float *Bs = smem + 2 * YPA;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15;
    const int lk = lane >> 4;

    // ---- workgroup -> (filter tile, tile group): bijective XCD remap, groups of GT tile groups share the filters'
    //      U slices in one XCD's L2 ----
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    constexpr int GT = 8;
    const int per_group = GT * p.tiles_m;
    const int tg = logical / per_group;
    const int rem_g = logical - tg * per_group;
    const int t_in_last = p.tiles_t - tg * GT;
    const int gsz = t_in_last < GT ? t_in_last : GT;
    const int tile_m = __builtin_amdgcn_readfirstlane(rem_g / gsz);
    const int tile_t = __builtin_amdgcn_readfirstlane(tg * GT + (rem_g - tile_m * gsz));
    const int m0 = tile_m * YBM;
    const int t0 = tile_t * YBT;

    const int HW = p.H * p.W;
    const int CHW = p.C * HW;

    // ---- staging role: tile t_s, channel `wave` of every panel ----
    const int t_s = tid & 63;
    const int tg_s = t0 + t_s;
    const bool t_ok = tg_s < p.T;
    const int b_s = t_ok ? tg_s / p.tpi : 0;
    const int r_s = tg_s - b_s * p.tpi;
    const int ti_s = r_s / p.tw;
    const int tj_s = r_s - ti_s * p.tw;

    const int b_first = __builtin_amdgcn_readfirstlane(t0 / p.tpi);
    const float *tile_base = p.in + (size_t)b_first * CHW - (ptrdiff_t)(p.W + 1);
    size_t rec = ((size_t)p.B - b_first) * CHW * sizeof(float) + (size_t)(p.W + 1) * sizeof(float);
    if (rec > 0xFFFFFFFEull) rec = 0xFFFFFFFEull;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)tile_base, 0, (int)(unsigned)rec, 0x00020000);
    int pvr[4];
    const bool left_s = (tj_s == 0);
    const bool inv2_s = (2 * tj_s + 1 >= p.W);
    const bool inv3_s = (2 * tj_s + 2 >= p.W);
    {
        const unsigned base = ((unsigned)(b_s - b_first) * (unsigned)CHW + (unsigned)(2 * ti_s) * (unsigned)p.W +
                               (unsigned)(2 * tj_s) + (left_s ? 1u : 0u)) * 4u;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int iy = 2 * ti_s - 1 + rr;
            const bool ok = t_ok && iy >= 0 && iy < p.H;
            pvr[rr] = ok ? (int)(base + (unsigned)(rr * p.W) * 4u) : -1;          // halo rows: range check -> 0.0
        }
    }
    const float *u_tile = p.u + (size_t)tile_m * p.nkb * YPA;

    float xr[16];
    float ur[2][4];

#define Y_LOAD_X(KB)                                                                               \
    {                                                                                              \
        const int s0 = ((KB) * YBK + wave) * HW * 4;                                               \
        _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) {                                         \
            const u32x4v q0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, pvr[rr], s0, 0);         \
            xr[rr * 4 + 0] = __uint_as_float(q0[0]); xr[rr * 4 + 1] = __uint_as_float(q0[1]);      \
            xr[rr * 4 + 2] = __uint_as_float(q0[2]); xr[rr * 4 + 3] = __uint_as_float(q0[3]);      \
        }                                                                                          \
    }
#define Y_LOAD_U(KB)                                                                               \
    {                                                                                              \
        const float4 *src = reinterpret_cast<const float4 *>(u_tile + (size_t)(KB) * YPA);         \
        _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                            \
            const float4 t4 = src[tid + e * 256];                                                  \
            ur[e][0] = t4.x; ur[e][1] = t4.y; ur[e][2] = t4.z; ur[e][3] = t4.w;                    \
        }                                                                                          \
    }
    // B[xi/4][k = wave][t_s][xi & 3]: one ds_write_b128 per plane quad, consecutive lanes -> consecutive 16 bytes
#define Y_STORE_X(BUF)                                                                             \
    {                                                                                              \
        float va[16];                                                                              \
        fix_rows16(xr, left_s, inv2_s, inv3_s);                                                    \
        input_transform16(xr, va);                                                                 \
        float *dst = Bs + (BUF) * YPB + (wave * 64 + t_s) * 4;                                     \
        _Pragma("unroll") for (int qd = 0; qd < 4; ++qd)                                           \
            *reinterpret_cast<float4 *>(dst + qd * 1024) =                                         \
                make_float4(va[4 * qd], va[4 * qd + 1], va[4 * qd + 2], va[4 * qd + 3]);           \
    }
#define Y_STORE_U(BUF)                                                                             \
    {                                                                                              \
        float4 *dst = reinterpret_cast<float4 *>(As + (BUF) * YPA);                                \
        _Pragma("unroll") for (int e = 0; e < 2; ++e)                                              \
            dst[tid + e * 256] = make_float4(ur[e][0], ur[e][1], ur[e][2], ur[e][3]);              \
    }

    f32x4 acc[16][2];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[xi][fb][e] = 0.f;

    // fragments of one plane half: fa[pp] = planes (8 h + 2 pp, +1) x filter blocks (0, 1); fb_[qq] = planes 8 h + 4 qq .. +3
    float4 fa[2][4];
    float4 fbv[2][2];
#define Y_READ_FRAGS(HALF, BUF)                                                                    \
    {                                                                                              \
        const float *Ab = As + (BUF) * YPA + (HALF) * 1024 + lane * 4;                             \
        const float *Bb = Bs + (BUF) * YPB + (HALF) * 2048 + (lk * 64 + wave * 16 + l15) * 4;      \
        _Pragma("unroll") for (int pp = 0; pp < 4; ++pp)                                           \
            fa[HALF][pp] = *reinterpret_cast<const float4 *>(Ab + pp * 256);                       \
        _Pragma("unroll") for (int qq = 0; qq < 2; ++qq)                                           \
            fbv[HALF][qq] = *reinterpret_cast<const float4 *>(Bb + qq * 1024);                     \
    }
    // the 16 MFMAs of one plane half: plane xi = 8 h + 2 pp + pr, filter block fb
#define Y_MFMAS(HALF)                                                                              \
    _Pragma("unroll") for (int pp = 0; pp < 4; ++pp)                                               \
        _Pragma("unroll") for (int pr = 0; pr < 2; ++pr) {                                         \
            const int xi_ = 8 * (HALF) + 2 * pp + pr;                                              \
            const float4 bq = fbv[HALF][pp >> 1];                                                  \
            const int bi = 2 * (pp & 1) + pr;                                                      \
            const float bv_ = bi == 0 ? bq.x : (bi == 1 ? bq.y : (bi == 2 ? bq.z : bq.w));         \
            const float a0 = pr ? fa[HALF][pp].y : fa[HALF][pp].x;                                 \
            const float a1 = pr ? fa[HALF][pp].w : fa[HALF][pp].z;                                 \
            acc[xi_][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv_, acc[xi_][0], 0, 0, 0);     \
            acc[xi_][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv_, acc[xi_][1], 0, 0, 0);     \
        }

    // ---- prologue: panel 0 -> LDS stage 0 -> fragments of planes 0-7; panel 1 -> registers ----
    // (nkb = C/4 is even and >= 4: the launcher requires C % 8 == 0, C >= 16)
    Y_LOAD_X(0)
    Y_LOAD_U(0)
    Y_STORE_X(0)
    Y_STORE_U(0)
    Y_LOAD_X(1)
    Y_LOAD_U(1)
    __syncthreads();
    Y_READ_FRAGS(0, 0)

    // sched_group_barrier masks: 0x008 MFMA, 0x002 VALU, 0x020 VMEM read, 0x080 DS, 0x100 DS read, 0x200 DS write
#define Y_PIPE(MASK, N) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(MASK, N, 0);
    // One panel, entered with the fragments of planes 0-7 of panel kb in set 0.
    //   first half  (planes 0-7): registers (panel kb+1) -> transform -> LDS[buf^1]; fragments of planes 8-15 of
    //               panel kb (LDS[buf]) -> set 1
    //   barrier     panel kb+1 is complete in LDS[buf^1]; every read of LDS[buf] has been issued before it and is
    //               waited for in front of the second half, i.e. before anyone writes LDS[buf] again (next panel)
    //   second half (planes 8-15): fragments of planes 0-7 of panel kb+1 -> set 0; panel kb+2 -> registers
#define Y_ITER(KB, DO_STORE, DO_LOAD)                                                              \
    {                                                                                              \
        const int buf = (KB) & 1;                                                                  \
        if (DO_STORE) Y_STORE_U(buf ^ 1)                                                           \
        if (DO_STORE) Y_STORE_X(buf ^ 1)                                                           \
        Y_READ_FRAGS(1, buf)                                                                       \
        Y_MFMAS(0)                                                                                 \
        if (DO_STORE) {                                                                            \
            _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {                                    \
                Y_PIPE(0x002, 4) __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);                \
            }                                                                                      \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        __syncthreads();                                                                           \
        if (DO_STORE) Y_READ_FRAGS(0, buf ^ 1)                                                     \
        if (DO_LOAD) Y_LOAD_X((KB) + 2)                                                            \
        if (DO_LOAD) Y_LOAD_U((KB) + 2)                                                            \
        Y_MFMAS(1)                                                                                 \
        if (DO_STORE) {                                                                            \
            _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                                     \
                Y_PIPE(0x100, 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                \
            }                                                                                      \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }

    int kb = 0;
    for (; kb + 4 <= p.nkb; kb += 2) {
        Y_ITER(kb, true, true)
        Y_ITER(kb + 1, true, true)
    }
    Y_ITER(kb, true, false)
    Y_ITER(kb + 1, false, false)
#undef Y_ITER
#undef Y_PIPE
#undef Y_MFMAS
#undef Y_READ_FRAGS
#undef Y_STORE_U
#undef Y_STORE_X
#undef Y_LOAD_U
#undef Y_LOAD_X

    wino16_epilogue<APF>(p, acc, wave, lane, m0, t0);
}

// ---------------------------------------------------------------------------------------------------------------
// K1w, warp-specialised form (round 3).  Same tile, same packing, same arithmetic in the same order as the kernel
// above (bit-identical results) -- what changes is WHO does what:
//
//   waves 0-3  "matrix" waves, one per SIMD: nothing but ds_read_b128 fragments + 32 MFMAs per panel + the epilogue
//   waves 4-7  "staging" waves, one per SIMD: patch loads, fix-up, B^T d B, LDS stores of panel kb+2, U loads/stores
//
// Why (same-box ablation of the round-2 kernel, profiles/r3_wino_ablation.txt): with loads, transform and LDS stores
// compiled out the kernel runs 0.78 ms on the [512,2304,1444] layer against 1.13 ms with them -- the staging work,
// issued by the SAME waves between their MFMAs, costs 30 % of the time; barriers 1-2 %; a second workgroup per CU
// adds only 17 % over a single one.  An in-order wave that stalls on a patch load or an LDS store cannot issue its
// next MFMA; a wave that issues nothing but MFMAs and fragment reads never stalls, and the VALU / VMEM / LDS-store
// work of a different wave co-issues beside it (separate pipes).  One workgroup of 8 waves per CU; LDS is a ring of
// THREE 24 KB stages so that one barrier per panel is enough and no LDS latency is ever in front of an MFMA:
//
//   staging, iteration j:  registers (panel j+2) -> transform -> stage (j+2)%3 ; panel j+3 -> registers ; barrier j
//   matrix,  iteration j:  planes 0-7 of panel j (fragments fetched during j-1) while fetching planes 8-15 of panel j;
//                          planes 8-15 while fetching planes 0-7 of panel j+1 (published by barrier j-1) ; barrier j
//   stage (j+2)%3 last held panel j-1, whose last fragment reads were issued before barrier j-1.
//
// The staging waves keep two register sets: a load issued at the start of iteration j is consumed in iteration j+1.
template <bool APF>
__global__ __launch_bounds__(512) void conv_f32_wino16ws_kernel(ConvWino16Dev p)
{
    constexpr int STAGE = YPA + YPB;                                            // 6144 floats = 24 KB
    __shared__ __attribute__((aligned(16))) float smem[3 * STAGE];              // 72 KB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15;
    const int lk = lane >> 4;

    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    constexpr int GT = 8;
    const int per_group = GT * p.tiles_m;
    const int tg = logical / per_group;
    const int rem_g = logical - tg * per_group;
    const int t_in_last = p.tiles_t - tg * GT;
    const int gsz = t_in_last < GT ? t_in_last : GT;
    const int tile_m = __builtin_amdgcn_readfirstlane(rem_g / gsz);
    const int tile_t = __builtin_amdgcn_readfirstlane(tg * GT + (rem_g - tile_m * gsz));
    const int m0 = tile_m * YBM;
    const int t0 = tile_t * YBT;
    const int nkb = p.nkb;

    if (wave >= 4) {
        // ================================================================== staging waves
        const int ch = wave - 4;                       // channel of every panel this wave stages
        const int HW = p.H * p.W;
        const int CHW = p.C * HW;
        const int t_s = lane;
        const int tg_s = t0 + t_s;
        const bool t_ok = tg_s < p.T;
        const int b_s = t_ok ? tg_s / p.tpi : 0;
        const int r_s = tg_s - b_s * p.tpi;
        const int ti_s = r_s / p.tw;
        const int tj_s = r_s - ti_s * p.tw;
        const int b_first = __builtin_amdgcn_readfirstlane(t0 / p.tpi);
        const float *tile_base = p.in + (size_t)b_first * CHW - (ptrdiff_t)(p.W + 1);
        size_t rec = ((size_t)p.B - b_first) * CHW * sizeof(float) + (size_t)(p.W + 1) * sizeof(float);
        if (rec > 0xFFFFFFFEull) rec = 0xFFFFFFFEull;
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void *)tile_base, 0, (int)(unsigned)rec, 0x00020000);
        int pvr[4];
        const bool left_s = (tj_s == 0);
        const bool inv2_s = (2 * tj_s + 1 >= p.W);
        const bool inv3_s = (2 * tj_s + 2 >= p.W);
        {
            const unsigned base = ((unsigned)(b_s - b_first) * (unsigned)CHW + (unsigned)(2 * ti_s) * (unsigned)p.W +
                                   (unsigned)(2 * tj_s) + (left_s ? 1u : 0u)) * 4u;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int iy = 2 * ti_s - 1 + rr;
                const bool ok = t_ok && iy >= 0 && iy < p.H;
                pvr[rr] = ok ? (int)(base + (unsigned)(rr * p.W) * 4u) : -1;      // halo rows: range check -> 0.0
            }
        }
        const float *u_tile = p.u + (size_t)tile_m * nkb * YPA;
        const int stid = tid - 256;                    // 0..255 within the staging half
        // two register sets: the loads of panel j+3 are issued at the START of iteration j and consumed in iteration
        // j+1, so a whole iteration (>= 1024 cycles of the matrix waves) hides their latency.  (First version: one
        // set, loaded at the end of an iteration and consumed right behind the barrier -- every iteration paid a
        // full L2 / HBM round trip and the matrix waves waited at the barrier: 0.41 of the MFMA peak.)
        float xa[16], xb[16];
        float ua[2][4], ub[2][4];
#define W_LOAD(KB, XR, UR)                                                                         \
    {                                                                                              \
        const int s0 = ((KB) * YBK + ch) * HW * 4;                                                 \
        _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) {                                         \
            const u32x4v q0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, pvr[rr], s0, 0);         \
            XR[rr * 4 + 0] = __uint_as_float(q0[0]); XR[rr * 4 + 1] = __uint_as_float(q0[1]);      \
            XR[rr * 4 + 2] = __uint_as_float(q0[2]); XR[rr * 4 + 3] = __uint_as_float(q0[3]);      \
        }                                                                                          \
        const float4 *src = reinterpret_cast<const float4 *>(u_tile + (size_t)(KB) * YPA);         \
        _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                            \
            const float4 t4 = src[stid + e * 256];                                                 \
            UR[e][0] = t4.x; UR[e][1] = t4.y; UR[e][2] = t4.z; UR[e][3] = t4.w;                    \
        }                                                                                          \
    }
#define W_STORE(STG, XR, UR)                                                                       \
    {                                                                                              \
        float *As_ = smem + (STG) * STAGE;                                                         \
        float4 *adst = reinterpret_cast<float4 *>(As_);                                            \
        _Pragma("unroll") for (int e = 0; e < 2; ++e)                                              \
            adst[stid + e * 256] = make_float4(UR[e][0], UR[e][1], UR[e][2], UR[e][3]);            \
        float va[16];                                                                              \
        fix_rows16(XR, left_s, inv2_s, inv3_s);                                                    \
        input_transform16(XR, va);                                                                 \
        float *dst = As_ + YPA + (ch * 64 + t_s) * 4;                                              \
        _Pragma("unroll") for (int qd = 0; qd < 4; ++qd)                                           \
            *reinterpret_cast<float4 *>(dst + qd * 1024) =                                         \
                make_float4(va[4 * qd], va[4 * qd + 1], va[4 * qd + 2], va[4 * qd + 3]);           \
    }
        // prologue: panels 0, 1 -> stages 0, 1; panel 2 -> set A   (nkb = C/4 is even and >= 4)
        W_LOAD(0, xa, ua)
        W_LOAD(1, xb, ub)
        W_STORE(0, xa, ua)
        W_STORE(1, xb, ub)
        W_LOAD(2, xa, ua)
        __syncthreads();                               // barrier -1: panels 0 and 1 are published
        int st = 2;                                    // stage of panel j + 2
        // Straight-line body, no guards: behind the last panel the loads re-read panel nkb-1 and the stores refill ring
        // slots nobody reads any more (the slot of panel j+2 is free by construction).  A guard would put a branch join
        // in front of every s_waitcnt and the compiler then assumes the guarded loads were NOT issued: it waits for
        // the new loads instead of the old ones (vmcnt(1) where vmcnt(7) is meant) and the prefetch is lost.
        const int last = nkb - 1;
        for (int j = 0; j < nkb; j += 2) {
            // iteration j: set A holds panel j+2
            { const int kb_ = j + 3 < last ? j + 3 : last; W_LOAD(kb_, xb, ub) }
            __builtin_amdgcn_sched_barrier(0);         // the loads go out FIRST (the scheduler would sink them below the stores)
            W_STORE(st, xa, ua)
            st = (st == 2) ? 0 : st + 1;
            __syncthreads();                           // barrier j
            // iteration j+1: set B holds panel j+3
            { const int kb_ = j + 4 < last ? j + 4 : last; W_LOAD(kb_, xa, ua) }
            __builtin_amdgcn_sched_barrier(0);
            W_STORE(st, xb, ub)
            st = (st == 2) ? 0 : st + 1;
            __syncthreads();                           // barrier j+1
        }
#undef W_STORE
#undef W_LOAD
        return;
    }

    // ====================================================================== matrix waves
    f32x4 acc[16][2];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[xi][fb][e] = 0.f;
    float4 fa[2][4];
    float4 fbv[2][2];
    const int a_off = lane * 4;
    const int b_off = YPA + (lk * 64 + wave * 16 + l15) * 4;
#define W_READ_FRAGS(HALF, STG)                                                                    \
    {                                                                                              \
        const float *Ab = smem + (STG) * STAGE + (HALF) * 1024 + a_off;                            \
        const float *Bb = smem + (STG) * STAGE + (HALF) * 2048 + b_off;                            \
        _Pragma("unroll") for (int pp = 0; pp < 4; ++pp)                                           \
            fa[HALF][pp] = *reinterpret_cast<const float4 *>(Ab + pp * 256);                       \
        _Pragma("unroll") for (int qq = 0; qq < 2; ++qq)                                           \
            fbv[HALF][qq] = *reinterpret_cast<const float4 *>(Bb + qq * 1024);                     \
    }
#define W_MFMAS(HALF)                                                                              \
    _Pragma("unroll") for (int pp = 0; pp < 4; ++pp)                                               \
        _Pragma("unroll") for (int pr = 0; pr < 2; ++pr) {                                         \
            const int xi_ = 8 * (HALF) + 2 * pp + pr;                                              \
            const float4 bq = fbv[HALF][pp >> 1];                                                  \
            const int bi = 2 * (pp & 1) + pr;                                                      \
            const float bv_ = bi == 0 ? bq.x : (bi == 1 ? bq.y : (bi == 2 ? bq.z : bq.w));         \
            const float a0 = pr ? fa[HALF][pp].y : fa[HALF][pp].x;                                 \
            const float a1 = pr ? fa[HALF][pp].w : fa[HALF][pp].z;                                 \
            acc[xi_][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv_, acc[xi_][0], 0, 0, 0);     \
            acc[xi_][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv_, acc[xi_][1], 0, 0, 0);     \
        }
    __syncthreads();                                   // barrier -1
    W_READ_FRAGS(0, 0)
    int st = 0, st1 = 1;                               // stages of panels j and j + 1
    for (int j = 0; j < nkb; ++j) {
        W_READ_FRAGS(1, st)                            // planes 8-15 of panel j: used by the second half below
        W_MFMAS(0)
#pragma unroll
        for (int i_ = 0; i_ < 6; ++i_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // planes 0-7 of panel j+1 (published by barrier j-1).  Unconditional: behind the last panel the stage holds stale
        // data nobody uses, and a branch here makes the compiler merge LDS wait counts across the join (it then waits
        // for these reads in front of the MFMAs below)
        W_READ_FRAGS(0, st1)
        W_MFMAS(1)
#pragma unroll
        for (int i_ = 0; i_ < 6; ++i_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        st = st1;
        st1 = (st1 == 2) ? 0 : st1 + 1;
        __syncthreads();                               // barrier j
    }
#undef W_MFMAS
#undef W_READ_FRAGS
    wino16_epilogue<APF>(p, acc, wave, lane, m0, t0);
}

size_t wino16_packed_floats(int C, int M)
{
    const int tiles_m = (M + YBM - 1) / YBM;
    return (size_t)tiles_m * (C / YBK) * YPA;
}

// U = G g G^T (double, rounded once), packed [tile_m][panel][xi/2][k][m16][fb][xi&1]: the lane-linear image of the
// A stage (a lane's ds_read_b128 = planes (2 xp, 2 xp + 1) x filter blocks (0, 1) of its (m16, k))
void wino16_pack_weights(const float *w, int C, int M, float *dst)
{
    static const double G[4][3] = {{1., 0., 0.}, {.5, .5, .5}, {.5, -.5, .5}, {0., 0., 1.}};
    const int tiles_m = (M + YBM - 1) / YBM;
    const int nkb = C / YBK;
    for (int tm = 0; tm < tiles_m; ++tm)
        for (int kb = 0; kb < nkb; ++kb) {
            float *panel = dst + ((size_t)tm * nkb + kb) * YPA;
            for (int ml = 0; ml < YBM; ++ml) {
                const int m = tm * YBM + ml;
                const int fb = ml >> 4, m16 = ml & 15;
                for (int k = 0; k < YBK; ++k) {
                    const int c = kb * YBK + k;
                    double u[4][4];
                    if (m < M) {
                        const float *g = w + ((size_t)m * C + c) * 9;
                        double t[4][3];
                        for (int i = 0; i < 4; ++i)
                            for (int b = 0; b < 3; ++b)
                                t[i][b] = G[i][0] * g[0 * 3 + b] + G[i][1] * g[1 * 3 + b] + G[i][2] * g[2 * 3 + b];
                        for (int i = 0; i < 4; ++i)
                            for (int j = 0; j < 4; ++j)
                                u[i][j] = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
                    } else {
                        for (int i = 0; i < 4; ++i)
                            for (int j = 0; j < 4; ++j) u[i][j] = 0.;
                    }
                    for (int xi = 0; xi < 16; ++xi)
                        panel[(((xi >> 1) * 4 + k) * 16 + m16) * 4 + fb * 2 + (xi & 1)] = (float)u[xi >> 2][xi & 3];
                }
            }
        }
}

int launch_conv_f32_wino16(const ConvF32Args &a, const float *u_packed, int variant, void *stream, char *name, size_t name_len)
{
    if (!wino_applicable(a.C, a.M, a.size, a.stride, a.pad) || a.OH != a.H || a.OW != a.W || a.H < 4 || a.W < 4)
        return (int)hipErrorInvalidValue;
    ConvWino16Dev d;
    d.in = a.in; d.u = u_packed; d.bias = a.bias; d.add = a.add; d.out_add = a.out_add; d.out = a.out;
    d.B = a.B; d.C = a.C; d.H = a.H; d.W = a.W; d.M = a.M;
    d.th = (a.H + 1) / 2; d.tw = (a.W + 1) / 2; d.tpi = d.th * d.tw;
    const long long T = (long long)a.B * d.tpi;
    if (T > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    if (!wino32_fits(a.B, a.M, a.H, a.W)) return (int)hipErrorInvalidValue;
    d.T = (int)T;
    d.tiles_m = (a.M + YBM - 1) / YBM;
    d.tiles_t = (int)((T + YBT - 1) / YBT);
    d.nkb = a.C / YBK;
    if (d.nkb < 4 || (d.nkb & 1)) return (int)hipErrorInvalidValue;
    d.act = a.act;
    const long long blocks = (long long)d.tiles_m * d.tiles_t;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (variant & 64) {          // warp-specialised: 4 matrix + 4 staging waves, one workgroup per CU
        const dim3 block8(512);
        if (variant & 2) hipLaunchKernelGGL(conv_f32_wino16ws_kernel<true>, grid, block8, 0, s, d);
        else hipLaunchKernelGGL(conv_f32_wino16ws_kernel<false>, grid, block8, 0, s, d);
    } else if (variant & 2) hipLaunchKernelGGL(conv_f32_wino16_kernel<true>, grid, block, 0, s, d);
    else hipLaunchKernelGGL(conv_f32_wino16_kernel<false>, grid, block, 0, s, d);
    if (name) snprintf(name, name_len, "conv_f32_wino<32x64t,f2x2,p16%s%s>", (variant & 64) ? ",ws" : "", (variant & 2) ? ",apf" : "");
    return (int)hipGetLastError();
}

}  // namespace yl
