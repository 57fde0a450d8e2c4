// conv_f32_wino.hip -- K1w: 3x3 / stride 1 / pad 1 FP32 convolution as Winograd F(2x2,3x3) on
// v_mfma_f32_32x32x2_f32, input transform, 16 plane GEMMs and output transform fused in one kernel.
//
// Same layer as conv_f32_mfma_v2.hip computes (forward_convolutional_layer_cpu FP32 branch,
// src/yolov2_forward_network.c:204-261): out = act(conv3x3(in, w) + bias).  The reference does
// im2col + gemm_nn (9 multiplies per output, channel and filter); F(2x2,3x3) needs 16 multiplies
// per 2x2 output tile = 4 per output -- 2.25x fewer MFMA flops, which matters because K1 already
// runs at ~92 % of what the FP32 matrix pipe delivers at the clock it sustains.
// 32 of yolov3's 75 convolutions (77 % of its FLOPs) have this shape.
//
//   U[xi]   = G g G^T              per (filter m, channel c): 4x4, packed by the host once
//   V[xi]   = B^T d B              per (tile t, channel c): d = 4x4 input patch, zero outside
//   M[xi]   = sum_c U[xi][m][c] * V[xi][c][t]        16 independent GEMMs, xi = 4*i + j
//   Y       = A^T M A              2x2 outputs of tile t for filter m; + bias, leaky, [shortcut]
//
// FP32 error: the transforms only add/subtract and halve, measured max |err| = 4e-6 of the layer
// RMS against 2.5e-6 for the direct kernel (bar 1e-4, tests/common.py::fp32_close).
//
// Workgroup = 4 waves (one per SIMD, 512 registers each), tile = 64 filters x 64 tiles x all 16
// planes: each wave owns a 32x32 (m, t) block of every plane = 16 accumulators of 16 registers,
// all of them in AccVGPRs.  K loop over channels in panels of 8:
//   LDS (2 stages, 128 KB):  A[xi][half][m][kk]   kk = 0..3  -> one ds_read_b128 per plane/panel
//                            B[xi][half][kp][t][2]           -> two ds_read_b64 per plane/panel
//     (`half` = lane >> 5 = which of the two k of a 32x32x2 step the lane feeds, k = 2*kk + half)
//   weights: the host packs U in exactly the stage layout, a panel is 32 KB of contiguous float4
//   input  : every thread gathers the 4x4 patches of 2 channels of one tile (one 16-byte buffer
//            load per patch row; halo rows -> voffset -1 -> 0.0 from the range check, halo columns
//            by lane selects), transforms them in registers (32 add/sub per patch) and writes 16
//            float2 -- im2col AND the V tensor never exist in HBM
// Schedule: registers hold the raw loads of panel kb+1 while panel kb is multiplied; the
// transform/LDS writes and the loads of panel kb+2 are pinned between the four 16-MFMA groups of
// the panel (sched_barrier); one workgroup barrier per panel (64 MFMAs = 4096 pipe cycles apart).
#include <hip/hip_runtime.h>
#include <cstdio>

#include "kernels.h"
#include "../../include/yolo2_hip.h"

namespace yl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// the return type of __builtin_amdgcn_raw_buffer_load_b128.  NB: __builtin_bit_cast(float, q[i]) on a
// vector element is miscompiled by this clang (every i reads element 0): use __uint_as_float
typedef unsigned int u32x4v __attribute__((__vector_size__(16)));

namespace {

constexpr int WBM = 64;                 // filters per workgroup
constexpr int WBT = 64;                 // 2x2-output tiles per workgroup
constexpr int WBK = 8;                  // channels per panel
constexpr int PANEL = 16 * WBK * 64;    // floats per operand panel (A or B) = 8192 = 32 KB
constexpr int PLANE = 2 * 64 * 4;       // floats per plane inside a panel = 512

struct ConvWinoDev {
    const float *in;
    const float *u;        // packed U: [tile_m][panel][xi][half][m 64][kk 4]
    const float *bias;
    const float *add;
    float *out_add;
    float *out;
    int B, C, H, W, M;
    int th, tw, tpi, T;    // tiles per column / row / image, total
    int tiles_m, tiles_t, nkb;
    int act;
};

// rows as loaded -> rows of the zero-padded patch: left-edge tiles loaded cols 0..3 and need
// (0, c0, c1, c2); cols >= W are zero
__device__ __forceinline__ void fix_patch_rows(float (&d)[16], bool left, bool inv2, bool inv3)
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

// B^T d B for one 4x4 patch d (row-major) -> v (row-major), 32 add/sub
__device__ __forceinline__ void input_transform(const float (&d)[16], float (&v)[16])
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

__global__ __launch_bounds__(256) void conv_f32_wino_kernel(ConvWinoDev p)
{
    __shared__ __attribute__((aligned(16))) float smem[4 * PANEL];
    float *As = smem;
    float *Bs = smem + 2 * PANEL;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int half = lane >> 5;

    // XCD-aware bijective remap (consecutive logical ids share an XCD's L2), then
    // logical -> (t group of 8, m tile, t in group): 8 neighbours share the U slice, the next 8 the
    // next filter tile over the same input tiles
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    constexpr int GT = 8;
    const int per_group = GT * p.tiles_m;
    const int tg = logical / per_group;
    const int rem_g = logical - tg * per_group;
    const int t_in_last = p.tiles_t - tg * GT;                 // t tiles in this group (last may be short)
    const int gsz = t_in_last < GT ? t_in_last : GT;
    const int tile_m = __builtin_amdgcn_readfirstlane(rem_g / gsz);
    const int tile_t = __builtin_amdgcn_readfirstlane(tg * GT + (rem_g - tile_m * gsz));
    const int m0 = tile_m * WBM;
    const int t0 = tile_t * WBT;

    const int HW = p.H * p.W;
    const int CHW = p.C * HW;

    // ---- staging role: tile t_s, channels (4*kp_s + half_s) and (4*kp_s + half_s + 2) of a panel ----
    const int t_s = tid & 63;
    const int half_s = wave & 1;
    const int kp_s = wave >> 1;
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
    // One 16-byte load per patch row (4 instead of 16 VMEM instructions per patch: the dword gather
    // kept the texture path, not the MFMA pipe, busy).  A row is cols 2*tj-1 .. 2*tj+2; rows outside
    // the image are dropped with voffset = -1 (range check -> 0.0).  The first tile of an image row
    // would start at col -1: it loads cols 0..3 instead and rotates (left), so no load ever starts
    // before the tensor; cols >= W on the right are loaded (they are the next row's first floats, or
    // fall to the range check at the very end of the tensor) and zeroed afterwards (inv2/inv3).
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
            pvr[rr] = ok ? (int)(base + (unsigned)(rr * p.W) * 4u) : -1;
        }
    }
    const int c_s = 4 * kp_s + half_s;                         // first channel inside a panel
    const float *u_tile = p.u + (size_t)tile_m * p.nkb * PANEL;

    float xr[2][16];
    float ur[8][4];

#define W_LOAD_X(KB)                                                                               \
    {                                                                                              \
        const int s0 = ((KB) * WBK + c_s) * HW * 4;                                                \
        const int s1 = s0 + 2 * HW * 4;                                                            \
        _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) {                                         \
            const u32x4v q0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, pvr[rr], s0, 0);          \
            xr[0][rr * 4 + 0] = __uint_as_float(q0[0]); xr[0][rr * 4 + 1] = __uint_as_float(q0[1]); \
            xr[0][rr * 4 + 2] = __uint_as_float(q0[2]); xr[0][rr * 4 + 3] = __uint_as_float(q0[3]); \
        }                                                                                          \
        _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) {                                         \
            const u32x4v q1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, pvr[rr], s1, 0);          \
            xr[1][rr * 4 + 0] = __uint_as_float(q1[0]); xr[1][rr * 4 + 1] = __uint_as_float(q1[1]); \
            xr[1][rr * 4 + 2] = __uint_as_float(q1[2]); xr[1][rr * 4 + 3] = __uint_as_float(q1[3]); \
        }                                                                                          \
    }
#define W_LOAD_U(KB)                                                                               \
    {                                                                                              \
        const float4 *src = reinterpret_cast<const float4 *>(u_tile + (size_t)(KB) * PANEL);       \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                            \
            const float4 t4 = src[tid + e * 256];                                                  \
            ur[e][0] = t4.x; ur[e][1] = t4.y; ur[e][2] = t4.z; ur[e][3] = t4.w;                    \
        }                                                                                          \
    }
#define W_STORE_X(BUF)                                                                             \
    {                                                                                              \
        float va[16], vb[16];                                                                      \
        fix_patch_rows(xr[0], left_s, inv2_s, inv3_s);                                             \
        fix_patch_rows(xr[1], left_s, inv2_s, inv3_s);                                             \
        input_transform(xr[0], va);                                                                \
        input_transform(xr[1], vb);                                                                \
        float *dst = Bs + (BUF) * PANEL + half_s * 256 + kp_s * 128 + t_s * 2;                     \
        _Pragma("unroll") for (int xi = 0; xi < 16; ++xi)                                          \
            *reinterpret_cast<float2 *>(dst + xi * PLANE) = make_float2(va[xi], vb[xi]);           \
    }
#define W_STORE_U(BUF)                                                                             \
    {                                                                                              \
        float4 *dst = reinterpret_cast<float4 *>(As + (BUF) * PANEL);                              \
        _Pragma("unroll") for (int e = 0; e < 8; ++e)                                              \
            dst[tid + e * 256] = make_float4(ur[e][0], ur[e][1], ur[e][2], ur[e][3]);              \
    }

    f32x16 acc[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[xi][e] = 0.f;

    const int wm = wave >> 1;
    const int wt = wave & 1;

    // ---- prologue: panel 0 -> LDS stage 0, panel 1 -> registers ----
    W_LOAD_X(0)
    W_LOAD_U(0)
    W_STORE_X(0)
    W_STORE_U(0)
    if (p.nkb > 1) {
        W_LOAD_X(1)
        W_LOAD_U(1)
    }
    __syncthreads();

    float4 fa[2][4];
    float2 fb[2][4][2];
#define W_READ_FRAGS_FROM(SET, G, AP, BP)                                                          \
    _Pragma("unroll") for (int pp = 0; pp < 4; ++pp) {                                             \
        fa[SET][pp] = *reinterpret_cast<const float4 *>((AP) + (4 * (G) + pp) * PLANE);            \
        fb[SET][pp][0] = *reinterpret_cast<const float2 *>((BP) + (4 * (G) + pp) * PLANE);         \
        fb[SET][pp][1] = *reinterpret_cast<const float2 *>((BP) + (4 * (G) + pp) * PLANE + 128);   \
    }
#define W_READ_FRAGS(SET, G) W_READ_FRAGS_FROM(SET, G, Ab, Bb)
#define W_MFMA_GROUP(SET, G)                                                                       \
    {                                                                                                            \
        _Pragma("unroll") for (int pp = 0; pp < 4; ++pp)                                           \
            acc[4 * (G) + pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][pp].x, fb[SET][pp][0].x, acc[4 * (G) + pp], 0, 0, 0); \
        _Pragma("unroll") for (int pp = 0; pp < 4; ++pp)                                           \
            acc[4 * (G) + pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][pp].y, fb[SET][pp][0].y, acc[4 * (G) + pp], 0, 0, 0); \
        _Pragma("unroll") for (int pp = 0; pp < 4; ++pp)                                           \
            acc[4 * (G) + pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][pp].z, fb[SET][pp][1].x, acc[4 * (G) + pp], 0, 0, 0); \
        _Pragma("unroll") for (int pp = 0; pp < 4; ++pp)                                           \
            acc[4 * (G) + pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][pp].w, fb[SET][pp][1].y, acc[4 * (G) + pp], 0, 0, 0); \
    }

    // one panel; DO_STORE: registers (panel kb+1) -> LDS[buf^1]; DO_LOAD: panel kb+2 -> registers.
    // Entered with the group-0 fragments of panel kb already in set 0.  The barrier sits after
    // group 2: by then every wave has written its share of panel kb+1 (groups 0/1) and issued its
    // last reads of panel kb (group-3 fragments, group 2), so behind it group 3 runs while the
    // group-0 fragments of panel kb+1 are fetched -- no LDS latency is exposed at the loop top.
    // sched_group_barrier masks: 0x008 MFMA, 0x002 VALU, 0x020 VMEM read, 0x100 DS read, 0x200 DS write.
    // One wave per SIMD: a cluster of fillers longer than an MFMA (64 cycles) drains the matrix
    // pipe, so every group spreads its fillers evenly: N fillers behind each of its 16 MFMAs.
#define W_PIPE(MASK, N) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(MASK, N, 0);
#define W_ITER(KB, DO_STORE, DO_LOAD)                                                              \
    {                                                                                              \
        const int buf = (KB) & 1;                                                                  \
        const float *Ab = As + buf * PANEL + half * 256 + (wm * 32 + l31) * 4;                     \
        const float *Bb = Bs + buf * PANEL + half * 256 + (wt * 32 + l31) * 2;                     \
        W_READ_FRAGS(1, 1)                                                                  \
        if (DO_STORE) W_STORE_X(buf ^ 1)                                                                   \
        W_MFMA_GROUP(0, 0)                                                                         \
        if (DO_STORE) {                                                                            \
            _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) {                                    \
                W_PIPE(0x002, 10) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);               \
            }                                                                                      \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { W_PIPE(0x200, 4) }                  \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        W_READ_FRAGS(0, 2)                                                                  \
        if (DO_LOAD) W_LOAD_X((KB) + 2)                                                                    \
        if (DO_STORE) W_STORE_U(buf ^ 1)                                                                   \
        W_MFMA_GROUP(1, 1)                                                                         \
        if (DO_LOAD) {                                                                             \
            _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {                                    \
                W_PIPE(0x020, 1) __builtin_amdgcn_sched_group_barrier(0x300, 1, 0);                \
            }                                                                                      \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        W_READ_FRAGS(1, 3)                                                                  \
        if (DO_LOAD) W_LOAD_U((KB) + 2)                                                                    \
        W_MFMA_GROUP(0, 2)                                                                         \
        if (DO_LOAD) {                                                                             \
            _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {                                    \
                W_PIPE(0x020, 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                \
            }                                                                                      \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        __syncthreads();                                                                           \
        if (DO_STORE) {                                                                            \
            const float *An = As + (buf ^ 1) * PANEL + half * 256 + (wm * 32 + l31) * 4;           \
            const float *Bn = Bs + (buf ^ 1) * PANEL + half * 256 + (wt * 32 + l31) * 2;           \
            W_READ_FRAGS_FROM(0, 0, An, Bn)                                                        \
        }                                                                                          \
        W_MFMA_GROUP(1, 3)                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }

    {
        const float *Ab = As + half * 256 + (wm * 32 + l31) * 4;
        const float *Bb = Bs + half * 256 + (wt * 32 + l31) * 2;
        W_READ_FRAGS(0, 0)
    }
    int kb = 0;
    for (; kb + 2 < p.nkb; ++kb) W_ITER(kb, true, true)
    if (kb + 1 < p.nkb) { W_ITER(kb, true, false) ++kb; }
    if (kb < p.nkb) W_ITER(kb, false, false)
#undef W_ITER
#undef W_PIPE
#undef W_MFMA_GROUP
#undef W_READ_FRAGS
#undef W_READ_FRAGS_FROM
#undef W_STORE_U
#undef W_STORE_X
#undef W_LOAD_U
#undef W_LOAD_X

    // ---- epilogue: Y = A^T M A per (filter, tile), + bias, activation, [shortcut], 2x2 stores.
    //      lane = tile column of the C/D layout: 32 consecutive tiles -> 64 consecutive pixels ----
    const int tg_e = t0 + wt * 32 + l31;
    if (tg_e >= p.T) return;
    const int b_e = tg_e / p.tpi;
    const int r_e = tg_e - b_e * p.tpi;
    const int ti_e = r_e / p.tw;
    const int tj_e = r_e - ti_e * p.tw;
    const int oy = 2 * ti_e, ox = 2 * tj_e;
    const bool row1 = oy + 1 < p.H;
    const bool col1 = ox + 1 < p.W;
    const bool vec2 = col1 && ((p.W & 1) == 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (m >= p.M) continue;
        float tmp[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tmp[0][j] = (acc[0 * 4 + j][e] + acc[1 * 4 + j][e]) + acc[2 * 4 + j][e];
            tmp[1][j] = (acc[1 * 4 + j][e] - acc[2 * 4 + j][e]) - acc[3 * 4 + j][e];
        }
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
        const size_t o0 = (((size_t)b_e * p.M + m) * p.H + oy) * p.W + ox;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i == 1 && !row1) break;
            const size_t o = o0 + (size_t)i * p.W;
            if (vec2) {
                if (p.out) *reinterpret_cast<float2 *>(p.out + o) = make_float2(y[i][0], y[i][1]);
                if (p.add) {
                    const float2 a = *reinterpret_cast<const float2 *>(p.add + o);
                    *reinterpret_cast<float2 *>(p.out_add + o) = make_float2(__fadd_rn(y[i][0], a.x), __fadd_rn(y[i][1], a.y));
                }
            } else {
                if (p.out) { p.out[o] = y[i][0]; if (col1) p.out[o + 1] = y[i][1]; }
                if (p.add) {
                    p.out_add[o] = __fadd_rn(y[i][0], p.add[o]);
                    if (col1) p.out_add[o + 1] = __fadd_rn(y[i][1], p.add[o + 1]);
                }
            }
        }
    }
}

// U = G g G^T per (m, c), packed [tile_m][panel][xi][half][m 64][kk 4]; k = panel*8 + 2*kk + half.
// Computed in double, rounded once.  dst must hold wino_packed_floats(C, M) floats.
size_t wino_packed_floats(int C, int M)
{
    const int tiles_m = (M + WBM - 1) / WBM;
    return (size_t)tiles_m * (C / WBK) * PANEL;
}

void wino_pack_weights(const float *w, int C, int M, float *dst)
{
    static const double G[4][3] = {{1., 0., 0.}, {.5, .5, .5}, {.5, -.5, .5}, {0., 0., 1.}};
    const int tiles_m = (M + WBM - 1) / WBM;
    const int nkb = C / WBK;
    for (int tm = 0; tm < tiles_m; ++tm)
        for (int kb = 0; kb < nkb; ++kb) {
            float *panel = dst + ((size_t)tm * nkb + kb) * PANEL;
            for (int ml = 0; ml < WBM; ++ml) {
                const int m = tm * WBM + ml;
                for (int kl = 0; kl < WBK; ++kl) {
                    const int c = kb * WBK + kl;
                    const int hf = kl & 1, kk = kl >> 1;
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
                        panel[xi * PLANE + hf * 256 + ml * 4 + kk] = (float)u[xi >> 2][xi & 3];
                }
            }
        }
}

bool wino_applicable(int C, int M, int size, int stride, int pad)
{
    return size == 3 && stride == 1 && pad == 1 && C % WBK == 0 && C >= 16 && M >= 1;      // + H, W >= 4 (launcher)
}

int launch_conv_f32_wino(const ConvF32Args &a, const float *u_packed, void *stream, char *name, size_t name_len)
{
    if (!wino_applicable(a.C, a.M, a.size, a.stride, a.pad) || a.OH != a.H || a.OW != a.W || a.H < 4 || a.W < 4)
        return (int)hipErrorInvalidValue;
    ConvWinoDev d;
    d.in = a.in; d.u = u_packed; d.bias = a.bias; d.add = a.add; d.out_add = a.out_add; d.out = a.out;
    d.B = a.B; d.C = a.C; d.H = a.H; d.W = a.W; d.M = a.M;
    d.th = (a.H + 1) / 2; d.tw = (a.W + 1) / 2; d.tpi = d.th * d.tw;
    const long long T = (long long)a.B * d.tpi;
    if (T > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    d.T = (int)T;
    d.tiles_m = (a.M + WBM - 1) / WBM;
    d.tiles_t = (int)((T + WBT - 1) / WBT);
    d.nkb = a.C / WBK;
    d.act = a.act;
    const long long blocks = (long long)d.tiles_m * d.tiles_t;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(conv_f32_wino_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d);
    if (name) snprintf(name, name_len, "conv_f32_wino<64x64t,f2x2>");
    return (int)hipGetLastError();
}

}  // namespace yl
