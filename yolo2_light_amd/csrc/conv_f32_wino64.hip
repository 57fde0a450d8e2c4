// conv_f32_wino64.hip -- K1w, 64-filter form: the Winograd F(2x2,3x3) kernel of conv_f32_wino32.hip with a
// 64-filter x 64-tile workgroup of EIGHT waves (one workgroup per CU, two waves per SIMD).
//
// Same function (forward_convolutional_layer_cpu FP32 branch, src/yolov2_forward_network.c:204-261), same MFMA
// (v_mfma_f32_32x32x2_f32), same per-wave block (8 planes of a 32x32 (filter, tile) block = 128 accumulators), the
// same arithmetic in the same order: results are bit-identical to conv_f32_wino32.hip.  What changes is how much
// staging work one MFMA has to carry.  Same-box ablation of the 32-filter kernel (profiles/r3_wino_ablation.txt,
// [512,2304,1444] layer, batch 64): 1.14 ms as shipped, 1.01 ms without its global loads, 0.78 ms without loads,
// input transform and LDS stores (0.62 ms = the 16-multiply MFMA work at 2.4 GHz); barriers cost 1-2 %, a second
// workgroup per CU adds 17 % over one, and moving the staging work to other waves does not help while only one wave
// per SIMD issues MFMAs (conv_f32_wino16.hip, warp-specialised form).  So the lever is less staging per MFMA:
//
//                                     32f x 64t (2 workgroups = 8 waves)      64f x 64t (1 workgroup = 8 waves)
//   MFMAs per panel                   2 x 64                                  128
//   patches loaded + transformed      2 x 256                                 256      (every patch feeds 64 filters)
//   global -> LDS bytes               2 x (8 + 16) KB                         16 + 16 KB
//   fragment reads                    the same
//
// A patch is staged by TWO threads: the waves of plane half hv = wave >> 2 load the three patch rows that rows
// 2 hv, 2 hv + 1 of B^T d need, and produce exactly the 8 planes [8 hv, +8) -- 3 x 16-byte loads, 18 selects and 16
// adds per thread and panel instead of 4 / 24 / 32.
//
//   waves: wt = wave & 1 -> tiles [32 wt, +32); wf = (wave >> 1) & 1 -> filters [32 wf, +32); ph = wave >> 2 -> planes [8 ph, +8)
//   LDS:   A[xi][half][m 64][kk 2]  (16 KB),  B[xi/2][half][kk 2][t 64][xi&1]  (16 KB), two stages: 64 KB
//   epilogue: as conv_f32_wino32.hip -- the two waves of a block (wave, wave ^ 4) swap partial row sums of A^T M
//          through the dead stages (8 x 8 KB), each finishing 8 of the 16 accumulator rows.
#include <hip/hip_runtime.h>
#include <cstdio>

#include "kernels.h"
#include "../../include/yolo2_hip.h"

namespace yl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4v __attribute__((__vector_size__(16)));      // see conv_f32_wino32.hip on __uint_as_float

namespace {

constexpr int ZBM = 64;
constexpr int ZBT = 64;
constexpr int ZBK = 4;
constexpr int ZPA = 16 * ZBK * ZBM;      // floats per A panel = 4096 (16 KB)
constexpr int ZPB = 16 * ZBK * ZBT;      // floats per B panel = 4096 (16 KB)

struct ConvWino64Dev {
    const float *in;
    const float *u;        // packed U: [tile_m][panel][xi][half][m 64][kk 2]
    const float *bias;
    const float *add;
    float *out_add;
    float *out;
    int B, C, H, W, M;
    int th, tw, tpi, T;
    int tiles_m, tiles_t, nkb;
    int act;
};

// columns of three patch rows: the first tile of an image row is loaded one float to the right and rotated, columns
// beyond the image are zeroed
__device__ __forceinline__ void fix_rows64(float (&d)[12], bool left, bool inv2, bool inv3)
{
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float x = d[r * 4 + 0], y = d[r * 4 + 1], z = d[r * 4 + 2], w = d[r * 4 + 3];
        d[r * 4 + 0] = left ? 0.f : x;
        d[r * 4 + 1] = left ? x : y;
        const float c2 = left ? y : z;
        const float c3 = left ? z : w;
        d[r * 4 + 2] = inv2 ? 0.f : c2;
        d[r * 4 + 3] = inv3 ? 0.f : c3;
    }
}

// rows 2 HV, 2 HV + 1 of V = B^T d B from patch rows [HV, HV + 3): the same subtractions / additions, in the same
// order, as conv_f32_wino32.hip's input_transform32 performs for these eight elements
template <int HV>
__device__ __forceinline__ void input_transform64(const float (&d)[12], float (&v)[8])
{
    float w[8];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (HV == 0) {              // d rows 0, 1, 2:  w0 = d0 - d2,  w1 = d1 + d2
            w[0 * 4 + s] = d[0 * 4 + s] - d[2 * 4 + s];
            w[1 * 4 + s] = d[1 * 4 + s] + d[2 * 4 + s];
        } else {                    // d rows 1, 2, 3:  w2 = d2 - d1,  w3 = d1 - d3
            w[0 * 4 + s] = d[1 * 4 + s] - d[0 * 4 + s];
            w[1 * 4 + s] = d[0 * 4 + s] - d[2 * 4 + s];
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        v[i * 4 + 0] = w[i * 4 + 0] - w[i * 4 + 2];
        v[i * 4 + 1] = w[i * 4 + 1] + w[i * 4 + 2];
        v[i * 4 + 2] = w[i * 4 + 2] - w[i * 4 + 1];
        v[i * 4 + 3] = w[i * 4 + 1] - w[i * 4 + 3];
    }
}

}  // namespace

// Epilogue of one wave (the scheme of conv_f32_wino32.hip): wave (wt, wf, PH) holds M[i][j] for rows i = 2 PH, 2 PH + 1
// of its 32x32 block; the partner wave ^ 4 holds the other two rows.  Partial row sums travel through
// xch[wave][32][64] floats in the dead panel stages.
template <int PH, bool APF>
__device__ __forceinline__ void wino64_epilogue(const ConvWino64Dev &p, const f32x16 (&acc)[8], float *xch, int wave, int lane,
                                                int mw0, int tw0)
{
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const int tg_e = tw0 + l31;
    const bool t_ok_e = tg_e < p.T;
    const int b_e = t_ok_e ? tg_e / p.tpi : 0;
    const int r_e = tg_e - b_e * p.tpi;
    const int ti_e = r_e / p.tw;
    const int tj_e = r_e - ti_e * p.tw;
    const int oy = 2 * ti_e, ox = 2 * tj_e;
    const bool row1 = oy + 1 < p.H;
    const bool col1 = ox + 1 < p.W;
    const bool vec2 = col1 && ((p.W & 1) == 0);
    const unsigned HW4 = (unsigned)(p.H * p.W) * 4u;
    const unsigned W4 = (unsigned)p.W * 4u;
    // byte offset of (b_e, mw0 + 4*half, oy, ox): 32-bit (the launcher keeps Winograd to tensors below 4 GB)
    const unsigned obase = ((((unsigned)b_e * (unsigned)p.M + (unsigned)(mw0 + 4 * half)) * (unsigned)p.H + (unsigned)oy) *
                            (unsigned)p.W + (unsigned)ox) * 4u;
    const char *addb = reinterpret_cast<const char *>(p.add);
    char *outb = reinterpret_cast<char *>(p.out);
    char *oaddb = reinterpret_cast<char *>(p.out_add);
    float *mine = xch + wave * 2048 + lane;
    const float *theirs = xch + (wave ^ 4) * 2048 + lane;
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
        float apf[4][2][2];
        if constexpr (APF) {
            if (p.add) {
#pragma unroll
                for (int ee = 0; ee < 4; ++ee) {
                    const int e = 8 * rnd + (PH ? 4 + ee : ee);
                    const int mrow = (e & 3) + 8 * (e >> 2);            // + 4*half is in obase
#pragma unroll
                    for (int i = 0; i < 2; ++i) apf[ee][i][0] = apf[ee][i][1] = 0.f;
                    if (mw0 + mrow + 4 * half < p.M && t_ok_e) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            if (i == 1 && !row1) break;
                            const unsigned o = obase + (unsigned)mrow * HW4 + (unsigned)i * W4;
                            if (vec2) {
                                const float2 a = *reinterpret_cast<const float2 *>(addb + o);
                                apf[ee][i][0] = a.x; apf[ee][i][1] = a.y;
                            } else {
                                apf[ee][i][0] = *reinterpret_cast<const float *>(addb + o);
                                if (col1) apf[ee][i][1] = *reinterpret_cast<const float *>(addb + o + 4u);
                            }
                        }
                    }
                }
            }
        }
        // send: PH 1 gives (M2, M2 + M3) of rows e = 8*rnd .. +3; PH 0 gives (M0 + M1, M1) of e = 8*rnd+4 .. +7
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
            const int e = 8 * rnd + (PH ? ee : 4 + ee);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lo = acc[j][e], hi = acc[4 + j][e];
                mine[(ee * 8 + j) * 64] = PH ? lo : (lo + hi);
                mine[(ee * 8 + 4 + j) * 64] = PH ? (lo + hi) : hi;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
            const int e = 8 * rnd + (PH ? 4 + ee : ee);
            const int mrow = (e & 3) + 8 * (e >> 2);
            const int m = mw0 + mrow + 4 * half;
            float tmp[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float g0 = theirs[(ee * 8 + j) * 64], g1 = theirs[(ee * 8 + 4 + j) * 64];
                const float lo = acc[j][e], hi = acc[4 + j][e];
                if (PH) {            // have M2 = lo, M3 = hi; got M0 + M1, M1
                    tmp[0][j] = g0 + lo;
                    tmp[1][j] = g1 - (lo + hi);
                } else {             // have M0 = lo, M1 = hi; got M2, M2 + M3
                    tmp[0][j] = (lo + hi) + g0;
                    tmp[1][j] = hi - g1;
                }
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
                    const unsigned o = obase + (unsigned)mrow * HW4 + (unsigned)i * W4;
                    if (vec2) {
                        if (p.out) *reinterpret_cast<float2 *>(outb + o) = make_float2(y[i][0], y[i][1]);
                        if (p.add) {
                            float2 a;
                            if constexpr (APF) a = make_float2(apf[ee][i][0], apf[ee][i][1]);
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
                                __fadd_rn(y[i][0], APF ? apf[ee][i][0] : *reinterpret_cast<const float *>(addb + o));
                            if (col1)
                                *reinterpret_cast<float *>(oaddb + o + 4u) =
                                    __fadd_rn(y[i][1], APF ? apf[ee][i][1] : *reinterpret_cast<const float *>(addb + o + 4u));
                        }
                    }
                }
            }
        }
        if (rnd == 0) __syncthreads();
    }
}

// HV = plane half this wave stages AND accumulates (wave >> 2): instantiated twice and entered through one
// wave-uniform branch, so that every register array index below is a compile-time constant
template <int HV, bool APF>
__device__ __forceinline__ void wino64_body(const ConvWino64Dev &p, float *smem, int tid, int wave, int lane, int tile_m, int tile_t)
{
    float *As = smem;
    float *Bs = smem + 2 * ZPA;
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const int m0 = tile_m * ZBM;
    const int t0 = tile_t * ZBT;
    const int HW = p.H * p.W;
    const int CHW = p.C * HW;

    // ---- staging role: tile t_s, channel ch of every panel, patch rows [HV, HV + 3) ----
    const int t_s = lane;
    const int ch = wave & 3;
    const int half_s = ch & 1;
    const int kk_s = ch >> 1;
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
    int pvr[3];
    const bool left_s = (tj_s == 0);
    const bool inv2_s = (2 * tj_s + 1 >= p.W);
    const bool inv3_s = (2 * tj_s + 2 >= p.W);
    {
        const unsigned base = ((unsigned)(b_s - b_first) * (unsigned)CHW + (unsigned)(2 * ti_s) * (unsigned)p.W +
                               (unsigned)(2 * tj_s) + (left_s ? 1u : 0u)) * 4u;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const int prow = HV + rr;                                        // patch row 0..3
            const int iy = 2 * ti_s - 1 + prow;
            const bool ok = t_ok && iy >= 0 && iy < p.H;
            pvr[rr] = ok ? (int)(base + (unsigned)(prow * p.W) * 4u) : -1;     // halo rows: range check -> 0.0
        }
    }
    const float *u_tile = p.u + (size_t)tile_m * p.nkb * ZPA;

    float xr[12];
    float ur[2][4];

#define Z_LOAD_X(KB)                                                                               \
    {                                                                                              \
        const int s0 = ((KB) * ZBK + ch) * HW * 4;                                                 \
        _Pragma("unroll") for (int rr = 0; rr < 3; ++rr) {                                         \
            const u32x4v q0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, pvr[rr], s0, 0);         \
            xr[rr * 4 + 0] = __uint_as_float(q0[0]); xr[rr * 4 + 1] = __uint_as_float(q0[1]);      \
            xr[rr * 4 + 2] = __uint_as_float(q0[2]); xr[rr * 4 + 3] = __uint_as_float(q0[3]);      \
        }                                                                                          \
    }
#define Z_LOAD_U(KB)                                                                               \
    {                                                                                              \
        const float4 *src = reinterpret_cast<const float4 *>(u_tile + (size_t)(KB) * ZPA);         \
        _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                            \
            const float4 t4 = src[tid + e * 512];                                                  \
            ur[e][0] = t4.x; ur[e][1] = t4.y; ur[e][2] = t4.z; ur[e][3] = t4.w;                    \
        }                                                                                          \
    }
    // planes 8 HV + 2 pr, + 1 of (tile t_s, channel ch): B[xi/2][half][kk][t][xi&1]
#define Z_STORE_X(BUF)                                                                             \
    {                                                                                              \
        float va[8];                                                                               \
        fix_rows64(xr, left_s, inv2_s, inv3_s);                                                    \
        input_transform64<HV>(xr, va);                                                             \
        float *dst = Bs + (BUF) * ZPB + (4 * HV) * 512 + half_s * 256 + kk_s * 128 + t_s * 2;      \
        _Pragma("unroll") for (int pr = 0; pr < 4; ++pr)                                           \
            *reinterpret_cast<float2 *>(dst + pr * 512) = make_float2(va[2 * pr], va[2 * pr + 1]); \
    }
#define Z_STORE_U(BUF)                                                                             \
    {                                                                                              \
        float4 *dst = reinterpret_cast<float4 *>(As + (BUF) * ZPA);                                \
        _Pragma("unroll") for (int e = 0; e < 2; ++e)                                              \
            dst[tid + e * 512] = make_float4(ur[e][0], ur[e][1], ur[e][2], ur[e][3]);              \
    }

    f32x16 acc[8];
#pragma unroll
    for (int pp = 0; pp < 8; ++pp)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[pp][e] = 0.f;

    const int wt = wave & 1;
    const int wf = (wave >> 1) & 1;

    float2 fa[2][8];
    float fb[2][8][2];
#define Z_READ_FRAGS(SET, BUF)                                                                     \
    {                                                                                              \
        const float *Ab = As + (BUF) * ZPA + (8 * HV) * 256 + half * 128 + (wf * 32 + l31) * 2;    \
        const float *Bb = Bs + (BUF) * ZPB + (4 * HV) * 512 + half * 256 + (wt * 32 + l31) * 2;    \
        _Pragma("unroll") for (int pp = 0; pp < 8; ++pp)                                           \
            fa[SET][pp] = *reinterpret_cast<const float2 *>(Ab + pp * 256);                        \
        _Pragma("unroll") for (int pr = 0; pr < 4; ++pr)                                           \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                     \
                const float2 v2 = *reinterpret_cast<const float2 *>(Bb + pr * 512 + kk * 128);     \
                fb[SET][2 * pr][kk] = v2.x;                                                        \
                fb[SET][2 * pr + 1][kk] = v2.y;                                                    \
            }                                                                                      \
    }

    // ---- prologue: panel 0 -> LDS stage 0 -> fragment set 0; panel 1 -> registers ----
    // (nkb = C/4 is even and >= 4: the launcher requires C % 8 == 0, C >= 16)
    Z_LOAD_X(0)
    Z_LOAD_U(0)
    Z_STORE_X(0)
    Z_STORE_U(0)
    Z_LOAD_X(1)
    Z_LOAD_U(1)
    __syncthreads();
    Z_READ_FRAGS(0, 0)

    // sched_group_barrier masks: 0x008 MFMA, 0x002 VALU, 0x020 VMEM read, 0x100 DS read, 0x200 DS write
#define Z_PIPE(MASK, N) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(MASK, N, 0);
    // One panel (the schedule of conv_f32_wino32.hip).  Entered with the fragments of panel kb in set SET.
    //   first half  (8 MFMAs, k 0/1 of the panel): registers (panel kb+1) -> transform -> LDS[buf^1]
    //   barrier
    //   second half (8 MFMAs, k 2/3): fragments of panel kb+1 -> set SET^1, panel kb+2 -> registers
#define Z_ITER(KB, SET, DO_STORE, DO_LOAD)                                                         \
    {                                                                                              \
        const int buf = (KB) & 1;                                                                  \
        if (DO_STORE) Z_STORE_U(buf ^ 1)                                                           \
        if (DO_STORE) Z_STORE_X(buf ^ 1)                                                           \
        _Pragma("unroll") for (int pp = 0; pp < 8; ++pp)                                           \
            acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][pp].x, fb[SET][pp][0], acc[pp], 0, 0, 0); \
        if (DO_STORE) {                                                                            \
            _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                     \
                Z_PIPE(0x002, 5) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                \
            }                                                                                      \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        __syncthreads();                                                                           \
        if (DO_STORE) Z_READ_FRAGS((SET) ^ 1, buf ^ 1)                                             \
        if (DO_LOAD) Z_LOAD_X((KB) + 2)                                                            \
        if (DO_LOAD) Z_LOAD_U((KB) + 2)                                                            \
        _Pragma("unroll") for (int pp = 0; pp < 8; ++pp)                                           \
            acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][pp].y, fb[SET][pp][1], acc[pp], 0, 0, 0); \
        if (DO_STORE) {                                                                            \
            _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                     \
                Z_PIPE(0x100, 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                \
            }                                                                                      \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }

    int kb = 0;
    for (; kb + 4 <= p.nkb; kb += 2) {
        Z_ITER(kb, 0, true, true)
        Z_ITER(kb + 1, 1, true, true)
    }
    Z_ITER(kb, 0, true, false)
    Z_ITER(kb + 1, 1, false, false)
    __syncthreads();            // the epilogue reuses the stages: every wave must be done reading them
#undef Z_ITER
#undef Z_PIPE
#undef Z_READ_FRAGS
#undef Z_STORE_U
#undef Z_STORE_X
#undef Z_LOAD_U
#undef Z_LOAD_X

    wino64_epilogue<HV, APF>(p, acc, smem, wave, lane, m0 + 32 * wf, t0 + 32 * wt);
}

template <bool APF>
__global__ __launch_bounds__(512) void conv_f32_wino64_kernel(ConvWino64Dev p)
{
    __shared__ __attribute__((aligned(16))) float smem[2 * ZPA + 2 * ZPB];      // 64 KB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

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

    if (wave >> 2) wino64_body<1, APF>(p, smem, tid, wave, lane, tile_m, tile_t);
    else wino64_body<0, APF>(p, smem, tid, wave, lane, tile_m, tile_t);
}

size_t wino64_packed_floats(int C, int M)
{
    const int tiles_m = (M + ZBM - 1) / ZBM;
    return (size_t)tiles_m * (C / ZBK) * ZPA;
}

// U = G g G^T (double, rounded once), packed [tile_m][panel][xi][half][m 64][kk 2]; k = panel*4 + 2*kk + half
void wino64_pack_weights(const float *w, int C, int M, float *dst)
{
    static const double G[4][3] = {{1., 0., 0.}, {.5, .5, .5}, {.5, -.5, .5}, {0., 0., 1.}};
    const int tiles_m = (M + ZBM - 1) / ZBM;
    const int nkb = C / ZBK;
    for (int tm = 0; tm < tiles_m; ++tm)
        for (int kb = 0; kb < nkb; ++kb) {
            float *panel = dst + ((size_t)tm * nkb + kb) * ZPA;
            for (int ml = 0; ml < ZBM; ++ml) {
                const int m = tm * ZBM + ml;
                for (int kl = 0; kl < ZBK; ++kl) {
                    const int c = kb * ZBK + kl;
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
                        panel[xi * 256 + hf * 128 + ml * 2 + kk] = (float)u[xi >> 2][xi & 3];
                }
            }
        }
}

int launch_conv_f32_wino64(const ConvF32Args &a, const float *u_packed, int variant, void *stream, char *name, size_t name_len)
{
    if (!wino_applicable(a.C, a.M, a.size, a.stride, a.pad) || a.OH != a.H || a.OW != a.W || a.H < 4 || a.W < 4)
        return (int)hipErrorInvalidValue;
    ConvWino64Dev d;
    d.in = a.in; d.u = u_packed; d.bias = a.bias; d.add = a.add; d.out_add = a.out_add; d.out = a.out;
    d.B = a.B; d.C = a.C; d.H = a.H; d.W = a.W; d.M = a.M;
    d.th = (a.H + 1) / 2; d.tw = (a.W + 1) / 2; d.tpi = d.th * d.tw;
    const long long T = (long long)a.B * d.tpi;
    if (T > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    if (!wino32_fits(a.B, a.M, a.H, a.W)) return (int)hipErrorInvalidValue;
    d.T = (int)T;
    d.tiles_m = (a.M + ZBM - 1) / ZBM;
    d.tiles_t = (int)((T + ZBT - 1) / ZBT);
    d.nkb = a.C / ZBK;
    if (d.nkb < 4 || (d.nkb & 1)) return (int)hipErrorInvalidValue;
    d.act = a.act;
    const long long blocks = (long long)d.tiles_m * d.tiles_t;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks), block(512);
    hipStream_t s = (hipStream_t)stream;
    if (variant & 2) hipLaunchKernelGGL(conv_f32_wino64_kernel<true>, grid, block, 0, s, d);
    else hipLaunchKernelGGL(conv_f32_wino64_kernel<false>, grid, block, 0, s, d);
    if (name) snprintf(name, name_len, "conv_f32_wino<64x64t,f2x2%s>", (variant & 2) ? ",apf" : "");
    return (int)hipGetLastError();
}

}  // namespace yl
