// conv_f32_wino32.hip -- K1w: 3x3 / stride 1 / pad 1 FP32 convolution as Winograd F(2x2,3x3) on
// v_mfma_f32_32x32x2_f32; input transform, 16 plane GEMMs and output transform fused in one kernel.
//
// Same layer as conv_f32_mfma.hip computes (forward_convolutional_layer_cpu FP32 branch,
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
// FP32 error: the transforms only add/subtract and halve (G's halves are folded into U in double),
// measured max |err| = 4e-6 of the layer RMS against 2.5e-6 for the direct kernel (bar 1e-4,
// tests/common.py::fp32_close).  im2col AND the V / M tensors never exist in HBM: every thread
// gathers the 4x4 patches of one channel of one tile (one 16-byte buffer load per patch row; halo
// rows -> voffset -1 -> 0.0 from the range check, halo columns by lane selects), transforms them in
// registers and writes them to the LDS stage.
//
// Tiling: a first version gave every wave all 16 planes of a 32x32 block (256 accumulator
// registers): one wave per SIMD and one workgroup per CU, so nothing overlapped a workgroup's
// prologue, its epilogue or a memory stall (PMC: matrix pipe busy 47 % of the time,
// profiles/r1_rocprofv3_pmc_wino64_layer_512x256x38.txt).  Here a workgroup is 32 filters x 64
// tiles, a wave owns HALF of the planes (8 x 16 = 128 AccVGPRs <= 256 registers in total), panels
// are 4 channels and the LDS stage is 24 KB: two workgroups are resident per CU (2 waves per SIMD)
// and run out of phase.  Price: the U slice is re-read per 32 instead of 64 filters.
//
//   waves: wt = wave & 1 -> tiles [32*wt, +32);  ph = wave >> 1 -> planes [8*ph, +8)
//   LDS:   A[xi][half][m 32][kk 2]   one ds_read_b64 per plane and panel
//          B[xi/2][half][kk 2][t 64][xi&1]   planes in pairs: the transform stores 8 float2 per patch
//                                    (ds_write_b32 moves only 64 B/clk) and a wave reads two of its
//                                    planes per ds_read_b64
//   epilogue: row i of A^T M A needs planes of both halves, so the two waves of a tile half swap
//          partial row sums through LDS (the dead panel buffers), each finishing 8 of the 16
//          accumulator rows: tmp0 = (M0 + M1) + M2, tmp1 = M1 - (M2 + M3), then the column pass.
#include <hip/hip_runtime.h>
#include <cstdio>

#include "kernels.h"
#include "../../include/yolo2_hip.h"

// (The -DX_DBG timing-experiment hooks of rounds 3 / 4 -- no loads, no transform, no MFMAs, one workgroup per CU ... -- left the
// file in round 5 with the kernel's demotion to the pooled layers; their measurements are in DESIGN.md, K1w, and profiles/r3_wino_ablation.txt,
// profiles/r4_ab_wino_64x32_ablation_clock.txt; the tree that built them is 446af69.)

namespace yl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// the return type of __builtin_amdgcn_raw_buffer_load_b128.  NB: __builtin_bit_cast(float, q[i]) on a
// vector element is miscompiled by this clang (every i reads element 0): use __uint_as_float
typedef unsigned int u32x4v __attribute__((__vector_size__(16)));

namespace {

constexpr int XBM = 32;
constexpr int XBT = 64;
constexpr int XBK = 4;
constexpr int XPA = 16 * XBK * XBM;      // floats per A panel = 2048 (8 KB)
constexpr int XPB = 16 * XBK * XBT;      // floats per B panel = 4096 (16 KB)

struct ConvWino32Dev {
    const float *in;
    const float *u;        // packed U: [tile_m][panel][xi][half][m 32][kk 2]
    const float *bias;
    const float *add;
    float *out_add;
    float *out;
    float *pool_out;       // fused [maxpool] 2x2 / stride 2 behind the layer (H, W even): [B][M][H/2][W/2], or nullptr
    int B, C, H, W, M;
    int th, tw, tpi, T;
    int tiles_m, tiles_t, nkb;
    int act;
    unsigned *tile_ctr;    // persistent form (VAR bit 2): 8 per-XCD work counters, all zero between launches
    int total;             // workgroup tiles of the launch = tiles_m * tiles_t
};

__device__ __forceinline__ void fix_rows32(float (&d)[16], bool left, bool inv2, bool inv3)
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

__device__ __forceinline__ void input_transform32(const float (&d)[16], float (&v)[16])
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

// The same transform with the column masks folded in (round 4, VAR bit 3): the patch comes as loaded -- for a left-edge
// tile from one column early, so its column 0 holds whatever precedes the row (finite: real data or the zeroed front pad
// of the tensor) -- and the masks m0 / m2 / m3 (1.0 or 0.0 per lane: column 0 of a left-edge tile, columns 2 / 3 beyond
// the right edge) enter the SECOND stage as multipliers: masking column s of d scales w[.][s] by m_s exactly, so
// fma(w0, m0, -w2) is the old (m0*w0) - w2 with one rounding, bit for bit (a difference is possible only in the SIGN of
// an exact zero of V).  24 v_cndmask per patch become 0 (even widths) or 4 v_mul (odd widths, column 2): on gfx950 the
// f32 MFMA shares its datapath with the VALU -- every VALU instruction of ANY wave on the SIMD costs the matrix pipe
// ~4.9 cycles (tools/mfma_f32_bench.hip, profiles/r4_mfma_f32_skeleton_bench.txt) -- so the kernel's efficiency is
// 1024 / (1024 + 4.9 * VALU per panel and wave) and the instruction count of the staging code is what bounds it.
template <bool ODD_W>
__device__ __forceinline__ void input_transform32m(const float (&d)[16], float (&v)[16], float m0, float m2, float m3)
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
        const float w2 = ODD_W ? __fmul_rn(w[i * 4 + 2], m2) : w[i * 4 + 2];      // even widths have no column-2 mask
        v[i * 4 + 0] = __fmaf_rn(w[i * 4 + 0], m0, -w2);
        v[i * 4 + 1] = w[i * 4 + 1] + w2;
        v[i * 4 + 2] = w2 - w[i * 4 + 1];
        v[i * 4 + 3] = __fmaf_rn(-w[i * 4 + 3], m3, w[i * 4 + 1]);
    }
}

}  // namespace

// Epilogue of one wave.  Wave (wt, PH) holds M[i][j] for rows i = 2*PH, 2*PH+1 (planes 8*PH + 4*(i&1) + j) of the
// 32x32 (m, t) block of tile half wt.  Row i of A^T M A needs planes of both halves, so the two waves of a tile
// half swap partial row sums through the dead panel stages (xch[wave][32][64] floats, the loop's last barrier has
// passed), each finishing 8 of the 16 accumulator rows.  Addresses: one 32-bit BYTE offset per lane from the
// tensor base (the launcher keeps Winograd to tensors below 4 GB), row / filter strides added as constants.
template <int PH, bool APF>
__device__ __forceinline__ void wino32_epilogue(const ConvWino32Dev &p, const f32x16 (&acc)[8], float *smem, int wave,
                                                int lane, int m0, int t0)
{
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const int wt = wave & 1;
    float *xch = smem;
    const int tg_e = t0 + wt * 32 + l31;
    const bool t_ok_e = tg_e < p.T;
    const int b_e = t_ok_e ? tg_e / p.tpi : 0;
    const int r_e = tg_e - b_e * p.tpi;
    const int ti_e = r_e / p.tw;
    const int tj_e = r_e - ti_e * p.tw;
    const int oy = 2 * ti_e, ox = 2 * tj_e;
    const bool row1 = oy + 1 < p.H;
    const bool col1 = ox + 1 < p.W;
    const bool pairs = (p.W & 1) == 0;          // wave-uniform: even width -> col1 holds everywhere, a 2-pixel row is one 8-byte access
    const unsigned HW4 = (unsigned)(p.H * p.W) * 4u;
    const unsigned W4 = (unsigned)p.W * 4u;
    // byte offset of (b_e, m0 + 4*half, oy, ox); every access goes through a buffer descriptor over the whole
    // tensor with the offset forced to 0xFFFFFFFF where the row / column / filter / tile does not exist: the range
    // check drops those lanes, so the epilogue has no divergent branches (the first version spent ~50 mostly
    // exec-mask instructions per filter row on them)
    const unsigned obase = ((((unsigned)b_e * (unsigned)p.M + (unsigned)(m0 + 4 * half)) * (unsigned)p.H + (unsigned)oy) *
                            (unsigned)p.W + (unsigned)ox) * 4u;
    const unsigned tbytes = (unsigned)((size_t)p.B * p.M * p.H * p.W * 4);
    const float *dummy = p.bias;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)(p.out ? p.out : dummy), 0, p.out ? (int)tbytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc((void *)(p.add ? p.add : dummy), 0, p.add ? (int)tbytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_oadd = __builtin_amdgcn_make_buffer_rsrc((void *)(p.add ? p.out_add : dummy), 0, p.add ? (int)tbytes : 0, 0x00020000);
    const bool has_out = p.out != nullptr, has_add = p.add != nullptr;
    // fused 2x2 / stride-2 [maxpool] (forward_maxpool_layer_cpu, src/additionally.c:1448-1482): an F(2x2) output tile IS
    // one pooling window (H, W even: window origin 0, no out-of-range taps), so the lane that finishes a tile's four
    // outputs also owns its pooled value -- one dword per lane, 32 consecutive tiles of a row = 128 contiguous bytes
    const bool has_pool = p.pool_out != nullptr;
    const unsigned PHW4 = (unsigned)((p.H >> 1) * (p.W >> 1)) * 4u;
    const unsigned pbase = ((((unsigned)b_e * (unsigned)p.M + (unsigned)(m0 + 4 * half)) * (unsigned)(p.H >> 1) + (unsigned)ti_e) *
                            (unsigned)(p.W >> 1) + (unsigned)tj_e) * 4u;
    const __amdgpu_buffer_rsrc_t rs_pool = __builtin_amdgcn_make_buffer_rsrc((void *)(has_pool ? p.pool_out : dummy), 0,
                                                                              has_pool ? (int)(tbytes >> 2) : 0, 0x00020000);
    float *mine = xch + wave * 2048 + lane;
    const float *theirs = xch + (wave ^ 2) * 2048 + lane;
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
        unsigned off[4][2], off1[4][2];          // row offsets of this round (first / second pixel), 0xFFFFFFFF = absent
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
            const int e = 8 * rnd + (PH ? 4 + ee : ee);
            const int mrow = (e & 3) + 8 * (e >> 2);                  // + 4*half is in obase
            const bool ok = t_ok_e && (m0 + mrow + 4 * half < p.M);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool rok = ok && (i == 0 || row1);
                const unsigned o = obase + (unsigned)mrow * HW4 + (unsigned)i * W4;
                off[ee][i] = rok ? o : 0xFFFFFFFFu;
                off1[ee][i] = (rok && col1) ? o + 4u : 0xFFFFFFFFu;
            }
        }
        float apf[4][2][2];
        if (APF && has_add) {
#pragma unroll
            for (int ee = 0; ee < 4; ++ee)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (pairs) {
                        const v2u a = __builtin_bit_cast(v2u, __builtin_amdgcn_raw_buffer_load_b64(rs_add, (int)off[ee][i], 0, 0));
                        apf[ee][i][0] = __uint_as_float(a[0]); apf[ee][i][1] = __uint_as_float(a[1]);
                    } else {
                        apf[ee][i][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_add, (int)off[ee][i], 0, 0));
                        apf[ee][i][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_add, (int)off1[ee][i], 0, 0));
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
            const int m = m0 + mrow + 4 * half;
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
            const float bv = p.bias[m < p.M ? m : 0];
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
            if (has_pool) {
                // the reference's scan: max = -FLT_MAX; rows then columns; `if (val > max) max = val`
                float mx = -3.402823466e+38f;
                mx = (y[0][0] > mx) ? y[0][0] : mx;
                mx = (y[0][1] > mx) ? y[0][1] : mx;
                mx = (y[1][0] > mx) ? y[1][0] : mx;
                mx = (y[1][1] > mx) ? y[1][1] : mx;
                const bool okp = t_ok_e && (m < p.M);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mx), rs_pool, okp ? (int)(pbase + (unsigned)mrow * PHW4) : -1, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float a0 = 0.f, a1 = 0.f;
                if (has_add) {
                    if (APF) { a0 = apf[ee][i][0]; a1 = apf[ee][i][1]; }
                    else if (pairs) {
                        const v2u a = __builtin_bit_cast(v2u, __builtin_amdgcn_raw_buffer_load_b64(rs_add, (int)off[ee][i], 0, 0));
                        a0 = __uint_as_float(a[0]); a1 = __uint_as_float(a[1]);
                    } else {
                        a0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_add, (int)off[ee][i], 0, 0));
                        a1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_add, (int)off1[ee][i], 0, 0));
                    }
                }
                if (pairs) {
                    const int o2 = (int)off[ee][i];
                    if (has_out) {
                        v2u d; d[0] = __float_as_uint(y[i][0]); d[1] = __float_as_uint(y[i][1]);
                        __builtin_amdgcn_raw_buffer_store_b64(d, rs_out, o2, 0, 0);
                    }
                    if (has_add) {
                        v2u d; d[0] = __float_as_uint(__fadd_rn(y[i][0], a0)); d[1] = __float_as_uint(__fadd_rn(y[i][1], a1));
                        __builtin_amdgcn_raw_buffer_store_b64(d, rs_oadd, o2, 0, 0);
                    }
                } else {
                    if (has_out) {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y[i][0]), rs_out, (int)off[ee][i], 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y[i][1]), rs_out, (int)off1[ee][i], 0, 0);
                    }
                    if (has_add) {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__fadd_rn(y[i][0], a0)), rs_oadd, (int)off[ee][i], 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__fadd_rn(y[i][1], a1)), rs_oadd, (int)off1[ee][i], 0, 0);
                    }
                }
            }
        }
        if (rnd == 0) __syncthreads();
    }
}

// VAR bit 0 (UDMA): the U panels go global -> LDS directly (global_load_lds_dwordx4, the packed panel IS the
//   lane-linear LDS image): no VGPR round trip, no ds_write for the weights.  The stage a panel lands in was last
//   read (fragments of panel kb-2) before the barrier the DMA is issued after; every wave drains its own DMAs
//   (vmcnt(0)) in front of the next barrier, which publishes them.
// VAR bit 1 (APF): the fused [shortcut] operand of an epilogue round is requested BEFORE the round's LDS exchange
//   and barrier instead of after them (its HBM latency hides behind the exchange).
// VAR bit 2 (PERSIST, round 4): persistent workgroups.  The launch has two workgroups per CU; each draws its tiles from
//   the work counter of ITS XCD (block b runs on XCD b % 8 -- an observation used for L2 locality only: the eight
//   contiguous tile ranges are what the non-persistent form's XCD remap gives the same XCD, and a workgroup that lands
//   elsewhere still computes correct tiles).  The next tile is drawn during the last panel of the current one, and its
//   first patch rows / U panel are requested BEFORE the epilogue, so their HBM round trip, the tile decode and the
//   workgroup launch itself hide behind the epilogue's LDS exchange and stores.  The workgroup that draws the last value
//   of a counter (range length + workgroups of the XCD - 1: every other draw has happened) resets it for the next launch.
template <int VAR>
__global__ __launch_bounds__(256, 2) void conv_f32_wino32_kernel(ConvWino32Dev p)
{
    constexpr bool UDMA = (VAR & 1) != 0;
    constexpr bool APF = (VAR & 2) != 0;
    constexpr bool PERSIST = (VAR & 4) != 0;
    constexpr bool LS = (VAR & 8) != 0;          // left-edge patches one column early, column masks folded into the transform
    constexpr bool ODDW = (VAR & 16) != 0;       // (with LS) odd map width: the last tile column also masks patch column 2
    static_assert(!(UDMA && PERSIST), "the persistent form stages U through registers");
    __shared__ __attribute__((aligned(16))) float smem[2 * XPA + 2 * XPB];      // 48 KB
    __shared__ int s_next;
    float *As = smem;
    float *Bs = smem + 2 * XPA;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int half = lane >> 5;

    // the tile range of this workgroup's XCD: logical tiles [x_start, x_start + x_len)
    const int nwg = PERSIST ? p.total : (int)gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7;
    const int x_start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int x_len = q + (xcd < r ? 1 : 0);
    const int x_wgs = ((int)gridDim.x + 7 - xcd) >> 3;          // workgroups of this launch with bid % 8 == xcd
    // draw the next tile of this XCD: index within the range, or -1 when the range is used up (lane 0 of wave 0 only)
#define X_DRAW(DST)                                                                                \
    {                                                                                              \
        const unsigned d_ = atomicAdd(p.tile_ctr + xcd, 1u);                                       \
        if (d_ == (unsigned)(x_len + x_wgs - 1)) atomicExch(p.tile_ctr + xcd, 0u);                 \
        DST = d_ < (unsigned)x_len ? (int)d_ : -1;                                                 \
    }
    int cur = bid >> 3;                                         // index within the XCD's range
    if constexpr (PERSIST) {
        if (tid == 0) { int d; X_DRAW(d) s_next = d; }
        __syncthreads();
        cur = __builtin_amdgcn_readfirstlane(s_next);
        if (cur < 0) return;
    }
#ifndef XGT
#define XGT 8
#endif
    constexpr int GT = XGT;
    const int per_group = GT * p.tiles_m;
    const int HW = p.H * p.W;
    const int CHW = p.C * HW;

    // ---- staging role: tile t_s, channel `wave` of every panel (half = wave & 1, kk = wave >> 1) ----
    const int t_s = tid & 63;
    const int half_s = wave & 1;
    const int kk_s = wave >> 1;
    // per-tile staging state (re-derived for the next tile in front of the epilogue in the persistent form)
    int m0, t0;
    __amdgpu_buffer_rsrc_t rsrc;
    int pvr[4];
    bool left_s, inv2_s, inv3_s;
    float m0f = 1.f, m2f = 1.f, m3f = 1.f;
    const float *u_tile;
    // the packed U through a buffer descriptor: lane offset tid * 16, tile / panel offsets in the scalar soffset -> no VALU
    // address arithmetic in the K loop (64-bit global addresses cost 5 VALU per pair of panels)
    const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.u, 0, (int)((unsigned)p.tiles_m * (unsigned)p.nkb * (unsigned)(XPA * 4)), 0x00020000);
    const int u_lane = tid * 16;
    int u_off = 0;
#define X_SETUP_TILE(IDX)                                                                          \
    {                                                                                              \
        const int logical = x_start + (IDX);                                                       \
        const int tg = logical / per_group;                                                        \
        const int rem_g = logical - tg * per_group;                                                \
        const int t_in_last = p.tiles_t - tg * GT;                                                 \
        const int gsz = t_in_last < GT ? t_in_last : GT;                                           \
        const int tile_m = __builtin_amdgcn_readfirstlane(rem_g / gsz);                            \
        const int tile_t = __builtin_amdgcn_readfirstlane(tg * GT + (rem_g - tile_m * gsz));       \
        m0 = tile_m * XBM;                                                                         \
        t0 = tile_t * XBT;                                                                         \
        const int tg_s = t0 + t_s;                                                                 \
        const bool t_ok = tg_s < p.T;                                                              \
        const int b_s = t_ok ? tg_s / p.tpi : 0;                                                   \
        const int r_s = tg_s - b_s * p.tpi;                                                        \
        const int ti_s = r_s / p.tw;                                                               \
        const int tj_s = r_s - ti_s * p.tw;                                                        \
        const int b_first = __builtin_amdgcn_readfirstlane(t0 / p.tpi);                            \
        const float *tile_base = p.in + (size_t)b_first * CHW - (ptrdiff_t)(p.W + 1);              \
        size_t rec = ((size_t)p.B - b_first) * CHW * sizeof(float) + (size_t)(p.W + 1) * sizeof(float); \
        if (rec > 0xFFFFFFFEull) rec = 0xFFFFFFFEull;                                              \
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)tile_base, 0, (int)(unsigned)rec, 0x00020000); \
        left_s = (tj_s == 0);                                                                      \
        inv2_s = (2 * tj_s + 1 >= p.W);                                                            \
        inv3_s = (2 * tj_s + 2 >= p.W);                                                            \
        m0f = left_s ? 0.f : 1.f; m2f = inv2_s ? 0.f : 1.f; m3f = inv3_s ? 0.f : 1.f;              \
        const unsigned base = ((unsigned)(b_s - b_first) * (unsigned)CHW + (unsigned)(2 * ti_s) * (unsigned)p.W + \
                               (unsigned)(2 * tj_s) + ((!LS && left_s) ? 1u : 0u)) * 4u;           \
        _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) {                                         \
            const int iy = 2 * ti_s - 1 + rr;                                                      \
            const bool ok = t_ok && iy >= 0 && iy < p.H;                                           \
            pvr[rr] = ok ? (int)(base + (unsigned)(rr * p.W) * 4u) : -1;                           \
        }                                                                                          \
        u_tile = p.u + (size_t)tile_m * p.nkb * XPA;                                               \
        u_off = tile_m * p.nkb * (XPA * 4);                                                        \
    }
    X_SETUP_TILE(cur)

    float xr[16];
    float ur[2][4];

#define X_LOAD_X(KB, XR)                                                                            \
    {                                                                                              \
        const int s0 = ((KB) * XBK + wave) * HW * 4;                                               \
        _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) {                                         \
            const u32x4v q0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, pvr[rr], s0, 0);         \
            XR[rr * 4 + 0] = __uint_as_float(q0[0]); XR[rr * 4 + 1] = __uint_as_float(q0[1]);      \
            XR[rr * 4 + 2] = __uint_as_float(q0[2]); XR[rr * 4 + 3] = __uint_as_float(q0[3]);      \
        }                                                                                          \
    }
#define X_LOAD_U(KB, UR)                                                                           \
    {                                                                                              \
        _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                            \
            const u32x4v t4 = __builtin_amdgcn_raw_buffer_load_b128(rs_u, u_lane, u_off + (KB) * (XPA * 4) + e * 4096, 0); \
            UR[e][0] = __uint_as_float(t4[0]); UR[e][1] = __uint_as_float(t4[1]);                  \
            UR[e][2] = __uint_as_float(t4[2]); UR[e][3] = __uint_as_float(t4[3]);                  \
        }                                                                                          \
    }
#define X_DMA_U(KB, BUF)                                                                           \
    {                                                                                              \
        const float4 *src = reinterpret_cast<const float4 *>(u_tile + (size_t)(KB) * XPA);         \
        float4 *dst = reinterpret_cast<float4 *>(As + (BUF) * XPA);                                \
        _Pragma("unroll") for (int e = 0; e < 2; ++e)                                              \
            __builtin_amdgcn_global_load_lds(                                                      \
                (const __attribute__((address_space(1))) void *)(src + tid + e * 256),             \
                (__attribute__((address_space(3))) void *)(dst + tid + e * 256), 16, 0, 0);        \
    }
#define X_STORE_X(BUF, XR)                                                                             \
    {                                                                                              \
        float va[16];                                                                              \
        if constexpr (LS) {                                                                        \
            input_transform32m<ODDW>(XR, va, m0f, m2f, m3f);                                       \
        } else {                                                                                   \
            fix_rows32(XR, left_s, inv2_s, inv3_s);                                                \
            input_transform32(XR, va);                                                             \
        }                                                                                          \
        float *dst = Bs + (BUF) * XPB + half_s * 256 + kk_s * 128 + t_s * 2;                       \
        _Pragma("unroll") for (int pr = 0; pr < 8; ++pr)                                           \
            *reinterpret_cast<float2 *>(dst + pr * 512) = make_float2(va[2 * pr], va[2 * pr + 1]); \
    }
#define X_STORE_U(BUF, UR)                                                                             \
    {                                                                                              \
        float4 *dst = reinterpret_cast<float4 *>(As + (BUF) * XPA);                                \
        _Pragma("unroll") for (int e = 0; e < 2; ++e)                                              \
            dst[tid + e * 256] = make_float4(UR[e][0], UR[e][1], UR[e][2], UR[e][3]);              \
    }

    f32x16 acc[8];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    const int wt = wave & 1;
    const int ph = wave >> 1;

    float2 fa[2][8];
    float fb[2][8][2];
#define X_READ_FRAGS(SET, BUF)                                                                     \
    {                                                                                              \
        const float *Ab = As + (BUF) * XPA + (8 * ph) * 128 + half * 64 + l31 * 2;                 \
        const float *Bb = Bs + (BUF) * XPB + (4 * ph) * 512 + half * 256 + (wt * 32 + l31) * 2;    \
        _Pragma("unroll") for (int pp = 0; pp < 8; ++pp)                                           \
            fa[SET][pp] = *reinterpret_cast<const float2 *>(Ab + pp * 128);                        \
        _Pragma("unroll") for (int pr = 0; pr < 4; ++pr)                                           \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                     \
                const float2 v2 = *reinterpret_cast<const float2 *>(Bb + pr * 512 + kk * 128);     \
                fb[SET][2 * pr][kk] = v2.x;                                                        \
                fb[SET][2 * pr + 1][kk] = v2.y;                                                    \
            }                                                                                      \
    }

    // ---- prologue: panel 0 -> LDS stage 0 -> fragment set 0; panel 1 -> registers ----
    // (nkb = C/4 is even and >= 4: the launcher requires C % 8 == 0, C >= 16)
    // The panel-0 requests of the FIRST tile; in the persistent form those of every later tile are issued in front of
    // the previous tile's epilogue.
    if constexpr (UDMA) {
        X_DMA_U(0, 0)
        X_DMA_U(1, 1)
        X_LOAD_X(0, xr)
    } else {
        X_LOAD_X(0, xr)
        X_LOAD_U(0, ur)
    }
  for (;;) {
    if constexpr (UDMA) {
        X_STORE_X(0, xr)
        X_LOAD_X(1, xr)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        X_STORE_X(0, xr)
        X_STORE_U(0, ur)
        X_LOAD_X(1, xr)
        X_LOAD_U(1, ur)
    }
    __syncthreads();
    X_READ_FRAGS(0, 0)

    // sched_group_barrier masks: 0x008 MFMA, 0x002 VALU, 0x020 VMEM read, 0x100 DS read, 0x200 DS write.
    // The transform (~70 VALU + LDS stores per panel) must sit INSIDE the shadow of this wave's own
    // MFMAs: measured, a VALU block in front of the 8 MFMAs costs 0.24 ms of a 1.2 ms layer even with
    // a second workgroup on the CU.
#define X_PIPE(MASK, N) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(MASK, N, 0);
    // One panel.  Entered with the fragments of panel kb in set SET.
    //   first half  (8 MFMAs, k 0/1 of the panel): registers (panel kb+1) -> transform -> LDS[buf^1]
    //   barrier     every wave has written its share of panel kb+1; the last reads of LDS[buf^1]
    //               (fragments of panel kb-1) completed before the PREVIOUS barrier
    //   second half (8 MFMAs, k 2/3): fragments of panel kb+1 -> set SET^1, panel kb+2 -> registers
    // so no LDS latency is exposed in front of an MFMA block.
#define X_ITER(KB, SET, DO_STORE, DO_LOAD) X_ITER_(KB, SET, DO_STORE, DO_LOAD, false)
    /* FIRST: the accumulators start as the inline constant 0 of the MFMA's C operand instead of 128 v_mov per tile */ \
#define X_ITER_(KB, SET, DO_STORE, DO_LOAD, FIRST)                                                 \
    {                                                                                              \
        const int buf = (KB) & 1;                                                                  \
        if (DO_STORE && !UDMA) X_STORE_U(buf ^ 1, ur)                                              \
        if (DO_STORE) X_STORE_X(buf ^ 1, xr)                                                        \
        _Pragma("unroll") for (int pp = 0; pp < 8; ++pp)                                           \
            acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][pp].x, fb[SET][pp][0], (FIRST) ? zero16 : acc[pp], 0, 0, 0); \
        if (DO_STORE) {                                                                            \
            _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                     \
                X_PIPE(0x002, 9) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                \
            }                                                                                      \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (UDMA && DO_STORE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     \
        __syncthreads();                                                                           \
        if (DO_STORE) X_READ_FRAGS((SET) ^ 1, buf ^ 1)                                             \
        /* the DMA is issued BEFORE the patch loads: vmcnt retires in order, so waiting for patch row r   */ \
        /* (vmcnt(3 - r)) covers the older DMAs and the rows are still consumed one by one                */ \
        if (DO_LOAD && UDMA) X_DMA_U((KB) + 2, buf)                                                \
        if (DO_LOAD) X_LOAD_X((KB) + 2, xr)                                                        \
        if (DO_LOAD && !UDMA) X_LOAD_U((KB) + 2, ur)                                               \
        _Pragma("unroll") for (int pp = 0; pp < 8; ++pp)                                           \
            acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][pp].y, fb[SET][pp][1], acc[pp], 0, 0, 0); \
        if (DO_STORE) {                                                                            \
            _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                     \
                X_PIPE(0x100, 2) __builtin_amdgcn_sched_group_barrier(UDMA ? 0x010 : 0x020, 1, 0); \
            }                                                                                      \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }

    // (nkb >= 4: the first pair of panels always loads panels 2 and 3)
    X_ITER_(0, 0, true, true, true)
    X_ITER(1, 1, true, true)
    int kb = 2;
    for (; kb + 4 <= p.nkb; kb += 2) {
        X_ITER(kb, 0, true, true)
        X_ITER(kb + 1, 1, true, true)
    }
    int drawn = -1;
    if constexpr (PERSIST) {
        if (tid == 0) X_DRAW(drawn)           // the atomic's round trip hides behind the last two panels
    }
    X_ITER(kb, 0, true, false)
    X_ITER(kb + 1, 1, false, false)
    if constexpr (PERSIST) {
        if (tid == 0) s_next = drawn;
    }
    __syncthreads();            // the epilogue reuses the stages: every wave must be done reading them
    const int e_m0 = m0, e_t0 = t0;           // the tile the epilogue finishes
    int nxt = -1;
    if constexpr (PERSIST) {
        nxt = __builtin_amdgcn_readfirstlane(s_next);
        if (nxt >= 0) {                       // next tile: decode, first patch rows and U panel on their way before the epilogue
            X_SETUP_TILE(nxt)
            X_LOAD_X(0, xr)
            X_LOAD_U(0, ur)
        }
    }

    // ---- epilogue ----
    // The plane half `ph` is wave-uniform: branch once so that every accumulator index below is a compile-time
    // constant (with a runtime `ph` the compiler indexed the 128 accumulator registers dynamically: 72
    // s_set_gpr_idx pairs and 330 v_mov per wave).
    if (ph) wino32_epilogue<1, APF>(p, acc, smem, wave, lane, e_m0, e_t0);
    else wino32_epilogue<0, APF>(p, acc, smem, wave, lane, e_m0, e_t0);
    if (!PERSIST || nxt < 0) break;
    __syncthreads();            // the exchange strips are the panel stages of the next tile
  }
#undef X_READ_FRAGS
#undef X_ITER
#undef X_PIPE
#undef X_STORE_U
#undef X_DMA_U
#undef X_STORE_X
#undef X_LOAD_U
#undef X_LOAD_X
#undef X_SETUP_TILE
#undef X_DRAW
}

// the epilogue addresses the output (and the fused [shortcut] tensors of the same shape) with 32-bit byte offsets
bool wino32_fits(int B, int M, int H, int W)
{
    return (long long)B * M * H * W * 4 < 0xFFFFFFF0LL;
}

bool wino_applicable(int C, int M, int size, int stride, int pad)
{
    return size == 3 && stride == 1 && pad == 1 && C % 8 == 0 && C >= 16 && M >= 1;      // + H, W >= 4 (launcher)
}

size_t wino32_packed_floats(int C, int M)
{
    const int tiles_m = (M + XBM - 1) / XBM;
    return (size_t)tiles_m * (C / XBK) * XPA;
}

// U = G g G^T (double, rounded once), packed [tile_m][panel][xi][half][m 32][kk 2]; k = panel*4 + 2*kk + half
void wino32_pack_weights(const float *w, int C, int M, float *dst)
{
    static const double G[4][3] = {{1., 0., 0.}, {.5, .5, .5}, {.5, -.5, .5}, {0., 0., 1.}};
    const int tiles_m = (M + XBM - 1) / XBM;
    const int nkb = C / XBK;
    for (int tm = 0; tm < tiles_m; ++tm)
        for (int kb = 0; kb < nkb; ++kb) {
            float *panel = dst + ((size_t)tm * nkb + kb) * XPA;
            for (int ml = 0; ml < XBM; ++ml) {
                const int m = tm * XBM + ml;
                for (int kl = 0; kl < XBK; ++kl) {
                    const int c = kb * XBK + kl;
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
                        panel[xi * 128 + hf * 64 + ml * 2 + kk] = (float)u[xi >> 2][xi & 3];
                }
            }
        }
}

int launch_conv_f32_wino32(const ConvF32Args &a, const float *u_packed, int variant, void *stream, char *name, size_t name_len)
{
    if (!wino_applicable(a.C, a.M, a.size, a.stride, a.pad) || a.OH != a.H || a.OW != a.W || a.H < 4 || a.W < 4)
        return (int)hipErrorInvalidValue;
    ConvWino32Dev d;
    d.in = a.in; d.u = u_packed; d.bias = a.bias; d.add = a.add; d.out_add = a.out_add; d.out = a.out;
    d.pool_out = a.pool_out;
    if (a.pool_out && ((a.H | a.W) & 1)) return (int)hipErrorInvalidValue;      // a tile must be a whole pooling window
    d.B = a.B; d.C = a.C; d.H = a.H; d.W = a.W; d.M = a.M;
    d.th = (a.H + 1) / 2; d.tw = (a.W + 1) / 2; d.tpi = d.th * d.tw;
    const long long T = (long long)a.B * d.tpi;
    if (T > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    if (!wino32_fits(a.B, a.M, a.H, a.W)) return (int)hipErrorInvalidValue;
    d.T = (int)T;
    d.tiles_m = (a.M + XBM - 1) / XBM;
    d.tiles_t = (int)((T + XBT - 1) / XBT);
    d.nkb = a.C / XBK;
    if (d.nkb < 4 || (d.nkb & 1)) return (int)hipErrorInvalidValue;
    d.act = a.act;
    const long long blocks = (long long)d.tiles_m * d.tiles_t;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    d.total = (int)blocks;
    d.tile_ctr = a.tile_ctr;
    hipStream_t s = (hipStream_t)stream;
    const bool persist = (variant & 64) != 0 && a.tile_ctr != nullptr;
    // VAR bit 3: the input tensor has the library's front pad -> the transform with folded column masks
    const bool ls = a.in_front_pad && (variant & 128) == 0;          // variant bit 7: A/B switch, keep the register-shift form
    if (persist) {
        // two workgroups per CU (what the kernel's registers and LDS allow), or one per tile on small layers
        const int n_cu = device_cu_count();
        const long long slots = 2LL * n_cu;
        const dim3 grid((unsigned)(blocks < slots ? blocks : slots)), block(256);
        if (ls && (a.W & 1)) {
            if (variant & 2) hipLaunchKernelGGL(conv_f32_wino32_kernel<30>, grid, block, 0, s, d);
            else hipLaunchKernelGGL(conv_f32_wino32_kernel<28>, grid, block, 0, s, d);
        } else if (ls) {
            if (variant & 2) hipLaunchKernelGGL(conv_f32_wino32_kernel<14>, grid, block, 0, s, d);
            else hipLaunchKernelGGL(conv_f32_wino32_kernel<12>, grid, block, 0, s, d);
        } else {
            if (variant & 2) hipLaunchKernelGGL(conv_f32_wino32_kernel<6>, grid, block, 0, s, d);
            else hipLaunchKernelGGL(conv_f32_wino32_kernel<4>, grid, block, 0, s, d);
        }
    } else {
        const dim3 grid((unsigned)blocks), block(256);
        switch ((variant & 3) | (ls ? 8 : 0) | ((ls && (a.W & 1)) ? 16 : 0)) {
        case 0: hipLaunchKernelGGL(conv_f32_wino32_kernel<0>, grid, block, 0, s, d); break;
        case 1: hipLaunchKernelGGL(conv_f32_wino32_kernel<1>, grid, block, 0, s, d); break;
        case 2: hipLaunchKernelGGL(conv_f32_wino32_kernel<2>, grid, block, 0, s, d); break;
        case 3: hipLaunchKernelGGL(conv_f32_wino32_kernel<3>, grid, block, 0, s, d); break;
        case 8: hipLaunchKernelGGL(conv_f32_wino32_kernel<8>, grid, block, 0, s, d); break;
        case 9: hipLaunchKernelGGL(conv_f32_wino32_kernel<9>, grid, block, 0, s, d); break;
        case 10: hipLaunchKernelGGL(conv_f32_wino32_kernel<10>, grid, block, 0, s, d); break;
        case 11: hipLaunchKernelGGL(conv_f32_wino32_kernel<11>, grid, block, 0, s, d); break;
        case 24: hipLaunchKernelGGL(conv_f32_wino32_kernel<24>, grid, block, 0, s, d); break;
        case 25: hipLaunchKernelGGL(conv_f32_wino32_kernel<25>, grid, block, 0, s, d); break;
        case 26: hipLaunchKernelGGL(conv_f32_wino32_kernel<26>, grid, block, 0, s, d); break;
        default: hipLaunchKernelGGL(conv_f32_wino32_kernel<27>, grid, block, 0, s, d); break;
        }
    }
    if (name) snprintf(name, name_len, "conv_f32_wino<32x64t,f2x2%s%s%s%s>", (!persist && (variant & 1)) ? ",udma" : "", (variant & 2) ? ",apf" : "",
                       persist ? ",pers" : "", a.pool_out ? (a.out ? ",pool+" : ",pool") : "");
    return (int)hipGetLastError();
}

}  // namespace yl
