// conv_f32_mfma.hip -- K1: FP32 implicit-GEMM convolution for gfx950 (CDNA4), software-pipelined.
//
// Replaces forward_convolutional_layer_cpu's FP32 branch
// (src/yolov2_forward_network.c:30-38,103-111,204-215,243-261: im2col_cpu ->
// gemm_nn -> +bias -> activate_array_cpu_custom) and the reference GPU's
// cudnnConvolutionForward + add_bias_gpu + activate_array_ongpu
// (src/yolov2_forward_network_gpu.cu:113-138).
//
//   GEMM view:  C[M = filters][N = batch*out_h*out_w] = A[M][K] * B[K][N], K = c*size*size.
//   A  (weights)  : pre-packed once, k-major [Kpad][Mpad] so a BK x BM panel is BK rows of BM
//                   contiguous floats (coalesced float4 loads).  K order (c,ky,kx) like im2col_cpu
//                   (src/additionally.c:39-62), or tap-major in 16-channel blocks when C % 16 == 0.
//   B  (im2col)   : never materialised.  Each thread owns ONE output pixel of the tile for the
//                   whole K loop (its (b,oy,ox) decode, byte offset and halo-validity mask are
//                   computed once); the k -> (c,ky,kx) decode is wave-uniform SALU work; gathers go
//                   through a buffer descriptor whose range check turns out-of-image taps into 0.0.
//   math          : v_mfma_f32_32x32x2_f32 -- exact f32, 64 FLOP/clk/SIMD, 157.3 TFLOP/s chip
//                   peak (MI355X_MICROARCH.md).
//   LDS           : As[2][BK][BM], Bs[2][BK][BN] f32, double-buffered, one barrier per BK step;
//                   operand reads are conflict-free ds_read_b32.
//   epilogue      : +bias -> leaky (x>0 ? x : (float)(.1*(double)x), the scalar reference's
//                   arithmetic, src/additionally.h:91) -> optional fused [shortcut] -> row-wise
//                   NCHW stores through wave-private LDS strips (epilogue.h).
//   grid          : 1-D, XCD-aware bijective remap so that the M-tiles sharing one im2col N-tile
//                   run on the same XCD (shared L2).
//
// Schedule.  v_mfma_f32_32x32x2_f32 occupies the pipe for 64 cycles while the wave can keep
// issuing independent instructions, so the staging work of a panel is cut into BK/2 slices and
// each slice is placed between the MFMAs of one k-step (a first version issued the whole gather,
// ~100 SALU/VALU + 10 VMEM, as one burst in front of the 32-MFMA block and idled the matrix pipe
// ~37 % of the cycles; profiles/r1_*):
//
//   iteration kb (registers hold panel kb+1, loaded one iteration ago):
//     for ks in 0 .. BK/2-1:
//        ds_write   slice ks of panel kb+1  -> LDS buffer (kb+1)&1   (not read in this iteration)
//        buffer/global loads slice ks of panel kb+2 -> the SAME registers
//        ds_read    operands of k-step ks+1
//        TM*TN MFMAs of k-step ks
//        sched_barrier                        (pins the slice to its k-step)
//     one barrier
//
// Loads get a whole iteration (>= BK/2 * TM*TN * 64 cycles) of latency budget, waits are
// counted vmcnt(N) (loads return in order), and no instruction burst separates MFMA blocks.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>

#include "kernels.h"
#include "epilogue.h"
#include "../../include/yolo2_hip.h"

namespace yl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvF32Dev {
    const float *in;
    const float *wt;
    const float *bias;
    const float *add;
    float *out_add;
    float *out;
    int8_t *q_out;        // optional int8 side output for a following INT8 convolution (see epilogue.h)
    float q_mult;
    int q_G;
    int B, C, H, W, M, OH, OW;
    int K, Kpad, Mpad;
    int size, stride, pad;
    int act;
    int Ntotal;
    int OHW;
    int tiles_m;
    int yolo_entries;
};

// VEC4 (1x1 / stride 1 / pad 0 layers with OH*OW % 4 == 0 and C % BK == 0): a row of the B panel is BN contiguous
// floats of one channel plane, so a thread fetches 4 consecutive pixels with ONE 16-byte buffer load and stages
// them with one ds_write_b128 (4x fewer VMEM and LDS-write instructions than the per-pixel gather); the k row of
// a lane goes into its voffset, the panel's first channel into the scalar offset.
// YOLO: the [yolo] layer behind a linear 1x1 head convolution folded into the epilogue (forward_yolo_layer_cpu,
// src/yolov2_forward_network.c: logistic_activate on every entry of an anchor except the raw w/h, i.e. rows m with
// m % (classes + 5) not in {2, 3}); same expression as yolo_kernel (layers.hip), so the tensor is bit-identical.
template <int BM, int BN, int WM, int WN, int KS, int BK, int NWAVES, bool TAPMAJOR, bool VEC4 = false, bool YOLO = false>
__global__ __launch_bounds__(NWAVES * 64) void conv_f32_mfma_pipe_kernel(ConvF32Dev p)
{
    constexpr int NT = NWAVES * 64;
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr int KSTEPS = BK / 2;
    static_assert(WM * WN == NWAVES, "wave grid");
    static_assert(TM >= 1 && TN >= 1, "tile too small");
    static_assert(NWAVES * 8 * TN * 32 <= 2 * BK * (BM + BN), "epilogue strips fit in the panel buffers");
    // the quantise-on-store epilogue stages 16 rows per wave
    constexpr bool Q_OK = NWAVES * 16 * TN * 32 <= 2 * BK * (BM + BN);
    static_assert(BN % 64 == 0 && BN <= NT, "a wave must stay inside one k row of the B panel");
    constexpr int A_F4 = BK * BM / 4;
    constexpr int APT = (A_F4 + NT - 1) / NT;               // float4 per thread per panel
    constexpr bool A_FULL = (A_F4 % NT) == 0;
    constexpr int BPT = VEC4 ? 1 : BK * BN / NT;            // gathered floats per thread per panel
    constexpr int K_STEP = NT / BN;
    static_assert(!VEC4 || (KS == 1 && !TAPMAJOR), "float4 rows: 1x1 layers only");
    constexpr int RPP4 = NT / (BN / 4);                     // VEC4: k rows one pass of the workgroup covers
    constexpr int BPT4 = VEC4 ? BK / RPP4 : 1;              // VEC4: float4 per thread per panel
    static_assert(!VEC4 || (BK % RPP4 == 0 && BPT4 >= 1), "VEC4 panel split");

    __shared__ __attribute__((aligned(16))) float smem[2 * BK * BM + 2 * BK * BN];
    float *As = smem;
    float *Bs = smem + 2 * BK * BM;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int half = lane >> 5;

    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tile_m = logical % p.tiles_m;
    const int tile_n = logical / p.tiles_m;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    const int n_local = tid % BN;
    const int krow0 = __builtin_amdgcn_readfirstlane(tid / BN);
    const int HW = p.H * p.W;
    const int CHW = p.C * HW;
    const int n_g = n0 + n_local;
    const bool n_ok = n_g < p.Ntotal;
    const int bimg = n_g / p.OHW;
    const int pix = n_g - bimg * p.OHW;
    const int oy = pix / p.OW;
    const int ox = pix - oy * p.OW;
    const int iy0 = oy * p.stride - p.pad;
    const int ix0 = ox * p.stride - p.pad;

    const int b_first = n0 / p.OHW;
    const float *tile_base = p.in + (size_t)b_first * CHW - (ptrdiff_t)p.pad * (p.W + 1);
    size_t rec = ((size_t)p.B - b_first) * CHW * sizeof(float) + (size_t)p.pad * (p.W + 1) * sizeof(float);
    if (rec > 0xFFFFFFFEull) rec = 0xFFFFFFFEull;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)tile_base, 0, (int)(unsigned)rec, 0x00020000);
    const int voff = (int)(((unsigned)(bimg - b_first) * (unsigned)CHW +
                            (unsigned)(oy * p.stride) * (unsigned)p.W + (unsigned)(ox * p.stride)) * 4u);

    unsigned ntapmask = 0xFFFFFFFFu;
    if (n_ok) {
        if (KS == 1) {
            ntapmask = 0u;
        } else if (KS == 0 && TAPMAJOR) {
            unsigned m = 0;
            for (int ky = 0; ky < p.size; ++ky)
                for (int kx = 0; kx < p.size; ++kx) {
                    const int iy = iy0 + ky, ix = ix0 + kx;
                    if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) m |= 1u << (ky * p.size + kx);
                }
            ntapmask = ~m;
        } else if (KS == 3) {
            unsigned m = 0;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int iy = iy0 + ky, ix = ix0 + kx;
                    if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) m |= 1u << (ky * 3 + kx);
                }
            ntapmask = ~m;
        }
    }

    float a_reg[APT][4];
    float b_reg[BPT];
    float b_reg4[BPT4][4];
    // VEC4 staging role: float4 column n4 of k row r4 (+ e * RPP4)
    const int n4 = tid % (BN / 4);
    const int r4 = tid / (BN / 4);
    int voff4 = -1;
    if (VEC4) {
        const int n_g4 = n0 + 4 * n4;
        if (n_g4 < p.Ntotal) {
            const int b4 = n_g4 / p.OHW;
            const int pix4 = n_g4 - b4 * p.OHW;
            voff4 = (int)(((unsigned)(b4 - b_first) * (unsigned)CHW + (unsigned)pix4 + (unsigned)r4 * (unsigned)HW) * 4u);
        }
    }

    // ---- slice helpers (E = element index inside the thread's share of a panel) ----
#define YL_LOAD_A(KB, E)                                                                           \
    {                                                                                              \
        const int idx = tid + (E) * NT;                                                            \
        if (A_FULL || idx < A_F4) {                                                                \
            const int kr = idx / (BM / 4);                                                         \
            const int c4 = idx - kr * (BM / 4);                                                    \
            const float4 t4 = *reinterpret_cast<const float4 *>(                                   \
                p.wt + (size_t)((KB) * BK + kr) * p.Mpad + m0 + c4 * 4);                           \
            a_reg[E][0] = t4.x; a_reg[E][1] = t4.y; a_reg[E][2] = t4.z; a_reg[E][3] = t4.w;        \
        }                                                                                          \
    }
#define YL_LOAD_B(KB, E)                                                                           \
    {                                                                                              \
        const int k = (KB) * BK + krow0 + (E) * K_STEP;            /* wave-uniform */              \
        /* tap-major needs C % BK == 0, so K == Kpad and no K tail exists */                       \
        const int kinv = TAPMAJOR ? 0 : ((k >= p.K) ? -1 : 0);                                     \
        int soff, tinv;                                                                            \
        if (TAPMAJOR) {                                                                            \
            /* K order (tap, c): the whole panel shares one tap; decode hoisted to pn_* */         \
            soff = pn_soff + (krow0 + (E) * K_STEP) * HW * 4;                                      \
            tinv = pn_tinv;                                                                        \
        } else if (KS == 1) {                                                                      \
            soff = k * HW * 4;                                                                     \
            tinv = (int)ntapmask;                                                                  \
        } else if (KS == 3) {                                                                      \
            const int c = k / 9;                                                                   \
            const int rr = k - c * 9;                                                              \
            const int ky = rr / 3;                                                                 \
            const int kx = rr - ky * 3;                                                            \
            soff = (c * HW + ky * p.W + kx) * 4;                                                   \
            tinv = __builtin_amdgcn_sbfe((int)ntapmask, rr, 1);                                    \
        } else {                                                                                   \
            const int ss = p.size * p.size;                                                        \
            const int c = k / ss;                                                                  \
            const int rr = k - c * ss;                                                             \
            const int ky = rr / p.size;                                                            \
            const int kx = rr - ky * p.size;                                                       \
            const int iy = iy0 + ky, ix = ix0 + kx;                                                \
            soff = (c * HW + ky * p.W + kx) * 4;                                                   \
            tinv = (n_ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? 0 : -1;                  \
        }                                                                                          \
        if (kinv) soff = 0;                                                                        \
        b_reg[E] = __builtin_bit_cast(float,                                                       \
            __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff | tinv | kinv, soff, 0));              \
    }
#define YL_LOAD_B4(KB, E)                                                                          \
    {                                                                                              \
        const int soff = ((KB) * BK + (E) * RPP4) * HW * 4;                                        \
        const auto q4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff4, soff, 0);               \
        b_reg4[E][0] = __uint_as_float(q4[0]); b_reg4[E][1] = __uint_as_float(q4[1]);              \
        b_reg4[E][2] = __uint_as_float(q4[2]); b_reg4[E][3] = __uint_as_float(q4[3]);              \
    }
#define YL_STORE_B4(BUF, E)                                                                        \
    {                                                                                              \
        *reinterpret_cast<float4 *>(Bs + (BUF) * BK * BN + (r4 + (E) * RPP4) * BN + n4 * 4) =      \
            make_float4(b_reg4[E][0], b_reg4[E][1], b_reg4[E][2], b_reg4[E][3]);                   \
    }
#define YL_STORE_A(BUF, E)                                                                         \
    {                                                                                              \
        const int idx = tid + (E) * NT;                                                            \
        if (A_FULL || idx < A_F4)                                                                  \
            *reinterpret_cast<float4 *>(As + (BUF) * BK * BM + idx * 4) =                          \
                make_float4(a_reg[E][0], a_reg[E][1], a_reg[E][2], a_reg[E][3]);                   \
    }
#define YL_STORE_B(BUF, E)                                                                         \
    { Bs[(BUF) * BK * BN + (krow0 + (E) * K_STEP) * BN + n_local] = b_reg[E]; }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int wm = wave / WN;
    const int wn = wave - wm * WN;
    const int wm0 = wm * TM * 32;
    const int wn0 = wn * TN * 32;
    const int nkb = p.Kpad / BK;

    // tap-major panel state: tap index / first channel of the NEXT panel to be loaded
    int pn_tap = 0, pn_c0 = 0, pn_soff = 0, pn_tinv = 0;
#define YL_PANEL_SETUP()                                                                           \
    if (TAPMAJOR) {                                                                                \
        const int ky = (KS == 3) ? ((pn_tap * 11) >> 5) : (pn_tap / p.size);                       \
        const int kx = pn_tap - ky * ((KS == 3) ? 3 : p.size);                                     \
        pn_soff = (pn_c0 * HW + ky * p.W + kx) * 4;                                                \
        pn_tinv = __builtin_amdgcn_sbfe((int)ntapmask, pn_tap, 1);                                 \
    }
#define YL_PANEL_ADVANCE()                                                                         \
    if (TAPMAJOR) {                                                                                \
        /* K order = (16-channel block, tap, channel in block): the taps of one channel block   */ \
        /* are consecutive panels, so their (shifted) input lines are re-read while still in    */ \
        /* L1/L2 instead of after a sweep over all C channels                                    */ \
        ++pn_tap;                                                                                  \
        if (pn_tap >= p.size * p.size) { pn_tap = 0; pn_c0 += BK; }                                \
    }

    // ---- prologue: panel 0 -> LDS buffer 0, panel 1 -> registers ----
    YL_PANEL_SETUP()
#pragma unroll
    for (int e = 0; e < APT; ++e) YL_LOAD_A(0, e)
    if constexpr (VEC4) {
#pragma unroll
        for (int e = 0; e < BPT4; ++e) YL_LOAD_B4(0, e)
    } else {
#pragma unroll
        for (int e = 0; e < BPT; ++e) YL_LOAD_B(0, e)
    }
#pragma unroll
    for (int e = 0; e < APT; ++e) YL_STORE_A(0, e)
    if constexpr (VEC4) {
#pragma unroll
        for (int e = 0; e < BPT4; ++e) YL_STORE_B4(0, e)
    } else {
#pragma unroll
        for (int e = 0; e < BPT; ++e) YL_STORE_B(0, e)
    }
    YL_PANEL_ADVANCE()
    if (nkb > 1) {
        YL_PANEL_SETUP()
#pragma unroll
        for (int e = 0; e < APT; ++e) YL_LOAD_A(1, e)
        if constexpr (VEC4) {
#pragma unroll
            for (int e = 0; e < BPT4; ++e) YL_LOAD_B4(1, e)
        } else {
#pragma unroll
            for (int e = 0; e < BPT; ++e) YL_LOAD_B(1, e)
        }
        YL_PANEL_ADVANCE()
    }
    __syncthreads();

    // one k-block; DO_STORE: registers (panel kb+1) -> LDS[buf^1]; DO_LOAD: panel kb+2 -> registers
    // LDS addressing.  Left alone, hipcc keeps one register per fragment and adds the (runtime) stage offset to each of them
    // in every iteration: ~0.9 v_add_u32 per MFMA, and on gfx950 a VALU instruction costs the f32 matrix pipe ~4.9 cycles
    // (tools/mfma_f32_bench.hip) -- 6 % of a 64-cycle v_mfma_f32_32x32x2_f32 -- plus 16 VGPRs.  Two remedies:
    //  * the steady-state loop is unrolled by two, BUF is then a compile-time constant and every address is a loop-invariant
    //    register + an immediate offset (3-7 VALU per 64 MFMAs).  Costs up to 35 VGPRs of overlap between the two bodies: the
    //    8-wave tiles (two workgroups per CU = 128 VGPRs) and the 32-deep panels would lose a wave per SIMD, so only the 4-wave
    //    tiles with 16-deep panels are unrolled;
    //  * the others, and the tail iterations of all, keep the runtime stage but pass the two fragment offsets through an empty
    //    asm: opaque to the compiler, they stay ONE register each and the reads become base + immediate (15 VALU per 32 MFMAs).
#define YL_ITER(KB, DO_STORE, DO_LOAD) YL_ITER_(KB, (KB) & 1, true, DO_STORE, DO_LOAD)
#define YL_ITER_(KB, BUF, OPAQUE, DO_STORE, DO_LOAD)                                               \
    {                                                                                              \
        const int buf = (BUF);                                                                     \
        if (DO_LOAD) { YL_PANEL_SETUP() }                                                          \
        int a_off = buf * BK * BM + wm0 + l31 + half * BM;                                         \
        int b_off = buf * BK * BN + wn0 + l31 + half * BN;                                         \
        if (OPAQUE) asm volatile("" : "+v"(a_off), "+v"(b_off));                                   \
        const float *Ab = As + a_off;                                                              \
        const float *Bb = Bs + b_off;                                                              \
        float av[2][TM], bv[2][TN];                                                                \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) av[0][i] = Ab[i * 32];                      \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) bv[0][j] = Bb[j * 32];                      \
        _Pragma("unroll") for (int ks = 0; ks < KSTEPS; ++ks) {                                    \
            const int cur = ks & 1, nxt = cur ^ 1;                                                 \
            _Pragma("unroll") for (int e = ks * APT / KSTEPS; e < (ks + 1) * APT / KSTEPS; ++e) {  \
                if (DO_STORE) YL_STORE_A(buf ^ 1, e)                                               \
                if (DO_LOAD) YL_LOAD_A((KB) + 2, e)                                                \
            }                                                                                      \
            if constexpr (VEC4) {                                                                  \
                _Pragma("unroll") for (int e = ks * BPT4 / KSTEPS; e < (ks + 1) * BPT4 / KSTEPS; ++e) { \
                    if (DO_STORE) YL_STORE_B4(buf ^ 1, e)                                          \
                    if (DO_LOAD) YL_LOAD_B4((KB) + 2, e)                                           \
                }                                                                                  \
            } else {                                                                               \
                _Pragma("unroll") for (int e = ks * BPT / KSTEPS; e < (ks + 1) * BPT / KSTEPS; ++e) { \
                    if (DO_STORE) YL_STORE_B(buf ^ 1, e)                                           \
                    if (DO_LOAD) YL_LOAD_B((KB) + 2, e)                                            \
                }                                                                                  \
            }                                                                                      \
            if (ks + 1 < KSTEPS) {                                                                 \
                _Pragma("unroll") for (int i = 0; i < TM; ++i)                                     \
                    av[nxt][i] = Ab[2 * (ks + 1) * BM + i * 32];                                   \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                     \
                    bv[nxt][j] = Bb[2 * (ks + 1) * BN + j * 32];                                   \
            }                                                                                      \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                         \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                     \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][j],       \
                                                                     acc[i][j], 0, 0, 0);          \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
        if (DO_LOAD) { YL_PANEL_ADVANCE() }                                                        \
        __syncthreads();                                                                           \
    }

    int kb = 0;
    if constexpr (NWAVES == 4 && BK == 16) {
        for (; kb + 3 < nkb; kb += 2) {
            YL_ITER_(kb, 0, false, true, true)
            YL_ITER_(kb + 1, 1, false, true, true)
        }
    }
    for (; kb + 2 < nkb; ++kb) YL_ITER(kb, true, true)
    if (kb + 1 < nkb) { YL_ITER(kb, true, false) ++kb; }
    if (kb < nkb) YL_ITER(kb, false, false)
#undef YL_ITER
#undef YL_ITER_
#undef YL_PANEL_SETUP
#undef YL_PANEL_ADVANCE
#undef YL_LOAD_A
#undef YL_LOAD_B
#undef YL_STORE_A
#undef YL_STORE_B
#undef YL_LOAD_B4
#undef YL_STORE_B4

    // ---- fused epilogue: +bias, activation (identical arithmetic to v1), then row-wise stores
    //      through a wave-private LDS strip (epilogue.h); the main loop's last barrier has passed,
    //      so the panel buffers are dead and are reused as the strips ----
    float *strip = smem + wave * ((Q_OK ? 16 : 8) * TN * 32);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float vals[TN][16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
            const float bv = (m < p.M) ? p.bias[m] : 0.f;
            bool logistic = false;
            if constexpr (YOLO) {
                const int entry = m % p.yolo_entries;
                logistic = entry != 2 && entry != 3;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float v = acc[i][j][e] + bv;
                if (p.act == YL_LEAKY) v = (v > 0.f) ? v : (float)(.1 * (double)v);
                if constexpr (YOLO) { if (logistic) v = (float)(1. / (1. + exp((double)(-v)))); }
                vals[j][e] = v;
            }
        }
        if (p.q_out && !p.out && !p.add) {          // only the int8 tensor is wanted (yolov3 layer 0 under -quantized)
            store_q_from_cd<TN>(vals, m0 + wm0 + i * 32, p.M, n0 + wn0, p.Ntotal, p.OHW, p.q_out, p.q_mult, p.q_G, lane);
            continue;
        }
        if constexpr (Q_OK) {
            if (p.q_out) {
                store_rows_via_lds_q<TN>(strip, vals, m0 + wm0 + i * 32, p.M, n0 + wn0, p.Ntotal, p.OHW,
                                         p.out, p.add, p.out_add, p.q_out, p.q_mult, p.q_G, lane);
                continue;
            }
        }
        store_rows_via_lds<TN>(strip, vals, m0 + wm0 + i * 32, p.M, n0 + wn0, p.Ntotal, p.OHW,
                               p.out, p.add, p.out_add, lane);
    }
}

template <int BM, int BN, int WM, int WN, int BK, int NWAVES>
static int launch_pipe(const ConvF32Dev &d, int ks, bool tapmajor, hipStream_t s, bool vec4 = false)
{
    // the folded [yolo] epilogue exists for the two tiles head convolutions take (64x64, 128x128r)
    constexpr bool YOLO_TILE = (BM == 64 && BN == 64) || (BM == 128 && BN == 128 && WM == 4) || (BM == 128 && BN == 256 && WM == 4);
    if (d.yolo_entries > 0 && !(YOLO_TILE && ks == 1 && !tapmajor)) return (int)hipErrorInvalidValue;
    ConvF32Dev p = d;
    p.tiles_m = (p.M + BM - 1) / BM;
    const int tiles_n = (p.Ntotal + BN - 1) / BN;
    const long long blocks = (long long)p.tiles_m * tiles_n;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    // the packed weights are zero-padded to a multiple of 32 rows: use only the panels K needs
    if ((p.K + BK - 1) / BK * BK > p.Kpad) return (int)hipErrorInvalidValue;
    p.Kpad = (p.K + BK - 1) / BK * BK;
    dim3 grid((unsigned)blocks), block(NWAVES * 64);
    if (tapmajor) {
        if (p.C % BK != 0 || p.size > 5) return (int)hipErrorInvalidValue;
        if (ks == 3) hipLaunchKernelGGL((conv_f32_mfma_pipe_kernel<BM, BN, WM, WN, 3, BK, NWAVES, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((conv_f32_mfma_pipe_kernel<BM, BN, WM, WN, 0, BK, NWAVES, true>), grid, block, 0, s, p);
    } else if (ks == 1 && vec4) {
        if (p.C % BK != 0 || p.OHW % 4 != 0 || p.stride != 1) return (int)hipErrorInvalidValue;
        if constexpr (YOLO_TILE) {
            if (p.yolo_entries > 0) {
                hipLaunchKernelGGL((conv_f32_mfma_pipe_kernel<BM, BN, WM, WN, 1, BK, NWAVES, false, true, true>), grid, block, 0, s, p);
                return (int)hipGetLastError();
            }
        }
        hipLaunchKernelGGL((conv_f32_mfma_pipe_kernel<BM, BN, WM, WN, 1, BK, NWAVES, false, true>), grid, block, 0, s, p);
    } else if (ks == 1) {
        if constexpr (YOLO_TILE) {
            if (p.yolo_entries > 0) {
                hipLaunchKernelGGL((conv_f32_mfma_pipe_kernel<BM, BN, WM, WN, 1, BK, NWAVES, false, false, true>), grid, block, 0, s, p);
                return (int)hipGetLastError();
            }
        }
        hipLaunchKernelGGL((conv_f32_mfma_pipe_kernel<BM, BN, WM, WN, 1, BK, NWAVES, false>), grid, block, 0, s, p);
    }
    else if (ks == 3) hipLaunchKernelGGL((conv_f32_mfma_pipe_kernel<BM, BN, WM, WN, 3, BK, NWAVES, false>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv_f32_mfma_pipe_kernel<BM, BN, WM, WN, 0, BK, NWAVES, false>), grid, block, 0, s, p);
    return (int)hipGetLastError();
}

// cfg: 1 128x128/4w  2 64x128/4w  3 32x256/4w  4 64x64/4w  5 128x128 BK32/4w  6 256x128/8w
//      7 128x256/8w  8 256x128 BK32/8w  9 128x128/8w(TM1)
static int launch_conv_f32_direct(const ConvF32Args &a, int cfg, int variant, void *stream, char *name, size_t name_len)
{
    ConvF32Dev d;
    d.in = a.in; d.wt = a.wt; d.bias = a.bias; d.add = a.add; d.out_add = a.out_add; d.out = a.out;
    d.q_out = a.q_out; d.q_mult = a.q_mult; d.q_G = a.q_G;
    if (a.q_out) {
        if (a.M % 16 != 0) return (int)hipErrorInvalidValue;
        if (a.out || a.add) {           // 16-row strips of the LDS quantise-on-store epilogue do not fit these two
            if (cfg == 10) cfg = 7;
            if (cfg == 11) cfg = 6;
        }
    }
    d.B = a.B; d.C = a.C; d.H = a.H; d.W = a.W; d.M = a.M; d.OH = a.OH; d.OW = a.OW;
    d.K = a.K; d.Kpad = a.Kpad; d.Mpad = a.Mpad;
    d.size = a.size; d.stride = a.stride; d.pad = a.pad; d.act = a.act;
    d.OHW = a.OH * a.OW;
    const long long nt = (long long)a.B * d.OHW;
    if (nt > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    d.Ntotal = (int)nt;
    d.tiles_m = 0;
    d.yolo_entries = a.yolo_entries;
    if (a.yolo_entries > 0) {
        if (a.size != 1 || a.pad != 0 || a.tapmajor || a.q_out || a.add) return (int)hipErrorInvalidValue;
        if (cfg != 4 && cfg != 12 && cfg != 10) cfg = 12;
    }
    hipStream_t s = (hipStream_t)stream;
    int ks = 0;
    if (a.size == 1 && a.pad == 0) ks = 1;
    else if (a.size == 3) ks = 3;
    const bool vec4 = (variant & 4) && ks == 1 && a.stride == 1 && !a.tapmajor && a.C % 32 == 0 && (d.OHW % 4) == 0;
    const char *t = "?";
    int rc;
    switch (cfg) {
    case 1: t = "128x128";      rc = launch_pipe<128, 128, 2, 2, 16, 4>(d, ks, a.tapmajor != 0, s, vec4); break;
    case 2: t = "64x128";       rc = launch_pipe<64, 128, 2, 2, 16, 4>(d, ks, a.tapmajor != 0, s, vec4); break;
    case 3: t = "32x256";       rc = launch_pipe<32, 256, 1, 4, 16, 4>(d, ks, a.tapmajor != 0, s, vec4); break;
    case 4: t = "64x64";        rc = launch_pipe<64, 64, 2, 2, 16, 4>(d, ks, a.tapmajor != 0, s, vec4); break;
    case 5: t = "128x128k32";   rc = launch_pipe<128, 128, 2, 2, 32, 4>(d, ks, a.tapmajor != 0, s, vec4); break;
    case 6: t = "256x128w8";    rc = launch_pipe<256, 128, 4, 2, 16, 8>(d, ks, a.tapmajor != 0, s, vec4); break;
    case 7: t = "128x256w8";    rc = launch_pipe<128, 256, 2, 4, 16, 8>(d, ks, a.tapmajor != 0, s, vec4); break;
    case 8: t = "256x128w8k32"; rc = launch_pipe<256, 128, 4, 2, 32, 8>(d, ks, a.tapmajor != 0, s, vec4); break;
    case 9: t = "128x128w8";    rc = launch_pipe<128, 128, 4, 2, 16, 8>(d, ks, a.tapmajor != 0, s, vec4); break;
    case 10: t = "128x256w8r";  rc = launch_pipe<128, 256, 4, 2, 16, 8>(d, ks, a.tapmajor != 0, s, vec4); break;
    case 11: t = "256x128w8r";  rc = launch_pipe<256, 128, 8, 1, 16, 8>(d, ks, a.tapmajor != 0, s, vec4); break;
    case 12: t = "128x128r";    rc = launch_pipe<128, 128, 4, 1, 16, 4>(d, ks, a.tapmajor != 0, s, vec4); break;
    default: return (int)hipErrorInvalidValue;
    }
    if (name) snprintf(name, name_len, "conv_f32_mfma_pipe<%s,ks%d%s%s%s>", t, ks, a.tapmajor ? ",tap" : "", vec4 ? ",v4" : "",
                       a.yolo_entries > 0 ? ",yolo" : "");
    return rc;
}

// The two branches of launch_conv_f32 below whose kernels have a pooled output (kept next to it on purpose).
int device_cu_count()
{
    constexpr int MAX_DEV = 64;
    static std::atomic<int> cache[MAX_DEV];          // zero-initialised; a racing first call stores the same value twice
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev >= 0 && dev < MAX_DEV) {
        v = cache[dev].load(std::memory_order_relaxed);
        if (v > 0) return v;
    }
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    if (dev >= 0 && dev < MAX_DEV) cache[dev].store(v, std::memory_order_relaxed);
    return v;
}

bool conv_f32_pool_fusable(const ConvF32Args &a0, const ConvF32Opts &o_in)
{
    ConvF32Opts o = o_in;
    if (o.force_tile >= 61 && o.force_tile <= 70) o.force_tile = 0;      // K1r's tile choice: every other layer keeps the heuristic
    ConvF32Args a = a0;
    a.pool_out = nullptr;
    if (a.q_out || a.bits_out || a.add || a.yolo_entries > 0 || ((a.H | a.W) & 1) || a.OH != a.H || a.OW != a.W) return false;
    if (o.force_tile == 0 && (o.variant & 8) && first_layer_valu_applicable(a)) return true;
    return a.wino32_u && (o.force_tile == 31 || (o.force_tile == 0 && o.winograd && a.C >= ((o.variant & 32) ? 16 : ((o.variant & 16) ? 32 : 64)) &&
                                                wino32_fits(a.B, a.M, a.H, a.W))) &&
           wino_applicable(a.C, a.M, a.size, a.stride, a.pad) && a.H >= 4 && a.W >= 4 && a.C / 4 >= 4 && ((a.C / 4) & 1) == 0;
}

bool conv_f32_two_source_now(const ConvF32Args &a, const ConvF32Opts &o)
{
    return o.force_tile == 0 && (o.variant & 1024) && !(o.variant & 4096) && x3_two_source_ok(a);
}

// Kernel choice for one FP32 convolution.  o.force_tile: 0 = heuristic, 11..22 = direct tile
// 1..12 of launch_conv_f32_direct, 31 = Winograd (error if the layer has no packed U).
int launch_conv_f32(const ConvF32Args &a, const ConvF32Opts &o_in, void *stream, char *name, size_t name_len)
{
    // force_tile 61..70 picks the tile / schedule of K1r WHERE a layer qualifies for it; every other layer of the network
    // keeps the heuristic (a whole network can be swept with one setting)
    ConvF32Opts o = o_in;
    int row3_tile = 0;
    if (o.force_tile >= 61 && o.force_tile <= 70) { row3_tile = o.force_tile - 60; o.force_tile = 0; }
    // measured on MI355X (tools/sweep_conv.py, yolov3-608 shapes, B=64): with 32 input channels
    // ([64,288,92416]) Winograd wins stand-alone (1.59 vs 2.16 ms) but not in the network, where the
    // layer carries a fused shortcut and is bound by 3 GB of epilogue traffic (2.31 vs 2.2 ms): C >= 64.
    if (a.in2)                                                        // two-source 1x1: only K1x reads it (the runtime asked conv_f32_two_source_now)
        return conv_f32_two_source_now(a, o) ? launch_conv_f32_x3(a, 0, stream, name, name_len, false) : (int)hipErrorInvalidValue;
    if (a.q_out && a.wino32_u) return (int)hipErrorInvalidValue;      // the planner gives q_out to direct layers only
    // a fused [maxpool] needs a kernel that has the pooled output (the runtime checks conv_f32_pool_fusable before every launch
    // and degrades to the stand-alone pooling kernel when a knob has moved the layer elsewhere: this is the backstop)
    if (a.pool_out && !conv_f32_pool_fusable(a, o)) return (int)hipErrorInvalidValue;
    // RGB first layers with <= 16 filters: the VALU kernel (force_tile 41 keeps the MFMA first-layer kernel for A/B)
    if (o.force_tile == 0 && (o.variant & 8) && first_layer_valu_applicable(a)) {
        if ((o.variant & 16384) && first_layer_mfma_applicable(a)) return launch_conv_f32_firstm(a, stream, name, name_len);
        return launch_conv_f32_first(a, stream, name, name_len);
    }
    if (a.bits_out)                                                   // sign-word side output: only the first-layer kernels have it
        return smallk_applicable(a) ? launch_conv_f32_smallk(a, stream, name, name_len) : (int)hipErrorInvalidValue;
    // K1x: the layer on the BF16 matrix pipe with three-piece operands (variant bit 10; force_tile 51..54 = its tiles)
    // (measured in yolov3-608: 0.60-0.72 of the FP32-MFMA kernel's time from 64 filters up, 1.13 at 32 filters, where the
    //  layer is bound by its tensors and the split's extra instructions only cost; profiles/r4_x3_per_layer.txt)
    const bool wino_takes = a.wino32_u && (o.force_tile == 31 || (o.force_tile == 0 && o.winograd &&
        a.C >= ((o.variant & 32) ? 16 : ((o.variant & 16) ? 32 : 64)) && wino32_fits(a.B, a.M, a.H, a.W)));
    // K1r: the 3x3 / stride-1 layers as row-wise Winograd F(2,3) on the BF16 matrix pipe with three-piece operands (variant bit
    // 11; force_tile 61..70 = its tiles); the layers with a pooled output keep the 2-D Winograd kernel (an F(2x2) tile is a window)
    if (a.row3_w && a.in_front_pad && !a.pool_out && !a.q_out && !a.bits_out && a.yolo_entries == 0 && (a.out || a.add) && o.force_tile == 0 &&
        (row3_tile || row3_fits(a.B, a.C, a.M, a.H, a.W)) && ((wino_takes && (o.variant & 2048)) || row3_tile))
        return launch_conv_f32_row3(a, row3_tile, stream, name, name_len, (o.variant & 8192) != 0);      // (a forced tile on a layer beyond its offsets fails there)
    if (!wino_takes && a.x3_w && (a.out || a.add) && !a.q_out && !a.pool_out && !a.bits_out && (a.yolo_entries == 0 || (a.size == 1 && !a.add)) &&
        ((o.force_tile == 0 && (o.variant & 1024) && a.M > 32) || (o.force_tile >= 51 && o.force_tile <= 55)))
        return launch_conv_f32_x3(a, o.force_tile >= 51 ? o.force_tile - 50 : 0, stream, name, name_len, (o.variant & 4096) != 0);
    if (o.force_tile >= 51 && o.force_tile <= 55) return (int)hipErrorInvalidValue;
    if (a.yolo_entries > 0)                                           // folded [yolo]: 1x1 direct kernel, two tiles
        return launch_conv_f32_direct(a, (o.force_tile == 14 || o.force_tile == 22 || o.force_tile == 20) ? o.force_tile - 10 :
                                      (((long long)((a.M + 127) / 128) * (((long long)a.B * a.OH * a.OW + 255) / 256) >= 384) ? 10 :
                                       (((long long)((a.M + 127) / 128) * (((long long)a.B * a.OH * a.OW + 127) / 128) >= 512) ? 12 : 4)),
                                      o.variant, stream, name, name_len);
    if (a.wino32_u && (o.force_tile == 31 || (o.force_tile == 0 && o.winograd && a.C >= ((o.variant & 32) ? 16 : ((o.variant & 16) ? 32 : 64)) &&
                                             wino32_fits(a.B, a.M, a.H, a.W))))
        return launch_conv_f32_wino32(a, a.wino32_u, o.variant, stream, name, name_len);
    if (o.force_tile == 31) return (int)hipErrorInvalidValue;   // forced on a layer without packed U
    if (o.force_tile == 41 || (o.force_tile == 0 && (o.variant & 8) && smallk_applicable(a)))
        return launch_conv_f32_smallk(a, stream, name, name_len);
    int cfg = o.force_tile >= 10 ? o.force_tile - 10 : 0;
    if (cfg == 0) {
        const long long ntot = (long long)a.B * a.OH * a.OW;
        auto nblocks = [&](int bm, int bn) { return (long long)((a.M + bm - 1) / bm) * ((ntot + bn - 1) / bn); };
        // measured on MI355X with tools/sweep_conv.py (profiles/r1_sweep_conv_tiles_b64*.txt):
        // tiles whose waves own 128 consecutive pixels (512-byte rows in the LDS-staged
        // epilogue) win: 4-wave 128x128 (TM=1,TN=4) for M <= 256 and for 1x1, 8-wave 128x256
        // (TM=1,TN=4) for wider 3x3 layers; 64x128 / 32x256 for the narrow-M early layers;
        // layers too small to give every CU two workgroups fall back to 64x64 tiles.
        if (a.M <= 32) cfg = 3;
        // 3x3 stride 2, M = 64 (yolov3 layer 1): 64x128 2.04 ms vs 64x64 2.16 since the unrolled K loop (round 4,
        // profiles/r4_sweep_direct_tiles_steady_state.txt; before it 2.54 vs 2.34)
        else if (a.M <= 64) cfg = 2;
        // round 4 sweep (profiles/r4_sweep_stride2_tiles.txt): 3x3 layers from 128 filters up take the 8-wave 128x256 tile
        // as well (M = 128, 256 at stride 2: 1.92 -> 1.82 ms, 1.81 -> 1.74 ms)
        // round 4, in-network A/B (profiles/r4_ab_1x1_tile.txt): 1x1 layers from 128 filters on the 8-wave 128x256 tile,
        // 22 launches 5.17 -> 4.82 ms, +0.7 % on the step
        else if (a.size == 1 && a.M >= 128 && nblocks(128, 256) >= 384) cfg = 10;
        else if (a.size == 1 || a.M < 128) cfg = (nblocks(128, 128) >= 512) ? 12 : 4;
        // same sweep: 3x3 stride 2 from 256 filters up on the 4-wave 128x128 tile (1.64 / 1.68 / 1.68 ms vs 1.73 / 1.70 / 1.72
        // on the 8-wave 128x256 one, which keeps M = 128: 1.73 vs 1.75)
        else if (a.M >= 256 && nblocks(128, 128) >= 512) cfg = 12;
        else cfg = (nblocks(128, 256) >= 384) ? 10 : ((nblocks(128, 128) >= 512) ? 12 : 4);
        if (cfg <= 3 && nblocks(cfg == 2 ? 64 : 32, cfg == 3 ? 256 : 128) < 512) cfg = 4;
    }
    // BK=32 variants need C % 32 == 0 in tap-major order
    if ((cfg == 5 || cfg == 8) && a.tapmajor) cfg = (cfg == 5) ? 1 : 6;   // tap-major blocks are 16 channels
    return launch_conv_f32_direct(a, cfg, o.variant, stream, name, name_len);
}

}  // namespace yl
