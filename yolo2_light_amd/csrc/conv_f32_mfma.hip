// conv_f32_mfma.hip -- K1: FP32 implicit-GEMM convolution for gfx950 (CDNA4).
//
// Replaces forward_convolutional_layer_cpu's FP32 branch
// (src/yolov2_forward_network.c:30-38,103-111,204-215,243-261: im2col_cpu ->
// gemm_nn -> +bias -> activate_array_cpu_custom) and the reference GPU's
// cudnnConvolutionForward + add_bias_gpu + activate_array_ongpu
// (src/yolov2_forward_network_gpu.cu:113-138).
//
//   GEMM view:  C[M = filters][N = batch*out_h*out_w] = A[M][K] * B[K][N],
//               K = c*size*size ordered (c, ky, kx) exactly like im2col_cpu
//               (src/additionally.c:39-62).
//   A  (weights)  : pre-packed once, k-major [Kpad][Mpad] so a BK x BM panel is
//                   BK rows of BM contiguous floats (coalesced float4 loads).
//   B  (im2col)   : never materialised.  Each thread owns ONE output pixel of
//                   the tile for the whole K loop (its (b,oy,ox) decode, base
//                   pointer and 3x3 halo-validity mask are computed once); the
//                   k -> (c,ky,kx) decode is wave-uniform and runs on the SALU.
//   math          : v_mfma_f32_32x32x2_f32 -- exact f32, 64 FLOP/clk/SIMD,
//                   157.3 TFLOP/s chip peak (MI355X_MICROARCH.md).  One wave
//                   per SIMD keeps the pipe full, so a 256-thread workgroup
//                   computes BM x BN with 4 waves of (TM x TN) 32x32 tiles.
//   LDS           : As[2][BK][BM], Bs[2][BK][BN] f32, double-buffered, one
//                   barrier per BK=16 step; operand reads are conflict-free
//                   ds_read_b32 (lanes 0-31 -> 32 consecutive dwords of row
//                   k, lanes 32-63 -> row k+1).
//   epilogue      : fused  +bias -> leaky (x>0 ? x : (float)(.1*(double)x),
//                   the scalar reference's arithmetic, src/additionally.h:91)
//                   -> optional residual add -> NCHW stores (32 lanes = 128 B
//                   contiguous per row).
//   grid          : 1-D, XCD-aware bijective remap so that the M-tiles sharing
//                   one im2col N-tile run on the same XCD (shared L2).
#include <hip/hip_runtime.h>
#include <cstdio>

#include "kernels.h"
#include "../../include/yolo2_hip.h"

namespace yl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;
constexpr int NTHREADS = 256;

struct ConvF32Dev {
    const float *in;
    const float *wt;
    const float *bias;
    const float *add;
    float *out_add;
    float *out;
    int B, C, H, W, M, OH, OW;
    int K, Kpad, Mpad;
    int size, stride, pad;
    int act;
    int Ntotal;       // B*OH*OW
    int OHW;
    int tiles_m;
};

template <int BM, int BN, int WM, int WN, int KS>
__global__ __launch_bounds__(NTHREADS) void conv_f32_mfma_kernel(ConvF32Dev p)
{
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(TM >= 1 && TN >= 1, "tile too small");
    static_assert(BN % 64 == 0 && BN <= NTHREADS, "a wave must stay inside one k row of the B panel");
    constexpr int A_F4 = BK * BM / 4;                            // float4 per A panel
    constexpr int A_PER_THREAD = (A_F4 + NTHREADS - 1) / NTHREADS;
    constexpr bool A_FULL = (A_F4 % NTHREADS) == 0;
    constexpr int B_PER_THREAD = BK * BN / NTHREADS;             // gathered floats per thread per panel
    constexpr int K_STEP = NTHREADS / BN;                        // k rows between a thread's gathers

    __shared__ __attribute__((aligned(16))) float smem[2 * BK * BM + 2 * BK * BN];
    float *As = smem;                    // [2][BK][BM]
    float *Bs = smem + 2 * BK * BM;      // [2][BK][BN]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int half = lane >> 5;

    // ---- XCD-aware bijective block remap (cdna_hip_programming.md T1) ----
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tile_m = logical % p.tiles_m;
    const int tile_n = logical / p.tiles_m;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- per-thread im2col state: one output pixel for the whole K loop ----
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

    // Buffer descriptor over the input, based at the first image this tile touches and shifted
    // back by pad*(W+1) elements so every lane's voffset is >= 0 (the hardware range check looks
    // at voffset only).  Taps outside the image / pixels past the end get voffset = 0xFFFFFFFF,
    // which is out of range for any num_records and therefore loads 0.0f: zero padding for free.
    const int b_first = n0 / p.OHW;                                   // wave-uniform
    const float *tile_base = p.in + (size_t)b_first * CHW - (ptrdiff_t)p.pad * (p.W + 1);
    const size_t total_bytes = (size_t)p.B * CHW * sizeof(float);
    const size_t base_off_bytes = ((size_t)b_first * CHW) * sizeof(float);
    size_t rec = total_bytes - base_off_bytes + (size_t)p.pad * (p.W + 1) * sizeof(float);
    if (rec > 0xFFFFFFFEull) rec = 0xFFFFFFFEull;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)tile_base, 0, (int)(unsigned)rec, 0x00020000);
    const int voff = (int)(((unsigned)(bimg - b_first) * (unsigned)CHW +
                            (unsigned)(oy * p.stride) * (unsigned)p.W + (unsigned)(ox * p.stride)) * 4u);

    // inverted tap-validity mask: bit (ky*KS+kx) set <=> that tap is OUTSIDE the image
    unsigned ntapmask = 0xFFFFFFFFu;
    if (n_ok) {
        if (KS == 1) {
            ntapmask = 0u;
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

    float a_reg[A_PER_THREAD][4];
    float b_reg[B_PER_THREAD];

#define YL_LOAD_PANEL(KB)                                                                          \
    {                                                                                              \
        _Pragma("unroll") for (int i = 0; i < A_PER_THREAD; ++i) {                                 \
            const int idx = tid + i * NTHREADS;                                                    \
            if (A_FULL || idx < A_F4) {                                                            \
                const int kr = idx / (BM / 4);                                                     \
                const int c4 = idx - kr * (BM / 4);                                                \
                const float4 t4 = *reinterpret_cast<const float4 *>(                               \
                    p.wt + (size_t)((KB) * BK + kr) * p.Mpad + m0 + c4 * 4);                       \
                a_reg[i][0] = t4.x; a_reg[i][1] = t4.y; a_reg[i][2] = t4.z; a_reg[i][3] = t4.w;    \
            }                                                                                      \
        }                                                                                          \
        _Pragma("unroll") for (int i = 0; i < B_PER_THREAD; ++i) {                                 \
            const int k = (KB) * BK + krow0 + i * K_STEP;          /* wave-uniform */              \
            const int kinv = (k >= p.K) ? -1 : 0;                                                  \
            int soff, tinv;                                                                        \
            if (KS == 1) {                                                                         \
                soff = k * HW * 4;                                                                 \
                tinv = (int)ntapmask;                                                              \
            } else if (KS == 3) {                                                                  \
                const int c = k / 9;                                                               \
                const int rr = k - c * 9;                                                          \
                const int ky = rr / 3;                                                             \
                const int kx = rr - ky * 3;                                                        \
                soff = (c * HW + ky * p.W + kx) * 4;                                               \
                tinv = __builtin_amdgcn_sbfe((int)ntapmask, rr, 1);                                \
            } else {                                                                               \
                const int ss = p.size * p.size;                                                    \
                const int c = k / ss;                                                              \
                const int rr = k - c * ss;                                                         \
                const int ky = rr / p.size;                                                        \
                const int kx = rr - ky * p.size;                                                   \
                const int iy = iy0 + ky, ix = ix0 + kx;                                            \
                soff = (c * HW + ky * p.W + kx) * 4;                                               \
                tinv = (n_ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? 0 : -1;              \
            }                                                                                      \
            if (kinv) soff = 0;                                                                    \
            const unsigned raw = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff | tinv | kinv, soff, 0); \
            b_reg[i] = __builtin_bit_cast(float, raw);                                                                          \
        }                                                                                          \
    }

#define YL_STORE_PANEL(BUF)                                                                        \
    {                                                                                              \
        float *Ab_ = As + (BUF) * BK * BM;                                                         \
        float *Bb_ = Bs + (BUF) * BK * BN;                                                         \
        _Pragma("unroll") for (int i = 0; i < A_PER_THREAD; ++i) {                                 \
            const int idx = tid + i * NTHREADS;                                                    \
            if (A_FULL || idx < A_F4)                                                              \
                *reinterpret_cast<float4 *>(Ab_ + idx * 4) =                                       \
                    make_float4(a_reg[i][0], a_reg[i][1], a_reg[i][2], a_reg[i][3]);               \
        }                                                                                          \
        _Pragma("unroll") for (int i = 0; i < B_PER_THREAD; ++i)                                   \
            Bb_[(krow0 + i * K_STEP) * BN + n_local] = b_reg[i];                                   \
    }

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

    YL_LOAD_PANEL(0)
    YL_STORE_PANEL(0)
    __syncthreads();

    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        const bool more = (kb + 1 < nkb);
        if (more) YL_LOAD_PANEL(kb + 1)

        const float *Ab = As + buf * BK * BM + wm0 + l31;
        const float *Bb = Bs + buf * BK * BN + wn0 + l31;
        // operand reads are software-pipelined one k-step ahead of the MFMAs that use them
        float av[2][TM], bv[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) av[0][i] = Ab[half * BM + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[0][j] = Bb[half * BN + j * 32];
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) av[nxt][i] = Ab[(2 * (ks + 1) + half) * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[nxt][j] = Bb[(2 * (ks + 1) + half) * BN + j * 32];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
        }

        if (more) YL_STORE_PANEL(buf ^ 1)
        __syncthreads();
    }
#undef YL_LOAD_PANEL
#undef YL_STORE_PANEL

    // ---- fused epilogue: +bias, activation, optional residual, NCHW store ----
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + l31;
        if (n >= p.Ntotal) continue;
        const int ob = n / p.OHW;
        const int opix = n - ob * p.OHW;
        const size_t obase = (size_t)ob * p.M * p.OHW + opix;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (m < p.M) {
                    float v = acc[i][j][e] + p.bias[m];
                    if (p.act == YL_LEAKY) v = (v > 0.f) ? v : (float)(.1 * (double)v);
                    const size_t o = obase + (size_t)m * p.OHW;
                    if (p.out) p.out[o] = v;
                    if (p.add) p.out_add[o] = v + p.add[o];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------
// host-side tile selection + launch
// ------------------------------------------------------------------------
static int g_force_tile = 0;
static char g_last_tile[64] = "";
static void set_tile_name(const char *t, int ks) { snprintf(g_last_tile, sizeof(g_last_tile), "conv_f32_mfma<%s,ks%d>", t, ks); }
void conv_f32_force_tile(int cfg) { g_force_tile = cfg; }
int conv_f32_forced_tile() { return g_force_tile; }
const char *conv_f32_last_tile_name() { return g_last_tile; }

template <int BM, int BN, int WM, int WN>
static int launch_tile(const ConvF32Dev &d, int ks, hipStream_t s)
{
    ConvF32Dev p = d;
    p.tiles_m = (p.M + BM - 1) / BM;
    const int tiles_n = (p.Ntotal + BN - 1) / BN;
    const long long blocks = (long long)p.tiles_m * tiles_n;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    dim3 grid((unsigned)blocks), block(NTHREADS);
    if (ks == 1) hipLaunchKernelGGL((conv_f32_mfma_kernel<BM, BN, WM, WN, 1>), grid, block, 0, s, p);
    else if (ks == 3) hipLaunchKernelGGL((conv_f32_mfma_kernel<BM, BN, WM, WN, 3>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv_f32_mfma_kernel<BM, BN, WM, WN, 0>), grid, block, 0, s, p);
    return (int)hipGetLastError();
}

static int g_variant = 1;      // 0 = v1 burst schedule, 1 = v2 software-pipelined schedule (default)
void conv_f32_set_variant(int v) { g_variant = v; }
int conv_f32_get_variant() { return g_variant; }

static int g_winograd = 1;
void conv_f32_set_winograd(int mode) { g_winograd = mode; }
int conv_f32_get_winograd() { return g_winograd; }

int launch_conv_f32(const ConvF32Args &a, void *stream)
{
    // measured on MI355X (tools/sweep_conv.py, yolov3-608 shapes, B=64): the 32-filter tiling (two
    // workgroups per CU) beats the 64-filter one by ~5 % everywhere.  With 32 input channels
    // ([64,288,92416]) it wins stand-alone (1.59 vs 2.16 ms) but not in the network, where the layer
    // carries a fused shortcut and is bound by 3 GB of epilogue traffic (2.31 vs 2.2 ms): C >= 64.
    if (a.wino32_u && (g_force_tile == 31 || (g_force_tile == 0 && g_winograd && a.C >= 64)))
        return launch_conv_f32_wino32(a, a.wino32_u, stream, g_last_tile, sizeof(g_last_tile));
    if (a.wino_u && g_force_tile == 30)
        return launch_conv_f32_wino(a, a.wino_u, stream, g_last_tile, sizeof(g_last_tile));
    if (g_force_tile == 30 || g_force_tile == 31) return (int)hipErrorInvalidValue;   // forced on a layer without packed U
    if (a.tapmajor || g_variant >= 1) {
        int cfg = g_force_tile >= 10 ? g_force_tile - 10 : 0;
        if (cfg == 0) {
            const long long ntot = (long long)a.B * a.OH * a.OW;
            auto nblocks = [&](int bm, int bn) { return (long long)((a.M + bm - 1) / bm) * ((ntot + bn - 1) / bn); };
            // measured on MI355X with tools/sweep_conv.py (profiles/r1_sweep_conv_tiles_b64*.txt):
            // tiles whose waves own 128 consecutive pixels (512-byte rows in the LDS-staged
            // epilogue) win: 4-wave 128x128 (TM=1,TN=4) for M <= 256 and for 1x1, 8-wave 128x256
            // (TM=1,TN=4) for wider 3x3 layers; 64x128 / 32x256 for the narrow-M early layers;
            // layers too small to give every CU two workgroups fall back to 64x64 tiles.
            if (a.M <= 32) cfg = 3;
            else if (a.M <= 64) cfg = (a.size == 3) ? 4 : 2;      // 3x3 stride 2, M = 64: 64x64 2.34 ms vs 64x128 2.54
            else if (a.size == 1 || a.M <= 256) cfg = (nblocks(128, 128) >= 512) ? 12 : 4;
            else cfg = (nblocks(128, 256) >= 384) ? 10 : ((nblocks(128, 128) >= 512) ? 12 : 4);
            if (cfg <= 3 && nblocks(cfg == 2 ? 64 : 32, cfg == 3 ? 256 : 128) < 512) cfg = 4;
        }
        // BK=32 variants need C % 32 == 0 in tap-major order
        if ((cfg == 5 || cfg == 8) && a.tapmajor) cfg = (cfg == 5) ? 1 : 6;   // tap-major blocks are 16 channels
        return launch_conv_f32_v2(a, cfg, stream, g_last_tile, sizeof(g_last_tile));
    }
    ConvF32Dev d;
    d.in = a.in; d.wt = a.wt; d.bias = a.bias; d.add = a.add; d.out_add = a.out_add; d.out = a.out;
    d.B = a.B; d.C = a.C; d.H = a.H; d.W = a.W; d.M = a.M; d.OH = a.OH; d.OW = a.OW;
    d.K = a.K; d.Kpad = a.Kpad; d.Mpad = a.Mpad;
    d.size = a.size; d.stride = a.stride; d.pad = a.pad;
    d.act = a.act;
    d.OHW = a.OH * a.OW;
    const long long nt = (long long)a.B * d.OHW;
    if (nt > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    d.Ntotal = (int)nt;
    d.tiles_m = 0;
    hipStream_t s = (hipStream_t)stream;
    // im2col validity masks are precomputed for 1x1 (pad 0) and 3x3; anything else -> generic
    int ks = 0;
    if (a.size == 1 && a.pad == 0) ks = 1;
    else if (a.size == 3) ks = 3;

    int cfg = g_force_tile;
    if (cfg == 0) {
        // heuristic: widest tile that still gives >= ~2 workgroups per CU
        const long long want = 512;
        auto nblocks = [&](int bm, int bn) {
            return (long long)((a.M + bm - 1) / bm) * ((d.Ntotal + bn - 1) / bn);
        };
        if (a.M <= 32) cfg = 3;
        else if (a.M <= 64) cfg = 2;
        else cfg = 1;
        if (nblocks(cfg == 1 ? 128 : (cfg == 2 ? 64 : 32), cfg == 3 ? 256 : 128) < want) {
            cfg = (a.M <= 32) ? 5 : 4;
        }
    }
    switch (cfg) {
    case 1: set_tile_name("128x128", ks); return launch_tile<128, 128, 2, 2>(d, ks, s);
    case 2: set_tile_name("64x128", ks); return launch_tile<64, 128, 2, 2>(d, ks, s);
    case 3: set_tile_name("32x256", ks); return launch_tile<32, 256, 1, 4>(d, ks, s);
    case 4: set_tile_name("64x64", ks); return launch_tile<64, 64, 2, 2>(d, ks, s);
    case 5: set_tile_name("32x128", ks); return launch_tile<32, 128, 1, 4>(d, ks, s);
    case 6: set_tile_name("128x64", ks); return launch_tile<128, 64, 4, 1>(d, ks, s);
    case 7: set_tile_name("256x64", ks); return launch_tile<256, 64, 4, 1>(d, ks, s);
    case 8: set_tile_name("128x256", ks); return launch_tile<128, 256, 2, 2>(d, ks, s);
    default: return (int)hipErrorInvalidValue;
    }
}

}  // namespace yl
