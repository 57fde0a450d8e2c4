// mfma_valu_contention.hip -- what does a VALU instruction of ANOTHER wave on the same SIMD cost the matrix pipe?
//
// Workgroups of 8 waves on every CU: waves 0-3 (one per SIMD) issue 16 independent MFMAs per iteration from registers,
// waves 4-7 (their SIMD neighbours) issue NV v_add_f32 per iteration.  For each MFMA type the table gives the time per
// iteration of the matrix waves against NV; the slope is the matrix-pipe time one wave64 VALU instruction takes away.
// tools/mfma_f32_bench.hip measured 4.9 cycles for v_mfma_f32_32x32x2_f32; this tool repeats it for the INT8 and BF16
// MFMAs the other two convolution kernels use (conv_i8_mfma.hip, conv_bf16_mfma.hip).
//
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu_contention tools/mfma_valu_contention.hip && /tmp/mfma_valu_contention
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// KIND 0: v_mfma_f32_32x32x2_f32, 1: v_mfma_i32_32x32x32_i8, 2: v_mfma_f32_32x32x16_bf16
template <int KIND, int NV>
__global__ __launch_bounds__(512, 2) void contention_kernel(float *sink, int iters)
{
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wave < 4) {
        if constexpr (KIND == 0) {
            f32x16 acc[8];
            for (int p = 0; p < 8; ++p)
                for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
            float a[8], b[8];
            for (int p = 0; p < 8; ++p) { a[p] = 1.f + p + (tid & 3); b[p] = 0.5f - p; }
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int p = 0; p < 8; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p], b[p], acc[p], 0, 0, 0);
            }
            float s = 0.f;
            for (int p = 0; p < 8; ++p)
                for (int e = 0; e < 16; ++e) s += acc[p][e];
            if (s == 12345.678f) sink[0] = s;
        } else if constexpr (KIND == 1) {
            v16i acc[8];
            for (int p = 0; p < 8; ++p)
                for (int e = 0; e < 16; ++e) acc[p][e] = 0;
            v4i a[8], b[8];
            for (int p = 0; p < 8; ++p)
                for (int e = 0; e < 4; ++e) { a[p][e] = 0x01020304 * (p + 1) + tid; b[p][e] = 0x04030201 + p * 77 + e; }
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int p = 0; p < 8; ++p) acc[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[p], b[p], acc[p], 0, 0, 0);
            }
            int s = 0;
            for (int p = 0; p < 8; ++p)
                for (int e = 0; e < 16; ++e) s += acc[p][e];
            if (s == 123456789) sink[0] = (float)s;
        } else {
            f32x16 acc[8];
            for (int p = 0; p < 8; ++p)
                for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
            v4i a[8], b[8];
            for (int p = 0; p < 8; ++p)
                for (int e = 0; e < 4; ++e) { a[p][e] = 0x3f803f80 + (p << 16) + (tid & 7); b[p][e] = 0x3f003e80 + p + (e << 16); }
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int p = 0; p < 8; ++p)
                        acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[p]), __builtin_bit_cast(bf16x8, b[p]), acc[p], 0, 0, 0);
            }
            float s = 0.f;
            for (int p = 0; p < 8; ++p)
                for (int e = 0; e < 16; ++e) s += acc[p][e];
            if (s == 12345.678f) sink[0] = s;
        }
    } else {
        float x[8];
        for (int p = 0; p < 8; ++p) x[p] = 3.f + p + (tid & 63);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int v = 0; v < NV / 8; ++v)
#pragma unroll
                for (int p = 0; p < 8; ++p) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[p]) : "v"(x[(p + 3) & 7]));
        }
        float s = 0.f;
        for (int p = 0; p < 8; ++p) s += x[p];
        if (s == 12345.678f) sink[1] = s;
    }
}

template <int KIND, int NV>
static double run(int n_cu, float *d_sink)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((contention_kernel<KIND, NV>), dim3(n_cu), dim3(512), 0, 0, d_sink, 2000);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((contention_kernel<KIND, NV>), dim3(n_cu), dim3(512), 0, 0, d_sink, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return (double)ms * 1e-3 / iters;          // seconds per iteration (16 MFMAs per matrix wave)
}

template <int KIND>
static void table(int n_cu, float *d_sink, const char *what, double ops_per_mfma, double clk_ghz)
{
    const double t0 = run<KIND, 0>(n_cu, d_sink), t16 = run<KIND, 16>(n_cu, d_sink), t32 = run<KIND, 32>(n_cu, d_sink),
                 t64 = run<KIND, 64>(n_cu, d_sink), t128 = run<KIND, 128>(n_cu, d_sink);
    const double per_mfma0 = t0 / 16 * clk_ghz * 1e9;
    printf("%-28s alone: %.1f cycles per MFMA (%.0f T op/s on %d CUs)\n", what, per_mfma0, n_cu * 4 * 16 * ops_per_mfma / t0 / 1e12, n_cu);
    const double ts[4] = {t16, t32, t64, t128};
    const int nv[4] = {16, 32, 64, 128};
    for (int k = 0; k < 4; ++k)
        printf("    + %3d v_add_f32 per 16 MFMAs on the neighbour wave: %.1f cycles per MFMA, %.3f of the rate alone, %.2f cycles per VALU instruction\n",
               nv[k], ts[k] / 16 * clk_ghz * 1e9, t0 / ts[k], (ts[k] - t0) * clk_ghz * 1e9 / nv[k]);
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    float *d_sink;
    CHECK(hipMalloc(&d_sink, 2 * sizeof(float)));
    const double clk = 2.35;          // GHz: the clock rocm-smi reports under these loops (profiles/r4_ab_wino_64x32_ablation_clock.txt)
    printf("# one matrix wave + one VALU wave per SIMD, %d CUs, cycles at %.2f GHz\n", n_cu, clk);
    table<0>(n_cu, d_sink, "v_mfma_f32_32x32x2_f32", 2.0 * 32 * 32 * 2, clk);
    table<1>(n_cu, d_sink, "v_mfma_i32_32x32x32_i8", 2.0 * 32 * 32 * 32, clk);
    table<2>(n_cu, d_sink, "v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16, clk);
    return 0;
}
