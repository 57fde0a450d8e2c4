// mfma_f32_bench.hip -- what does the loop SKELETON of the Winograd kernel (conv_f32_wino32.hip) cost on its own?
//
// The kernel issues 16 v_mfma_f32_32x32x2_f32 per 4-channel panel and wave (8 accumulators x 2 k-pairs), reads 16
// 8-byte operand fragments from LDS for the next panel and crosses two workgroup barriers per panel; two workgroups
// of 4 waves share a CU.  Round 3's ablation put the "skeleton" (MFMAs + fragment reads + barriers + epilogue) at
// 0.74 of the FP32 matrix peak.  This microbenchmark separates the pieces on every CU:
//   mode 0  16 MFMAs per iteration, nothing else
//   mode 1  + 16 ds_read_b64 of the next iteration's operands (double-buffered registers)
//   mode 2  + one barrier per iteration (after 16 MFMAs)
//   mode 3  + two barriers per iteration (after every 8 MFMAs: the kernel's structure)
//   mode 4  mode 3 + 10 ds_write_b64 per iteration (the staging stores, without the transform)
//   mode 5  mode 3 + the same 80 bytes per lane as 20 ds_write_b32
//   mode 6  ... as 5 ds_write_b128
//   mode 7  ... as 20 ds_write_addtid_b32 (address = M0 + offset + 4 * lane, no address VGPR)
//   mode 8  mode 4 with the ten stores in one block BEHIND the 8 MFMAs instead of one per MFMA
//   mode 9  mode 4 with store data from registers no MFMA reads
//   mode 10 mode 9 + 72 dependent-free v_add_f32 / v_sub_f32 per iteration (the size of the input transform), 9 per MFMA in the first half
//   mode 11 mode 10 + 6 16-byte global loads per iteration from a 64 MB L2/MALL-resident buffer, consumed by the adds
//   mode 12 mode 11 with the 72 VALU spread over all 16 MFMAs
// each with 1 and 2 workgroups per CU.  Output: TFLOP/s (2 * 32*32*2 per MFMA) and the fraction of 157.3.
//
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f32_bench tools/mfma_f32_bench.hip && /tmp/mfma_f32_bench
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

template <int MODE>
__global__ __launch_bounds__(256, 2) void skeleton_kernel(float *sink, int iters, float seed, const float4 *gsrc, size_t gmask)
{
    float4 ld[6], lprev[6];
#pragma unroll
    for (int w = 0; w < 6; ++w) lprev[w] = make_float4(seed, seed, seed, seed);
    const size_t goff = (size_t)blockIdx.x * 4099 + threadIdx.x;
    __shared__ __attribute__((aligned(16))) float lds[12288];          // 48 KB, like the kernel's two panel stages
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    for (int i = tid; i < 12288; i += 256) lds[i] = seed + i * 1e-6f;
    __syncthreads();
    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
    float2 fa[2][8], fb[2][8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        fa[0][p] = make_float2(seed + p, seed - p);
        fb[0][p] = make_float2(seed * 0.5f + p, seed * 0.25f - p);
        fa[1][p] = fa[0][p];
        fb[1][p] = fb[0][p];
    }
    const float *base = lds + lane * 2;
    float *wbase = lds + 6144 + tid * 2;
    float *wbase1 = lds + 6144 + tid;
    float *wbase4 = lds + 6144 + tid * 4;
    const unsigned m0v = (unsigned)((tid >> 6) * 256 + 0);       // wave-uniform LDS byte base of the addtid stores
    float extra[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) extra[p] = seed * 3.f + p + lane;
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int buf = s;
            if (MODE >= 11) {
#pragma unroll
                for (int w = 0; w < 6; ++w) ld[w] = gsrc[(goff + (size_t)(it + s) * 1536 + w * 256) & gmask];
            }
            if (MODE >= 10) {
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    // 9 adds per group: a small butterfly on the extra registers (results feed the stores)
                    const float a0 = extra[w], a1 = extra[(w + 1) & 7], a2 = extra[(w + 2) & 7], a3 = extra[(w + 3) & 7];
                    const float g0 = MODE >= 11 ? (w < 6 ? lprev[w].x : lprev[w - 6].y) : 0.f;
                    const float t0 = a0 - a2, t1 = a1 + a2, t2 = a2 - a1, t3 = a1 - a3;
                    const float u0 = t0 - t2, u1 = t1 + t2, u2 = t2 - t1, u3 = t1 - t3;
                    extra[w] = (u0 + u1) * 0.5f + (u2 - u3) + g0;
                }
            }
            if (MODE >= 11) {
#pragma unroll
                for (int w = 0; w < 6; ++w) lprev[w] = ld[w];
            }
            if (MODE == 4 || MODE == 8 || MODE >= 9) {
#pragma unroll
                for (int w = 0; w < 10; ++w) {
                    const float2 v = MODE >= 9 ? make_float2(extra[w & 7], extra[(w + 1) & 7]) : make_float2(fa[s][w & 7].x, fb[s][w & 7].y);
                    *reinterpret_cast<float2 *>(wbase + ((w * 512 + buf * 4) & 4095)) = v;
                }
            }
            if (MODE == 5) {
#pragma unroll
                for (int w = 0; w < 20; ++w) wbase1[((w * 256 + buf * 4) & 4095)] = (w & 1) ? fa[s][(w >> 1) & 7].x : fb[s][(w >> 1) & 7].y;
            }
            if (MODE == 6) {
#pragma unroll
                for (int w = 0; w < 5; ++w)
                    *reinterpret_cast<float4 *>(wbase4 + ((w * 1024 + buf * 4) & 4095)) = make_float4(fa[s][w].x, fb[s][w].y, fa[s][w + 1].x, fb[s][w + 1].y);
            }
            if (MODE == 7) {
#pragma unroll
                for (int w = 0; w < 20; ++w) {
                    const float v = (w & 1) ? fa[s][(w >> 1) & 7].x : fb[s][(w >> 1) & 7].y;
                    asm volatile("ds_write_addtid_b32 %0 offset:%1" : : "v"(v), "n"(24576 + (MODE == 7 ? 0 : 0)), "{m0}"(m0v) : "memory");
                }
            }
#pragma unroll
            for (int p = 0; p < 8; ++p)
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][p].x, fb[s][p].x, acc[p], 0, 0, 0);
            if (MODE == 4 || MODE == 9) {
#pragma unroll
                for (int i_ = 0; i_ < 8; ++i_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            }
            if (MODE == 10 || MODE == 11) {
#pragma unroll
                for (int i_ = 0; i_ < 8; ++i_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 11, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            }
            if (MODE == 12) {
#pragma unroll
                for (int i_ = 0; i_ < 8; ++i_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            }
            if (MODE == 5) {
#pragma unroll
                for (int i_ = 0; i_ < 8; ++i_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
                }
            }
            if (MODE == 8) {
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 10, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE >= 3) __syncthreads();
            if (MODE >= 1) {
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    fa[s ^ 1][p] = *reinterpret_cast<const float2 *>(base + p * 128 + buf * 2048);
                    fb[s ^ 1][p] = *reinterpret_cast<const float2 *>(base + 1024 + p * 128 + buf * 2048);
                }
            }
#pragma unroll
            for (int p = 0; p < 8; ++p)
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][p].y, fb[s][p].y, acc[p], 0, 0, 0);
            if (MODE >= 1) {
#pragma unroll
                for (int i_ = 0; i_ < 8; ++i_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    if (MODE == 12) __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                    if (MODE >= 11 && i_ < 6) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE >= 2) __syncthreads();
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) sum += acc[p][e];
#pragma unroll
    for (int p = 0; p < 8; ++p) sum += extra[p];
    if (sum == 12345.678f) sink[0] = sum;
}


// Wave-specialised probe: 512-thread workgroups, waves 0-3 run the mode-3 skeleton (MFMAs + fragment reads + two barriers
// among THEMSELVES are replaced by none: no cross-role sync), waves 4-7 run NV VALU instructions per iteration and no MFMA.
// Does VALU work issued by ANOTHER wave of the same SIMD cost the matrix wave its MFMA issue slots?
template <int NV>
__global__ __launch_bounds__(512, 2) void specialised_kernel(float *sink, int iters, float seed)
{
    __shared__ __attribute__((aligned(16))) float lds[12288];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    for (int i = tid; i < 12288; i += 512) lds[i] = seed + i * 1e-6f;
    __syncthreads();
    if (wave < 4) {
        f32x16 acc[8];
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
        float2 fa[2][8], fb[2][8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            fa[0][p] = make_float2(seed + p, seed - p); fb[0][p] = make_float2(seed * 0.5f + p, seed * 0.25f - p);
            fa[1][p] = fa[0][p]; fb[1][p] = fb[0][p];
        }
        const float *base = lds + lane * 2;
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int p = 0; p < 8; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][p].x, fb[s][p].x, acc[p], 0, 0, 0);
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    fa[s ^ 1][p] = *reinterpret_cast<const float2 *>(base + p * 128 + s * 2048);
                    fb[s ^ 1][p] = *reinterpret_cast<const float2 *>(base + 1024 + p * 128 + s * 2048);
                }
#pragma unroll
                for (int p = 0; p < 8; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][p].y, fb[s][p].y, acc[p], 0, 0, 0);
#pragma unroll
                for (int i_ = 0; i_ < 8; ++i_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float sum = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += acc[p][e];
        if (sum == 12345.678f) sink[0] = sum;
    } else {
        float x[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) x[p] = seed * 3.f + p + lane;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int v = 0; v < NV / 8; ++v) {
#pragma unroll
                for (int p = 0; p < 8; ++p) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[p]) : "v"(x[(p + 3) & 7]));
            }
        }
        float sum = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p) sum += x[p];
        if (sum == 12345.678f) sink[1] = sum;
    }
}

template <int NV>
static void run_spec(int n_cu, float *d_sink)
{
    const int iters = 2048;
    for (int wgs : {1, 2}) {
        const int blocks = n_cu * wgs;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(specialised_kernel<NV>, dim3(blocks), dim3(512), 0, 0, d_sink, 64, 1.f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(specialised_kernel<NV>, dim3(blocks), dim3(512), 0, 0, d_sink, iters, 1.f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)blocks * 4 * iters * 16 * (2.0 * 32 * 32 * 2);
        const double tf = flops / (ms * 1e-3) / 1e12;
        printf("specialised: 4 matrix waves + 4 waves of %3d v_add_f32 per iteration          %d workgroup(s) per CU  %8.3f ms  %7.1f TFLOP/s  %.3f of 157.3\n",
               NV, wgs, ms, tf, tf / 157.3);
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
}

// Sustained rate of the bare MFMA loop over ~1 s of back-to-back launches, with constant operands (every lane the same
// small integers) and with pseudo-random operands in [-1, 1) per lane and register: the clock the chip sustains depends on
// the switching activity of the matrix pipe, so the second number -- not 157.3 -- is the ceiling a convolution on real
// activations can reach.
template <bool RANDOM>
__global__ __launch_bounds__(256, 2) void sustained_kernel(float *sink, int iters)
{
    const int tid = threadIdx.x;
    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
    float2 fa[2][8], fb[2][8];
    unsigned h = (blockIdx.x * 256u + tid) * 2654435761u + 12345u;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            float r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                h = h * 1664525u + 1013904223u;
                r[k] = RANDOM ? (float)(int)(h >> 8) * (1.f / 8388608.f) - 1.f : (float)(p + k + s);
            }
            fa[s][p] = make_float2(r[0], r[1]);
            fb[s][p] = make_float2(r[2], r[3]);
        }
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int p = 0; p < 8; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][p].x, fb[s][p].x, acc[p], 0, 0, 0);
#pragma unroll
            for (int p = 0; p < 8; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][p].y, fb[s][p].y, acc[p], 0, 0, 0);
        }
    }
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) t += acc[p][e];
    if (t == 123.456f) sink[0] = t;
}

template <bool RANDOM>
static void run_sustained(int n_cu, float *d_sink)
{
    const int iters = 100000, launches = 14;
    const int blocks = n_cu * 2;
    hipEvent_t ev[launches + 1];
    for (auto &e : ev) CHECK(hipEventCreate(&e));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(ev[0]));
    for (int l = 0; l < launches; ++l) {
        hipLaunchKernelGGL(sustained_kernel<RANDOM>, dim3(blocks), dim3(256), 0, 0, d_sink, iters);
        CHECK(hipEventRecord(ev[l + 1]));
    }
    CHECK(hipEventSynchronize(ev[launches]));
    const double flops = (double)blocks * 4 * iters * 16 * (2.0 * 32 * 32 * 2);
    printf("sustained, %s operands: 2 workgroups per CU, %d launches of %d iterations back to back; TFLOP/s per launch:",
           RANDOM ? "pseudo-random [-1,1)" : "constant small-integer", launches, iters);
    double last4 = 0.;
    for (int l = 0; l < launches; ++l) {
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, ev[l], ev[l + 1]));
        const double tf = flops / (ms * 1e-3) / 1e12;
        printf(" %.1f", tf);
        if (l >= launches - 4) last4 += tf / 4.;
    }
    printf("\n  -> last four: %.1f TFLOP/s = %.3f of 157.3 (%.2f GHz equivalent of the 2.4 GHz peak clock)\n", last4, last4 / 157.3, last4 / 157.3 * 2.4);
    for (auto &e : ev) CHECK(hipEventDestroy(e));
}

static const float4 *g_src = nullptr;
static size_t g_mask = 0;

template <int MODE>
static void run(int n_cu, float *d_sink, const char *what)
{
    const int iters = 2048;
    for (int wgs : {1, 2}) {
        const int blocks = n_cu * wgs;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(skeleton_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d_sink, 64, 1.f, g_src, g_mask);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(skeleton_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d_sink, iters, 1.f, g_src, g_mask);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)blocks * 4 /*waves*/ * iters * 16 /*mfma*/ * (2.0 * 32 * 32 * 2);
        const double tf = flops / (ms * 1e-3) / 1e12;
        printf("mode %d  %-64s  %d workgroup(s) per CU  %8.3f ms  %7.1f TFLOP/s  %.3f of 157.3\n", MODE, what, wgs, ms, tf, tf / 157.3);
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("# %d CUs; 256-thread workgroups (one wave per SIMD each), 8 x 32x32 accumulators per wave, 16 v_mfma_f32_32x32x2_f32 per iteration\n", n_cu);
    float *d_sink;
    CHECK(hipMalloc(&d_sink, sizeof(float)));
    {
        float4 *g;
        const size_t n = (size_t)1 << 22;          // 4 M float4 = 64 MB
        CHECK(hipMalloc(&g, n * sizeof(float4)));
        CHECK(hipMemset(g, 0, n * sizeof(float4)));
        g_src = g; g_mask = n - 1;
    }
    if (getenv("MFMA_SUSTAINED_ONLY") == nullptr || atoi(getenv("MFMA_SUSTAINED_ONLY")) >= 0) {
        run_sustained<false>(n_cu, d_sink);
        run_sustained<true>(n_cu, d_sink);
        run_sustained<false>(n_cu, d_sink);
        run_sustained<true>(n_cu, d_sink);
        if (getenv("MFMA_SUSTAINED_ONLY") && atoi(getenv("MFMA_SUSTAINED_ONLY")) > 0) return 0;
    }
    run<0>(n_cu, d_sink, "16 MFMAs per iteration, nothing else");
    run<1>(n_cu, d_sink, "+ 16 ds_read_b64 of the next operands");
    run<2>(n_cu, d_sink, "+ one barrier per iteration");
    run<3>(n_cu, d_sink, "+ two barriers per iteration (the kernel's structure)");
    run<4>(n_cu, d_sink, "+ 10 ds_write_b64 per iteration (staging stores, no transform)");
    run<5>(n_cu, d_sink, "+ 20 ds_write_b32 (same bytes)");
    run<6>(n_cu, d_sink, "+ 5 ds_write_b128 (same bytes)");
    run<7>(n_cu, d_sink, "+ 20 ds_write_addtid_b32 (same bytes)");
    run<8>(n_cu, d_sink, "+ 10 ds_write_b64 in one block behind the first 8 MFMAs");
    run<9>(n_cu, d_sink, "+ 10 ds_write_b64, data from registers no MFMA reads");
    run<10>(n_cu, d_sink, "+ 72+ VALU adds per iteration in the first half (9-11 per MFMA)");
    run<11>(n_cu, d_sink, "+ 6 x 16-byte global loads per iteration (64 MB buffer)");
    run<12>(n_cu, d_sink, "mode 11 with the VALU spread over all 16 MFMAs");
    run_spec<0>(n_cu, d_sink);
    run_spec<72>(n_cu, d_sink);
    run_spec<144>(n_cu, d_sink);
    run_spec<288>(n_cu, d_sink);
    return 0;
}
