// Where does the dispatcher put the workgroups of a launch?  One workgroup per CU (100 KB of LDS), 1024 workgroups: prints, per
// workgroup, XCC / SE / CU / SIMD / wave slot of its first wave and its start and end time (100 MHz ticks since the first start).
//   hipcc --offload-arch=gfx950 -O3 tools/hwid_probe.hip -o build/hwid_probe && build/hwid_probe [ticks of work per workgroup]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(512) void probe(unsigned long long *o, int ticks)
{
    __shared__ char big[100 * 1024];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    big[threadIdx.x] = (char)hw;
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) {
        o[4 * blockIdx.x] = hw; o[4 * blockIdx.x + 1] = xcc; o[4 * blockIdx.x + 2] = t0; o[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime() + big[1];
    }
}
int main(int argc, char **argv)
{
    const int n = 1024, ticks = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned long long *d;
    hipMalloc(&d, n * 32);
    hipLaunchKernelGGL(probe, dim3(n), dim3(512), 0, 0, d, ticks);
    hipLaunchKernelGGL(probe, dim3(n), dim3(512), 0, 0, d, ticks);
    std::vector<unsigned long long> h(n * 4);
    hipMemcpy(h.data(), d, n * 32, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull;
    for (int i = 0; i < n; ++i) if (h[4 * i + 2] < tmin) tmin = h[4 * i + 2];
    for (int i = 0; i < n; ++i) {
        const unsigned hw = (unsigned)h[4 * i], xcc = (unsigned)h[4 * i + 1];
        printf("wg %4d xcc %u se %u sh %u cu %2u simd %u slot %u  start %6llu end %6llu\n", i, xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15,
               (hw >> 4) & 3, hw & 15, h[4 * i + 2] - tmin, h[4 * i + 3] - tmin);
    }
    return 0;
}
