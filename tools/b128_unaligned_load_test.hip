// micro test: raw buffer_load b128 at dword-aligned (not 16-byte aligned) offsets, and range checking
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int i32x4 __attribute__((__vector_size__(16)));
__global__ void k(const float *in, float *out, int nrec_bytes)
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, nrec_bytes, 0x00020000);
    const int t = threadIdx.x;
    int off = t * 4;                    // lane t starts at float t: unaligned for t % 4 != 0
    if (t == 60) off = -1;
    if (t == 61) off = nrec_bytes - 8;  // last two floats valid, two beyond
    if (t == 62) off = nrec_bytes;      // fully beyond
    const i32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    out[t * 4 + 0] = __uint_as_float(q[0]);
    out[t * 4 + 1] = __uint_as_float(q[1]);
    out[t * 4 + 2] = __uint_as_float(q[2]);
    out[t * 4 + 3] = __uint_as_float(q[3]);
}
int main()
{
    const int N = 1024;
    float h[N]; for (int i = 0; i < N; ++i) h[i] = (float)i;
    float *din, *dout; hipMalloc(&din, N * 4); hipMalloc(&dout, 256 * 4);
    hipMemcpy(din, h, N * 4, hipMemcpyHostToDevice);
    const int nrec = 100 * 4;
    k<<<1, 64>>>(din, dout, nrec);
    float o[256]; hipMemcpy(o, dout, 256 * 4, hipMemcpyDeviceToHost);
    for (int t : {0, 1, 2, 3, 5, 60, 61, 62}) printf("lane %d: %g %g %g %g\n", t, o[t*4], o[t*4+1], o[t*4+2], o[t*4+3]);
    return 0;
}
