// valu_issue_bench.hip -- how many VALU wave-instructions per clock does one gfx950 SIMD issue?
//
// The roofs bench.py quotes the XNOR convolutions (v_xnor_b32 + v_bcnt_u32_b32 with accumulate) and the first-layer
// kernel K1f (v_fma_f32) against are instruction-issue roofs: T instructions x lanes x work per lane / cycles.  The
// microarchitecture guide says a wave64 VALU instruction occupies a SIMD for 2 cycles (SIMD-32); the round-3 kernels
// all landed near 5 cycles per instruction, so the figure is MEASURED here instead of assumed:
//
//   * every CU gets `wps` workgroups of 256 threads (4 waves, one per SIMD) -> wps waves per SIMD, wps = 1, 2, 4, 8;
//   * each wave runs ITERS iterations of an unrolled body of 64 instructions of one kind: one dependent chain, or 8
//     independent chains (what the kernels' inner loops look like), operands in VGPRs or with an SGPR source (how
//     conv_xnor feeds the weight words);
//   * cycles per wave from s_memtime (shader clock), wall time from HIP events.
//
// Output: one line per (instruction mix, wps): wave-instructions per SIMD-clock by the in-kernel clock (median over
// the waves), the same figure from wall time at 2.4 GHz, and the effective clock = cycles / wall time.
//
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o tools/valu_issue_bench tools/valu_issue_bench.hip && tools/valu_issue_bench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

enum Mix {
    FMA_DEP = 0,       // v_fma_f32, one dependent chain
    FMA_IND8,          // v_fma_f32, 8 independent chains
    PKFMA_IND8,        // v_pk_fma_f32, 8 independent chains (2 fma per lane and instruction)
    XNOR_BCNT_DEP,     // v_xnor_b32 + v_bcnt_u32_b32 (accumulating), one chain
    XNOR_BCNT_IND8,    // ... 8 independent accumulators, weights in VGPRs
    XNOR_BCNT_IND8_S,  // ... 8 independent accumulators, weights as SGPR operands (conv_xnor's form)
    BCNT_IND8,         // v_bcnt_u32_b32 only
    XNOR_IND8,         // v_xnor_b32 only
    ADD_IND8,          // v_add_u32 only (plain integer VALU)
    XOR_IND8,          // v_xor_b32 only: is the plain logic op full rate where v_xnor_b32 is not?
    XOR_BCNT_IND8_S,   // v_xor_b32 + v_bcnt_u32_b32 (acc), SGPR weights: the mismatch-count form of the XNOR convolution
    AND_IND8,          // v_and_b32 only
    XOR8_BCNT8_S,      // 8 x v_xor_b32 then 8 x v_bcnt_u32_b32 (acc): the same work as XOR_BCNT_IND8_S, batched by instruction kind
    XNOR8_BCNT8_S,     // 8 x v_xnor_b32 then 8 x v_bcnt_u32_b32 (acc)
    XOR32_BCNT32_S,    // 32 x v_xor_b32 then 32 x v_bcnt_u32_b32 (acc): does a longer run of one kind pay?
    N_MIX
};
static const char *mix_name[N_MIX] = {
    "v_fma_f32 dependent chain", "v_fma_f32 8 independent chains", "v_pk_fma_f32 8 independent chains",
    "v_xnor_b32+v_bcnt_u32_b32(acc) dependent", "v_xnor_b32+v_bcnt_u32_b32(acc) 8 independent, VGPR weights",
    "v_xnor_b32+v_bcnt_u32_b32(acc) 8 independent, SGPR weights", "v_bcnt_u32_b32(acc) 8 independent",
    "v_xnor_b32 8 independent", "v_add_u32 8 independent", "v_xor_b32 8 independent",
    "v_xor_b32+v_bcnt_u32_b32(acc) 8 independent, SGPR weights", "v_and_b32 8 independent",
    "8 x v_xor_b32 then 8 x v_bcnt_u32_b32(acc), SGPR weights", "8 x v_xnor_b32 then 8 x v_bcnt_u32_b32(acc), SGPR weights",
    "32 x v_xor_b32 then 32 x v_bcnt_u32_b32(acc), SGPR weights"};
// VALU instructions per unrolled body (all bodies are 64 instructions)
constexpr int BODY = 64;

// 8 repetitions of an 8-instruction group = 64 instructions
#define REP8(S) S S S S S S S S
#define REP4(S) S S S S

template <int MIX>
__global__ __launch_bounds__(256) void bench_kernel(unsigned long long *cycles, float *sink, int iters, float seed_f, unsigned seed_u)
{
    const int lane_global = blockIdx.x * 256 + threadIdx.x;
    float a0 = seed_f + lane_global * 1e-7f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,
          a6 = a0 + 6.f, a7 = a0 + 7.f;
    float b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;       // upper halves of the packed pairs
    const float m = 0.999f + seed_f * 1e-9f, c = 1e-3f;
    unsigned x = seed_u * 2654435761u + lane_global;
    unsigned w0 = seed_u ^ 0x1234567u, w1 = w0 * 3u, w2 = w0 * 5u, w3 = w0 * 7u, w4 = w0 * 9u, w5 = w0 * 11u, w6 = w0 * 13u,
             w7 = w0 * 15u;
    unsigned n0 = 0, n1 = 0, n2 = 0, n3 = 0, n4 = 0, n5 = 0, n6 = 0, n7 = 0;
    unsigned t0, t1, t2, t3, t4, t5, t6, t7;
    // wave-uniform copies for the SGPR-operand form
    const unsigned s0 = __builtin_amdgcn_readfirstlane(w0), s1 = __builtin_amdgcn_readfirstlane(w1),
                   s2 = __builtin_amdgcn_readfirstlane(w2), s3 = __builtin_amdgcn_readfirstlane(w3),
                   s4 = __builtin_amdgcn_readfirstlane(w4), s5 = __builtin_amdgcn_readfirstlane(w5),
                   s6 = __builtin_amdgcn_readfirstlane(w6), s7 = __builtin_amdgcn_readfirstlane(w7);

    __syncthreads();
    const unsigned long long t_begin = __builtin_readcyclecounter();        // s_memtime
    for (int it = 0; it < iters; ++it) {
        if (MIX == FMA_DEP) {
            asm volatile(REP8(REP8("v_fma_f32 %0, %0, %1, %2\n")) : "+v"(a0) : "v"(m), "v"(c));
        } else if (MIX == FMA_IND8) {
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n"
                              "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                              "v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(m), "v"(c));
        } else if (MIX == PKFMA_IND8) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, b0}, p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3}, p4 = {a4, b4}, p5 = {a5, b5}, p6 = {a6, b6}, p7 = {a7, b7};
            const f2 mm = {m, m}, cc = {c, c};
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n"
                              "v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n"
                              "v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                         : "v"(mm), "v"(cc));
            a0 = p0.x; b0 = p0.y; a1 = p1.x; b1 = p1.y; a2 = p2.x; b2 = p2.y; a3 = p3.x; b3 = p3.y;
            a4 = p4.x; b4 = p4.y; a5 = p5.x; b5 = p5.y; a6 = p6.x; b6 = p6.y; a7 = p7.x; b7 = p7.y;
        } else if (MIX == XNOR_BCNT_DEP) {
            asm volatile(REP8(REP8("v_xnor_b32 %1, %2, %0\n v_bcnt_u32_b32 %0, %1, %0\n"))      // 128 instr: counted below
                         : "+v"(n0), "=&v"(t0) : "v"(w0));
        } else if (MIX == XNOR_BCNT_IND8) {
            asm volatile(REP4("v_xnor_b32 %8, %16, %24\n v_bcnt_u32_b32 %0, %8, %0\n v_xnor_b32 %9, %17, %24\n v_bcnt_u32_b32 %1, %9, %1\n"
                              "v_xnor_b32 %10, %18, %24\n v_bcnt_u32_b32 %2, %10, %2\n v_xnor_b32 %11, %19, %24\n v_bcnt_u32_b32 %3, %11, %3\n"
                              "v_xnor_b32 %12, %20, %24\n v_bcnt_u32_b32 %4, %12, %4\n v_xnor_b32 %13, %21, %24\n v_bcnt_u32_b32 %5, %13, %5\n"
                              "v_xnor_b32 %14, %22, %24\n v_bcnt_u32_b32 %6, %14, %6\n v_xnor_b32 %15, %23, %24\n v_bcnt_u32_b32 %7, %15, %7\n")
                         : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(n4), "+v"(n5), "+v"(n6), "+v"(n7),
                           "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                         : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(w5), "v"(w6), "v"(w7), "v"(x));
        } else if (MIX == XNOR_BCNT_IND8_S) {
            asm volatile(REP4("v_xnor_b32 %8, %16, %24\n v_bcnt_u32_b32 %0, %8, %0\n v_xnor_b32 %9, %17, %24\n v_bcnt_u32_b32 %1, %9, %1\n"
                              "v_xnor_b32 %10, %18, %24\n v_bcnt_u32_b32 %2, %10, %2\n v_xnor_b32 %11, %19, %24\n v_bcnt_u32_b32 %3, %11, %3\n"
                              "v_xnor_b32 %12, %20, %24\n v_bcnt_u32_b32 %4, %12, %4\n v_xnor_b32 %13, %21, %24\n v_bcnt_u32_b32 %5, %13, %5\n"
                              "v_xnor_b32 %14, %22, %24\n v_bcnt_u32_b32 %6, %14, %6\n v_xnor_b32 %15, %23, %24\n v_bcnt_u32_b32 %7, %15, %7\n")
                         : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(n4), "+v"(n5), "+v"(n6), "+v"(n7),
                           "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                         : "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(s6), "s"(s7), "v"(x));
        } else if (MIX == BCNT_IND8) {
            asm volatile(REP8("v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %8, %1\n v_bcnt_u32_b32 %2, %8, %2\n v_bcnt_u32_b32 %3, %8, %3\n"
                              "v_bcnt_u32_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %8, %5\n v_bcnt_u32_b32 %6, %8, %6\n v_bcnt_u32_b32 %7, %8, %7\n")
                         : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(n4), "+v"(n5), "+v"(n6), "+v"(n7) : "v"(x));
        } else if (MIX == XNOR_IND8) {
            asm volatile(REP8("v_xnor_b32 %0, %8, %0\n v_xnor_b32 %1, %8, %1\n v_xnor_b32 %2, %8, %2\n v_xnor_b32 %3, %8, %3\n"
                              "v_xnor_b32 %4, %8, %4\n v_xnor_b32 %5, %8, %5\n v_xnor_b32 %6, %8, %6\n v_xnor_b32 %7, %8, %7\n")
                         : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(n4), "+v"(n5), "+v"(n6), "+v"(n7) : "v"(x));
        } else if (MIX == XOR8_BCNT8_S || MIX == XNOR8_BCNT8_S) {
#define YL_B8(OP)                                                                                                              \
            asm volatile(REP4(OP " %8, %16, %24\n " OP " %9, %17, %24\n " OP " %10, %18, %24\n " OP " %11, %19, %24\n "          \
                              OP " %12, %20, %24\n " OP " %13, %21, %24\n " OP " %14, %22, %24\n " OP " %15, %23, %24\n"         \
                              "v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %9, %1\n v_bcnt_u32_b32 %2, %10, %2\n v_bcnt_u32_b32 %3, %11, %3\n" \
                              "v_bcnt_u32_b32 %4, %12, %4\n v_bcnt_u32_b32 %5, %13, %5\n v_bcnt_u32_b32 %6, %14, %6\n v_bcnt_u32_b32 %7, %15, %7\n") \
                         : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(n4), "+v"(n5), "+v"(n6), "+v"(n7),                      \
                           "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)               \
                         : "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(s6), "s"(s7), "v"(x));
            if (MIX == XOR8_BCNT8_S) { YL_B8("v_xor_b32") } else { YL_B8("v_xnor_b32") }
#undef YL_B8
        } else if (MIX == XOR32_BCNT32_S) {
            unsigned u0, u1, u2, u3, u4, u5, u6, u7, u8, u9, u10, u11, u12, u13, u14, u15, u16, u17, u18, u19, u20, u21, u22, u23;
            // 32 temporaries: t0-t7 + u0-u23; xors of x with 8 SGPR weights (each weight used 4 times with shifted copies)
            asm volatile(
                "v_xor_b32 %8, %40, %48\n v_xor_b32 %9, %41, %48\n v_xor_b32 %10, %42, %48\n v_xor_b32 %11, %43, %48\n"
                "v_xor_b32 %12, %44, %48\n v_xor_b32 %13, %45, %48\n v_xor_b32 %14, %46, %48\n v_xor_b32 %15, %47, %48\n"
                "v_xor_b32 %16, %40, %0\n v_xor_b32 %17, %41, %1\n v_xor_b32 %18, %42, %2\n v_xor_b32 %19, %43, %3\n"
                "v_xor_b32 %20, %44, %4\n v_xor_b32 %21, %45, %5\n v_xor_b32 %22, %46, %6\n v_xor_b32 %23, %47, %7\n"
                "v_xor_b32 %24, %41, %0\n v_xor_b32 %25, %42, %1\n v_xor_b32 %26, %43, %2\n v_xor_b32 %27, %44, %3\n"
                "v_xor_b32 %28, %45, %4\n v_xor_b32 %29, %46, %5\n v_xor_b32 %30, %47, %6\n v_xor_b32 %31, %40, %7\n"
                "v_xor_b32 %32, %42, %0\n v_xor_b32 %33, %43, %1\n v_xor_b32 %34, %44, %2\n v_xor_b32 %35, %45, %3\n"
                "v_xor_b32 %36, %46, %4\n v_xor_b32 %37, %47, %5\n v_xor_b32 %38, %40, %6\n v_xor_b32 %39, %41, %7\n"
                "v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %9, %1\n v_bcnt_u32_b32 %2, %10, %2\n v_bcnt_u32_b32 %3, %11, %3\n"
                "v_bcnt_u32_b32 %4, %12, %4\n v_bcnt_u32_b32 %5, %13, %5\n v_bcnt_u32_b32 %6, %14, %6\n v_bcnt_u32_b32 %7, %15, %7\n"
                "v_bcnt_u32_b32 %0, %16, %0\n v_bcnt_u32_b32 %1, %17, %1\n v_bcnt_u32_b32 %2, %18, %2\n v_bcnt_u32_b32 %3, %19, %3\n"
                "v_bcnt_u32_b32 %4, %20, %4\n v_bcnt_u32_b32 %5, %21, %5\n v_bcnt_u32_b32 %6, %22, %6\n v_bcnt_u32_b32 %7, %23, %7\n"
                "v_bcnt_u32_b32 %0, %24, %0\n v_bcnt_u32_b32 %1, %25, %1\n v_bcnt_u32_b32 %2, %26, %2\n v_bcnt_u32_b32 %3, %27, %3\n"
                "v_bcnt_u32_b32 %4, %28, %4\n v_bcnt_u32_b32 %5, %29, %5\n v_bcnt_u32_b32 %6, %30, %6\n v_bcnt_u32_b32 %7, %31, %7\n"
                "v_bcnt_u32_b32 %0, %32, %0\n v_bcnt_u32_b32 %1, %33, %1\n v_bcnt_u32_b32 %2, %34, %2\n v_bcnt_u32_b32 %3, %35, %3\n"
                "v_bcnt_u32_b32 %4, %36, %4\n v_bcnt_u32_b32 %5, %37, %5\n v_bcnt_u32_b32 %6, %38, %6\n v_bcnt_u32_b32 %7, %39, %7\n"
                : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(n4), "+v"(n5), "+v"(n6), "+v"(n7),
                  "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7),
                  "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4), "=&v"(u5), "=&v"(u6), "=&v"(u7),
                  "=&v"(u8), "=&v"(u9), "=&v"(u10), "=&v"(u11), "=&v"(u12), "=&v"(u13), "=&v"(u14), "=&v"(u15),
                  "=&v"(u16), "=&v"(u17), "=&v"(u18), "=&v"(u19), "=&v"(u20), "=&v"(u21), "=&v"(u22), "=&v"(u23)
                : "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(s6), "s"(s7), "v"(x));
        } else if (MIX == XOR_IND8) {
            asm volatile(REP8("v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n"
                              "v_xor_b32 %4, %8, %4\n v_xor_b32 %5, %8, %5\n v_xor_b32 %6, %8, %6\n v_xor_b32 %7, %8, %7\n")
                         : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(n4), "+v"(n5), "+v"(n6), "+v"(n7) : "v"(x));
        } else if (MIX == AND_IND8) {
            asm volatile(REP8("v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n"
                              "v_and_b32 %4, %8, %4\n v_and_b32 %5, %8, %5\n v_and_b32 %6, %8, %6\n v_and_b32 %7, %8, %7\n")
                         : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(n4), "+v"(n5), "+v"(n6), "+v"(n7) : "v"(x));
        } else if (MIX == XOR_BCNT_IND8_S) {
            asm volatile(REP4("v_xor_b32 %8, %16, %24\n v_bcnt_u32_b32 %0, %8, %0\n v_xor_b32 %9, %17, %24\n v_bcnt_u32_b32 %1, %9, %1\n"
                              "v_xor_b32 %10, %18, %24\n v_bcnt_u32_b32 %2, %10, %2\n v_xor_b32 %11, %19, %24\n v_bcnt_u32_b32 %3, %11, %3\n"
                              "v_xor_b32 %12, %20, %24\n v_bcnt_u32_b32 %4, %12, %4\n v_xor_b32 %13, %21, %24\n v_bcnt_u32_b32 %5, %13, %5\n"
                              "v_xor_b32 %14, %22, %24\n v_bcnt_u32_b32 %6, %14, %6\n v_xor_b32 %15, %23, %24\n v_bcnt_u32_b32 %7, %15, %7\n")
                         : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(n4), "+v"(n5), "+v"(n6), "+v"(n7),
                           "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                         : "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(s6), "s"(s7), "v"(x));
        } else {
            asm volatile(REP8("v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n"
                              "v_add_u32 %4, %8, %4\n v_add_u32 %5, %8, %5\n v_add_u32 %6, %8, %6\n v_add_u32 %7, %8, %7\n")
                         : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(n4), "+v"(n5), "+v"(n6), "+v"(n7) : "v"(x));
        }
    }
    const unsigned long long t_end = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t_end - t_begin;
    const float s = (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7) + (b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7) +
                    (float)(n0 + n1 + n2 + n3 + n4 + n5 + n6 + n7);
    if (s == 1234.5678f) sink[0] = s;           // keeps every chain alive
}

template <int MIX>
static void run(int n_cu, double clock_ghz, unsigned long long *d_cycles, float *d_sink)
{
    // instructions per unrolled body: the dependent xnor+bcnt body holds 64 PAIRS
    const int body = (MIX == XNOR_BCNT_DEP) ? 2 * BODY : BODY;
    const int iters = 4096;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = n_cu * wps;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(bench_kernel<MIX>, dim3(blocks), dim3(256), 0, 0, d_cycles, d_sink, 64, 1.f, 3u);     // warm-up
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(bench_kernel<MIX>, dim3(blocks), dim3(256), 0, 0, d_cycles, d_sink, iters, 1.f, 3u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> cyc((size_t)blocks * 4);
        CHECK(hipMemcpy(cyc.data(), d_cycles, cyc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::sort(cyc.begin(), cyc.end());
        const double med = (double)cyc[cyc.size() / 2];
        const double instr_per_wave = (double)iters * body;
        // s_memtime counts at a fixed 100 MHz on gfx9: convert with the wall time of the launch instead when the
        // counter is obviously not the shader clock
        const double ipc_cycles = wps * instr_per_wave / med;
        const double ipc_wall = wps * instr_per_wave / (ms * 1e-3 * clock_ghz * 1e9);
        printf("%-62s wps %d  %8.3f ms  counter ticks/wave %12.0f  instr/SIMD/tick %7.4f  instr/SIMD/clk@%.1fGHz(wall) %6.4f  -> %5.2f clk per wave-instr\n",
               mix_name[MIX], wps, ms, med, ipc_cycles, clock_ghz, ipc_wall, 1.0 / ipc_wall);
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const double clock_ghz = 2.4;
    printf("# %s, %d CUs, clockRate %.0f MHz; one 256-thread workgroup = one wave per SIMD, wps workgroups per CU\n", prop.name, n_cu,
           prop.clockRate / 1000.0);
    printf("# `clk per wave-instr` = SIMD cycles (at %.1f GHz, from wall time) per issued wave64 VALU instruction, all waves of the SIMD together\n",
           clock_ghz);
    unsigned long long *d_cycles;
    float *d_sink;
    CHECK(hipMalloc(&d_cycles, sizeof(unsigned long long) * (size_t)n_cu * 8 * 4));
    CHECK(hipMalloc(&d_sink, sizeof(float)));
    run<FMA_DEP>(n_cu, clock_ghz, d_cycles, d_sink);
    run<FMA_IND8>(n_cu, clock_ghz, d_cycles, d_sink);
    run<PKFMA_IND8>(n_cu, clock_ghz, d_cycles, d_sink);
    run<XNOR_BCNT_DEP>(n_cu, clock_ghz, d_cycles, d_sink);
    run<XNOR_BCNT_IND8>(n_cu, clock_ghz, d_cycles, d_sink);
    run<XNOR_BCNT_IND8_S>(n_cu, clock_ghz, d_cycles, d_sink);
    run<BCNT_IND8>(n_cu, clock_ghz, d_cycles, d_sink);
    run<XNOR_IND8>(n_cu, clock_ghz, d_cycles, d_sink);
    run<ADD_IND8>(n_cu, clock_ghz, d_cycles, d_sink);
    run<XOR_IND8>(n_cu, clock_ghz, d_cycles, d_sink);
    run<XOR_BCNT_IND8_S>(n_cu, clock_ghz, d_cycles, d_sink);
    run<AND_IND8>(n_cu, clock_ghz, d_cycles, d_sink);
    run<XOR8_BCNT8_S>(n_cu, clock_ghz, d_cycles, d_sink);
    run<XNOR8_BCNT8_S>(n_cu, clock_ghz, d_cycles, d_sink);
    run<XOR32_BCNT32_S>(n_cu, clock_ghz, d_cycles, d_sink);
    return 0;
}
