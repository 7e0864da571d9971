// How fast does ONE wave issue FP64 instructions on gfx950?  Dependent chain vs independent streams, plain v_fma_f64 vs
// v_fmac_f64_dpp row_newbcast, v_rsq_f64, v_readlane.  Cycles from s_memtime (100 MHz -> scaled by the shader clock is
// not needed: we use s_memrealtime? no: __builtin_readcyclecounter = s_memtime counts shader clocks on gfx9).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_IT 256
__global__ void k(double* out, unsigned long long* cyc, double seed) {
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + i + threadIdx.x * 1e-3;
    double b = 1.0000001, c = 1e-9;
    unsigned long long t0, t1;
    // 1) dependent v_fma_f64 chain
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N_IT; it++) {
        asm volatile("v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\t"
                     "v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2"
                     : "+v"(a[0]) : "v"(b), "v"(c));
    }
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    // 2) 8 independent v_fma_f64 streams
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N_IT; it++) {
        asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));
    }
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[1] = t1 - t0;
    // 3) 8 independent v_fmac_f64_dpp row_newbcast
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N_IT; it++) {
        asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c), "v"(b));
    }
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[2] = t1 - t0;
    // 4) dependent v_fmac_f64_dpp chain (8 per iteration)
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N_IT; it++) {
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %0, %1, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xf"
                     : "+v"(a[0]) : "v"(c), "v"(b));
    }
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[3] = t1 - t0;
    // 5) dependent: fma whose result feeds the DPP source of the next fmac (with the s_nop 1 the hazard needs)
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N_IT; it++) {
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fmac_f64_dpp %1, %0, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                     "v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fmac_f64_dpp %1, %0, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                     : "+v"(a[0]), "+v"(a[1]) : "v"(b));
    }
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[4] = t1 - t0;
    // 6) dependent v_rsq_f64 chain (4 per iteration)
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N_IT; it++) {
        asm volatile("v_rsq_f64 %0, %0\n\tv_rsq_f64 %0, %0\n\tv_rsq_f64 %0, %0\n\tv_rsq_f64 %0, %0" : "+v"(a[2]));
    }
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[5] = t1 - t0;
    // 7) v_readlane pair + dependent VALU use (SGPR operand), 4 per iteration
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N_IT; it++) {
        asm volatile("v_readlane_b32 s20, %0, 3\n\tv_readlane_b32 s21, %1, 3\n\tv_fma_f64 %2, s[20:21], %2, %2\n\t"
                     "v_readlane_b32 s20, %0, 5\n\tv_readlane_b32 s21, %1, 5\n\tv_fma_f64 %2, s[20:21], %2, %2\n\t"
                     "v_readlane_b32 s20, %0, 7\n\tv_readlane_b32 s21, %1, 7\n\tv_fma_f64 %2, s[20:21], %2, %2\n\t"
                     "v_readlane_b32 s20, %0, 9\n\tv_readlane_b32 s21, %1, 9\n\tv_fma_f64 %2, s[20:21], %2, %2"
                     :: "v"(__double2loint(a[3])), "v"(__double2hiint(a[3])), "v"(a[4]) : "s20", "s21");
    }
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[6] = t1 - t0;
    // 8) 2 independent chains interleaved (does one dependent chain leave room for a second one?)
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N_IT; it++) {
        asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3\n\tv_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3\n\t"
                     "v_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3\n\tv_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3"
                     : "+v"(a[0]), "+v"(a[1]) : "v"(b), "v"(c));
    }
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[7] = t1 - t0;
    // 9) dependent MFMA f64 16x16x4 chain (4 per iteration) and 10) 4 independent
    typedef double v4d __attribute__((ext_vector_type(4)));
    v4d m0 = {a[0], a[1], a[2], a[3]}, m1 = m0, m2 = m0, m3 = m0;
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N_IT; it++) {
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0"
                     : "+v"(m0) : "v"(b), "v"(c));
    }
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[8] = t1 - t0;
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N_IT; it++) {
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\n\tv_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n\tv_mfma_f64_16x16x4_f64 %2, %4, %5, %2\n\tv_mfma_f64_16x16x4_f64 %3, %4, %5, %3"
                     : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(b), "v"(c));
    }
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[9] = t1 - t0;
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    s += m0[0] + m1[1] + m2[2] + m3[3];
    out[threadIdx.x] = s;
}
int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 1.5);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 1.5);
    unsigned long long h[16];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"dependent v_fma_f64", "8 independent v_fma_f64", "8 independent v_fmac_f64_dpp", "dependent v_fmac_f64_dpp",
                           "fmac_dpp -> DPP source of the next (s_nop 1)", "dependent v_rsq_f64", "readlane x2 + dependent fma (SGPR)",
                           "2 interleaved dependent chains", "dependent mfma_f64_16x16x4", "4 independent mfma_f64_16x16x4"};
    const int per_it[] = {8, 8, 8, 8, 4, 4, 4, 8, 4, 4};
    for (int i = 0; i < 10; i++) printf("%-48s %7.1f cycles per instruction (group)\n", names[i], (double)h[i] / (N_IT * per_it[i]));
    return 0;
}
