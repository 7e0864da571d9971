// Shader clock actually delivered in two regimes: a chain of short one-workgroup launches (the shape of the Cholesky panel chain)
// and a launch that keeps every CU's matrix pipe busy.  clock64() counts shader cycles, wall_clock64() a constant-rate timer.
// hipcc --offload-arch=gfx950 -O3 -o tools/r04/sclk_probe tools/r04/sclk_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void probe(long long* out, int iters, int slot) {
    const long long c0 = clock64(), w0 = wall_clock64();
    v4d acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0}, acc4 = {0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; i++) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
        acc4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc4, 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[2 * slot] = c1 - c0; out[2 * slot + 1] = w1 - w0; }
    if (acc[0] + acc2[0] + acc3[0] + acc4[0] == 12345.678) out[0] = 0;
}
// how many FP64 MFMAs per cycle a SIMD delivers against waves per SIMD and independent accumulators per wave
template <int NACC>
__global__ void mfma_rate(long long* out, int iters, int slot) {
    v4d acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = (v4d){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    const long long c0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    }
    const long long c1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i][0];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[slot] = c1 - c0;
    if (s == 12345.678) out[0] = 0;
}
template <int NACC>
static void rate(long long* d, int threads, int wgs = 1, int total = 4096) {
    const int iters = total / NACC;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_rate<NACC>, dim3(wgs), dim3(threads), 0, 0, d, iters, 0);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_rate<NACC>, dim3(wgs), dim3(threads), 0, 0, d, iters, 1);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double tf = (double)wgs * (threads / 64) * iters * NACC * 2048.0 / (ms * 1e-3) / 1e12;
    long long h[2];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // (the cycle count of wave 0 is that of the OLDEST wave of its SIMD, which is served first: only with one wave per SIMD is it
    // the SIMD's rate -- the event time is what counts)
    const double per_wave = (double)h[1] / (iters * NACC);
    printf("  %4d workgroup(s) of %4d threads (%d wave(s) per SIMD), %2d accumulators per wave, %d MFMAs per wave: %.3f ms by events = %.1f TFLOP/s"
           " (oldest wave: %.1f cycles per MFMA)\n", wgs, threads, threads / 256, NACC, iters * NACC, ms, tf, per_wave);
}
int main() {
    int wall_khz = 0;
    (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    int sclk_khz = 0;
    (void)hipDeviceGetAttribute(&sclk_khz, hipDeviceAttributeClockRate, 0);
    printf("wall clock rate %d kHz, nominal shader clock %d kHz\n", wall_khz, sclk_khz);
    long long* d; (void)hipMalloc(&d, 4096 * sizeof(long long));
    std::vector<long long> h(4096);
    auto report = [&](const char* what, int n, double mfmas) {
        (void)hipMemcpy(h.data(), d, 2 * n * sizeof(long long), hipMemcpyDeviceToHost);
        double cs = 0, ws = 0;
        for (int i = n / 2; i < n; i++) { cs += h[2 * i]; ws += h[2 * i + 1]; }
        (void)mfmas;
        printf("%s: %.0f shader cycles per %.2f us => %.0f MHz\n", what, cs / (n - n / 2), ws / (n - n / 2) / wall_khz * 1e3, cs / ws * wall_khz / 1e3);
    };
    // (a) chain of short one-workgroup launches: 64 x 4 MFMAs ~ 16k cycles each
    for (int rep = 0; rep < 3; rep++) {
        for (int i = 0; i < 512; i++) hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, d, 64, i);
        (void)hipDeviceSynchronize();
        report("chain of 1-workgroup launches (256 MFMAs per wave)", 512, 256);
    }
    // (b) chain of launches with 64 workgroups
    for (int i = 0; i < 512; i++) hipLaunchKernelGGL(probe, dim3(64), dim3(256), 0, 0, d, 64, i);
    (void)hipDeviceSynchronize();
    report("chain of 64-workgroup launches", 512, 256);
    // (c) every CU busy for ~ 50 ms per launch
    for (int i = 0; i < 8; i++) hipLaunchKernelGGL(probe, dim3(1024), dim3(256), 0, 0, d, 200000, i);
    (void)hipDeviceSynchronize();
    report("1024 workgroups x 800k MFMAs per wave", 8, 800000);
    // (d) the chain again right after the busy phase
    for (int i = 0; i < 512; i++) hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, d, 64, i);
    (void)hipDeviceSynchronize();
    report("chain of 1-workgroup launches after the busy phase", 512, 256);
    printf("v_mfma_f64_16x16x4_f64 (64 cycles per instruction and SIMD = 78.6 TFLOP/s at 2.4 GHz): waves per SIMD, accumulators per wave\n");
    rate<1>(d, 256); rate<4>(d, 256); rate<16>(d, 256);
    for (int wgs : {32, 128, 256}) { rate<4>(d, 256, wgs, 262144); rate<4>(d, 512, wgs, 262144); rate<4>(d, 1024, wgs, 262144); }
    rate<4>(d, 256, 1024, 262144);
    return 0;
}
