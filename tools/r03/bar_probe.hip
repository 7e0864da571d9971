// Can the host store directly into fine-grained device memory (large BAR), and how fast does a resident kernel see it?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void echo(volatile unsigned long long* flag_dev, volatile unsigned long long* out_host, int n) {
    unsigned long long last = 0;
    for (int it = 0; it < n;) {
        unsigned long long v = __hip_atomic_load((unsigned long long*)flag_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v != last) { last = v; __hip_atomic_store((unsigned long long*)out_host, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); it++; }
    }
}
int main() {
    unsigned long long* dflag = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&dflag, 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s ptr %p\n", hipGetErrorString(e), (void*)dflag);
    if (e != hipSuccess) return 1;
    hipMemset(dflag, 0, 4096);
    unsigned long long* hout = nullptr;
    hipHostMalloc((void**)&hout, 4096, hipHostMallocMapped | hipHostMallocCoherent);
    hout[0] = 0;
    hipPointerAttribute_t at; e = hipPointerGetAttributes(&at, dflag);
    printf("attributes: %s type %d host %p dev %p\n", hipGetErrorString(e), (int)at.type, at.hostPointer, at.devicePointer);
    const int n = 20000;
    hipLaunchKernelGGL(echo, dim3(1), dim3(1), 0, 0, dflag, hout, n);
    volatile unsigned long long* hf = dflag;       // direct CPU store into VRAM: faults if the BAR does not expose it
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 1; i <= n; i++) {
        hf[0] = (unsigned long long)i; __builtin_ia32_sfence();
        while (((volatile unsigned long long*)hout)[0] != (unsigned long long)i) { }
    }
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
    printf("ping-pong host store -> VRAM flag -> kernel -> host memory: %.2f us per round trip\n", us);
    hipDeviceSynchronize();
    // the same with the flag in pinned host memory (what server.hip does today)
    unsigned long long* hflag = nullptr;
    hipHostMalloc((void**)&hflag, 4096, hipHostMallocMapped | hipHostMallocCoherent);
    hflag[0] = 0; hout[0] = 0;
    hipLaunchKernelGGL(echo, dim3(1), dim3(1), 0, 0, hflag, hout, n);
    t0 = std::chrono::steady_clock::now();
    for (int i = 1; i <= n; i++) {
        ((volatile unsigned long long*)hflag)[0] = (unsigned long long)i;
        while (((volatile unsigned long long*)hout)[0] != (unsigned long long)i) { }
    }
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
    printf("ping-pong host store -> HOST flag -> kernel polls over PCIe -> host memory: %.2f us per round trip\n", us);
    hipDeviceSynchronize();
    return 0;
}
