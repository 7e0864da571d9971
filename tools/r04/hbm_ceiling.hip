// What store / copy rate does this MI355X reach, and with which launch shape?  (the library's own fill
// micro-benchmark reads 4.6-4.8 TB/s, /opt/skills/guides/MI355X_MICROARCH.md documents 6.0-6.2 TB/s for plain stores and
// 6.29 TB/s for a float4 copy.)  Sweeps: bytes per lane (4 / 8 / 16), cache policy (default / non-temporal), workgroups per
// CU (grid-stride with 1 ... 32 x 256 workgroups, or one workgroup per contiguous chunk), buffer size (inside and beyond the
// 256 MB Infinity Cache), and the write pattern of the covariance build: 512-byte row segments of 64 x 64 tiles at a row
// stride of 8 N bytes (power of two at N = 4096) against N + 32.
//   hipcc -O3 --offload-arch=gfx950 tools/r04/hbm_ceiling.hip -o /tmp/hbm_ceiling && /tmp/hbm_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <functional>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <typename T, bool NT>
__global__ __launch_bounds__(256) void fill_kernel(T* __restrict__ p, int64_t n, T v) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
    }
}
// one workgroup per contiguous chunk of `per` elements (no grid stride: the hardware dispatcher walks the buffer)
template <typename T, bool NT>
__global__ __launch_bounds__(256) void fill_chunk_kernel(T* __restrict__ p, int64_t n, int per, T v) {
    const int64_t b0 = (int64_t)blockIdx.x * per;
    for (int i = threadIdx.x; i < per; i += 256) {
        const int64_t j = b0 + i;
        if (j < n) { if (NT) __builtin_nontemporal_store(v, p + j); else p[j] = v; }
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const v4f* __restrict__ s, v4f* __restrict__ d, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        v4f v = NT ? __builtin_nontemporal_load(s + i) : s[i];
        if (NT) __builtin_nontemporal_store(v, d + i); else d[i] = v;
    }
}
// the covariance build's pattern: workgroup = 64 x 64 tile (bi >= bj) of an N x N matrix of doubles with leading dimension
// ld; every wave stores 16-byte pieces so that 8 lanes cover 128 contiguous bytes of a row (as kernel_train_q_kernel), both
// images of an off-diagonal tile; no arithmetic
template <bool NT>
__global__ __launch_bounds__(256) void tile_store_kernel(double* __restrict__ K, int64_t ld, int nb) {
    int t = blockIdx.x;
    int bi = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((int64_t)(bi + 1) * (bi + 2) / 2 <= t) bi++;
    while ((int64_t)bi * (bi + 1) / 2 > t) bi--;
    const int bj = t - bi * (bi + 1) / 2;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, lx = lane & 7, ly = lane >> 3;
    const int r0 = 32 * (w >> 1) + 2 * ly, c0 = 32 * (w & 1) + 2 * lx;
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d v = {1.0, 2.0};
    for (int img = 0; img < (bi == bj ? 1 : 2); img++) {
        const int64_t rb = (int64_t)(img ? bj : bi) * 64, cb = (int64_t)(img ? bi : bj) * 64;
#pragma unroll
        for (int ap = 0; ap < 2; ap++)
#pragma unroll
            for (int a2 = 0; a2 < 2; a2++) {
                v2d* p = reinterpret_cast<v2d*>(K + (rb + r0 + 16 * ap + a2) * ld + cb + c0);
                if (NT) { __builtin_nontemporal_store(v, p); __builtin_nontemporal_store(v, p + 8); } else { p[0] = v; p[8] = v; }
            }
    }
}

static double time_ms(hipStream_t st, int reps, const std::function<void()>& f) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f(); f();
    hipEventRecord(a, st);
    for (int r = 0; r < reps; r++) f();
    hipEventRecord(b, st);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms / reps;
}

int main() {
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    const int64_t maxb = (int64_t)4 << 30;
    char *A = nullptr, *B = nullptr;
    CHECK(hipMalloc((void**)&A, maxb));
    CHECK(hipMalloc((void**)&B, maxb));
    CHECK(hipMemset(A, 1, maxb));
    CHECK(hipMemset(B, 1, maxb));
    printf("## fill: grid-stride, TB/s  (rows: buffer size; columns: workgroups per CU x [16 B | 16 B nt | 8 B | 4 B])\n");
    for (int64_t mb : {64, 134, 512, 2048}) {
        const int64_t bytes = mb << 20;
        printf("%5lld MB:", (long long)mb);
        for (int wpc : {1, 2, 4, 8, 16, 32}) {
            const int grid = 256 * wpc, reps = mb >= 512 ? 5 : 20;
            v4f v4 = {1, 2, 3, 4}; v2f v2 = {1, 2};
            double t16 = time_ms(st, reps, [&] { hipLaunchKernelGGL((fill_kernel<v4f, false>), dim3(grid), dim3(256), 0, st, (v4f*)A, bytes / 16, v4); });
            double t16n = time_ms(st, reps, [&] { hipLaunchKernelGGL((fill_kernel<v4f, true>), dim3(grid), dim3(256), 0, st, (v4f*)A, bytes / 16, v4); });
            double t8 = time_ms(st, reps, [&] { hipLaunchKernelGGL((fill_kernel<v2f, false>), dim3(grid), dim3(256), 0, st, (v2f*)A, bytes / 8, v2); });
            double t4 = time_ms(st, reps, [&] { hipLaunchKernelGGL((fill_kernel<float, false>), dim3(grid), dim3(256), 0, st, (float*)A, bytes / 4, 1.0f); });
            printf("  x%-2d [%.2f %.2f %.2f %.2f]", wpc, bytes / t16 / 1e9, bytes / t16n / 1e9, bytes / t8 / 1e9, bytes / t4 / 1e9);
        }
        printf("\n");
    }
    printf("## fill: one workgroup per chunk, TB/s  (columns: chunk bytes x [16 B | 16 B nt])\n");
    for (int64_t mb : {134, 512, 2048}) {
        const int64_t bytes = mb << 20;
        printf("%5lld MB:", (long long)mb);
        for (int kb : {4, 16, 64, 256, 1024}) {
            const int per = kb * 1024 / 16;
            const int64_t n = bytes / 16;
            const unsigned grid = (unsigned)((n + per - 1) / per);
            v4f v4 = {1, 2, 3, 4};
            const int reps = mb >= 512 ? 5 : 20;
            double t = time_ms(st, reps, [&] { hipLaunchKernelGGL((fill_chunk_kernel<v4f, false>), dim3(grid), dim3(256), 0, st, (v4f*)A, n, per, v4); });
            double tn = time_ms(st, reps, [&] { hipLaunchKernelGGL((fill_chunk_kernel<v4f, true>), dim3(grid), dim3(256), 0, st, (v4f*)A, n, per, v4); });
            printf("  %4d KB [%.2f %.2f]", kb, bytes / t / 1e9, bytes / tn / 1e9);
        }
        printf("\n");
    }
    printf("## copy (float4, read + write counted), TB/s  (columns: workgroups per CU x [default | nt])\n");
    for (int64_t mb : {134, 512, 2048}) {
        const int64_t bytes = mb << 20;
        printf("%5lld MB:", (long long)mb);
        for (int wpc : {2, 4, 8, 16, 32}) {
            const int grid = 256 * wpc, reps = mb >= 512 ? 5 : 20;
            double t = time_ms(st, reps, [&] { hipLaunchKernelGGL((copy_kernel<false>), dim3(grid), dim3(256), 0, st, (const v4f*)A, (v4f*)B, bytes / 16); });
            double tn = time_ms(st, reps, [&] { hipLaunchKernelGGL((copy_kernel<true>), dim3(grid), dim3(256), 0, st, (const v4f*)A, (v4f*)B, bytes / 16); });
            printf("  x%-2d [%.2f %.2f]", wpc, 2.0 * bytes / t / 1e9, 2.0 * bytes / tn / 1e9);
        }
        printf("\n");
    }
    printf("## covariance-build store pattern (64 x 64 tiles, both images, no arithmetic): us and TB/s of 8 N^2 bytes\n");
    for (int N : {4096, 8192}) {
        const int nb = N / 64, nt = nb * (nb + 1) / 2;
        for (int pad : {0, 16, 32, 64, 160}) {
            const int64_t ld = N + pad;
            double t = time_ms(st, 20, [&] { hipLaunchKernelGGL((tile_store_kernel<false>), dim3(nt), dim3(256), 0, st, (double*)A, ld, nb); });
            double tn = time_ms(st, 20, [&] { hipLaunchKernelGGL((tile_store_kernel<true>), dim3(nt), dim3(256), 0, st, (double*)A, ld, nb); });
            printf("N=%d ld=N+%-3d: default %.1f us = %.2f TB/s | nt %.1f us = %.2f TB/s\n", N, pad, t * 1e3, 8.0 * N * N / t / 1e9,
                   tn * 1e3, 8.0 * N * N / tn / 1e9);
        }
    }
    return 0;
}
