// One wave, one 16 x 16 Cholesky (the link of every factorisation chain in this library): cycles per call of the
// shipped c16::chol16_wave and of candidate instruction orders, and whether the factors agree bit for bit.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -I gpry_amd/csrc tools/r03/chol16_bench.hip -o tools/r03/chol16_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "chol16.h"

#define LD 66

// ---- experiment (measured, not adopted): the producer publishes every column as it is finished -- the column, then its
// reciprocal pivot into rd[j] (preset to a sentinel) -- so that the wave that factors the next diagonal block can run its
// row substitution one column behind instead of after all sixteen (trsm16_rows_follow in the kernel, removed again).
// The two unconditional LDS writes per column cost the producer 520 cycles per block (3996 vs 3474; 4142 with a third
// write for a counter, 4603 when the writes sit under `if (lane < 16)`), the follower pays an LDS round trip per column
// unless it is a column behind, and every wave further down that follows too takes issue slots from the producer it
// shares a SIMD with: chol of N = 128 in lml_small 53.5k -> 51.4-52.3k cycles at best (57k with all waves following),
// i.e. under 1 us of a 60-us evaluation.
namespace c16 {
#define C16_UNPUBLISHED (-1.0)      // no reciprocal pivot is negative (a failed pivot gives NaN or +inf)
typedef __attribute__((address_space(3))) void* c16_lptr;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(c16_lptr)p; }
__device__ __forceinline__ void publish_column(double* col_i, double lij, double* rd_j, double rinv) {
    asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %2, %3" :: "v"(lds_addr(col_i)), "v"(lij), "v"(lds_addr(rd_j)), "v"(rinv) : "memory");
}

// chol16_wave that publishes its columns (rd[0..15] == C16_UNPUBLISHED before the call)
__device__ __forceinline__ int chol16_wave_pub(double* S, double* rd, int lane) {
    const int i = lane & 15;
    double x[16];
#pragma unroll
    for (int c = 0; c < 16; c++) x[c] = S[i * LD + c];
    int bad = 0;
    double my_d = 1.0, my_r = 1.0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const double djj = readlane_f64(x[j], j);
        if (!(djj > 0.0) && bad == 0) bad = j + 1;
        const double rinv = pivot_rsqrt(djj);
        if (i == j) { my_d = djj; my_r = rinv; }
        double lij = x[j] * rinv;
        publish_column(S + i * LD + j, lij, rd + j, rinv);
        asm volatile("s_nop 1" : "+v"(lij));
#pragma unroll
        for (int c = j + 1; c < 16; c++) fmsub_row_bcast(x[c], lij, lij, c);
        x[j] = lij;
    }
    double piv = my_d * my_r;
    piv = fma(fma(-piv, piv, my_d), 0.5 * my_r, piv);
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; c++) S[i * LD + c] = x[c];      // (the published entries again, and the ones above the diagonal)
        S[i * LD + i] = piv;
        rd[i] = my_r;
    }
    return bad;
}

}  // namespace c16
#define SB __builtin_amdgcn_sched_barrier(0)

// candidate: the next pivot's entry is updated first, its reciprocal root is under way while the rest of the column's
// updates issue (a lone wave issues in order: what follows a v_rsq_f64 or a v_readlane in program order and depends
// on it waits, whatever else could have run)
template <int A, int B, int C>
__device__ __forceinline__ int chol16_sched(double* S, double* rd, int lane) {
    using namespace c16;
    const int i = lane & 15;
    double x[16];
#pragma unroll
    for (int c = 0; c < 16; c++) x[c] = S[i * LD + c];
    int bad = 0;
    double my_d = 1.0, my_r = 1.0;
    double djj = readlane_f64(x[0], 0);
    double rinv = pivot_rsqrt(djj);
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if (!(djj > 0.0) && bad == 0) bad = j + 1;
        if (i == j) { my_d = djj; my_r = rinv; }
        double lij = x[j] * rinv;
        asm volatile("s_nop 1" : "+v"(lij));
        double djn = 1.0, rn = 1.0;
        if (j < 15) {
            constexpr int unused = 0; (void)unused;
            const int c1 = j + 2, c2 = c1 + A < 16 ? c1 + A : 16, c3 = c2 + B < 16 ? c2 + B : 16, c4 = c3 + C < 16 ? c3 + C : 16;
            fmsub_row_bcast(x[j + 1], lij, lij, j + 1);
            SB;
            djn = readlane_f64(x[j + 1], j + 1);
            SB;
#pragma unroll
            for (int c = c1; c < c2; c++) fmsub_row_bcast(x[c], lij, lij, c);
            SB;
            const double r = __builtin_amdgcn_rsq(djn);
            SB;
#pragma unroll
            for (int c = c2; c < c3; c++) fmsub_row_bcast(x[c], lij, lij, c);
            SB;
            const double e = fma(-djn * r, r, 1.0);
            SB;
#pragma unroll
            for (int c = c3; c < c4; c++) fmsub_row_bcast(x[c], lij, lij, c);
            SB;
            const double p = fma(0.375, e, 0.5);
            const double q = r * e;
            rn = fma(q, p, r);
            SB;
#pragma unroll
            for (int c = c4; c < 16; c++) fmsub_row_bcast(x[c], lij, lij, c);
            SB;
        }
        x[j] = lij;
        djj = djn; rinv = rn;
    }
    double piv = my_d * my_r;
    piv = fma(fma(-piv, piv, my_d), 0.5 * my_r, piv);
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; c++) S[i * LD + c] = x[c];
        S[i * LD + i] = piv;
        rd[i] = my_r;
    }
    return bad;
}

template <int VAR>
__global__ __launch_bounds__(64) void bench(const double* A, double* out, unsigned long long* cyc, int reps) {
    __shared__ double P[16 * LD], S[16 * LD], rd[16];
    __shared__ int cnt;
    const int lane = threadIdx.x;
    for (int e = lane; e < 256; e += 64) P[(e >> 4) * LD + (e & 15)] = A[e];
    __syncthreads();
    int bad = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < reps; it++) {
        if (lane < 16)
#pragma unroll
            for (int c = 0; c < 16; c++) S[lane * LD + c] = P[lane * LD + c];
        c16::wave_fence();
        if (VAR == 0) bad |= c16::chol16_wave<LD>(S, rd, lane);
        else if (VAR == 1) bad |= chol16_sched<2, 3, 0>(S, rd, lane);
        else if (VAR == 2) bad |= chol16_sched<1, 3, 1>(S, rd, lane);
        else if (VAR == 3) bad |= chol16_sched<2, 2, 2>(S, rd, lane);
        else if (VAR == 4) bad |= chol16_sched<0, 3, 0>(S, rd, lane);
        else if (VAR == 5) { if (lane < 16) rd[lane] = C16_UNPUBLISHED; bad |= c16::chol16_wave_pub(S, rd, lane); }
        c16::wave_fence();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = bad; }
    for (int e = lane; e < 256; e += 64) out[e] = S[(e >> 4) * LD + (e & 15)];
    if (lane < 16) out[256 + lane] = rd[lane];
}

int main() {
    double hA[256], hO[6][272];
    srand(3);
    double B[256];
    for (int e = 0; e < 256; e++) B[e] = (double)rand() / RAND_MAX - 0.5;
    for (int a = 0; a < 16; a++)
        for (int b = 0; b < 16; b++) {
            double s = a == b ? 0.5 : 0.0;
            for (int k = 0; k < 16; k++) s += B[a * 16 + k] * B[b * 16 + k];
            hA[a * 16 + b] = s;
        }
    double *dA, *dO; unsigned long long* dC;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dO, 272 * 8); hipMalloc(&dC, 16);
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
    const int reps = 2000;
    for (int var = 0; var < 6; var++) {
        unsigned long long c[2], best = ~0ull;
        for (int run = 0; run < 3; run++) {
            switch (var) {
                case 0: hipLaunchKernelGGL(bench<0>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps); break;
                case 1: hipLaunchKernelGGL(bench<1>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps); break;
                case 2: hipLaunchKernelGGL(bench<2>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps); break;
                case 3: hipLaunchKernelGGL(bench<3>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps); break;
                case 4: hipLaunchKernelGGL(bench<4>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps); break;
                default: hipLaunchKernelGGL(bench<5>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps); break;
            }
            hipDeviceSynchronize();
            hipMemcpy(c, dC, 16, hipMemcpyDeviceToHost);
            if (c[0] < best) best = c[0];
        }
        hipMemcpy(hO[var], dO, 272 * 8, hipMemcpyDeviceToHost);
        // check against a host factorisation (sanity), and bits against variant 0
        double maxerr = 0.0;
        for (int a = 0; a < 16; a++)
            for (int b = 0; b <= a; b++) {
                double s = 0.0;
                for (int k = 0; k <= b; k++) s += hO[var][a * 16 + k] * hO[var][b * 16 + k];
                maxerr = fmax(maxerr, fabs(s - hA[a * 16 + b]));
            }
        int same = 1;
        for (int a = 0; a < 16; a++)
            for (int b = 0; b <= a; b++) same &= !memcmp(&hO[var][a * 16 + b], &hO[0][a * 16 + b], 8);
        same &= !memcmp(&hO[var][256], &hO[0][256], 128);
        printf("variant %d: %.0f cycles per call (copy + fences included), |L L^T - A| %.1e, lower triangle + reciprocal pivots bit-identical to variant 0: %s, bad %llu\n",
               var, (double)best / reps, maxerr, same ? "yes" : "NO", c[1]);
    }
    return 0;
}
