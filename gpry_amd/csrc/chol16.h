// 16 x 16 building blocks of the in-LDS factorisation, templated on the row stride LD of the LDS image: MFMA
// tile products in the three operand layouts, the single-wave Cholesky of a 16 x 16 diagonal block and the 16-wide
// triangular solve with DPP row broadcasts.  Same arithmetic, instruction for instruction, as the functions of the
// same names in chol_panel.hip (stride 66, which that file keeps as its own copies: its kernels sit at 176 / 334
// VGPRs without scratch and their register allocation has proved sensitive to how the code reaches them);
// used by the single-launch objective of small training sets (lml_small.hip, stride 130).
#pragma once
#include "common.h"

namespace c16 {

__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// C/D fragment of v_mfma_f64_16x16x4_f64: lane (g = lane >> 4, r = lane & 15) holds rows g + 4q (q = 0..3), column r
template <int LD>
__device__ __forceinline__ v4d tile_load(const double* T, int lane) {
    const int r = lane & 15, g = lane >> 4;
    v4d v;
#pragma unroll
    for (int q = 0; q < 4; q++) v[q] = T[(g + 4 * q) * LD + r];
    return v;
}
template <int LD>
__device__ __forceinline__ void tile_store(double* T, v4d v, int lane) {
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; q++) T[(g + 4 * q) * LD + r] = v[q];
}
// acc += sign * A[16 x K] * B[16 x K]^T     (both operands row-wise: A[m][k], B[n][k])
template <int LD, bool NEG>
__device__ __forceinline__ v4d mfma_nt(v4d acc, const double* A, const double* B, int K, int lane) {
    const int r = lane & 15, g = lane >> 4;
    for (int k0 = 0; k0 < K; k0 += 16) {          // K is a multiple of 16: eight fragment reads in flight, then four MFMAs
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { a[u] = A[r * LD + k0 + 4 * u + g]; b[u] = B[r * LD + k0 + 4 * u + g]; }
#pragma unroll
        for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? -a[u] : a[u], b[u], acc, 0, 0, 0);
    }
    return acc;
}
// acc += sign * A[16 x K] * B[K x 16]       (A[m][k] row-wise, B[k][n] row-wise)
template <int LD, bool NEG>
__device__ __forceinline__ v4d mfma_nn(v4d acc, const double* A, const double* B, int K, int lane) {
    const int r = lane & 15, g = lane >> 4;
    for (int k0 = 0; k0 < K; k0 += 16) {          // K is a multiple of 16: eight fragment reads in flight, then four MFMAs
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { a[u] = A[r * LD + k0 + 4 * u + g]; b[u] = B[(k0 + 4 * u + g) * LD + r]; }
#pragma unroll
        for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? -a[u] : a[u], b[u], acc, 0, 0, 0);
    }
    return acc;
}
// acc += A[K x 16]^T * B[K x 16]            (A[k][m], B[k][n], both row-wise)
template <int LD>
__device__ __forceinline__ v4d mfma_tn(v4d acc, const double* A, const double* B, int K, int lane) {
    const int r = lane & 15, g = lane >> 4;
    for (int k0 = 0; k0 < K; k0 += 16) {          // K is a multiple of 16: eight fragment reads in flight, then four MFMAs
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { a[u] = A[(k0 + 4 * u + g) * LD + r]; b[u] = B[(k0 + 4 * u + g) * LD + r]; }
#pragma unroll
        for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
    return acc;
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src_lane);
    hi = __builtin_amdgcn_readlane(hi, src_lane);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(x): v_rsq_f64 + one third-order step (chol_panel.hip: pivot_rsqrt)
__device__ __forceinline__ double pivot_rsqrt(double x) {
    const double r = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * r, r, 1.0);
    const double p = fma(0.375, e, 0.5);
    const double q = r * e;
    return fma(q, p, r);
}

#define C16_FMAC_CASE(C) case C: asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #C " row_mask:0xf bank_mask:0xf" \
                                              : "+v"(acc) : "v"(a), "v"(b)); break;
// acc += a[lane c of this lane's 16-lane row] * b
__device__ __forceinline__ void fmac_row_bcast(double& acc, double a, double b, int c) {
    switch (c) {
        C16_FMAC_CASE(0) C16_FMAC_CASE(1) C16_FMAC_CASE(2) C16_FMAC_CASE(3) C16_FMAC_CASE(4) C16_FMAC_CASE(5)
        C16_FMAC_CASE(6) C16_FMAC_CASE(7) C16_FMAC_CASE(8) C16_FMAC_CASE(9) C16_FMAC_CASE(10)
        C16_FMAC_CASE(11) C16_FMAC_CASE(12) C16_FMAC_CASE(13) C16_FMAC_CASE(14) C16_FMAC_CASE(15)
        default: break;
    }
}
#undef C16_FMAC_CASE
// acc -= a[lane c of this lane's 16-lane row] * b   (negation as the source modifier of the DPP operand)
#define C16_FMSUB_CASE(C) case C: asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:" #C " row_mask:0xf bank_mask:0xf" \
                                               : "+v"(acc) : "v"(a), "v"(b)); break;
__device__ __forceinline__ void fmsub_row_bcast(double& acc, double a, double b, int c) {
    switch (c) {
        C16_FMSUB_CASE(0) C16_FMSUB_CASE(1) C16_FMSUB_CASE(2) C16_FMSUB_CASE(3) C16_FMSUB_CASE(4) C16_FMSUB_CASE(5)
        C16_FMSUB_CASE(6) C16_FMSUB_CASE(7) C16_FMSUB_CASE(8) C16_FMSUB_CASE(9) C16_FMSUB_CASE(10)
        C16_FMSUB_CASE(11) C16_FMSUB_CASE(12) C16_FMSUB_CASE(13) C16_FMSUB_CASE(14) C16_FMSUB_CASE(15)
        default: break;
    }
}
#undef C16_FMSUB_CASE

// Cholesky of the 16 x 16 block at S by one wave, one row per lane (lanes 16..63 repeat lanes 0..15); writes L
// (lower, zeros above) and the reciprocal pivots rd[16]; returns the first failing column + 1 (0 if ok).
template <int LD>
__device__ __forceinline__ int chol16_wave(double* S, double* rd, int lane) {
    const int i = lane & 15;
    double x[16];
#pragma unroll
    for (int c = 0; c < 16; c++) x[c] = S[i * LD + c];
    int bad = 0;
    double my_d = 1.0, my_r = 1.0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const double djj = readlane_f64(x[j], j);
        if (!(djj > 0.0) && bad == 0) bad = j + 1;              // dpotf2: ajj <= 0 or NaN
        const double rinv = pivot_rsqrt(djj);
        if (i == j) { my_d = djj; my_r = rinv; }
        double lij = x[j] * rinv;
        asm volatile("s_nop 1" : "+v"(lij));                    // VALU write -> DPP read of the same VGPR
#pragma unroll
        for (int c = j + 1; c < 16; c++) fmsub_row_bcast(x[c], lij, lij, c);   // x[c] -= L[c][j] L[i][j]
        x[j] = lij;                                             // (rows i <= j: values nobody reads; the diagonal follows)
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

// x <- x * L^-T for the 16 rows at Xr against the 16 x 16 lower factor at L (reciprocal pivots rd): one row per lane
// (lanes 16..63 repeat lanes 0..15)
template <int LD>
__device__ __forceinline__ void trsm16_rows(double* Xr, const double* L, const double* rd, int lane) {
    const int i = lane & 15;
    double x[16], lr[16];
#pragma unroll
    for (int c = 0; c < 16; c++) x[c] = Xr[i * LD + c];
#pragma unroll
    for (int c = 0; c < 16; c++) lr[c] = L[i * LD + c];
    const double myr = rd[i];
#pragma unroll
    for (int c = 0; c < 16; c++) {
        double xc = 0.0;
        fmac_row_bcast(xc, myr, x[c], c);                       // x[c] / L[c][c]
        x[c] = xc;
        const double nx = -xc;
#pragma unroll
        for (int c2 = c + 1; c2 < 16; c2++) fmac_row_bcast(x[c2], lr[c], nx, c2);   // x[c2] -= x[c] L[c2][c]
    }
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; c++) Xr[i * LD + c] = x[c];
    }
}

}  // namespace c16
