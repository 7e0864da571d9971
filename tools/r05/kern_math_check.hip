// Accuracy of the straight-line sqrt / exp / correlation of kern_math.h against libm on the device, and cycles per evaluation:
//   hipcc -O3 --offload-arch=gfx950 -I gpry_amd/csrc tools/r05/kern_math_check.hip -o tools/r05/kern_math_check && tools/r05/kern_math_check
#include "kern_math.h"
#include <cstdio>
#include <vector>
#include <random>
#include <cmath>

__device__ __forceinline__ double sqrt_one_correction(double x) {
    const double s = __builtin_amdgcn_rsq(x);
    double g = x * s, h = 0.5 * s;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    const double e = fma(-g, g, x);
    return fma(e, h, g);
}
__device__ __forceinline__ double sqrt_newton_only(double x) {       // rsq, then ONE coupled step and one correction without h refinement
    const double s = __builtin_amdgcn_rsq(x);
    const double g = x * s, h = 0.5 * s;
    const double e = fma(-g, g, x);
    return fma(e, h, g);
}
__global__ void eval(const double* x, double* out, int n, int what) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i], r = 0.0;
    switch (what) {
        case 0: r = sqrt(v); break;
        case 1: r = fast_sqrt_pos(v); break;
        case 2: r = fast_sqrt_nz(v); break;
        case 3: r = sqrt_one_correction(v); break;
        case 4: r = sqrt_newton_only(v); break;
        case 5: r = exp(-v); break;
        case 6: r = fast_exp_neg(v); break;
        case 7: r = fast_exp_neg11(v); break;
        case 8: r = corr_r2<GPRY_MATERN52>(v); break;
        case 9: r = corr_r2_fast<GPRY_MATERN52>(v); break;
        case 10: r = corr_scaled_fast<GPRY_MATERN52>(5.0 * v); break;
    }
    out[i] = r;
}
template <int WHAT>
__global__ void cycles(const double* x, double* out, long long* cyc) {
    double v = x[threadIdx.x], acc = 0.0;
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 256; it++) {
        double r;
        if (WHAT == 1) r = fast_sqrt_pos(v); else if (WHAT == 2) r = fast_sqrt_nz(v); else if (WHAT == 3) r = sqrt_one_correction(v);
        else if (WHAT == 6) r = fast_exp_neg(v); else if (WHAT == 7) r = fast_exp_neg11(v);
        else if (WHAT == 9) r = corr_r2_fast<GPRY_MATERN52>(v); else if (WHAT == 10) r = corr_scaled_fast<GPRY_MATERN52>(v);
        else r = corr_r2<GPRY_MATERN52>(v);
        acc += r; v += 1e-3;
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
static double ulps(double a, double b) {
    if (a == b) return 0.0;
    const double u = std::nextafter(std::fabs(b), INFINITY) - std::fabs(b);
    return std::fabs(a - b) / u;
}
int main() {
    const int n = 1 << 22;
    std::vector<double> hx(n), ha(n), hb(n);
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> lg(-12.0, 4.5);
    for (int i = 0; i < n; i++) hx[i] = std::exp(lg(rng) * 2.302585092994046);     // 1e-12 .. 3e4, log-uniform
    double *dx, *dout; long long* dc;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, n * 8); hipMalloc(&dc, 8);
    hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice);
    auto run = [&](int what, std::vector<double>& h) {
        hipLaunchKernelGGL(eval, dim3(n / 256), dim3(256), 0, 0, dx, dout, n, what);
        hipMemcpy(h.data(), dout, n * 8, hipMemcpyDeviceToHost);
    };
    const char* names[] = {"libm sqrt", "fast_sqrt_pos (round 3)", "fast_sqrt_nz (two corrections, no zero test)", "one correction",
                           "no coupled step, one correction", "libm exp(-t)", "fast_exp_neg (Taylor 13)", "fast_exp_neg11 (degree 11)",
                           "corr_r2<M52> (libm)", "corr_r2_fast<M52>", "corr_scaled_fast<M52>(5 r^2)"};
    for (int base : {0, 5, 8}) {
        run(base, ha);
        // reference: long double on the host
        const int last = base == 0 ? 4 : base == 5 ? 7 : 10;
        for (int w = base; w <= last; w++) {
            if (w != base) run(w, hb); else hb = ha;
            double worst = 0.0, mean = 0.0; long cnt = 0;
            for (int i = 0; i < n; i++) {
                long double x = hx[i], ref;
                if (base == 0) ref = sqrtl(x);
                else if (base == 5) ref = expl(-x);
                else { long double t = sqrtl(5.0L * x); ref = (1.0L + t + t * t / 3.0L) * expl(-t); }
                if ((double)ref < 1e-300) continue;
                const double u = ulps(hb[i], (double)ref);
                worst = u > worst ? u : worst; mean += u; cnt++;
            }
            printf("%-48s max %.2f ulp, mean %.3f ulp (vs the correctly rounded value, %ld arguments)\n", names[w], worst, mean / cnt, cnt);
        }
    }
    long long c;
#define CYC(W) hipLaunchKernelGGL(cycles<W>, dim3(1), dim3(64), 0, 0, dx, dout, dc); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); \
    printf("%-48s %.1f cycles per evaluation (one wave, dependent accumulate)\n", names[W], c / 256.0);
    CYC(1) CYC(2) CYC(3) CYC(6) CYC(7) CYC(8) CYC(9) CYC(10)
    return 0;
}
