// gelu_erf(u) (csrc/common.hpp: u Phi(u) through erfc = 2^(-a Q(a))) against the exact function in fp64 for EVERY fp32 bit
// pattern, with the form rounds 1-3 used (0.5 u (1 + erff(u / sqrt 2)), libm's erff) beside it.  Prints, for each: the
// largest |error|, the largest |error| / max(|gelu|, 1e-2) (what a GEMM that sums thousands of these sees), the largest and
// the mean error in ulps of the result over |gelu| > 1e-6, and checks gelu(+inf) = +inf, NaN -> NaN, gelu(+-0) = +-0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/gelu_check.hip -o gpurun_out/gelu_check && gpurun_out/gelu_check
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include "../../visiontransformer_amd/csrc/common.hpp"

struct Stat {
    double max_abs, max_weighted, max_ulp, sum_ulp;
    unsigned long long n_ulp, bad_special;
    float at_abs, at_weighted, at_ulp;
};

__device__ float gelu_libm(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f)); }

template <int WHICH>
__global__ void sweep(Stat* out) {
    Stat s = {};
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < (1ull << 32);
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const float u = __uint_as_float((unsigned)i);
        const float g = WHICH == 0 ? vitseg::gelu_erf(u) : gelu_libm(u);
        if (u != u) { s.bad_special += !(g != g); continue; }
        if (isinf(u)) { if (u > 0) s.bad_special += !(g == u); continue; }
        if (u == 0.f) { s.bad_special += !(g == 0.f); continue; }
        const double ud = (double)u, ex = 0.5 * ud * erfc(-ud * 0.70710678118654752440);
        const double err = fabs((double)g - ex);
        if (err > s.max_abs) { s.max_abs = err; s.at_abs = u; }
        const double wt = err / fmax(fabs(ex), 1e-2);
        if (wt > s.max_weighted) { s.max_weighted = wt; s.at_weighted = u; }
        if (fabs(ex) > 1e-6) {
            const float exf = (float)fabs(ex);
            const double ulp = (double)(__uint_as_float(__float_as_uint(exf) + 1) - exf);
            const double e = err / ulp;
            if (e > s.max_ulp) { s.max_ulp = e; s.at_ulp = u; }
            s.sum_ulp += e;
            s.n_ulp++;
        }
    }
    out[blockIdx.x * (size_t)blockDim.x + threadIdx.x] = s;
}

int main() {
    const int G = 2048, T = 256;
    Stat* d;
    if (hipMalloc(&d, sizeof(Stat) * G * T) != hipSuccess) return 2;
    Stat* h = (Stat*)malloc(sizeof(Stat) * G * T);
    int rc = 0;
    for (int which = 0; which < 2; ++which) {
        if (which == 0) sweep<0><<<G, T>>>(d); else sweep<1><<<G, T>>>(d);
        if (hipMemcpy(h, d, sizeof(Stat) * G * T, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        Stat t = {};
        for (int i = 0; i < G * T; ++i) {
            if (h[i].max_abs > t.max_abs) { t.max_abs = h[i].max_abs; t.at_abs = h[i].at_abs; }
            if (h[i].max_weighted > t.max_weighted) { t.max_weighted = h[i].max_weighted; t.at_weighted = h[i].at_weighted; }
            if (h[i].max_ulp > t.max_ulp) { t.max_ulp = h[i].max_ulp; t.at_ulp = h[i].at_ulp; }
            t.sum_ulp += h[i].sum_ulp; t.n_ulp += h[i].n_ulp; t.bad_special += h[i].bad_special;
        }
        printf("%-28s max |err| %.3e (u = %g); max |err| / max(|gelu|, 1e-2) %.3e (u = %g); ulps over |gelu| > 1e-6: max %.1f "
               "(u = %g), mean %.3f; special values wrong: %llu\n",
               which == 0 ? "gelu_erf (erfc form):" : "0.5 u (1 + erff) (libm):", t.max_abs, t.at_abs, t.max_weighted,
               t.at_weighted, t.max_ulp, t.at_ulp, t.sum_ulp / (double)t.n_ulp, t.bad_special);
        if (which == 0 && (t.bad_special || t.max_weighted > 1.5e-6)) rc = 1;
    }
    return rc;
}
