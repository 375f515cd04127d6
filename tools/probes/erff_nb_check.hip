// erff_nb(x) (csrc/common.hpp) against the device library's erff(x) for EVERY fp32 bit pattern; and the two GELUs built on
// them.  Prints the number of differing inputs (NaN payloads compared as NaN == NaN).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/erff_nb_check.hip -o gpurun_out/erff_nb_check && gpurun_out/erff_nb_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../visiontransformer_amd/csrc/common.hpp"

__global__ void sweep(unsigned long long* bad) {
    unsigned long long n = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < (1ull << 32);
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)i);
        const float a = erff(x), b = vitseg::erff_nb(x);
        const float ga = vitseg::gelu_erf(x), gb = vitseg::gelu_erf_nb(x);
        const bool same = (__float_as_uint(a) == __float_as_uint(b) || (a != a && b != b)) &&
                          (__float_as_uint(ga) == __float_as_uint(gb) || (ga != ga && gb != gb));
        n += !same;
    }
    if (n) atomicAdd(bad, n);
}

int main() {
    unsigned long long *d, h = 0;
    if (hipMalloc(&d, 8) != hipSuccess || hipMemset(d, 0, 8) != hipSuccess) return 2;
    sweep<<<4096, 256>>>(d);
    if (hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    printf("erff_nb / gelu_erf_nb: %llu of 2^32 inputs differ from erff / gelu_erf\n", h);
    return h != 0;
}
