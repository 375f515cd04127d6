// VALU issue rate of the 32-bit integer multiply against the 24-bit multiply-add on gfx950 (decides what a counter-based
// dropout hash may cost): 8 independent chains per thread, grid = 256 CUs x 32 waves.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/imul_rate.hip -o gpurun_out/imul_rate && gpurun_out/imul_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned seed, int iters) {
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 8 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = a[i] * 0x85EBCA6Bu;                       // v_mul_lo_u32
            if (OP == 1) a[i] = __umul24(a[i], 0x85EBCBu) + 0x9E3779u;    // v_mad_u32_u24
            if (OP == 2) a[i] = a[i] ^ (a[i] >> 15);                      // shift + xor (2 ops)
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    if (s == 0x12345678u) out[0] = s;
}
int main() {
    unsigned* d;
    hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4096, blocks = 256 * 8;
    const char* names[3] = {"v_mul_lo_u32", "v_mad_u32_u24", "lshr+xor"};
    for (int op = 0; op < 3; ++op) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (op == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, 1u, iters);
            if (op == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, 1u, iters);
            if (op == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, 1u, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double ops = (double)blocks * 256 * iters * 8;
        printf("%-14s %8.3f ms  %7.1f Gop/s (lane-ops)  -> %.2f lane-ops/clk/CU at 2.4 GHz\n", names[op], ms, ops / ms / 1e6,
               ops / (ms * 1e-3) / 2.4e9 / 256);
    }
    return 0;
}
