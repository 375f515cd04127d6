// Probe of ds_read_b64_tr_b16 lane semantics on gfx950 (run once on the GPU box; documents the
// address/result mapping the bf16 attention kernel relies on).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 64];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;  // value = key*64 + d
    __syncthreads();
    const int lane = threadIdx.x, g = lane & 15, grp = lane >> 4;
    const int q = g >> 2, p = g & 3;
    const int key0 = 8 * (grp >> 1), d0 = 16 * (grp & 1);
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s4*)(lds + (key0 + q) * 64 + d0 + 4 * p));
    *(s4*)(out + lane * 4) = v;
}
int main() {
    short* d;
    hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) printf(" (k%d,d%d)", h[l * 4 + e] / 64, h[l * 4 + e] % 64);
        printf("\n");
    }
    return 0;
}
