// VALU issue rates that bound the attention softmax on gfx950: v_exp_f32, v_exp_f16, v_cvt_pk_bf16_f32, v_pk_mul_f32,
// v_fma_f32.  8 independent chains per thread, 256 CUs x 32 waves.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16;
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float seed, int iters) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (1.0f + 1e-3f * (threadIdx.x * 8 + i));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = __builtin_amdgcn_exp2f(a[i]) - 1.0f;                                  // v_exp_f32 + v_add
            if (OP == 2) a[i] = fmaf(a[i], 0.999f, 1e-4f);                                             // v_fma_f32
            if (OP == 3) a[i] = a[i] - 1.0f;                                                           // v_add (baseline of OP 0)
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    if (s == 0.12345f) out[0] = s;
}
__global__ __launch_bounds__(256) void kh(float* out, float seed, int iters) {   // v_exp_f16 chain (+ one f16 add)
    f16 a[8];
    for (int i = 0; i < 8; ++i) a[i] = (f16)(seed * (1.0f + 1e-3f * (threadIdx.x * 8 + i)));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_exp2f16(a[i]) - (f16)1.0f;
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += (float)a[i];
    if (s == 0.12345f) out[0] = s;
}
__global__ __launch_bounds__(256) void kpk(float* out, float seed, int iters) {  // v_pk_mul_f32: 2 elements per op
    f32x2 a[4];
    for (int i = 0; i < 4; ++i) a[i] = f32x2{seed + i, seed - i};
    const f32x2 c = {0.999f, 1.001f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = a[i] * c;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = a[i] * c;
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += a[i][0] + a[i][1];
    if (s == 0.12345f) out[0] = s;
}
int main() {
    float* d;
    hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2048, blocks = 256 * 8;
    const char* names[6] = {"v_exp_f32+add", "-", "v_fma_f32", "v_add_f32", "v_exp_f16+add", "v_pk_mul_f32 (8 ops = 16 elts)"};
    for (int op = 0; op < 6; ++op) {
        if (op == 1) continue;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (op == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, 0.5f, iters);
            if (op == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, 0.5f, iters);
            if (op == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, 0.5f, iters);
            if (op == 4) hipLaunchKernelGGL(kh, dim3(blocks), dim3(256), 0, 0, d, 0.5f, iters);
            if (op == 5) hipLaunchKernelGGL(kpk, dim3(blocks), dim3(256), 0, 0, d, 0.5f, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double ops = (double)blocks * 256 * iters * 8;
        printf("%-32s %8.3f ms  -> %.2f chain-steps/clk/CU at 2.4 GHz\n", names[op], ms, ops / (ms * 1e-3) / 2.4e9 / 256);
    }
    return 0;
}
