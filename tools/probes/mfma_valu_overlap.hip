// Do MFMA and plain VALU instructions of two different waves on the SAME SIMD overlap on gfx950?  A block of 8 waves
// (wave w and w + 4 share a SIMD): mode 1 = waves 0-3 run an MFMA chain, waves 4-7 idle; mode 2 = waves 4-7 run a VALU
// (v_exp + fma) chain, waves 0-3 idle; mode 3 = both.  time(3) ~ max(time(1), time(2)) -> they overlap.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap.hip -o gpurun_out/mvo && gpurun_out/mvo
#include <hip/hip_runtime.h>
#include <stdio.h>
#ifndef VOP
#define VOP 0
#endif
__device__ __forceinline__ float vop(float v) {
#if VOP == 0
    return fmaf(__builtin_amdgcn_exp2f(v), 0.25f, 0.1f);               // v_exp (8 clk) + v_fma (2 clk)
#elif VOP == 1
    v = fmaf(v, 0.999f, 0.001f); v = fmaf(v, 0.999f, 0.001f); v = fmaf(v, 0.999f, 0.001f); v = fmaf(v, 0.999f, 0.001f);
    return fmaf(v, 0.999f, 0.001f);                                    // 5 x v_fma (10 clk)
#else
    unsigned u = __float_as_uint(v); u = u * 0x9E3779B1u + 0x7F4A7C15u; u ^= u >> 15; u = u * 0x85EBCA6Bu + 1u;
    return __uint_as_float((u & 0x007fffffu) | 0x3f000000u);           // integer multiply / shift / xor
#endif
}
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(512) void k(float* out, int mode, int iters) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
            bf16x8 x, y;
            for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f); y[i] = (__bf16)0.5f; }
            for (int it = 0; it < iters; ++it) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
            }
            r = a0[0] + a1[1] + a2[2] + a3[3];
        }
    } else if (mode & 2) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = 0.5f + 1e-3f * (threadIdx.x + i);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = vop(v[i]);
        }
        for (int i = 0; i < 8; ++i) r += v[i];
    }
    if (r == 0.123456f) out[0] = r;
}
// same-wave shadow: each MFMA is followed IN PROGRAM ORDER by `NV` independent VALU ops (v_exp + fma pairs) of the same
// wave; one wave per SIMD.  sel: 1 = MFMA only, 2 = VALU only, 3 = interleaved.
template <int NV, int sel>
__global__ __launch_bounds__(256) void ks(float* out, int iters) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f); y[i] = (__bf16)0.5f; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.5f + 1e-3f * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#define STEP(acc, base)                                                                                     \
        if (sel & 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);                      \
        if (sel & 2) {                                                                                       \
            _Pragma("unroll") for (int i = 0; i < NV; ++i)                                                   \
                v[(base + i) & 7] = vop(v[(base + i) & 7]);            \
        }
        STEP(a0, 0) STEP(a1, 2) STEP(a2, 4) STEP(a3, 6)
#undef STEP
    }
    float r = a0[0] + a1[1] + a2[2] + a3[3];
    for (int i = 0; i < 8; ++i) r += v[i];
    if (r == 0.123456f) out[0] = r;
}
int main() {
    float* d;
    hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256;   // one block per CU
    for (int mode = 1; mode <= 3; ++mode) {
        // per iteration: 4 MFMAs = 128 matrix cycles; VALU: 8 x (8 + 2) = 80 cycles -> scale VALU iterations by 1.6 for balance
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, mode, 20000);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d (%s): %.3f ms\n", mode, mode == 1 ? "MFMA waves only" : mode == 2 ? "VALU waves only" : "both", ms);
    }
    for (int sel = 1; sel <= 3; ++sel) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (sel == 1) hipLaunchKernelGGL((ks<2, 1>), dim3(blocks), dim3(256), 0, 0, d, 20000);
            if (sel == 2) hipLaunchKernelGGL((ks<2, 2>), dim3(blocks), dim3(256), 0, 0, d, 20000);
            if (sel == 3) hipLaunchKernelGGL((ks<2, 3>), dim3(blocks), dim3(256), 0, 0, d, 20000);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("same wave, 2 (exp+fma) per MFMA, sel %d (%s): %.3f ms\n", sel, sel == 1 ? "MFMA only" : sel == 2 ? "VALU only" : "interleaved", ms);
    }
    return 0;
}
