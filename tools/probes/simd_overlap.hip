// Do the matrix pipe and the vector ALU of ONE SIMD work at the same time on gfx950 -- for instructions of two
// different waves, and for instructions of one wave?  Hand-written instruction streams (no compiler scheduling in the
// measured loops), every stream issue-bound (no dependency closer than 12 instructions):
//   M: 8 x v_mfma_f32_32x32x16_bf16 per iteration (4 independent accumulators)       = 256 matrix cycles
//   V: 48 x v_fma_f32 (VOP 0)  or  24 x v_exp_f32 (VOP 1) per iteration              ~ 192 issue cycles
// Block of 512 threads = 2 waves per SIMD (wave w and w + 4 share one), one block per CU:
//   mode 1: waves 0-3 run M, waves 4-7 exit      mode 2: waves 4-7 run V, waves 0-3 exit      mode 3: both
//   mode 4: ONE wave per SIMD (waves 4-7 exit) runs M with 6 v_fma (or 3 v_exp) placed behind every MFMA
// time(3) ~ max(time(1), time(2)) <=> the two pipes overlap across waves; time(4) ~ time(1) <=> inside a wave.
// Run under rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE
// for the counter view (one kernel dispatch per mode; the mode is the kernel's template argument).
//   hipcc --offload-arch=gfx950 -O3 [-DVOP=1] tools/probes/simd_overlap.hip -o gpurun_out/simd_overlap && gpurun_out/simd_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#ifndef VOP
#define VOP 0
#endif
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#ifdef F32MFMA   // the fp32 matrix instruction (64 cycles; 8 per iteration = 512 matrix cycles): does IT leave the vector ALU free?
#define MFMA(acc) "v_mfma_f32_32x32x2_f32 %" #acc ", %18, %19, %" #acc "\n\t"
#else
#define MFMA(acc) "v_mfma_f32_32x32x16_bf16 %" #acc ", %16, %17, %" #acc "\n\t"
#endif
#if VOP == 0
#define V1(r) "v_fma_f32 %" #r ", %" #r ", %18, %19\n\t"
#define VPER 6
#else
#define V1(r) "v_exp_f32 %" #r ", %" #r "\n\t"
#define VPER 3
#endif
#define V12 V1(4) V1(5) V1(6) V1(7) V1(8) V1(9) V1(10) V1(11) V1(12) V1(13) V1(14) V1(15)

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    f32x4 x = {1.f, 2.f, 3.f, 4.f}, y = {.5f, .25f, .125f, 1.f};
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = 0.5f + 1e-3f * (threadIdx.x + i);
    const float ka = 0.999f, kb = 0.001f;
    const bool m_role = wave < 4 && (MODE == 1 || MODE == 3 || MODE == 4);
    const bool v_role = wave >= 4 && (MODE == 2 || MODE == 3);
#define OPS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]),   \
            "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]) : "v"(x), "v"(y), "v"(ka), "v"(kb)
    if (m_role && MODE != 4) {
        for (int it = 0; it < iters; ++it)
            asm volatile(MFMA(0) MFMA(1) MFMA(2) MFMA(3) MFMA(0) MFMA(1) MFMA(2) MFMA(3) : OPS);
    } else if (m_role) {   // 8 MFMAs, VPER vector instructions behind each (48 v_fma / 24 v_exp per iteration)
        for (int it = 0; it < iters; ++it) {
#if VOP == 0
#define VA6 V1(4) V1(5) V1(6) V1(7) V1(8) V1(9)
#define VB6 V1(10) V1(11) V1(12) V1(13) V1(14) V1(15)
            asm volatile(MFMA(0) VA6 MFMA(1) VB6 MFMA(2) VA6 MFMA(3) VB6 MFMA(0) VA6 MFMA(1) VB6 MFMA(2) VA6 MFMA(3) VB6 : OPS);
#else
            asm volatile(MFMA(0) V1(4) V1(5) V1(6) MFMA(1) V1(7) V1(8) V1(9) MFMA(2) V1(10) V1(11) V1(12) MFMA(3) V1(13) V1(14) V1(15)
                         MFMA(0) V1(4) V1(5) V1(6) MFMA(1) V1(7) V1(8) V1(9) MFMA(2) V1(10) V1(11) V1(12) MFMA(3) V1(13) V1(14) V1(15) : OPS);
#endif
        }
    } else if (v_role) {
        for (int it = 0; it < iters; ++it) {
#if VOP == 0
            asm volatile(V12 V12 V12 V12 : OPS);
#else
            asm volatile(V12 V12 : OPS);
#endif
        }
    }
    float r = a0[0] + a1[1] + a2[2] + a3[3];
    for (int i = 0; i < 12; ++i) r += v[i];
    if (r == 0.123456f) out[0] = r;
}

template <int MODE>
static float run(float* d, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}
int main() {
    float* d;
    (void)hipMalloc(&d, 4);
    const int iters = 20000;
    const float t1 = run<1>(d, iters), t2 = run<2>(d, iters), t3 = run<3>(d, iters), t4 = run<4>(d, iters);
    printf("VOP %d (%s), %d iterations, one 512-thread block per CU\n", VOP, VOP ? "24 v_exp_f32" : "48 v_fma_f32", iters);
    printf("mode 1  matrix waves only (8 MFMA / iteration)          : %.3f ms\n", t1);
    printf("mode 2  vector waves only                                : %.3f ms\n", t2);
    printf("mode 3  both, two waves per SIMD                         : %.3f ms   (max %.3f, sum %.3f)\n", t3, t1 > t2 ? t1 : t2, t1 + t2);
    printf("mode 4  one wave per SIMD, %d vector ops behind each MFMA : %.3f ms\n", VPER, t4);
    return 0;
}
