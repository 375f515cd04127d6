// Where do the idle matrix-pipe cycles of csrc/attention_f32.hip go?  Standalone harness: includes a (possibly patched)
// copy of the kernel source (ATTN_SRC) and times the patch-query kernel on the headline shape (32 images x 1024 patches,
// 12 heads).  tools/probes/attn_f32_where.sh builds the variants (garbage results there; only the time matters).
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <stdio.h>
#include <stdlib.h>
#include ATTN_SRC

namespace vitseg {
int hip_fail(hipError_t e, const char* what) {
    fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e));
    return 1;
}
void set_error(const char* fmt, ...) { fprintf(stderr, "%s\n", fmt); }
int launch_attention_x3_main(const float*, float*, int, int, int, hipStream_t) { return 1; }
}   // namespace vitseg

__global__ void fill_kernel(float* x, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        x[i] = ((int)(h & 0xffff) - 32768) * (scale / 32768.f);
    }
}

int main() {
    const int B = 32, Np = 1024, A = 12, D = 768;
    const size_t rows = (size_t)B * Np + B;
    float *qkv, *ctx;
    hipMalloc(&qkv, rows * 3 * D * 4);
    hipMalloc(&ctx, rows * D * 4);
    fill_kernel<<<2048, 256>>>(qkv, rows * 3 * D, 1u, 2.f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    vitseg::DropArgs dr = {};
    float best = 1e9f;
    for (int rnd = 0; rnd < 4; ++rnd) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 10; ++i)
            if (vitseg::launch_attention_f32(qkv, ctx, nullptr, B, Np, A, dr, 0, false)) return 1;
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (rnd && ms / 10 < best) best = ms / 10;
    }
    // (the launch includes the small CLS-query kernel: ~36 us)
    printf("attention fp32 (patch + CLS query kernels) %8.1f us   %6.1f TFLOP/s\n", best * 1e3,
           4.0 * B * A * (double)(Np + 1) * (Np + 1) * 64 / (best * 1e-3) * 1e-12);
    return 0;
}
