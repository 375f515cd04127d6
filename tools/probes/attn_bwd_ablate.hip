// Ablations of the bf16 attention backward kernels (csrc/attention_bwd_bf16.hip) at the training geometry (B 64, 1024
// patches, 12 heads, dropout 0.1 with precomputed mask words): tools/probes/attn_bwd_ablate.sh builds one binary per
// variant from a patched copy of the source (ATTN_SRC) and this harness times the dQ and the dK/dV kernel separately.
// Timing only: the variants compute wrong gradients on purpose.
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <stdio.h>
#include <stdlib.h>
static bool g_skip_dq = false, g_skip_dkv = false;
#include ATTN_SRC

namespace vitseg {
int hip_fail(hipError_t e, const char* what) {
    fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e));
    return 1;
}
void set_error(const char* fmt, ...) { fprintf(stderr, "%s\n", fmt); }
}   // namespace vitseg

__global__ void fill16(unsigned short* x, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float f = ((int)(h & 0xffff) - 32768) * (scale / 32768.f);
        x[i] = (unsigned short)(__float_as_uint(f) >> 16);
    }
}
__global__ void fillw(unsigned* x, size_t n) {   // ~90 % ones
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + 77u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        unsigned g = h * 2654435761u; g ^= g >> 16;
        x[i] = h | g | (h << 7) | (g >> 9);
    }
}
__global__ void fillf(float* x, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = v;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int B = 64, Np = 1024, A = 12, D = 768;
    const size_t rows = (size_t)B * Np + B;
    unsigned short *qkv, *ctx, *dctx, *dqkv;
    float *lse, *dvec;
    unsigned* maskw;
    const size_t nstat = (size_t)B * A * (Np + 1), nscr = vitseg::attention_bwd_bf16_scratch_floats(B, Np, A);
    const size_t nwords = vitseg::attn_dropmask_words(B, Np, A);
    CK(hipMalloc(&qkv, rows * 3 * D * 2)); CK(hipMalloc(&ctx, rows * D * 2)); CK(hipMalloc(&dctx, rows * D * 2));
    CK(hipMalloc(&dqkv, rows * 3 * D * 2)); CK(hipMalloc(&lse, nstat * 4)); CK(hipMalloc(&dvec, nscr * 4));
    CK(hipMalloc(&maskw, nwords * 4));
    fill16<<<2048, 256>>>(qkv, rows * 3 * D, 1u, 1.5f);
    fill16<<<2048, 256>>>(ctx, rows * D, 2u, 1.f);
    fill16<<<2048, 256>>>(dctx, rows * D, 3u, 1.f);
    fillw<<<2048, 256>>>(maskw, nwords);
    fillf<<<256, 256>>>(lse, nstat, 12.f);   // a plausible log-sum-exp: p stays finite
    fillf<<<256, 256>>>(dvec, nscr, 0.f);
    CK(hipDeviceSynchronize());
    vitseg::DropArgs dr = {};
    dr.thresh = 6554; dr.seed = 1; dr.stream = 2; dr.scale = 1.f / 0.9f;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* label = argc > 1 ? argv[1] : "?";
    for (int which = 0; which < 2; ++which) {
        g_skip_dq = which == 1;
        g_skip_dkv = which == 0;
        for (int it = 0; it < 3; ++it)
            if (vitseg::launch_attention_bwd_bf16(qkv, ctx, dctx, lse, dvec, dqkv, B, Np, A, dr, 0, maskw)) return 1;
        CK(hipDeviceSynchronize());
        const int n = 10;
        CK(hipEventRecord(e0, 0));
        for (int it = 0; it < n; ++it)
            if (vitseg::launch_attention_bwd_bf16(qkv, ctx, dctx, lse, dvec, dqkv, B, Np, A, dr, 0, maskw)) return 1;
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-14s %-6s %8.1f us per launch (with the CLS finish kernel)\n", label, which == 0 ? "dQ" : "dK/dV", ms * 1000.f / n);
    }
    return 0;
}
