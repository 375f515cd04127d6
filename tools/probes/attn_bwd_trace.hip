// Cycle stamps inside the bf16 attention dK/dV kernel (csrc/attention_bwd_bf16.hip): where does one wave spend a 64-query
// tile?  tools/probes/attn_bwd_trace.sh patches s_memtime stamps into a copy of the kernel (ATTN_SRC), this harness runs
// it at the training geometry and prints, for the four waves of one block, the mean cycles between consecutive stamps.
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__device__ unsigned long long g_trace[4 * 20 * 16];   // [wave][tile][stamp]
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
__global__ void fillf(float* x, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = v;
}

int main() {
    const int B = 64, Np = 1024, A = 12, D = 768;
    const size_t rows = (size_t)B * Np + B;
    unsigned short *qkv, *ctx, *dctx, *dqkv;
    float *lse, *dvec;
    hipMalloc(&qkv, rows * 3 * D * 2); hipMalloc(&ctx, rows * D * 2); hipMalloc(&dctx, rows * D * 2); hipMalloc(&dqkv, rows * 3 * D * 2);
    hipMalloc(&lse, (size_t)B * A * (Np + 1) * 4);
    hipMalloc(&dvec, vitseg::attention_bwd_bf16_scratch_floats(B, Np, A) * 4);   // delta + the per-block CLS partials
    fill16<<<2048, 256>>>(qkv, rows * 3 * D, 1u, 1.5f);
    fill16<<<2048, 256>>>(ctx, rows * D, 2u, 1.f);
    fill16<<<2048, 256>>>(dctx, rows * D, 3u, 1.f);
    fillf<<<256, 256>>>(lse, (size_t)B * A * (Np + 1), 12.f);   // a plausible log-sum-exp (log2 units): p stays finite
    hipDeviceSynchronize();
    vitseg::DropArgs dr = {};
    for (int it = 0; it < 3; ++it)
        if (vitseg::launch_attention_bwd_bf16(qkv, ctx, dctx, lse, dvec, dqkv, B, Np, A, dr, 0, nullptr)) return 1;
    hipDeviceSynchronize();
    std::vector<unsigned long long> t(4 * 20 * 16);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_trace), t.size() * 8);
    const char* names[] = {"tile start -> S/dP MFMAs issued (q block 0)", "-> exp / pack done", "-> dV/dK MFMAs issued",
                           "-> S/dP MFMAs issued (q block 1)", "-> exp / pack done", "-> dV/dK MFMAs issued",
                           "-> stats written, own DMA landed", "-> past the barrier (next tile start)"};
    for (int w = 0; w < 4; ++w) {
        printf("wave %d: cycles per 64-query tile (mean over tiles 2..13)\n", w);
        double tot = 0;
        for (int k = 0; k < 8; ++k) {
            double s = 0;
            for (int tile = 2; tile < 14; ++tile) {
                const unsigned long long a = t[(w * 20 + tile) * 16 + k], b = k < 7 ? t[(w * 20 + tile) * 16 + k + 1] : t[(w * 20 + tile + 1) * 16];
                s += (double)(b - a);
            }
            printf("   %-48s %8.0f\n", names[k], s / 12);
            tot += s / 12;
        }
        printf("   %-48s %8.0f\n", "total", tot);
    }
    return 0;
}
