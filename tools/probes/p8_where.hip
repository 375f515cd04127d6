// Where does the time of csrc/gemm_p8.hip (16-bit persistent GEMM) go?  Standalone harness: includes a (possibly patched)
// copy of the kernel source (P8_SRC) and times it on the bf16 training shapes (batch 64).  tools/probes/p8_where.sh builds
// the variants: the product kernel and copies without the epilogue / its stores / the barriers / the DMA (garbage results
// there; only the time matters).
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <stdio.h>
#include <stdlib.h>
#include P8_SRC
#include H16P_SRC   // gemm_h16p.hip: launch_gemm_p8 hands the NT forms to it unless VITSEG_NO_H16P is set

namespace vitseg {
int hip_fail(hipError_t e, const char* what) {
    fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e));
    return 1;
}
void set_error(const char* fmt, ...) { fprintf(stderr, "%s\n", fmt); }
static long g_probe_opts[OPT_COUNT] = {};   // the dispatcher switches of common.hpp, local to this harness
long opt(int id) { return g_probe_opts[id]; }
int launch_splitk_reduce(const float*, float*, size_t, int, hipStream_t) { return 0; }
}   // namespace vitseg

__global__ void fill16_kernel(unsigned short* x, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float f = ((int)(h & 0xffff) - 32768) * (scale / 32768.f);
        x[i] = (unsigned short)(__float_as_uint(f) >> 16);
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 65536;
    struct Shape { const char* name; int N, K, epi; bool aux; } shapes[] = {
        {"qkv (bias, 16-bit out)", 2304, 768, vitseg::EPI_BIAS, false},
        {"fc1 + gelu (+ saved gelu')", 3072, 768, vitseg::EPI_GELU, true},
        {"fc1 + gelu", 3072, 768, vitseg::EPI_GELU, false},
        {"fc2 + residual (fp32 out)", 768, 3072, vitseg::EPI_RESADD, false},
        {"o_proj + residual (fp32 out)", 768, 768, vitseg::EPI_RESADD, false},
        {"fc2 dgrad x gelu' (dgelu)", 3072, 768, vitseg::EPI_DGELU, false},
        {"fc2 dgrad x gelu' + column sums", 3072, 768, vitseg::EPI_DGELU, true},
        {"fc1 dgrad (bias)", 768, 3072, vitseg::EPI_BIAS, false},
        {"qkv dgrad (bias)", 768, 2304, vitseg::EPI_BIAS, false}};
    unsigned short *A, *W, *R16, *AUX;
    float *b, *C;
    const size_t big = (size_t)M * 3072;
    hipMalloc(&A, big * 2); hipMalloc(&W, (size_t)3072 * 3072 * 2); hipMalloc(&R16, big * 2); hipMalloc(&AUX, big * 2);
    hipMalloc(&b, 3072 * 4); hipMalloc(&C, big * 4);
    fill16_kernel<<<2048, 256>>>(A, big, 1u, 1.f);
    fill16_kernel<<<2048, 256>>>(W, (size_t)3072 * 3072, 2u, 0.05f);
    fill16_kernel<<<2048, 256>>>(R16, big, 3u, 1.f);
    hipMemset(b, 0, 3072 * 4);
    hipMemset(C, 0, big * 4);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const bool both = getenv("P8_WHERE_BOTH") != nullptr;
    if (getenv("P8_WHERE_NO_H16P")) vitseg::g_probe_opts[vitseg::OPT_NO_H16P] = 1;   // every NT shape on gemm_p8.hip
    for (int pass = 0; pass < (both ? 2 : 1); ++pass) {
    if (both) {
        vitseg::g_probe_opts[vitseg::OPT_NO_H16P] = pass == 0 ? 1 : 0;
        printf("-- %s\n", pass == 0 ? "gemm_p8 (VITSEG_NO_H16P=1)" : "gemm_h16p");
    }
    for (const Shape& s : shapes) {
        vitseg::GemmArgs a = {};
        a.A = A; a.W = W; a.bias = b; a.C = C;
        a.R = s.epi == vitseg::EPI_RESADD ? C : (s.epi == vitseg::EPI_DGELU ? (const float*)R16 : nullptr);
        a.aux = (s.aux && s.epi == vitseg::EPI_GELU) ? AUX : nullptr;
        a.colsum_scratch = (s.aux && s.epi == vitseg::EPI_DGELU) ? (float*)AUX : nullptr;
        if (s.epi == vitseg::EPI_DGELU) a.bias = nullptr;
        a.M = M; a.N = s.N; a.K = s.K; a.lda = s.K; a.ldc = s.N; a.ldw = s.K;
        // P8_WHERE_LDA=64: the activation rows overlap (row stride 128 B): an 8 MB operand that stays in the L2 / Infinity
        // Cache instead of 100-400 MB streamed from HBM -- same instruction stream, shorter load latency
        if (getenv("P8_WHERE_LDA")) a.lda = atoi(getenv("P8_WHERE_LDA"));
        static unsigned long long* clk = nullptr;   // "clock" variants of p8_where.sh: per-block (shader ticks, 100 MHz ticks)
        if (!clk) { hipMalloc(&clk, 4096 * 16); }
        hipMemset(clk, 0, 4096 * 16);
        a.thin_scratch = (float*)clk;
        float best = 1e9f;
        for (int rnd = 0; rnd < 4; ++rnd) {
            hipEventRecord(e0, 0);
            for (int i = 0; i < 10; ++i)
                if (vitseg::launch_gemm_p8(a, s.epi, 0, false)) return 1;
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            if (rnd && ms / 10 < best) best = ms / 10;
        }
        // per 64-deep K step of a 256 x 256 tile, one tile per CU and round (256 CUs): what the main loop costs
        const double ksteps = (double)((M + 255) / 256) * (s.N / 256) / 256.0 * (s.K / 64);
        printf("%-30s %8.1f us  %7.1f TFLOP/s  %6.3f us per K step (%.0f steps)", s.name, best * 1e3,
               2.0 * M * s.N * s.K / (best * 1e-3) * 1e-12, best * 1e3 / ksteps, ksteps);
        {   // in-kernel clock, if this build stamps it: median over the blocks of the last launch
            static unsigned long long h[512];
            hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
            double r[256];
            int n = 0;
            for (int b = 0; b < 256; ++b)
                if (h[2 * b + 1]) r[n++] = (double)h[2 * b] / (double)h[2 * b + 1] * 0.1;   // GHz
            if (n) {
                for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) if (r[j] < r[i]) { double t = r[i]; r[i] = r[j]; r[j] = t; }
                printf("  clock %.2f GHz (median of %d blocks; %.0f cycles per K step)", r[n / 2], n, best * 1e3 / ksteps * r[n / 2] * 1e3);
            }
        }
        printf("\n");
    }
    }
    {   // weight gradient (TT form): dW[768][3072] over 65536 tokens
        vitseg::GemmArgs a = {};
        a.A = A; a.W = R16; a.C = C; a.M = 768; a.N = 3072; a.K = M; a.lda = 768; a.ldw = 3072; a.ldc = 3072;
        float* scratch;
        hipMalloc(&scratch, (size_t)768 * 3072 * 4 * 16);
        static unsigned long long* clk2 = nullptr;
        if (!clk2) hipMalloc(&clk2, 4096 * 16);
        hipMemset(clk2, 0, 4096 * 16);
        a.thin_scratch = (float*)clk2;
        float best = 1e9f;
        for (int rnd = 0; rnd < 4; ++rnd) {
            hipEventRecord(e0, 0);
            for (int i = 0; i < 10; ++i)
                if (vitseg::launch_wgrad_p8(a, scratch, 0)) return 1;
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            if (rnd && ms / 10 < best) best = ms / 10;
        }
        const int splits = vitseg::wgrad_p8_splits(768, 3072, M);
        const double ksteps = (double)((M + 63) / 64) / splits;   // one item per CU: 64-token steps of a slice
        printf("%-30s %8.1f us  %7.1f TFLOP/s  %6.3f us per K step (%.0f steps, reduce kernel included)", "wgrad fc1 (TT, split-K)", best * 1e3,
               2.0 * M * 768 * 3072 / (best * 1e-3) * 1e-12, best * 1e3 / ksteps, ksteps);
        {
            static unsigned long long h[512];
            hipMemcpy(h, clk2, sizeof(h), hipMemcpyDeviceToHost);
            double r[256], cyc[256];
            int n = 0;
            for (int b = 0; b < 256; ++b)
                if (h[2 * b + 1]) { r[n] = (double)h[2 * b] / (double)h[2 * b + 1] * 0.1; cyc[n++] = (double)h[2 * b]; }
            if (n) {
                for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) if (r[j] < r[i]) { double t = r[i]; r[i] = r[j]; r[j] = t; }
                for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) if (cyc[j] < cyc[i]) { double t = cyc[i]; cyc[i] = cyc[j]; cyc[j] = t; }
                printf("  clock %.2f GHz (median of %d blocks; %.0f shader cycles per K step, kernel only)", r[n / 2], n, cyc[n / 2] / ksteps);
            }
        }
        printf("\n");
    }
    return 0;
}
