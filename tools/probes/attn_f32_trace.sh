#!/bin/bash
# s_memtime stamps in the key loop of the fp32 attention kernel (copy of csrc/attention_f32.hip under
# gpurun_out/attn_f32_trace/): mean cycles between consecutive stamps for the four waves of one block.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/attn_f32_trace
mkdir -p $out
python3 - visiontransformer_amd/csrc/attention_f32.hip "$out" <<'PY'
import os, sys
src, out = sys.argv[1], sys.argv[2]
s = open(src).read().replace('#include "kernels.hpp"', '#include "%s/visiontransformer_amd/csrc/kernels.hpp"' % os.getcwd())
def rep(t, a, b):
    assert t.count(a) == 1, (t.count(a), a)
    return t.replace(a, b)
ST = "        if (blockIdx.x == 200 && lane == 0 && kt < 16) g_trace[(wave * 16 + kt) * 8 + %d] = __builtin_readcyclecounter();\n"
s = rep(s, "        const int buf = kt & 1;\n        gload(min(kt + 1, nkt - 1));", ST % 0 + "        const int buf = kt & 1;\n        gload(min(kt + 1, nkt - 1));")
s = rep(s, "        scores(negm);\n        float psum = 0.f;\n", "        scores(negm);\n" + ST % 1 + "        float psum = 0.f;\n")
s = rep(s, "        l_run += psum;\n", "        l_run += psum;\n" + ST % 2)
s = rep(s, "        swrite(buf ^ 1);\n        __syncthreads();\n    }\n\n    // ---- normalise and store", ST % 3 + "        swrite(buf ^ 1);\n" + ST % 4 + "        __syncthreads();\n    }\n\n    // ---- normalise and store")
open(out + "/traced.hip", "w").write(s)
PY
cat > $out/main.hip <<'CPP'
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <stdio.h>
#include <vector>
__device__ unsigned long long g_trace[4 * 16 * 8];
#include "traced.hip"
namespace vitseg {
int hip_fail(hipError_t e, const char* what) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return 1; }
void set_error(const char* fmt, ...) { fprintf(stderr, "%s\n", fmt); }
int launch_attention_x3_main(const float*, float*, int, int, int, hipStream_t) { return 1; }
}
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
    hipMalloc(&qkv, rows * 3 * D * 4); hipMalloc(&ctx, rows * D * 4);
    fill_kernel<<<2048, 256>>>(qkv, rows * 3 * D, 1u, 2.f);
    vitseg::DropArgs dr = {};
    for (int i = 0; i < 3; ++i) if (vitseg::launch_attention_f32(qkv, ctx, nullptr, B, Np, A, dr, 0, false)) return 1;
    hipDeviceSynchronize();
    std::vector<unsigned long long> t(4 * 16 * 8);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_trace), t.size() * 8);
    const char* names[] = {"tile start -> QK MFMAs issued (incl. K/V global loads)", "-> exp / row sums done", "-> PV MFMAs issued",
                           "-> K/V of the next tile written to LDS", "-> past the barrier (next tile start)"};
    for (int w = 0; w < 4; ++w) {
        printf("wave %d: cycles per 64-key tile (mean over tiles 2..13)\n", w);
        double tot = 0;
        for (int k = 0; k < 5; ++k) {
            double s = 0;
            for (int tile = 2; tile < 14; ++tile) {
                const unsigned long long a = t[(w * 16 + tile) * 8 + k], b = k < 4 ? t[(w * 16 + tile) * 8 + k + 1] : t[(w * 16 + tile + 1) * 8];
                s += (double)(b - a);
            }
            printf("   %-56s %8.0f\n", names[k], s / 12); tot += s / 12;
        }
        printf("   %-56s %8.0f\n", "total", tot);
    }
    return 0;
}
CPP
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$out $out/main.hip -o $out/traced 2> $out/traced.err || { cat $out/traced.err; exit 1; }
if [ -z "$TRACE_BUILD_ONLY" ]; then $out/traced; fi
