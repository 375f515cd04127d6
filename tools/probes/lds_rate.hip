// LDS read INSTRUCTION throughput of one CU (gfx950): how many cycles does the LDS pipe spend per wave-level
// ds_read_b128 / ds_read_b64 / ds_read_b64_tr_b16 when 4 or 8 waves of a block issue nothing else?  (The weight-gradient
// GEMM gathers both operands with the transposing read: 2x the instruction count of the b128 form for the same bytes.)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_rate.hip -o tools/probes/_build/lds_rate && tools/probes/_build/lds_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int NR = 16;   // reads in flight per wave between two waits
template <int KIND>   // 0 b128, 1 b64, 2 b64_tr_b16
__global__ void probe(unsigned long long* cyc, float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((float*)lds)[i] = (float)i;   // 64 KiB
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // conflict-free: consecutive lanes read consecutive 16 / 8 bytes; every wave its own 8 KiB window
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (wave & 7) * 8192 + lane * (KIND == 0 ? 16 : 8);
    float acc = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 r[NR];
#pragma unroll
            for (int j = 0; j < NR; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[j]) : "v"(a), "n"(0) );
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < NR; ++j) acc += r[j][0];
        } else {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 r[NR];
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                if (KIND == 1) asm volatile("ds_read_b64 %0, %1" : "=v"(r[j]) : "v"(a));
                else asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r[j]) : "v"(a));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < NR; ++j) acc += r[j][0];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
    if (acc == 123.456f) sink[0] = acc;
}

template <int KIND>
int run(const char* name, int waves, unsigned long long* d_cyc, float* d_sink) {
    const int iters = 20000, blocks = 256;
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(64 * waves), 65536, 0, d_cyc, d_sink, iters);
    CK(hipDeviceSynchronize());
    unsigned long long h[256 * 16];
    CK(hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost));
    double s = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) s += (double)h[b * 16 + w];
    const double per_wave = s / (blocks * waves);               // cycles one wave needed for iters * NR reads
    const double per_instr_cu = per_wave / ((double)iters * NR * waves);   // LDS cycles per wave instruction, whole CU
    const int bytes = KIND == 0 ? 1024 : 512;
    printf("%-20s %d waves per CU: %6.2f cycles per wave instruction per CU = %6.1f B/clk/CU\n", name, waves, per_instr_cu,
           bytes / per_instr_cu);
    return 0;
}

int main() {
    unsigned long long* d_cyc;
    float* d_sink;
    CK(hipMalloc(&d_cyc, 256 * 16 * 8));
    CK(hipMalloc(&d_sink, 4));
    for (int waves = 4; waves <= 16; waves *= 2) {
        if (run<0>("ds_read_b128", waves, d_cyc, d_sink)) return 1;
        if (run<1>("ds_read_b64", waves, d_cyc, d_sink)) return 1;
        if (run<2>("ds_read_b64_tr_b16", waves, d_cyc, d_sink)) return 1;
    }
    return 0;
}
