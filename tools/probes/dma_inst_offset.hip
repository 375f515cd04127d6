// LDS-DMA (buffer_load_dwordx4 ... lds) with an instruction offset on gfx950: the immediate moves BOTH the memory address
// and the LDS destination (M0 + offset + 16 * lane).  gemm_p8 issues the second 1 KiB piece of a half through it.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/dma_inst_offset.hip -o tools/probes/_build/dma_inst_offset
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* base, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[1024];   // 4 KiB
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1.f;
    __syncthreads();
    const unsigned long long b = (unsigned long long)base;
    i32x4 r;
    r[0] = (int)(unsigned)b;
    r[1] = (int)(unsigned)((b >> 32) & 0xffffu);
    r[2] = 65536;
    r[3] = 0x00020000;
    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned voff = 4096 - 1024 + threadIdx.x * 16, soff = 0;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds\n\ts_waitcnt vmcnt(0)"
                 :: "s"(dst), "v"(voff), "s"(r), "s"(soff) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
    float *d, *o;
    hipMalloc(&d, 65536);
    hipMalloc(&o, 4096);
    float* h = (float*)malloc(65536);
    for (int i = 0; i < 16384; ++i) h[i] = (float)i;
    hipMemcpy(d, h, 65536, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    float r[1024];
    hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
    // expected: floats [256, 512) of the LDS (bytes 1024 .. 2047) = floats 1024 .. 1279 of the source (bytes 4096 ...)
    int ok = 0, untouched = 0, wrong = 0, first_written = -1;
    for (int i = 0; i < 1024; ++i) {
        if (r[i] == -1.f) ++untouched;
        else {
            if (first_written < 0) first_written = i;
            if (i >= 256 && i < 512 && r[i] == (float)(1024 + i - 256)) ++ok; else ++wrong;
        }
    }
    printf("LDS floats written as expected (LDS byte 1024 + 16 lane <- source byte 4096 + 16 lane): %d of 256; untouched %d of 768; wrong %d; "
           "first written float %d (value %g)\n", ok, untouched, wrong, first_written, first_written >= 0 ? r[first_written] : 0.f);
    return ok == 256 && wrong == 0 ? 0 : 1;
}
