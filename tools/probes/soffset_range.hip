// Does the range check of a raw buffer load (stride 0) on gfx950 include the scalar offset?  A 64 KiB allocation of 1.0f,
// a descriptor whose num_records covers only the first 1 KiB, 64 lanes x 16 bytes at voffset = 16 * lane, soffset swept:
// a lane whose voffset + soffset + 16 exceeds num_records returns 0 if the scalar offset is part of the check.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/soffset_range.hip -o tools/probes/_build/soffset_range
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* base, float* out, unsigned soff, int to_lds) {
    __shared__ __attribute__((aligned(16))) float lds[256];
    const unsigned long long b = (unsigned long long)base;
    i32x4 r;
    r[0] = (int)(unsigned)b;
    r[1] = (int)(unsigned)((b >> 32) & 0xffffu);
    r[2] = 1024;
    r[3] = 0x00020000;
    const unsigned voff = threadIdx.x * 16;
    f32x4 v = {-1.f, -1.f, -1.f, -1.f};
    if (!to_lds) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(voff), "s"(r), "s"(soff) : "memory");
    } else {   // the LDS-DMA form the GEMMs use
        lds[threadIdx.x * 4] = lds[threadIdx.x * 4 + 1] = lds[threadIdx.x * 4 + 2] = lds[threadIdx.x * 4 + 3] = -1.f;
        __syncthreads();
        const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_waitcnt vmcnt(0)"
                     :: "s"(dst), "v"(voff), "s"(r), "s"(soff) : "memory");
        __syncthreads();
        v = *(const f32x4*)&lds[threadIdx.x * 4];
    }
    out[threadIdx.x] = v[0] + v[1] + v[2] + v[3];
}
int main() {
    float *d, *o;
    hipMalloc(&d, 65536);
    hipMalloc(&o, 256);
    float* h = (float*)malloc(65536);
    for (int i = 0; i < 16384; ++i) h[i] = 1.0f;
    hipMemcpy(d, h, 65536, hipMemcpyHostToDevice);
    const unsigned soffs[] = {0, 512, 1008, 1024, 2048, 0x7fffff00u};
    for (int lds = 0; lds < 2; ++lds)
        for (unsigned so : soffs) {
            if (so > 60000) continue;   // stay inside the allocation whatever the answer is
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, so, lds);
            float r[64];
            hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
            int in = 0, zero = 0, other = 0;
            for (int l = 0; l < 64; ++l) { if (r[l] == 4.f) ++in; else if (r[l] == 0.f) ++zero; else ++other; }
            const int expect_in = so >= 1024 ? 0 : (int)((1024 - so) / 16);
            printf("%s soffset %5u: %2d lanes got data, %2d zeros, %d other  (if soffset is range-checked: %d with data)\n",
                   lds ? "lds-dma" : "vgpr   ", so, in, zero, other, expect_in);
        }
    return 0;
}
