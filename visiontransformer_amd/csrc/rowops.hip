// HBM-bound row kernels: LayerNorm, CLS-row initialisation, fp32 -> bf16 cast.
#include "kernels.hpp"

namespace vitseg {
namespace {

// nn.LayerNorm(D, eps=1e-12) (transformers/models/vit/modeling_vit.py:261-262,274,281,348,385;
// eps from configuration_vit.py:58).  Biased variance, two-pass in registers (a row of up to
// 2048 floats lives in one wave's registers), row reductions by wavefront butterfly.
// One wave per row, 4 rows per block.  Bound: HBM (read D*4 + write D*{4,2} bytes per row).
// NV = float4 vectors per lane (D <= 256*NV).  All loads of a row (x, then w and b) are issued
// unconditionally up front -- out-of-range lanes re-read the last vector and are masked out of the
// sums -- so a wave has its whole row in flight at once instead of one dependent load per branch.
template <typename OutT, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, OutT* __restrict__ y, int rows,
                                                        int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = D >> 2;
    const f32x4* xr = (const f32x4*)(x + (size_t)row * D);
    f32x4 v[NV], wv[NV], bv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = xr[min(lane + 64 * i, nv - 1)];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        wv[i] = ((const f32x4*)w)[min(lane + 64 * i, nv - 1)];
        bv[i] = ((const f32x4*)b)[min(lane + 64 * i, nv - 1)];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float t = (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        s += (lane + 64 * i < nv) ? t : 0.f;
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[i][e] -= mean;
            t = fmaf(v[i][e], v[i][e], t);
        }
        q += (lane + 64 * i < nv) ? t : 0.f;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[i][e] * rstd * wv[i][e] + bv[i][e];
            if constexpr (sizeof(OutT) == 4) {
                ((f32x4*)(y + (size_t)row * D))[c] = o;
            } else {
                uint2 h;
                h.x = H16<OutT>::pack2(o[0], o[1]);
                h.y = H16<OutT>::pack2(o[2], o[3]);
                ((uint2*)(y + (size_t)row * D))[c] = h;
            }
        }
    }
}

// embeddings: CLS rows  X[B*Np + b][:] = cls_token + position_embeddings[0]
// (transformers/models/vit/modeling_vit.py:146-158)
__global__ void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ X,
                                int B, int Np, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    const int b = i / D, d = i - b * D;
    X[((size_t)B * Np + b) * D + d] = cls[d] + pos[d];
}

template <typename H>
__global__ void cast_bf16_kernel(const float* __restrict__ src, H* __restrict__ dst, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        const f32x4 v = ((const f32x4*)src)[i];
        uint2 h;
        h.x = H16<H>::pack2(v[0], v[1]);
        h.y = H16<H>::pack2(v[2], v[3]);
        ((uint2*)dst)[i] = h;
    }
}

__global__ void cast_split_kernel(const float* __restrict__ src, uint4* __restrict__ dst, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        const f32x4 v = ((const f32x4*)src)[i];
        const _Float16 h0 = (_Float16)v[0], h1 = (_Float16)v[1], h2 = (_Float16)v[2], h3 = (_Float16)v[3];
        uint4 o;
        o.x = __builtin_bit_cast(unsigned, f16x2{h0, h1});
        o.y = __builtin_bit_cast(unsigned, f16x2{h2, h3});
        o.z = H16<f16_t>::pack2((v[0] - (float)h0) * 2048.f, (v[1] - (float)h1) * 2048.f);
        o.w = H16<f16_t>::pack2((v[2] - (float)h2) * 2048.f, (v[3] - (float)h3) * 2048.f);
        dst[i] = o;
    }
}

template <typename OutT>
__global__ __launch_bounds__(256) void dropout_rows_kernel(const float* __restrict__ src, OutT* __restrict__ dst,
                                                           int rows, int cols4, DropArgs d) {
    const size_t total = (size_t)rows * cols4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const unsigned row = (unsigned)(i / cols4), c0 = (unsigned)(i % cols4) * 4;
        const unsigned key = drop_key(d.seed, d.stream, row);
        f32x4 v = ((const f32x4*)src)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = drop_keep(key, c0 + e, d.thresh) ? v[e] * d.scale : 0.f;
        if constexpr (sizeof(OutT) == 4) {
            ((f32x4*)dst)[i] = v;
        } else {
            uint2 h;
            h.x = pack2_bf16(v[0], v[1]);
            h.y = pack2_bf16(v[2], v[3]);
            ((uint2*)dst)[i] = h;
        }
    }
}

}  // namespace

int launch_dropout_rows(const float* src, void* dst, int dst_bf16, int rows, int cols, DropArgs d, hipStream_t s) {
    VITSEG_CHECK_ARG(cols % 4 == 0, VITSEG_EINVAL, "dropout_rows: cols %% 4");
    const size_t n4 = (size_t)rows * (cols / 4);
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    if (dst_bf16)
        hipLaunchKernelGGL(dropout_rows_kernel<unsigned short>, dim3(blocks), dim3(256), 0, s, src, (unsigned short*)dst,
                           rows, cols / 4, d);
    else
        hipLaunchKernelGGL(dropout_rows_kernel<float>, dim3(blocks), dim3(256), 0, s, src, (float*)dst, rows, cols / 4, d);
    VITSEG_LAUNCH_CHECK("dropout_rows");
    return VITSEG_OK;
}

int launch_layernorm(const float* x, const float* w, const float* b, void* y, int rows, int D, float eps,
                     int out_fmt, hipStream_t s) {
    VITSEG_CHECK_ARG(x && w && b && y && rows > 0, VITSEG_EINVAL, "layernorm: bad arguments");
    VITSEG_CHECK_ARG(D % 4 == 0 && D <= 2048, VITSEG_ESHAPE, "layernorm: D=%d must be a multiple of 4 and <= 2048", D);
    const dim3 grid((rows + 3) / 4);
    const int nvl = (D / 4 + 63) / 64;  // vectors per lane
#define VITSEG_LN(NV)                                                                                              \
    do {                                                                                                           \
        if (out_fmt == 1)                                                                                          \
            hipLaunchKernelGGL((layernorm_kernel<bf16_t, NV>), grid, dim3(256), 0, s, x, w, b, (bf16_t*)y, rows, D, \
                               eps);                                                                               \
        else if (out_fmt == 2)                                                                                     \
            hipLaunchKernelGGL((layernorm_kernel<f16_t, NV>), grid, dim3(256), 0, s, x, w, b, (f16_t*)y, rows, D,  \
                               eps);                                                                               \
        else                                                                                                       \
            hipLaunchKernelGGL((layernorm_kernel<float, NV>), grid, dim3(256), 0, s, x, w, b, (float*)y, rows, D,  \
                               eps);                                                                               \
    } while (0)
    if (nvl <= 1) VITSEG_LN(1);
    else if (nvl <= 2) VITSEG_LN(2);
    else if (nvl <= 3) VITSEG_LN(3);
    else if (nvl <= 4) VITSEG_LN(4);
    else VITSEG_LN(8);
#undef VITSEG_LN
    VITSEG_LAUNCH_CHECK("layernorm");
    return VITSEG_OK;
}

int launch_cls_rows(const float* cls, const float* pos, float* X, int B, int Np, int D, hipStream_t s) {
    hipLaunchKernelGGL(cls_rows_kernel, dim3((B * D + 255) / 256), dim3(256), 0, s, cls, pos, X, B, Np, D);
    VITSEG_LAUNCH_CHECK("cls_rows");
    return VITSEG_OK;
}

int launch_cast_split(const float* src, void* dst, size_t n, hipStream_t s) {
    VITSEG_CHECK_ARG(n % 4 == 0, VITSEG_EINVAL, "cast_split: n %% 4");
    const size_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(cast_split_kernel, dim3(blocks), dim3(256), 0, s, src, (uint4*)dst, n4);
    VITSEG_LAUNCH_CHECK("cast_split");
    return VITSEG_OK;
}

int launch_cast_bf16(const float* src, void* dst, size_t n, hipStream_t s, bool f16) {
    VITSEG_CHECK_ARG(n % 4 == 0, VITSEG_EINVAL, "cast_bf16: n %% 4");
    const size_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    if (f16)
        hipLaunchKernelGGL(cast_bf16_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, src, (f16_t*)dst, n4);
    else
        hipLaunchKernelGGL(cast_bf16_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, src, (bf16_t*)dst, n4);
    VITSEG_LAUNCH_CHECK("cast_bf16");
    return VITSEG_OK;
}

}  // namespace vitseg
