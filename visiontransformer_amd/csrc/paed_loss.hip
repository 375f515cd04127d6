// Soft PAED loss for C classes, forward value and gradient w.r.t. the logits in one call.
//
// Replaces, in the 17-class LightningViTModel of model/PAED (classes.py:449-478), the chain
//   probs = softmax(logits, 1); onehot = one_hot(y); paed_loss_multiclass_soft(onehot, probs)   (classes.py:336-369)
// and its autograd: both maps are blurred per class with a normalised (6 sigma + 1)^2 Gaussian (zero padding), the L1
// difference is weighted by 2 * onehot * (1 - probs) and averaged over space, classes and batch.
//
// The blur is linear, so only E = G * (onehot - probs) is needed; the 2-D Gaussian is the outer product of the
// normalised 1-D one, so it runs as a row pass and a column pass (19 taps each instead of 361).  With
// w = 2 t (1 - p) (or 1 without the class penalty) and k = 1 / (B C H W):
//   loss      = k * sum w |E|
//   dL/dp     = k * ( -2 t |E|  -  G * (w sign E) )          (G is symmetric: its adjoint is itself)
//   dL/dz_c   = p_c (g_c - sum_j g_j p_j)                    (softmax backward)
// Everything is HBM-bound elementwise / short-stencil work on [B, C, H, W] fp32 maps (3.4 MB per image at 17 x 224^2);
// loss partials are summed in fp64 in a fixed order (deterministic).
#include <math.h>

#include "kernels.hpp"

namespace vitseg {
namespace {

constexpr int MAX_TAPS = 64;
struct Taps {
    int k;            // number of taps, odd
    float w[MAX_TAPS];
};

template <typename TargetT>
__global__ __launch_bounds__(256) void paed_softmax_diff_kernel(const float* __restrict__ logits,
                                                                const TargetT* __restrict__ target, float* __restrict__ P,
                                                                float* __restrict__ D, int B, int C, size_t plane) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * plane) return;
    const size_t b = i / plane, px = i - b * plane;
    const float* z = logits + b * C * plane + px;
    const int t = (int)target[i];
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, z[(size_t)c * plane]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(z[(size_t)c * plane] - m);
    const float inv = 1.0f / s;
    for (int c = 0; c < C; ++c) {
        const float p = expf(z[(size_t)c * plane] - m) * inv;
        const size_t o = b * C * plane + (size_t)c * plane + px;
        P[o] = p;
        D[o] = (c == t ? 1.f : 0.f) - p;
    }
}

// dst = 1-D Gaussian of src along x (ALONG_X) or y, zero padding; planes = B * C images of H x W
template <bool ALONG_X>
__global__ __launch_bounds__(256) void paed_blur_kernel(const float* __restrict__ src, float* __restrict__ dst, Taps g,
                                                        int H, int W, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const int r = g.k >> 1;
    float acc = 0.f;
    if (ALONG_X) {
        const float* row = src + (i - x);
        for (int j = 0; j < g.k; ++j) {
            const int xx = x + j - r;
            if ((unsigned)xx < (unsigned)W) acc = fmaf(row[xx], g.w[j], acc);
        }
    } else {
        const float* col = src + (i - (size_t)y * W);
        for (int j = 0; j < g.k; ++j) {
            const int yy = y + j - r;
            if ((unsigned)yy < (unsigned)H) acc = fmaf(col[(size_t)yy * W], g.w[j], acc);
        }
    }
    dst[i] = acc;
}

// loss partials of w |E| and, when U != nullptr, U = w sign(E)
template <typename TargetT>
__global__ __launch_bounds__(256) void paed_reduce_kernel(const float* __restrict__ E, const float* __restrict__ P,
                                                          const TargetT* __restrict__ target, float* __restrict__ U,
                                                          double* __restrict__ partial, int C, size_t plane, size_t total,
                                                          int class_penalty) {
    __shared__ double red[4];
    double local = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t bc = i / plane, px = i - bc * plane;
        const int c = (int)(bc % C);
        const size_t b = bc / C;
        const float e = E[i];
        float w = 1.f;
        if (class_penalty) w = ((int)target[b * plane + px] == c) ? 2.f * (1.f - P[i]) : 0.f;
        local += (double)(w * fabsf(e));
        if (U) U[i] = e > 0.f ? w : (e < 0.f ? -w : 0.f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void paed_finish_kernel(const double* __restrict__ partial, int n, double inv_count,
                                                          float* __restrict__ loss) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(red[0] * inv_count);
}

// grad_logits = softmax backward of g_c = k (-2 t_c |E_c| [class penalty] - V_c)
template <typename TargetT>
__global__ __launch_bounds__(256) void paed_grad_kernel(const float* __restrict__ P, const float* __restrict__ E,
                                                        const float* __restrict__ V, const TargetT* __restrict__ target,
                                                        float* __restrict__ G, int B, int C, size_t plane, float k,
                                                        int class_penalty) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * plane) return;
    const size_t b = i / plane, px = i - b * plane;
    const size_t base = b * C * plane + px;
    const int t = (int)target[i];
    float dot = 0.f;
    for (int c = 0; c < C; ++c) {
        const size_t o = base + (size_t)c * plane;
        float g = -V[o];
        if (class_penalty && c == t) g -= 2.f * fabsf(E[o]);
        dot = fmaf(g * k, P[o], dot);
    }
    for (int c = 0; c < C; ++c) {
        const size_t o = base + (size_t)c * plane;
        float g = -V[o];
        if (class_penalty && c == t) g -= 2.f * fabsf(E[o]);
        G[o] = P[o] * (g * k - dot);
    }
}

inline int reduce_blocks(size_t total) {
    const size_t b = (total + 256 * 8 - 1) / (256 * 8);
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

size_t paed_scratch_bytes(int B, int C, int H, int W) {
    const size_t total = (size_t)B * C * H * W;
    return 4 * total * sizeof(float) + (size_t)reduce_blocks(total) * sizeof(double) + 256;
}

int launch_paed_multiclass(const float* logits, const void* target, int target_is_u8, int B, int C, int H, int W, float sigma,
                           int class_penalty, void* scratch, float* loss, float* grad_logits, hipStream_t s) {
    VITSEG_CHECK_ARG(logits && target && scratch && loss && B > 0 && C > 0 && H > 0 && W > 0, VITSEG_EINVAL,
                     "paed_multiclass: bad arguments");
    Taps g;
    g.k = (int)(6 * sigma + 1);   // classes.py:345 kernel_size = int(6 * sigma + 1)
    VITSEG_CHECK_ARG(sigma > 0.f && (g.k & 1) && g.k <= MAX_TAPS, VITSEG_ESHAPE,
                     "paed_multiclass: sigma %f gives %d taps (odd, <= %d supported)", sigma, g.k, MAX_TAPS);
    float sum = 0.f;
    for (int j = 0; j < g.k; ++j) {
        const float a = (float)(j - g.k / 2);
        g.w[j] = expf(-(a * a) / (2.f * sigma * sigma));
        sum += g.w[j];
    }
    for (int j = 0; j < g.k; ++j) g.w[j] /= sum;   // (g1 x g1) / sum(g1 x g1) = (g1 / sum g1) x (g1 / sum g1)
    const size_t plane = (size_t)H * W, total = (size_t)B * C * plane;
    float* P = (float*)scratch;
    float* E = P + total;
    float* U = E + total;
    float* T = U + total;
    double* partial = (double*)(((uintptr_t)(T + total) + 255) & ~(uintptr_t)255);
    const unsigned px_blocks = (unsigned)(((size_t)B * plane + 255) / 256), el_blocks = (unsigned)((total + 255) / 256);
    const int rb = reduce_blocks(total);
    const double inv = 1.0 / (double)total;
#define TGT(T_) (const T_*)target
    if (target_is_u8)
        hipLaunchKernelGGL(paed_softmax_diff_kernel<uint8_t>, dim3(px_blocks), dim3(256), 0, s, logits, TGT(uint8_t), P, U, B, C,
                           plane);
    else
        hipLaunchKernelGGL(paed_softmax_diff_kernel<long long>, dim3(px_blocks), dim3(256), 0, s, logits, TGT(long long), P, U,
                           B, C, plane);
    VITSEG_LAUNCH_CHECK("paed_softmax_diff");
    hipLaunchKernelGGL(paed_blur_kernel<true>, dim3(el_blocks), dim3(256), 0, s, U, T, g, H, W, total);
    hipLaunchKernelGGL(paed_blur_kernel<false>, dim3(el_blocks), dim3(256), 0, s, T, E, g, H, W, total);
    VITSEG_LAUNCH_CHECK("paed_blur");
    float* Uout = grad_logits ? U : nullptr;
    if (target_is_u8)
        hipLaunchKernelGGL(paed_reduce_kernel<uint8_t>, dim3(rb), dim3(256), 0, s, E, P, TGT(uint8_t), Uout, partial, C, plane,
                           total, class_penalty);
    else
        hipLaunchKernelGGL(paed_reduce_kernel<long long>, dim3(rb), dim3(256), 0, s, E, P, TGT(long long), Uout, partial, C,
                           plane, total, class_penalty);
    VITSEG_LAUNCH_CHECK("paed_reduce");
    hipLaunchKernelGGL(paed_finish_kernel, dim3(1), dim3(256), 0, s, partial, rb, inv, loss);
    VITSEG_LAUNCH_CHECK("paed_finish");
    if (!grad_logits) return VITSEG_OK;
    hipLaunchKernelGGL(paed_blur_kernel<true>, dim3(el_blocks), dim3(256), 0, s, U, T, g, H, W, total);
    hipLaunchKernelGGL(paed_blur_kernel<false>, dim3(el_blocks), dim3(256), 0, s, T, U, g, H, W, total);   // U := V
    VITSEG_LAUNCH_CHECK("paed_blur(adjoint)");
    if (target_is_u8)
        hipLaunchKernelGGL(paed_grad_kernel<uint8_t>, dim3(px_blocks), dim3(256), 0, s, P, E, U, TGT(uint8_t), grad_logits, B, C,
                           plane, (float)inv, class_penalty);
    else
        hipLaunchKernelGGL(paed_grad_kernel<long long>, dim3(px_blocks), dim3(256), 0, s, P, E, U, TGT(long long), grad_logits,
                           B, C, plane, (float)inv, class_penalty);
    VITSEG_LAUNCH_CHECK("paed_grad");
#undef TGT
    return VITSEG_OK;
}

}  // namespace vitseg

extern "C" {

size_t vitseg_paed_scratch_bytes(int batch, int C, int H, int W) { return vitseg::paed_scratch_bytes(batch, C, H, W); }

int vitseg_paed_multiclass_loss(const float* logits, const void* target, int target_is_u8, int batch, int C, int H, int W,
                                float sigma, int class_penalty, void* scratch, float* loss, float* grad_logits, void* stream) {
    return vitseg::launch_paed_multiclass(logits, target, target_is_u8, batch, C, H, W, sigma, class_penalty, scratch, loss,
                                          grad_logits, (hipStream_t)stream);
}

}  // extern "C"
