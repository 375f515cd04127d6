// Row kernels of the small-batch fp32 route (small.hpp): what sits between two GEMMs of a pre-LN block when the
// producing GEMM leaves K-chunk slabs -- the chunk sum, the bias, the residual add and the next LayerNorm in ONE pass
// over the row (modeling_vit.py:266-286: x = x + attention(ln1(x)); x = x + mlp(ln2(x))), the embedding sum
// (modeling_vit.py:129-161: patch projection + bias + position embedding, CLS rows = cls + pos[0]) -- and the
// seg_head tail behind the 3x3 conv's nine tap slabs (ReLU + 1x1 conv, model/CE/classes.py:240-244).
// Bound: HBM / L2 (a row is read and written once); at 197-1576 rows the launch is latency, not bandwidth.
#include "small.hpp"

namespace vitseg {
namespace {

// One wave per row, the row in registers (NV float4 per lane, D <= 256 NV).
// Summation order (fixed, batch-independent): t = slab 0 + slab 1 + ... ; t += bias; x = residual + t.
// LayerNorm: two-pass, biased variance, the arithmetic of rowops.hip:layernorm_kernel.
template <int NV>
__global__ __launch_bounds__(256) void resln_kernel(const SRows p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int nv = p.D >> 2;
    const bool cls_row = p.embed && row >= p.Mp;
    f32x4 v[NV], wv[NV], bv[NV];
    float* xr = p.X + (size_t)row * p.D;
    // every load of the row is issued up front (out-of-range lanes re-read the last vector and are masked out below)
    f32x4 res[NV], acc[NV];
    const float* rsrc = p.embed ? (cls_row ? p.pos : p.pos + (size_t)(1 + row % p.Np) * p.D)
                                : (p.Xres ? p.Xres + (size_t)row * p.D : xr);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = min(lane + 64 * i, nv - 1);
        res[i] = ((const f32x4*)rsrc)[c];
        acc[i] = cls_row ? ((const f32x4*)p.cls)[c] : ((const f32x4*)(p.partial + (size_t)row * p.D))[c];
        wv[i] = ((const f32x4*)p.lnw)[c];
        bv[i] = ((const f32x4*)p.lnb)[c];
    }
    if (!cls_row) {
        // the remaining slabs five at a time (fc2's six chunks: one group), their loads in flight together with the row's
        // other loads (a dependent round trip per slab made this kernel 8 us at 788 rows), added in slab order
        constexpr int G = NV <= 4 ? 5 : 2;
        for (int s0 = 1; s0 < p.splits; s0 += G) {
            f32x4 t[G][NV];
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const f32x4* ps = (const f32x4*)(p.partial + (size_t)min(s0 + u, p.splits - 1) * p.split_stride + (size_t)row * p.D);
#pragma unroll
                for (int i = 0; i < NV; ++i) t[u][i] = ps[min(lane + 64 * i, nv - 1)];
            }
#pragma unroll
            for (int u = 0; u < G; ++u) {
                if (s0 + u < p.splits) {
#pragma unroll
                    for (int i = 0; i < NV; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][e] += t[u][i][e];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const f32x4 b4 = ((const f32x4*)p.bias)[min(lane + 64 * i, nv - 1)];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][e] += b4[e];
        }
        if (p.drop.thresh && !p.embed) {   // the branch is dropped, not the residual (the mask of gemm.hip's EPI_RESADD: same key, same bits)
            const unsigned key = drop_key(p.drop.seed, p.drop.stream, (unsigned)row);
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[i][e] = drop_keep(key, (unsigned)(4 * (lane + 64 * i) + e), p.drop.thresh) ? acc[i][e] * p.drop.scale : 0.f;
        }
    }
    // embedding form in training: the SUM is dropped, CLS rows included (dropout(embeddings), modeling_vit.py:159; the mask of
    // dropout_rows_kernel: key (seed, stream, row), element = column)
    const bool drop_sum = p.embed && p.drop.thresh;
    const unsigned ekey = drop_key(p.drop.seed, p.drop.stream, (unsigned)row);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[i][e] = res[i][e] + acc[i][e];
            if (drop_sum) v[i][e] = drop_keep(ekey, (unsigned)(4 * (lane + 64 * i) + e), p.drop.thresh) ? v[i][e] * p.drop.scale : 0.f;
        }
        if (lane + 64 * i < nv) ((f32x4*)xr)[lane + 64 * i] = v[i];
    }
    if (row >= p.ln_rows) return;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float t = (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        s += (lane + 64 * i < nv) ? t : 0.f;
    }
    const float mean = wave_sum(s) / (float)p.D;
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
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)p.D + p.eps);
    float* hr = p.H + (size_t)row * p.D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (lane + 64 * i < nv) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[i][e] * rstd * wv[i][e] + bv[i][e];
            if (p.h_fmt == 0) {
                ((f32x4*)hr)[lane + 64 * i] = o;
            } else {   // the 16-bit route: H is the next GEMM's operand
                uint2 h;
                if (p.h_fmt == 2) {
                    h.x = H16<f16_t>::pack2(o[0], o[1]);
                    h.y = H16<f16_t>::pack2(o[2], o[3]);
                } else {
                    h.x = pack2_bf16(o[0], o[1]);
                    h.y = pack2_bf16(o[2], o[3]);
                }
                ((uint2*)((unsigned short*)p.H + (size_t)row * p.D))[lane + 64 * i] = h;
            }
        }
    }
}

__global__ __launch_bounds__(256) void slabsum_kernel(const float* __restrict__ partial, size_t stride, int splits,
                                                      float* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 v = ((const f32x4*)partial)[i];
        for (int s = 1; s < splits; ++s) {
            const f32x4 t = ((const f32x4*)(partial + (size_t)s * stride))[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += t[e];
        }
        ((f32x4*)out)[i] = v;
    }
}

constexpr int MID = 256;   // seg_head.0 output channels (model/CE/classes.py:241)

// One wave per pixel: F = relu(tap slab 0 + ... + tap slab 8 + b0) (256 values = 64 lanes x 4), then the C class rows of
// seg_head.2 against it (wave reductions), Z[b, c, y, x] NCHW as head1x1_kernel writes it.
__global__ __launch_bounds__(256) void headfin_kernel(const float* __restrict__ partial, size_t split_stride,
                                                      const float* __restrict__ b0, const float* __restrict__ W2,
                                                      const float* __restrict__ b2, float* __restrict__ Z, int B, int Np, int C,
                                                      float* __restrict__ F_out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * Np) return;
    f32x4 t[9];
#pragma unroll
    for (int s = 0; s < 9; ++s) t[s] = ((const f32x4*)(partial + (size_t)s * split_stride + (size_t)row * MID))[lane];
    f32x4 f = t[0];
#pragma unroll
    for (int s = 1; s < 9; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] += t[s][e];
    const f32x4 bb = ((const f32x4*)b0)[lane];
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = fmaxf(f[e] + bb[e], 0.f);
    if (F_out) ((f32x4*)(F_out + (size_t)row * MID))[lane] = f;
    const int b = row / Np, tok = row - b * Np;
    for (int c = 0; c < C; ++c) {
        const f32x4 w = ((const f32x4*)(W2 + (size_t)c * MID))[lane];
        float s = (f[0] * w[0] + f[1] * w[1]) + (f[2] * w[2] + f[3] * w[3]);
        s = wave_sum(s);
        if (lane == 0) Z[((size_t)b * C + c) * Np + tok] = s + b2[c];
    }
}

}  // namespace

int launch_resln(const SRows& a, hipStream_t s) {
    VITSEG_CHECK_ARG(a.X && a.partial && a.bias && a.lnw && a.lnb && a.H && a.rows > 0 && a.splits >= 1, VITSEG_EINVAL, "resln: bad arguments");
    VITSEG_CHECK_ARG(a.D % 4 == 0 && a.D <= 2048, VITSEG_ESHAPE, "resln: D=%d must be a multiple of 4 and <= 2048", a.D);
    VITSEG_CHECK_ARG(!a.embed || (a.pos && a.cls), VITSEG_EINVAL, "resln: the embedding form needs pos and cls");
    const dim3 grid((a.rows + 3) / 4);
    const int nvl = (a.D / 4 + 63) / 64;
    if (nvl <= 1) hipLaunchKernelGGL(resln_kernel<1>, grid, dim3(256), 0, s, a);
    else if (nvl <= 2) hipLaunchKernelGGL(resln_kernel<2>, grid, dim3(256), 0, s, a);
    else if (nvl <= 3) hipLaunchKernelGGL(resln_kernel<3>, grid, dim3(256), 0, s, a);
    else if (nvl <= 4) hipLaunchKernelGGL(resln_kernel<4>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(resln_kernel<8>, grid, dim3(256), 0, s, a);
    VITSEG_LAUNCH_CHECK("resln");
    return VITSEG_OK;
}

int launch_slabsum(const float* partial, size_t split_stride, int splits, float* out, size_t n, hipStream_t s) {
    VITSEG_CHECK_ARG(partial && out && splits >= 1 && n % 4 == 0 && split_stride % 4 == 0, VITSEG_EINVAL, "slabsum: bad arguments");
    const size_t n4 = n / 4;
    const unsigned blocks = (unsigned)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(slabsum_kernel, dim3(blocks), dim3(256), 0, s, partial, split_stride, splits, out, n4);
    VITSEG_LAUNCH_CHECK("slabsum");
    return VITSEG_OK;
}

int launch_headfin(const float* partial, size_t split_stride, const float* b0, const float* W2, const float* b2, float* Z,
                   int B, int Np, int C, hipStream_t s, float* F_out) {
    VITSEG_CHECK_ARG(partial && b0 && W2 && b2 && Z && B > 0 && Np > 0 && C > 0, VITSEG_EINVAL, "headfin: bad arguments");
    hipLaunchKernelGGL(headfin_kernel, dim3((B * Np + 3) / 4), dim3(256), 0, s, partial, split_stride, b0, W2, b2, Z, B, Np, C, F_out);
    VITSEG_LAUNCH_CHECK("headfin");
    return VITSEG_OK;
}

}  // namespace vitseg
