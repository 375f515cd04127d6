// The steps either side of the model in the reference's scripts, on the device:
//
//  f3  transforms.Resize((S, S)) + transforms.ToTensor() on a PIL RGB image
//      (model/CE/trainCurrentViTmodel.py:48-51, model/CE/testViTModel.py:92-97): Pillow's antialiased two-pass
//      triangle-filter resampling in 8-bit fixed point (libImaging/Resample.c), reproduced bit for bit --
//      coefficient tables are computed on the host exactly as precompute_coeffs / normalize_coeffs_8bpc do
//      (IEEE double, no contraction), the passes run here: horizontal into a uint8 intermediate, vertical fused
//      with the /255 and the HWC -> CHW transposition.
//      Mask side: Resize(NEAREST) + value -> class remap + F.interpolate(nearest) (model/CE/classes.py:76-83,
//      273-274) as one gather through host-made index tables and a 256-entry LUT.
//  f4  per-image class statistics behind accuracy / IoU / Dice / class sets
//      (model/CE/datasetTestViTmodel.py:193-219): the ground truth is nearest-resized on the fly, counts are
//      accumulated as integers (LDS histograms, 64-bit global atomics), so the derived metrics are exact.
//
// All three are byte/integer work bound by HBM: one pass over the source, coalesced along x.
#include <math.h>

#include "kernels.hpp"

namespace vitseg {
namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;  // Resample.c

__device__ __forceinline__ unsigned char clip8(int v) {
    v >>= PRECISION_BITS;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// tmp[n][rows][S][3] = horizontal pass of img[n][H][W][3] rows [row_first, row_first + rows).
// A block owns R consecutive source rows: their bytes are one contiguous span of the image, pulled into LDS with
// aligned 16-byte loads (the taps of neighbouring outputs overlap ~2x `scale` pixels, and per-lane byte gathers from
// global memory touch a dozen cache lines per instruction: 1.0 TB/s measured); every thread then produces its
// outputs for all R rows from LDS, reading each coefficient once.
// MAXT = compile-time bound on the taps: the coefficients of an output sit in registers, loaded by independent
// (fully unrolled, predicated) loads -- a tap loop with one dependent L2 load per iteration was the whole run time.
constexpr int HROWS = 4;
template <int MAXT>
__global__ __launch_bounds__(256) void resize_h_kernel(const unsigned char* __restrict__ img, unsigned char* __restrict__ tmp,
                                                       const int* __restrict__ xb, const int* __restrict__ xk, int xks,
                                                       int H, int W, int S, int row_first, int rows, int R) {
    extern __shared__ __attribute__((aligned(16))) unsigned char span[];
    const int r0 = blockIdx.x * R, n = blockIdx.y;
    const int nr = min(R, rows - r0);
    const size_t row_bytes = (size_t)W * 3;
    const size_t begin = ((size_t)n * H + row_first + r0) * row_bytes, end = begin + (size_t)nr * row_bytes;
    const size_t abegin = begin & ~(size_t)15;
    const int skew = (int)(begin - abegin);
    const size_t img_end = (size_t)gridDim.y * H * row_bytes;  // bytes in the whole batch: never read past it
    // vector part: whole 16-byte chunks that exist in the batch buffer; loads are unconditional (clamped address),
    // only the LDS store is predicated, so U loads per thread are in flight at once
    const size_t vend = min(end, img_end & ~(size_t)15);
    constexpr int U = 4;
    for (size_t base = abegin + (size_t)threadIdx.x * 16; base < vend; base += (size_t)256 * 16 * U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *(const uint4*)(img + min(base + (size_t)u * 256 * 16, (vend - 1) & ~(size_t)15));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t o = base + (size_t)u * 256 * 16;
            if (o < vend) *(uint4*)&span[o - abegin] = v[u];
        }
    }
    for (size_t o = max(abegin, vend) + threadIdx.x; o < end; o += 256) span[o - abegin] = img[o];  // < 16 bytes, last block
    __syncthreads();
    // Taps beyond `cnt` carry a zero coefficient and rows beyond `nr` are computed but not stored: no branches in
    // the MAC chain (LDS reads past the span stay inside the allocation and are multiplied by 0).
    for (int xx = threadIdx.x; xx < S; xx += 256) {
        const int xmin = xb[2 * xx], cnt = xb[2 * xx + 1];
        const int* k = xk + (size_t)xx * xks;
        int kv[MAXT];
#pragma unroll
        for (int x = 0; x < MAXT; ++x) {
            const int t = k[min(x, xks - 1)];
            kv[x] = x < cnt ? t : 0;
        }
        const unsigned char* p = span + skew + (size_t)xmin * 3;
#pragma unroll
        for (int r = 0; r < HROWS; ++r) {
            const unsigned char* q = p + (size_t)r * row_bytes;
            unsigned a0 = 1u << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
#pragma unroll
            for (int x = 0; x < MAXT; ++x) {
                a0 = __umul24(q[3 * x], kv[x]) + a0;
                a1 = __umul24(q[3 * x + 1], kv[x]) + a1;
                a2 = __umul24(q[3 * x + 2], kv[x]) + a2;
            }
            if (r < nr) {
                unsigned char* dst = tmp + (((size_t)n * rows + r0 + r) * S + xx) * 3;
                dst[0] = clip8((int)a0);
                dst[1] = clip8((int)a1);
                dst[2] = clip8((int)a2);
            }
        }
    }
}

// same pass without staging, for rows too long for LDS
__global__ __launch_bounds__(256) void resize_h_direct_kernel(const unsigned char* __restrict__ img,
                                                              unsigned char* __restrict__ tmp, const int* __restrict__ xb,
                                                              const int* __restrict__ xk, int xks, int H, int W, int S,
                                                              int row_first, int rows) {
    const int xx = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y, n = blockIdx.z;
    if (xx >= S) return;
    const int xmin = xb[2 * xx], cnt = xb[2 * xx + 1];
    const int* k = xk + (size_t)xx * xks;
    const unsigned char* src = img + (((size_t)n * H + row_first + r) * W + xmin) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < cnt; ++x) {
        const int kv = k[x];
        s0 += src[3 * x] * kv;
        s1 += src[3 * x + 1] * kv;
        s2 += src[3 * x + 2] * kv;
    }
    unsigned char* dst = tmp + (((size_t)n * rows + r) * S + xx) * 3;
    dst[0] = clip8(s0);
    dst[1] = clip8(s1);
    dst[2] = clip8(s2);
}

// out[n][3][S][S] = ToTensor(vertical pass of src[n][rows][S][3]); yb == nullptr: no vertical pass (rows == S)
__global__ __launch_bounds__(256) void resize_v_tensor_kernel(const unsigned char* __restrict__ src, float* __restrict__ out,
                                                              const int* __restrict__ yb, const int* __restrict__ yk, int yks,
                                                              int rows, int S, int row_first) {
    const int xx = blockIdx.x * 256 + threadIdx.x;
    const int yy = blockIdx.y, n = blockIdx.z;
    if (xx >= S) return;
    int v0, v1, v2;
    if (yb) {
        const int ymin = yb[2 * yy] - row_first, cnt = yb[2 * yy + 1];
        const int* k = yk + (size_t)yy * yks;
        const unsigned char* p = src + (((size_t)n * rows + ymin) * S + xx) * 3;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int y = 0; y < cnt; ++y) {
            const int kv = k[y];
            s0 += p[0] * kv;
            s1 += p[1] * kv;
            s2 += p[2] * kv;
            p += (size_t)S * 3;
        }
        v0 = clip8(s0); v1 = clip8(s1); v2 = clip8(s2);
    } else {
        const unsigned char* p = src + (((size_t)n * rows + yy) * S + xx) * 3;
        v0 = p[0]; v1 = p[1]; v2 = p[2];
    }
    const size_t plane = (size_t)S * S, o = (size_t)n * 3 * plane + (size_t)yy * S + xx;
    out[o] = __fdiv_rn((float)v0, 255.0f);  // ToTensor: one correctly rounded division, as torch's .div(255)
    out[o + plane] = __fdiv_rn((float)v1, 255.0f);
    out[o + 2 * plane] = __fdiv_rn((float)v2, 255.0f);
}

template <typename InT, typename OutT>
__global__ __launch_bounds__(256) void nearest_lut_kernel(const InT* __restrict__ src, OutT* __restrict__ dst,
                                                          const int* __restrict__ yi, const int* __restrict__ xi,
                                                          const unsigned char* __restrict__ lut, int H, int W, int oh,
                                                          int ow) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y, n = blockIdx.z;
    if (x >= ow) return;
    InT v = src[((size_t)n * H + yi[y]) * W + xi[x]];
    if (lut) v = (InT)lut[(unsigned char)v];
    dst[((size_t)n * oh + y) * ow + x] = (OutT)v;
}

// counts[n][3][256]: per 8-bit label value |gt == v and pred == v|, |gt == v|, |pred == v| of image n
__global__ __launch_bounds__(256) void eval_counts_kernel(const unsigned char* __restrict__ pred,
                                                          const unsigned char* __restrict__ gt, const int* __restrict__ yi,
                                                          const int* __restrict__ xi, unsigned long long* __restrict__ counts,
                                                          int S, int Hg, int Wg) {
    __shared__ unsigned hist[3 * 256];
    const int n = blockIdx.y;
    for (int i = threadIdx.x; i < 3 * 256; i += 256) hist[i] = 0;
    __syncthreads();
    const size_t total = (size_t)S * S;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / S), x = (int)(i - (size_t)y * S);
        const unsigned p = pred[(size_t)n * total + i];
        const unsigned g = gt[((size_t)n * Hg + (yi ? yi[y] : y)) * Wg + (xi ? xi[x] : x)];
        if (p == g) atomicAdd(&hist[p], 1u);
        atomicAdd(&hist[256 + g], 1u);
        atomicAdd(&hist[512 + p], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * 256; i += 256)
        if (hist[i]) atomicAdd(&counts[(size_t)n * 768 + i], (unsigned long long)hist[i]);
}

double triangle(double x) {
#pragma clang fp contract(off)
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}

}  // namespace

// ---- host: Resample.c precompute_coeffs + normalize_coeffs_8bpc (triangle filter, box = whole axis) ----
int resize_taps(int in_size, int out_size) {
    double filterscale = (double)((float)in_size - 0.0f) / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    return (int)ceil(1.0 * filterscale) * 2 + 1;
}

int resize_coeffs(int in_size, int out_size, int* bounds, int* kk) {
#pragma clang fp contract(off)  // every double operation rounds on its own, as Pillow's build does (no fused a*b+c)
    VITSEG_CHECK_ARG(in_size > 0 && out_size > 0 && bounds && kk, VITSEG_EINVAL, "resize_coeffs: bad arguments");
    const double scale = (double)((float)in_size - 0.0f) / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    const double ss = 1.0 / filterscale;
    double* w = (double*)malloc(sizeof(double) * ksize);
    VITSEG_CHECK_ARG(w, VITSEG_EINVAL, "resize_coeffs: out of memory");
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            w[x] = triangle((x + xmin - center + 0.5) * ss);
            ww += w[x];
        }
        int* k = kk + (size_t)xx * ksize;
        for (int x = 0; x < ksize; ++x) {
            double v = 0.0;
            if (x < xmax) v = ww != 0.0 ? w[x] / ww : w[x];
            k[x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    free(w);
    return VITSEG_OK;
}

// mode 0: Image.resize(NEAREST) (Geometry.c ImagingScaleAffine: running double sum); 1: F.interpolate 'nearest'
int nearest_index(int in_size, int out_size, int mode, int* idx) {
#pragma clang fp contract(off)
    VITSEG_CHECK_ARG(in_size > 0 && out_size > 0 && idx && (mode == 0 || mode == 1), VITSEG_EINVAL,
                     "nearest_index: bad arguments");
    if (mode == 0) {
        const double a0 = (double)in_size / out_size;
        double xo = 0.0 + a0 * 0.5;
        for (int i = 0; i < out_size; ++i) {
            int v = xo < 0.0 ? -1 : (int)xo;
            idx[i] = v < 0 ? 0 : (v > in_size - 1 ? in_size - 1 : v);
            xo += a0;
        }
    } else {
        const float scale = (float)((double)in_size / out_size);
        for (int i = 0; i < out_size; ++i) {
            int v = (int)floorf((float)i * scale);
            idx[i] = v > in_size - 1 ? in_size - 1 : v;
        }
    }
    return VITSEG_OK;
}
int launch_preprocess_u8(const unsigned char* img, int n, int H, int W, int S, const int* xb, const int* xk, int xks,
                         const int* yb, const int* yk, int yks, int row_first, int rows, unsigned char* tmp, float* out,
                         hipStream_t s) {
    VITSEG_CHECK_ARG(img && out && n > 0 && H > 0 && W > 0 && S > 0, VITSEG_EINVAL, "preprocess: bad arguments");
    VITSEG_CHECK_ARG((uintptr_t)img % 16 == 0, VITSEG_EINVAL, "preprocess: image pointer must be 16-byte aligned");
    const bool has_h = W != S, has_v = H != S;
    VITSEG_CHECK_ARG(!has_h || (xb && xk && tmp && xks > 0), VITSEG_EINVAL, "preprocess: horizontal tables / scratch missing");
    VITSEG_CHECK_ARG(!has_v || (yb && yk && yks > 0), VITSEG_EINVAL, "preprocess: vertical tables missing");
    if (!has_v) { row_first = 0; rows = H; }
    VITSEG_CHECK_ARG(row_first >= 0 && rows > 0 && row_first + rows <= H, VITSEG_EINVAL, "preprocess: rows [%d, +%d) of %d",
                     row_first, rows, H);
    const unsigned char* vsrc = img;
    int vrows = H, vfirst = 0;
    if (has_h) {
        const size_t row_bytes = (size_t)W * 3;
        const int R = row_bytes * HROWS + 160 <= 64 * 1024 ? HROWS : 0;
        if (R >= 1 && xks <= 37) {
            const size_t smem = (size_t)HROWS * row_bytes + 32 + 3 * 40;  // rows beyond nr / taps beyond cnt are read, not used
            const dim3 grid((rows + R - 1) / R, n);
#define VITSEG_RH(T)                                                                                              \
    hipLaunchKernelGGL(resize_h_kernel<T>, grid, dim3(256), smem, s, img, tmp, xb, xk, xks, H, W, S, row_first, rows, R)
            if (xks <= 3) VITSEG_RH(3);          // up-scaling
            else if (xks <= 7) VITSEG_RH(7);     // up to 3x reduction
            else if (xks <= 13) VITSEG_RH(13);   // up to 6x
            else if (xks <= 19) VITSEG_RH(19);   // up to 9x (12 MP photo -> 512)
            else if (xks <= 27) VITSEG_RH(27);
            else VITSEG_RH(37);
#undef VITSEG_RH
        } else {  // rows too long for LDS or more than 37 taps (> 18x reduction): plain gather
            hipLaunchKernelGGL(resize_h_direct_kernel, dim3((S + 255) / 256, rows, n), dim3(256), 0, s, img, tmp, xb, xk,
                               xks, H, W, S, row_first, rows);
        }
        VITSEG_LAUNCH_CHECK("resize_h");
        vsrc = tmp;
        vrows = rows;
        vfirst = row_first;
    }
    hipLaunchKernelGGL(resize_v_tensor_kernel, dim3((S + 255) / 256, S, n), dim3(256), 0, s, vsrc, out, has_v ? yb : nullptr,
                       yk, yks, vrows, S, vfirst);
    VITSEG_LAUNCH_CHECK("resize_v_tensor");
    return VITSEG_OK;
}

// src_i64: the source holds int64 class indices (torch.long targets: model/CE/classes.py:273-274 resizes those)
int launch_nearest_lut(const void* src, int src_i64, int n, int H, int W, const int* yi, const int* xi, int oh, int ow,
                       const unsigned char* lut, int out_i64, void* out, hipStream_t s) {
    VITSEG_CHECK_ARG(src && out && yi && xi && n > 0 && oh > 0 && ow > 0, VITSEG_EINVAL, "resize_nearest: bad arguments");
    VITSEG_CHECK_ARG(!(src_i64 && lut), VITSEG_EINVAL, "resize_nearest: the value table applies to 8-bit sources");
    const dim3 grid((ow + 255) / 256, oh, n);
#define VITSEG_NEAR(IN, OUT) \
    hipLaunchKernelGGL((nearest_lut_kernel<IN, OUT>), grid, dim3(256), 0, s, (const IN*)src, (OUT*)out, yi, xi, lut, H, W, oh, ow)
    if (src_i64) {
        if (out_i64) VITSEG_NEAR(long long, long long); else VITSEG_NEAR(long long, unsigned char);
    } else {
        if (out_i64) VITSEG_NEAR(unsigned char, long long); else VITSEG_NEAR(unsigned char, unsigned char);
    }
#undef VITSEG_NEAR
    VITSEG_LAUNCH_CHECK("nearest_lut");
    return VITSEG_OK;
}

int launch_eval_counts(const unsigned char* pred, const unsigned char* gt, int n, int S, int Hg, int Wg, const int* yi,
                       const int* xi, long long* counts, hipStream_t s) {
    VITSEG_CHECK_ARG(pred && gt && counts && n > 0 && S > 0 && Hg > 0 && Wg > 0, VITSEG_EINVAL, "eval_counts: bad arguments");
    VITSEG_CHECK_ARG((yi && xi) || (Hg == S && Wg == S), VITSEG_ESHAPE,
                     "eval_counts: ground truth %dx%d needs index tables to meet the %dx%d prediction", Hg, Wg, S, S);
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)n * 768 * sizeof(long long), s);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(eval counts)");
    const int per = (int)(((size_t)S * S + 256 * 16 - 1) / (256 * 16));  // ~16 pixels per thread
    hipLaunchKernelGGL(eval_counts_kernel, dim3(per < 1 ? 1 : per, n), dim3(256), 0, s, pred, gt, yi, xi,
                       (unsigned long long*)counts, S, Hg, Wg);
    VITSEG_LAUNCH_CHECK("eval_counts");
    return VITSEG_OK;
}

}  // namespace vitseg

extern "C" {

int vitseg_resize_taps(int in_size, int out_size) { return vitseg::resize_taps(in_size, out_size); }
int vitseg_resize_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk) {
    return vitseg::resize_coeffs(in_size, out_size, bounds, kk);
}
int vitseg_nearest_index(int in_size, int out_size, int mode, int32_t* idx) {
    return vitseg::nearest_index(in_size, out_size, mode, idx);
}
int vitseg_preprocess_u8(const uint8_t* img, int n, int H, int W, int S, const int32_t* xbounds, const int32_t* xk, int xksize,
                         const int32_t* ybounds, const int32_t* yk, int yksize, int row_first, int rows, uint8_t* scratch,
                         float* out, void* stream) {
    return vitseg::launch_preprocess_u8(img, n, H, W, S, xbounds, xk, xksize, ybounds, yk, yksize, row_first, rows, scratch,
                                        out, (hipStream_t)stream);
}
int vitseg_resize_nearest_u8(const uint8_t* src, int n, int H, int W, const int32_t* yidx, const int32_t* xidx, int out_h,
                             int out_w, const uint8_t* lut, int out_is_i64, void* out, void* stream) {
    return vitseg::launch_nearest_lut(src, 0, n, H, W, yidx, xidx, out_h, out_w, lut, out_is_i64, out, (hipStream_t)stream);
}
int vitseg_resize_nearest_i64(const int64_t* src, int n, int H, int W, const int32_t* yidx, const int32_t* xidx, int out_h,
                              int out_w, int out_is_i64, void* out, void* stream) {
    return vitseg::launch_nearest_lut(src, 1, n, H, W, yidx, xidx, out_h, out_w, nullptr, out_is_i64, out,
                                      (hipStream_t)stream);
}
int vitseg_eval_counts(const uint8_t* pred, const uint8_t* gt, int n, int S, int gt_h, int gt_w, const int32_t* yidx,
                       const int32_t* xidx, int64_t* counts, void* stream) {
    return vitseg::launch_eval_counts(pred, gt, n, S, gt_h, gt_w, yidx, xidx, (long long*)counts, (hipStream_t)stream);
}

}  // extern "C"
