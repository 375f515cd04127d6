// Loss tail of the binary PAED trainer, value and gradient w.r.t. the logits in three launches (SURVEY.md 8(f) row f1,
// second half).  Replaces, in PAEDTrainer._forward_step_paed (/root/reference/model/PAED/classes.py:664-701):
//   preds = sigmoid(logits)
//   bce   = F.binary_cross_entropy(preds, masks)                                   (:679)
//   dice  = 1 - (2 sum(p m) + 1e-6) / (sum p + sum m + 1e-6)                       (dice_loss, :608-620)
//   paed  = mean(ext * edge) - 0.5 mean(int * p)                                   (paed_loss_soft, :623-661)
//           ext / int = the SDFs bilinearly resized to the prediction (align_corners=False),
//           edge = sqrt(gx^2 + gy^2 + 1e-6) / (per-image max + 1e-6),  gx / gy = Sobel cross-correlations of p, zero padding
//   loss  = bce + 0.1 dice + 5 |paed|                                              (:681)
// and autograd through all of it (~30 elementwise / conv launches on [B,1,S,S] maps in the reference).
//   pass 1 (paed_bin_fwd):    p, Sobel, E per pixel from a z tile with halo in LDS; fp64 block partials of every sum; per-block
//                             maximum of E with the FIRST index attaining it (what torch's max backward picks on ties)
//   pass 2 (paed_bin_finish): fixed-order reduction (deterministic), the loss terms, and the coefficients of the gradient
//   pass 3 (paed_bin_bwd):    dL/dp = bce' + 0.1 dice' + 5 sign(paed) paed', where the edge term flows back through the
//                             per-image normalisation (its argmax pixel collects -sum(ext E) / (max + eps)^2) and through the
//                             TRANSPOSED Sobel stencils of the neighbours; dL/dz = dL/dp * p (1 - p).
// HBM-bound elementwise / 3x3-stencil work: ~5 fp32 maps read, 1 written per pixel.
#include <math.h>

#include "kernels.hpp"

namespace vitseg {
namespace {

constexpr int TS = 32;          // tile edge
constexpr int NSUM = 10;        // bce, I = sum p m, P, T, sum int p, tp, fp, fn, eq, (spare)

__device__ __forceinline__ void taps(int d, float scale, int n_in, int& i0, int& i1, float& w0, float& w1) {
    float src = __fsub_rn(__fmul_rn(scale, __fadd_rn((float)d, 0.5f)), 0.5f);
    src = src < 0.f ? 0.f : src;
    i0 = min((int)floorf(src), n_in - 1);
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    w1 = fminf(fmaxf(__fsub_rn(src, (float)i0), 0.f), 1.f);
    w0 = __fsub_rn(1.f, w1);
}
// F.interpolate(sdf, size=(H, W), mode='bilinear', align_corners=False) at one pixel (ATen's arithmetic order)
__device__ __forceinline__ float bilerp(const float* __restrict__ s, int hs, int ws, int H, int W, int y, int x) {
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    taps(y, (float)hs / (float)H, hs, y0, y1, wy0, wy1);
    taps(x, (float)ws / (float)W, ws, x0, x1, wx0, wx1);
    const float top = __fmaf_rn(s[y0 * ws + x0], wx0, __fmul_rn(s[y0 * ws + x1], wx1));
    const float bot = __fmaf_rn(s[y1 * ws + x0], wx0, __fmul_rn(s[y1 * ws + x1], wx1));
    return __fmaf_rn(top, wy0, __fmul_rn(bot, wy1));
}
__device__ __forceinline__ float sigmoidf(float z) { return 1.0f / (1.0f + expf(-z)); }

struct BinMax {
    float e;
    int idx;   // pixel index inside the image (row-major); the smallest one among equal maxima
};
__device__ __forceinline__ BinMax bmax(BinMax a, BinMax b) {
    return (b.e > a.e || (b.e == a.e && b.idx < a.idx)) ? b : a;
}

// p on a (TS + 2 HALO)^2 tile, zero outside the image (conv2d padding=1 pads the PREDICTION with zeros)
template <int HALO>
__device__ __forceinline__ void load_p_tile(float (*pt)[TS + 2 * HALO], const float* __restrict__ z, int H, int W, int ty0,
                                            int tx0) {
    constexpr int E = TS + 2 * HALO;
    for (int i = threadIdx.x; i < E * E; i += 256) {
        const int ly = i / E, lx = i - ly * E;
        const int y = ty0 + ly - HALO, x = tx0 + lx - HALO;
        pt[ly][lx] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? sigmoidf(z[(size_t)y * W + x]) : 0.f;
    }
}
// Sobel cross-correlations at tile position (ly, lx) of a p tile (indices already include the halo)
template <int E>
__device__ __forceinline__ void sobel(const float (*pt)[E], int ly, int lx, float& gx, float& gy) {
    const float a = pt[ly - 1][lx - 1], b = pt[ly - 1][lx], c = pt[ly - 1][lx + 1];
    const float d = pt[ly][lx - 1], f = pt[ly][lx + 1];
    const float g = pt[ly + 1][lx - 1], h = pt[ly + 1][lx], k = pt[ly + 1][lx + 1];
    gx = (a - c) + 2.f * (d - f) + (g - k);       // [[1,0,-1],[2,0,-2],[1,0,-1]]
    gy = (a - g) + 2.f * (b - h) + (c - k);       // its transpose
}

__global__ __launch_bounds__(256) void paed_bin_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ mask,
                                                           const float* __restrict__ sdf_ext,
                                                           const float* __restrict__ sdf_int, int hs, int ws, int H, int W,
                                                           int tiles_x, int tiles_per_img, double* __restrict__ psum,
                                                           double* __restrict__ pext, BinMax* __restrict__ pmax) {
    __shared__ float pt[TS + 2][TS + 2];
    __shared__ double red[4][NSUM + 1];
    __shared__ BinMax redm[4];
    const int img = blockIdx.x / tiles_per_img, t = blockIdx.x - img * tiles_per_img;
    const int ty0 = (t / tiles_x) * TS, tx0 = (t - (t / tiles_x) * tiles_x) * TS;
    const float* z = logits + (size_t)img * H * W;
    load_p_tile<1>(pt, z, H, W, ty0, tx0);
    __syncthreads();
    double s[NSUM + 1];
#pragma unroll
    for (int i = 0; i <= NSUM; ++i) s[i] = 0.0;
    BinMax bm{-1.f, 0x7fffffff};
    for (int i = threadIdx.x; i < TS * TS; i += 256) {
        const int ly = i / TS, lx = i - ly * TS, y = ty0 + ly, x = tx0 + lx;
        if (y >= H || x >= W) continue;
        const float p = pt[ly + 1][lx + 1];
        const float m = mask[((size_t)img * H + y) * W + x];
        float gx, gy;
        sobel<TS + 2>(pt, ly + 1, lx + 1, gx, gy);
        const float e = sqrtf(gx * gx + gy * gy + 1e-6f);
        const float ext = bilerp(sdf_ext + (size_t)img * hs * ws, hs, ws, H, W, y, x);
        const float inn = bilerp(sdf_int + (size_t)img * hs * ws, hs, ws, H, W, y, x);
        // F.binary_cross_entropy: -(m max(log p, -100) + (1 - m) max(log1p(-p), -100))
        s[0] += (double)(-(m * fmaxf(logf(p), -100.f) + (1.f - m) * fmaxf(log1pf(-p), -100.f)));
        s[1] += (double)(p * m);
        s[2] += (double)p;
        s[3] += (double)m;
        s[4] += (double)(inn * p);
        const float bin = p > 0.5f ? 1.f : 0.f;
        s[5] += (double)(bin * m);
        s[6] += (double)(bin * (1.f - m));
        s[7] += (double)((1.f - bin) * m);
        s[8] += (double)(bin == m ? 1.f : 0.f);
        s[NSUM] += (double)(ext * e);             // numerator of the exterior term of this image
        bm = bmax(bm, BinMax{e, y * W + x});
    }
#pragma unroll
    for (int i = 0; i <= NSUM; ++i)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[i] += __shfl_xor(s[i], o, 64);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        BinMax other;
        other.e = __shfl_xor(bm.e, o, 64);
        other.idx = __shfl_xor(bm.idx, o, 64);
        bm = bmax(bm, other);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i <= NSUM; ++i) red[wave][i] = s[i];
        redm[wave] = bm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 0; i < NSUM; ++i) psum[(size_t)blockIdx.x * NSUM + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        pext[blockIdx.x] = (red[0][NSUM] + red[1][NSUM]) + (red[2][NSUM] + red[3][NSUM]);
        pmax[blockIdx.x] = bmax(bmax(redm[0], redm[1]), bmax(redm[2], redm[3]));
    }
}

// coefficient block handed to the backward pass (floats): [0] 1/N  [1] dice a  [2] dice b  [3] 5 sign(paed)
// then per image: [4 + 3 b] 1/(M_b + eps)  [5 + 3 b] S_b / (M_b + eps)^2  [6 + 3 b] argmax index (as int bits)
__global__ __launch_bounds__(256) void paed_bin_finish_kernel(const double* __restrict__ psum, const double* __restrict__ pext,
                                                              const BinMax* __restrict__ pmax, int batch,
                                                              int tiles_per_img, double n_px, float* __restrict__ coef,
                                                              float* __restrict__ out) {
    __shared__ double tot[NSUM];
    __shared__ double ext_term;
    if (threadIdx.x < NSUM) {   // fixed order per sum: deterministic
        double a = 0.0;
        for (int b = 0; b < batch * tiles_per_img; ++b) a += psum[(size_t)b * NSUM + threadIdx.x];
        tot[threadIdx.x] = a;
    }
    if (threadIdx.x == 32) {
        double et = 0.0;
        for (int b = 0; b < batch; ++b) {
            BinMax bm{-1.f, 0x7fffffff};
            double sb = 0.0;
            for (int t = 0; t < tiles_per_img; ++t) {
                bm = bmax(bm, pmax[b * tiles_per_img + t]);
                sb += pext[b * tiles_per_img + t];
            }
            const float inv = 1.0f / (bm.e + 1e-6f);     // edge / (max + 1e-6), fp32 as the reference
            et += sb * (double)inv;
            coef[4 + 3 * b] = inv;
            coef[5 + 3 * b] = (float)(sb * (double)inv * (double)inv);
            coef[6 + 3 * b] = __int_as_float(bm.idx);
        }
        ext_term = et;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double bce = tot[0] / n_px;
        const double dn = tot[2] + tot[3] + 1e-6, nn = 2.0 * tot[1] + 1e-6;
        const double dice = 1.0 - nn / dn;
        const double paed = ext_term / n_px - 0.5 * tot[4] / n_px;
        out[0] = (float)(bce + 0.1 * dice + 5.0 * fabs(paed));
        out[1] = (float)bce;
        out[2] = (float)dice;
        out[3] = (float)paed;
        out[4] = (float)tot[5];   // tp
        out[5] = (float)tot[6];   // fp
        out[6] = (float)tot[7];   // fn
        out[7] = (float)tot[8];   // pixels where (p > 0.5) == mask
        coef[0] = (float)(1.0 / n_px);
        coef[1] = (float)(-0.1 * 2.0 / dn);            // d dice / d p_i = -2 m_i / dn + nn / dn^2
        coef[2] = (float)(0.1 * nn / (dn * dn));
        coef[3] = paed > 0.0 ? 5.f : (paed < 0.0 ? -5.f : 0.f);
    }
}

__global__ __launch_bounds__(256) void paed_bin_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ mask,
                                                           const float* __restrict__ sdf_ext,
                                                           const float* __restrict__ sdf_int, int hs, int ws, int H, int W,
                                                           int tiles_x, int tiles_per_img, const float* __restrict__ coef,
                                                           float* __restrict__ grad) {
    __shared__ float pt[TS + 4][TS + 4];
    __shared__ float ax[TS + 2][TS + 2], ay[TS + 2][TS + 2];
    const int img = blockIdx.x / tiles_per_img, t = blockIdx.x - img * tiles_per_img;
    const int ty0 = (t / tiles_x) * TS, tx0 = (t - (t / tiles_x) * tiles_x) * TS;
    const float* z = logits + (size_t)img * H * W;
    load_p_tile<2>(pt, z, H, W, ty0, tx0);
    __syncthreads();
    const float inv_n = coef[0], da = coef[1], db = coef[2], sg = coef[3];
    const float inv_m = coef[4 + 3 * img], sb2 = coef[5 + 3 * img];
    const int amax = __float_as_int(coef[6 + 3 * img]);
    // a_k = w_k g_k / E_k on the tile plus one ring (the pixels whose stencil reaches into the tile); 0 outside the image
    for (int i = threadIdx.x; i < (TS + 2) * (TS + 2); i += 256) {
        const int ly = i / (TS + 2), lx = i - ly * (TS + 2);
        const int y = ty0 + ly - 1, x = tx0 + lx - 1;
        float vx = 0.f, vy = 0.f;
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
            float gx, gy;
            sobel<TS + 4>(pt, ly + 1, lx + 1, gx, gy);
            const float e = sqrtf(gx * gx + gy * gy + 1e-6f);
            const float ext = bilerp(sdf_ext + (size_t)img * hs * ws, hs, ws, H, W, y, x);
            float w = ext * inv_m;
            if (y * W + x == amax) w -= sb2;           // the normalising maximum itself depends on this pixel
            w *= inv_n / e;
            vx = w * gx;
            vy = w * gy;
        }
        ax[ly][lx] = vx;
        ay[ly][lx] = vy;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TS * TS; i += 256) {
        const int ly = i / TS, lx = i - ly * TS, y = ty0 + ly, x = tx0 + lx;
        if (y >= H || x >= W) continue;
        const float p = pt[ly + 2][lx + 2];
        const float m = mask[((size_t)img * H + y) * W + x];
        const float inn = bilerp(sdf_int + (size_t)img * hs * ws, hs, ws, H, W, y, x);
        // transposed stencils: d gx_k / d p_i = sx[i - k + 1]: neighbour k = i + (dy, dx) uses sx[1 - dy][1 - dx]
        const int cy = ly + 1, cx = lx + 1;
        const float tx = (ax[cy + 1][cx + 1] - ax[cy + 1][cx - 1]) + 2.f * (ax[cy][cx + 1] - ax[cy][cx - 1]) +
                         (ax[cy - 1][cx + 1] - ax[cy - 1][cx - 1]);
        const float tyv = (ay[cy + 1][cx + 1] - ay[cy - 1][cx + 1]) + 2.f * (ay[cy + 1][cx] - ay[cy - 1][cx]) +
                          (ay[cy + 1][cx - 1] - ay[cy - 1][cx - 1]);
        const float dpaed = tx + tyv - 0.5f * inn * inv_n;
        // binary_cross_entropy_backward: (p - m) / max((1 - p) p, 1e-12) / N
        const float dbce = (p - m) / fmaxf((1.f - p) * p, 1e-12f) * inv_n;
        const float dp = dbce + (da * m + db) + sg * dpaed;
        grad[((size_t)img * H + y) * W + x] = dp * p * (1.f - p);
    }
}

}  // namespace

size_t paed_binary_scratch_bytes(int batch, int H, int W) {
    const size_t tiles = (size_t)batch * ((H + TS - 1) / TS) * ((W + TS - 1) / TS);
    return tiles * (NSUM + 1) * sizeof(double) + tiles * sizeof(BinMax) + (size_t)(4 + 3 * batch) * sizeof(float) + 64;
}

int launch_paed_binary(const float* logits, const float* mask, const float* sdf_ext, const float* sdf_int, int hs, int ws,
                       int batch, int H, int W, void* scratch, float* out, float* grad, hipStream_t s) {
    VITSEG_CHECK_ARG(logits && mask && sdf_ext && sdf_int && scratch && out, VITSEG_EINVAL, "paed_binary: null pointer");
    VITSEG_CHECK_ARG(batch >= 1 && H >= 1 && W >= 1 && hs >= 1 && ws >= 1, VITSEG_EINVAL, "paed_binary: bad shape");
    const int tiles_x = (W + TS - 1) / TS, tiles_per_img = tiles_x * ((H + TS - 1) / TS);
    const size_t tiles = (size_t)batch * tiles_per_img;
    double* psum = (double*)scratch;
    double* pext = psum + tiles * NSUM;
    BinMax* pmax = (BinMax*)(pext + tiles);
    float* coef = (float*)(pmax + tiles);
    hipLaunchKernelGGL(paed_bin_fwd_kernel, dim3((unsigned)tiles), dim3(256), 0, s, logits, mask, sdf_ext, sdf_int, hs, ws, H,
                       W, tiles_x, tiles_per_img, psum, pext, pmax);
    VITSEG_LAUNCH_CHECK("paed_bin_fwd");
    hipLaunchKernelGGL(paed_bin_finish_kernel, dim3(1), dim3(256), 0, s, psum, pext, pmax, batch, tiles_per_img,
                       (double)batch * H * W, coef, out);
    VITSEG_LAUNCH_CHECK("paed_bin_finish");
    if (grad) {
        hipLaunchKernelGGL(paed_bin_bwd_kernel, dim3((unsigned)tiles), dim3(256), 0, s, logits, mask, sdf_ext, sdf_int, hs,
                           ws, H, W, tiles_x, tiles_per_img, coef, grad);
        VITSEG_LAUNCH_CHECK("paed_bin_bwd");
    }
    return VITSEG_OK;
}

}  // namespace vitseg

extern "C" {
size_t vitseg_paed_binary_scratch_bytes(int batch, int H, int W) { return vitseg::paed_binary_scratch_bytes(batch, H, W); }

int vitseg_paed_binary_loss(const float* logits, const float* mask, const float* sdf_ext, const float* sdf_int, int sdf_h,
                            int sdf_w, int batch, int H, int W, void* scratch, float* out8, float* grad_logits,
                            void* stream) {
    return vitseg::launch_paed_binary(logits, mask, sdf_ext, sdf_int, sdf_h, sdf_w, batch, H, W, scratch, out8, grad_logits,
                                      (hipStream_t)stream);
}
}
