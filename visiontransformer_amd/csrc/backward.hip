// Backward-pass kernels that are not GEMMs or attention (a13: what autograd runs behind
// LightningViTModel.training_step, /root/reference/model/CE/classes.py:276-285, and Adam, :296-297).
// All reductions are deterministic: per-block partial sums in a scratch buffer, finished in a fixed order.
#include <type_traits>

#include "kernels.hpp"
#include "small.hpp"

namespace vitseg {
namespace {

constexpr int MID = 256;

// ---- column sums: out[n] = sum_m X[m][n]  (bias gradients) -----------------------------------
// stage 1: block = 256 columns x a chunk of 256 rows -> partial[chunk][n]; stage 2: sum the chunks.
typedef unsigned short bf16_t;
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16_t* p) { return bf16_to_f32(*p); }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 ld4(const bf16_t* p) {
    const uint2 u = *(const uint2*)p;
    f32x4 r = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
               __uint_as_float(u.y & 0xffff0000u)};
    return r;
}

// Block = 256 rows x 256 columns: wave w sums rows r0 + 64 w .. + 63, lane owns 4 adjacent columns (one 8- or
// 16-byte load per row), the 4 waves are combined through LDS in a fixed order.
template <typename InT>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const InT* __restrict__ X, float* __restrict__ partial,
                                                             int M, int N, int ld) {
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 256 + lane * 4;
    const int r0 = blockIdx.y * 256 + wave * 64, r1 = min(r0 + 64, M);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (n + 3 < N) {
        int r = r0;
        f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1, a3 = a1;
        for (; r + 3 < r1; r += 4) {   // four rows in flight per lane
            const f32x4 v0 = ld4(X + (size_t)r * ld + n), v1 = ld4(X + (size_t)(r + 1) * ld + n);
            const f32x4 v2 = ld4(X + (size_t)(r + 2) * ld + n), v3 = ld4(X + (size_t)(r + 3) * ld + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[e] += v0[e];
                a1[e] += v1[e];
                a2[e] += v2[e];
                a3[e] += v3[e];
            }
        }
        for (; r < r1; ++r) {
            const f32x4 v = ld4(X + (size_t)r * ld + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = (acc[e] + a1[e]) + (a2[e] + a3[e]);
    } else {
        for (int r = r0; r < r1; ++r)
            for (int e = 0; e < 4; ++e)
                if (n + e < N) acc[e] += ldf(X + (size_t)r * ld + n + e);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][lane * 4 + e] = acc[e];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < N)
        partial[(size_t)blockIdx.y * N + c] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// out[n] = sum_chunks partial[chunk][n]: block = 64 columns x 16 chunk groups (1024 threads), combined through LDS in a
// fixed order.  (With 4 groups a thread walked 256 partials of the 1025-chunk training shapes on 24 CUs: 19-62 us for a
// few MB; 16 groups and unrolled independent loads bring it to the launch-latency scale.)
constexpr int FIN_GROUPS = 16;
__device__ __forceinline__ float finish_column(const float* __restrict__ partial, int chunks, size_t stride, int col, int grp,
                                               float (*red)[64]) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = grp;
    for (; c + 3 * FIN_GROUPS < chunks; c += 4 * FIN_GROUPS) {
        s0 += partial[(size_t)c * stride + col];
        s1 += partial[(size_t)(c + FIN_GROUPS) * stride + col];
        s2 += partial[(size_t)(c + 2 * FIN_GROUPS) * stride + col];
        s3 += partial[(size_t)(c + 3 * FIN_GROUPS) * stride + col];
    }
    for (; c < chunks; c += FIN_GROUPS) s0 += partial[(size_t)c * stride + col];
    red[grp][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    float v = 0.f;
    if (grp == 0)
#pragma unroll
        for (int k = 0; k < FIN_GROUPS; ++k) v += red[k][threadIdx.x];
    return v;
}
__global__ __launch_bounds__(1024) void colsum_finish_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                             int chunks, int N) {
    __shared__ float red[FIN_GROUPS][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    const float v = finish_column(partial, chunks, (size_t)N, min(col, N - 1), grp, red);
    if (grp == 0 && col < N) out[col] = v;
}

// ---- LayerNorm backward ------------------------------------------------------------------------
// y = xhat * w + b, xhat = (x - mu) * rstd.  With gw = g * w:
//   dx = rstd * (gw - mean(gw) - xhat * mean(gw * xhat));   dw = sum_rows g * xhat;   db = sum_rows g.
// dres_out[row] = (dres_in ? dres_in[row] : 0) + dx  (the residual branch's gradient is added here).
// One wave per row (statistics recomputed from the saved input), ~64 rows per block; the block's dw/db
// partial sums go to partial[block][SETS][D].
// BR (branch output): the gradient that enters the NEXT dropped residual branch of the backward walk is
// mask * dres_out (hidden dropout of that branch, DropArgs br_drop; thresh 0 = no dropout) -- written here as bf16
// (the operand format of the branch's GEMMs) together with its column sums (the branch's bias gradient, third
// partial set), so neither a dropout/cast pass nor a column-sum pass has to re-read dres_out.
template <int NV, typename GT, bool BR>
__global__ __launch_bounds__(256, NV <= 3 ? 3 : 2) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const GT* __restrict__ g, const float* dres_in,
                                                            float* dres_out, float* __restrict__ partial, int rows,
                                                            int D, float eps, bf16_t* __restrict__ br_out,
                                                            DropArgs br_drop) {
    constexpr int SETS = BR ? 3 : 2;
    __shared__ float red[SETS][4][NV * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = D >> 2;
    f32x4 wv[NV], dw[NV], db[NV], dbr[BR ? NV : 1];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        wv[i] = ((const f32x4*)w)[min(lane + 64 * i, nv - 1)];
#pragma unroll
        for (int e = 0; e < 4; ++e) dw[i][e] = db[i][e] = 0.f;
        if constexpr (BR) dbr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // Rows are dealt evenly over the grid (block b: rows [b R / G, (b + 1) R / G), wave w every fourth of them): the
    // launcher sizes the grid to what is resident at once, so no last partial round of a few blocks runs alone
    // (65 600 rows as 1 025 blocks of 64 on 512 resident slots was 2.002 rounds).
    // The next row's x / g are requested before the current row's four dependent wave reductions, so a wave always has
    // a row of loads in flight; g stays packed while it waits (16-bit GT: 2 registers per 4 values), and the residual
    // gradient of the CURRENT row is requested at the row's start and only read by its last stage -- its latency hides
    // behind the reductions without a second set of staging registers (192 -> 165 VGPRs: 3 waves per SIMD).
    constexpr bool G16 = sizeof(GT) == 2;
    typedef typename std::conditional<G16, uint2, f32x4>::type gpack_t;
    f32x4 nx[NV];
    gpack_t ng[NV];
    auto fetch = [&](int row) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = min(lane + 64 * i, nv - 1);
            nx[i] = ((const f32x4*)(x + (size_t)row * D))[c];
            ng[i] = *(const gpack_t*)(g + (size_t)row * D + 4 * c);
        }
    };
    const int row_begin = (int)((long long)blockIdx.x * rows / (int)gridDim.x);
    const int row_end = (int)((long long)(blockIdx.x + 1) * rows / (int)gridDim.x);
    if (row_begin + wave < row_end) fetch(row_begin + wave);
    for (int row = row_begin + wave; row < row_end; row += 4) {   // wave-uniform
        f32x4 xv[NV], gv[NV], dv[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            xv[i] = nx[i];
            if constexpr (G16) {
                gv[i] = f32x4{H16<bf16_t>::lo(ng[i].x), H16<bf16_t>::hi(ng[i].x), H16<bf16_t>::lo(ng[i].y), H16<bf16_t>::hi(ng[i].y)};
            } else {
                gv[i] = ng[i];
            }
        }
        if (row + 4 < row_end) fetch(row + 4);
        if (dres_in) {
#pragma unroll
            for (int i = 0; i < NV; ++i) dv[i] = ((const f32x4*)(dres_in + (size_t)row * D))[min(lane + 64 * i, nv - 1)];
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float t = (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
            s += (lane + 64 * i < nv) ? t : 0.f;
        }
        const float mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xv[i][e] -= mean;
                t = fmaf(xv[i][e], xv[i][e], t);
            }
            q += (lane + 64 * i < nv) ? t : 0.f;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const bool ok = lane + 64 * i < nv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xv[i][e] *= rstd;  // xhat
                const float gw = gv[i][e] * wv[i][e];
                if (ok) {
                    s1 += gw;
                    s2 = fmaf(gw, xv[i][e], s2);
                    dw[i][e] = fmaf(gv[i][e], xv[i][e], dw[i][e]);
                    db[i][e] += gv[i][e];
                }
            }
        }
        const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
        unsigned bkey = 0;
        if constexpr (BR) bkey = drop_key(br_drop.seed, br_drop.stream, (unsigned)row);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                if (dres_in) o = dv[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += rstd * (gv[i][e] * wv[i][e] - c1 - xv[i][e] * c2);
                ((f32x4*)(dres_out + (size_t)row * D))[c] = o;
                if constexpr (BR) {
                    if (br_drop.thresh) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            o[e] = drop_keep(bkey, (unsigned)(4 * c + e), br_drop.thresh) ? o[e] * br_drop.scale : 0.f;
                    }
                    uint2 h;
                    h.x = pack2_bf16(o[0], o[1]);
                    h.y = pack2_bf16(o[2], o[3]);
                    ((uint2*)(br_out + (size_t)row * D))[c] = h;
                    // the bias gradient sums what the GEMMs will read: the bf16-rounded values
                    dbr[i][0] += __uint_as_float(h.x << 16);
                    dbr[i][1] += __uint_as_float(h.x & 0xffff0000u);
                    dbr[i][2] += __uint_as_float(h.y << 16);
                    dbr[i][3] += __uint_as_float(h.y & 0xffff0000u);
                }
            }
        }
    }
    // block reduction of dw / db over the 4 waves, fixed order
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[0][wave][(lane + 64 * i) * 4 + e] = dw[i][e];
            red[1][wave][(lane + 64 * i) * 4 + e] = db[i][e];
            if constexpr (BR) red[2][wave][(lane + 64 * i) * 4 + e] = dbr[i][e];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
#pragma unroll
        for (int k = 0; k < SETS; ++k)
            partial[((size_t)blockIdx.x * SETS + k) * D + c] = (red[k][0][c] + red[k][1][c]) + (red[k][2][c] + red[k][3][c]);
    }
}
// partial[block][2][D] -> dw[D], db[D]: block = 64 columns of the 2D-wide matrix x 16 block groups
__global__ __launch_bounds__(1024) void layernorm_bwd_finish_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                    float* __restrict__ db, float* __restrict__ dbr,
                                                                    int blocks, int D, int sets) {
    __shared__ float red[FIN_GROUPS][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;  // col in [0, sets * D)
    const float v = finish_column(partial, blocks, (size_t)sets * D, min(col, sets * D - 1), grp, red);
    if (grp == 0 && col < sets * D) {
        if (col < D) dw[col] = v;
        else if (col < 2 * D) db[col - D] = v;
        else dbr[col - 2 * D] = v;
    }
}

// ---- LayerNorm backward of the small-batch fp32 training step (small.hpp) ----------------------------------------
// The arithmetic, the row-to-wave assignment and the partial-sum order of layernorm_bwd_kernel<NV, float, false> (same bits for
// dx / dw / db), with what the small route hangs on it so that no separate pass re-reads the rows:
//   * g may arrive as K-chunk slabs of the activation-gradient GEMM (gemm_f32s SE_PARTIAL): g = slab 0 + slab 1 + ... in
//     chunk order (launch_slabsum's order), summed here;
//   * the gradient that enters the NEXT dropped residual branch of the backward walk, mask * dres_out (hidden dropout of that
//     branch; thresh 0: dres_out itself, nothing written), and its column sums = that branch's bias gradient (third partial
//     set) -- the fp32 form of the BR mode above.
// One wave per row, every load of the row issued up front (788 rows: latency, not bandwidth).
struct LnbSmall {
    const float* x;
    const float* w;
    const float* g;
    size_t g_stride;
    int g_splits;
    const float* dres_in;
    float* dres_out;
    float* partial;
    int rows, D, sets;
    float eps;
    float* br_out;
    DropArgs br_drop;
};
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_small_kernel(const LnbSmall p) {
    __shared__ float red[3][4][NV * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D = p.D, nv = D >> 2;
    f32x4 wv[NV], dw[NV], db[NV], dbr[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        wv[i] = ((const f32x4*)p.w)[min(lane + 64 * i, nv - 1)];
#pragma unroll
        for (int e = 0; e < 4; ++e) dw[i][e] = db[i][e] = dbr[i][e] = 0.f;
    }
    const int row_begin = (int)((long long)blockIdx.x * p.rows / (int)gridDim.x);
    const int row_end = (int)((long long)(blockIdx.x + 1) * p.rows / (int)gridDim.x);
    for (int row = row_begin + wave; row < row_end; row += 4) {   // wave-uniform
        f32x4 xv[NV], gv[NV], dv[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = min(lane + 64 * i, nv - 1);
            xv[i] = ((const f32x4*)(p.x + (size_t)row * D))[c];
            gv[i] = ((const f32x4*)(p.g + (size_t)row * D))[c];
            if (p.dres_in) dv[i] = ((const f32x4*)(p.dres_in + (size_t)row * D))[c];
        }
        constexpr int G = NV <= 3 ? 5 : 2;
        for (int s0 = 1; s0 < p.g_splits; s0 += G) {
            f32x4 t[G][NV];
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const f32x4* ps = (const f32x4*)(p.g + (size_t)min(s0 + u, p.g_splits - 1) * p.g_stride + (size_t)row * D);
#pragma unroll
                for (int i = 0; i < NV; ++i) t[u][i] = ps[min(lane + 64 * i, nv - 1)];
            }
#pragma unroll
            for (int u = 0; u < G; ++u)
                if (s0 + u < p.g_splits) {
#pragma unroll
                    for (int i = 0; i < NV; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) gv[i][e] += t[u][i][e];
                }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float t = (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
            s += (lane + 64 * i < nv) ? t : 0.f;
        }
        const float mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xv[i][e] -= mean;
                t = fmaf(xv[i][e], xv[i][e], t);
            }
            q += (lane + 64 * i < nv) ? t : 0.f;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + p.eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const bool ok = lane + 64 * i < nv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xv[i][e] *= rstd;  // xhat
                const float gw = gv[i][e] * wv[i][e];
                if (ok) {
                    s1 += gw;
                    s2 = fmaf(gw, xv[i][e], s2);
                    dw[i][e] = fmaf(gv[i][e], xv[i][e], dw[i][e]);
                    db[i][e] += gv[i][e];
                }
            }
        }
        const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
        const unsigned bkey = drop_key(p.br_drop.seed, p.br_drop.stream, (unsigned)row);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                if (p.dres_in) o = dv[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += rstd * (gv[i][e] * wv[i][e] - c1 - xv[i][e] * c2);
                ((f32x4*)(p.dres_out + (size_t)row * D))[c] = o;
                if (p.sets == 3) {
                    if (p.br_drop.thresh) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            o[e] = drop_keep(bkey, (unsigned)(4 * c + e), p.br_drop.thresh) ? o[e] * p.br_drop.scale : 0.f;
                        ((f32x4*)(p.br_out + (size_t)row * D))[c] = o;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) dbr[i][e] += o[e];
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[0][wave][(lane + 64 * i) * 4 + e] = dw[i][e];
            red[1][wave][(lane + 64 * i) * 4 + e] = db[i][e];
            red[2][wave][(lane + 64 * i) * 4 + e] = dbr[i][e];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256)
        for (int k = 0; k < p.sets; ++k)
            p.partial[((size_t)blockIdx.x * p.sets + k) * D + c] = (red[k][0][c] + red[k][1][c]) + (red[k][2][c] + red[k][3][c]);
}

// ---- transposed bilinear upsample: G[B,C,S,S] -> dZ[B,C,g,g] -----------------------------------
// Gather form of the adjoint of upsample_kernel: cell (y, x) collects every output pixel that has it as a
// tap, with the same tap arithmetic as the forward.  One wave per low-res cell; deterministic.
__device__ __forceinline__ void taps_bwd(int d, float scale, int n_in, int& i0, int& i1, float& w0, float& w1) {
    float src = scale * ((float)d + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = min((int)floorf(src), n_in - 1);
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    w1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
    w0 = 1.f - w1;
}
// One wave per low-res cell (any S): each wave walks its cell's 3P x 3P window, 192-byte row pieces.
__global__ __launch_bounds__(256) void upsample_bwd_cell_kernel(const float* __restrict__ G, float* __restrict__ dZ, int B,
                                                                int C, int g, int S) {
    const int lane = threadIdx.x & 63;
    const size_t cell = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= (size_t)B * C * g * g) return;
    const int x = (int)(cell % g), y = (int)((cell / g) % g);
    const size_t bc = cell / ((size_t)g * g);
    const int P = S / g;
    const float scale = (float)g / (float)S;
    const int Y0 = max(0, (y - 1) * P), Y1 = min(S, (y + 2) * P);
    const int X0 = max(0, (x - 1) * P), X1 = min(S, (x + 2) * P);
    const float* Gp = G + bc * (size_t)S * S;
    float acc = 0.f;
    for (int X = X0 + lane; X < X1; X += 64) {
        int i0, i1;
        float w0, w1;
        taps_bwd(X, scale, g, i0, i1, w0, w1);
        const float wx = (i0 == x ? w0 : 0.f) + (i1 == x ? w1 : 0.f);
        if (wx == 0.f) continue;
        float col = 0.f;
        for (int Y = Y0; Y < Y1; ++Y) {
            taps_bwd(Y, scale, g, i0, i1, w0, w1);
            const float wy = (i0 == y ? w0 : 0.f) + (i1 == y ? w1 : 0.f);
            col = fmaf(wy, Gp[(size_t)Y * S + X], col);
        }
        acc = fmaf(wx, col, acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) dZ[cell] = acc;
}
// One block per ROW of cells (image-class plane bc, cell row y), S <= UPB_MAX_S (round 4; the cell form above read its
// window in 192-byte pieces, 9x over, at 0.5 TB/s): the rows of G that carry weight for cell row y are read ONCE by the
// block, whole and coalesced (thread t owns pixel columns t, t + 256, ...: col[X] = sum_Y wy(Y) G[Y][X], rows in
// increasing Y), the column sums are parked in LDS and each cell adds its window of them in a fixed order (8 threads per
// cell at g = 32, a fixed shuffle tree).  Deterministic; 403 MB instead of ~1.2 GB through the L2 at B = 64, C = 2.
constexpr int UPB_MAX_S = 2048;
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ G, float* __restrict__ dZ, int B,
                                                           int C, int g, int S) {
    __shared__ float colsum[UPB_MAX_S];
    const int tid = threadIdx.x;
    const int y = (int)(blockIdx.x % g);
    const size_t bc = blockIdx.x / g;
    const int P = S / g;
    const float scale = (float)g / (float)S;
    const int Y0 = max(0, (y - 1) * P), Y1 = min(S, (y + 2) * P);
    const float* Gp = G + bc * (size_t)S * S;
    constexpr int NX = UPB_MAX_S / 256;
    float acc[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) acc[i] = 0.f;
    for (int Y = Y0; Y < Y1; ++Y) {
        int i0, i1;
        float w0, w1;
        taps_bwd(Y, scale, g, i0, i1, w0, w1);
        const float wy = (i0 == y ? w0 : 0.f) + (i1 == y ? w1 : 0.f);   // block-uniform
        if (wy == 0.f) continue;
        const float* row = Gp + (size_t)Y * S;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int X = tid + 256 * i;
            if (X < S) acc[i] = fmaf(wy, row[X], acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int X = tid + 256 * i;
        if (X < S) colsum[X] = acc[i];
    }
    __syncthreads();
    // tpc threads per cell (a power of two, <= 64 and <= 256 / min(g, 256)); cells in passes of 256 / tpc
    int tpc = 1;
    while (tpc < 64 && tpc * 2 * g <= 256) tpc *= 2;
    const int sub = tid & (tpc - 1), cpp = 256 / tpc;
    for (int x0 = 0; x0 < g; x0 += cpp) {
        const int x = x0 + tid / tpc;
        float a = 0.f;
        if (x < g) {
            const int X0 = max(0, (x - 1) * P), X1 = min(S, (x + 2) * P);
            for (int X = X0 + sub; X < X1; X += tpc) {
                int i0, i1;
                float w0, w1;
                taps_bwd(X, scale, g, i0, i1, w0, w1);
                const float wx = (i0 == x ? w0 : 0.f) + (i1 == x ? w1 : 0.f);
                a = fmaf(wx, colsum[X], a);
            }
        }
        for (int o = tpc >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if (x < g && sub == 0) dZ[(bc * g + y) * (size_t)g + x] = a;
    }
}

// ---- seg_head.2 (1x1 conv) backward + ReLU backward --------------------------------------------
// dFpre[m][j] = (F[m][j] > 0) * sum_c dZ[b,c,t] W2[c][j];  dW2[c][j] = sum_m dZ[b,c,t] F[m][j].
// Thread = mid channel j, block = rpb rows (64; 8 for the few rows of the reference's batch: 98 blocks of 8 dependent rows
// instead of 13 of 64 -- 128 us at 784 rows); partial dW2 to partial[block][C][256].  C <= 32.
__global__ __launch_bounds__(256) void head1x1_bwd_kernel(const float* __restrict__ dZ, const float* __restrict__ F,
                                                          const float* __restrict__ W2, float* __restrict__ dFpre,
                                                          float* __restrict__ partial, int B, int Np, int C, int rpb) {
    const int j = threadIdx.x;
    float w[32], dw[32];
    for (int c = 0; c < C; ++c) {
        w[c] = W2[c * MID + j];
        dw[c] = 0.f;
    }
    const int M = B * Np;
    for (int it = 0; it < rpb; ++it) {
        const int m = blockIdx.x * rpb + it;
        if (m >= M) break;
        const int b = m / Np, t = m - b * Np;
        const float f = F[(size_t)m * MID + j];
        float s = 0.f;
        for (int c = 0; c < C; ++c) {
            const float dz = dZ[((size_t)b * C + c) * Np + t];
            s = fmaf(dz, w[c], s);
            dw[c] = fmaf(dz, f, dw[c]);
        }
        dFpre[(size_t)m * MID + j] = f > 0.f ? s : 0.f;
    }
    for (int c = 0; c < C; ++c) partial[((size_t)blockIdx.x * C + c) * MID + j] = dw[c];
}
// db2[c] = sum over (b, t) of dZ[b,c,t]; one block per class
__global__ __launch_bounds__(1024) void head_bias_bwd_kernel(const float* __restrict__ dZ, float* __restrict__ db2, int B,
                                                             int Np, int C) {
    // 1024 threads x 4 independent chains per class (256 threads x one chain was a 256-deep dependent load-add chain:
    // 85 us for 0.5 MB); fixed order: chain j of thread t takes elements t + 1024 (4 k + j), then a fixed tree
    __shared__ float red[1024];
    const int c = blockIdx.x, n = B * Np;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i0 = threadIdx.x; i0 < n; i0 += 4096) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + 1024 * j;
            if (i < n) {
                const int b = i / Np, t = i - b * Np;
                s[j] += dZ[((size_t)b * C + c) * Np + t];
            }
        }
    }
    red[threadIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) db2[c] = red[0];
}

// ---- im2col materialisation for the two conv weight gradients (T-form B operands of the wgrad GEMM) ----
// 3x3, pad 1 on the token-major map: T[m][tap*D + d] = H[(b, y+ky-1, x+kx-1)][d] or 0
template <typename HT>
__global__ __launch_bounds__(256) void im2col3x3_kernel(const HT* __restrict__ H, float* __restrict__ T, int B, int g,
                                                        int D) {
    const int nv = D >> 2;
    const size_t total = (size_t)B * g * g * 9 * nv;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int dv = (int)(i % nv);
        const int tap = (int)((i / nv) % 9);
        const size_t m = i / ((size_t)nv * 9);
        const int x = (int)(m % g), y = (int)((m / g) % g);
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)yy < (unsigned)g && (unsigned)xx < (unsigned)g)
            v = ld4(H + (m + (ptrdiff_t)(yy - y) * g + (xx - x)) * D + 4 * dv);
        ((f32x4*)T)[i] = v;
    }
}
// patch rows of the NCHW image: T[m][(c, py, px)]
__global__ __launch_bounds__(256) void im2col_patch_kernel(const float* __restrict__ img, float* __restrict__ T, int B,
                                                           int Cin, int S, int P) {
    const int g = S / P, Kp = Cin * P * P, nv = Kp >> 2;
    const size_t total = (size_t)B * g * g * nv;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int kv = (int)(i % nv);
        const size_t m = i / nv;
        const int gx = (int)(m % g), gy = (int)((m / g) % g), b = (int)(m / ((size_t)g * g));
        const int k = kv * 4, c = k / (P * P), rem = k - c * P * P, py = rem / P, px = rem - py * P;
        ((f32x4*)T)[i] = *(const f32x4*)(img + (((size_t)b * Cin + c) * S + gy * P + py) * S + gx * P + px);
    }
}
// seg_head.0 weights for the input-gradient conv: Wd[d][t][o] = W0[o][8 - t][d]  (flipped taps)
__global__ __launch_bounds__(256) void conv_dgrad_weight_kernel(const float* __restrict__ W0, float* __restrict__ Wd,
                                                                int D) {
    const size_t total = (size_t)D * 9 * MID;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int o = (int)(i % MID), t = (int)((i / MID) % 9), d = (int)(i / (MID * 9));
        Wd[i] = W0[((size_t)o * 9 + (8 - t)) * D + d];
    }
}

// ---- embeddings backward: dpos[1+t] = sum_b dX[b*Np+t];  dpos[0] = dcls = sum_b dX[B*Np+b] ----
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dX, float* __restrict__ dpos,
                                                        float* __restrict__ dcls, int B, int Np, int D) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)(Np + 1) * D) return;
    const int d = (int)(i % D), n = (int)(i / D);
    float s = 0.f;
    if (n == 0) {
        for (int b = 0; b < B; ++b) s += dX[((size_t)B * Np + b) * D + d];
        dcls[d] = s;
    } else {
        for (int b = 0; b < B; ++b) s += dX[((size_t)b * Np + n - 1) * D + d];
    }
    dpos[i] = s;
}

// ---- Adam (torch.optim.Adam defaults, single fused pass over the flat arena) --------------------
// exp_avg.lerp_(g, 1-b1); exp_avg_sq = b2*v + (1-b2) g^2; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// decay = 1 - lr * weight_decay: torch.optim.AdamW's decoupled weight decay, p.mul_(decay) in front of the update
// (model/PAED/classes.py:536-548); 1 (an exact multiply) for plain Adam
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, size_t n4, float b1,
                                                   float b2, float eps, float step_size, float inv_bc2_sqrt,
                                                   float grad_scale, float decay) {
#pragma clang fp contract(off)   // p.mul_(decay) is rounded before the update is subtracted, as torch does it
    // two 16-byte pieces of each stream per thread and trip (eight loads in flight); the gradient and the two moments are touched
    // once per step: non-temporal, so that they do not push the parameters (re-read by the next forward) out of the caches
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 2 * stride) {
        f32x4 pv[2], gv[2], mv[2], vv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const size_t i = i0 + u * stride < n4 ? i0 + u * stride : i0;
            pv[u] = ((f32x4*)p)[i];
            gv[u] = __builtin_nontemporal_load((const f32x4*)g + i);
            mv[u] = __builtin_nontemporal_load((f32x4*)m + i);
            vv[u] = __builtin_nontemporal_load((f32x4*)v + i);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const size_t i = i0 + u * stride;
            if (i >= n4) break;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gr = gv[u][e] * grad_scale;
                mv[u][e] = mv[u][e] + (gr - mv[u][e]) * (1.0f - b1);
                vv[u][e] = vv[u][e] * b2 + (1.0f - b2) * gr * gr;
                const float denom = sqrtf(vv[u][e]) * inv_bc2_sqrt + eps;
                pv[u][e] = pv[u][e] * decay - step_size * (mv[u][e] / denom);
            }
            ((f32x4*)p)[i] = pv[u];
            __builtin_nontemporal_store(mv[u], (f32x4*)m + i);
            __builtin_nontemporal_store(vv[u], (f32x4*)v + i);
        }
    }
}

// bf16 transpose with zero padding of the (new) contiguous dimension: out[c][r] = in[r][c], r < R; 0 for
// R <= r < Rpad.  Feeds the bf16 weight-gradient GEMMs, whose reduction index (the token row) must be
// contiguous and a multiple of 64.  64x64 tiles through LDS.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                             int R, int C, int ldin, int Rpad) {
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + i * 4 + ty, c = c0 + tx;
        tile[i * 4 + ty][tx] = (r < R && c < C) ? in[(size_t)r * ldin + c] : (bf16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + i * 4 + ty, r = r0 + tx;
        if (c < C && r < Rpad) out[(size_t)c * Rpad + r] = tile[tx][i * 4 + ty];
    }
}

// All dgrad operands of a training step in ONE launch: the four weight matrices of every encoder layer, [R][C] bf16 in
// the shadow arena -> [C][R] (48 small transposes between the big GEMMs cost ~10 us each of mostly launch latency).
// blockIdx.z = 4 * layer + kind; the grid covers the largest matrix, smaller ones leave early.
struct TransposeBatch {
    size_t src0[4], dst0[4];   // element offsets of layer 0's matrices in the shadow arena / in the output
    int R[4], C[4];
    size_t src_stride, dst_stride;   // per layer
};
__global__ __launch_bounds__(256) void transpose_layers_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                                    TransposeBatch tb) {
    __shared__ bf16_t tile[64][66];
    const int kind = blockIdx.z & 3, layer = blockIdx.z >> 2;
    const int R = tb.R[kind], C = tb.C[kind];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    if (r0 >= R || c0 >= C) return;   // block-uniform
    const bf16_t* src = in + tb.src0[kind] + (size_t)layer * tb.src_stride;
    bf16_t* dst = out + tb.dst0[kind] + (size_t)layer * tb.dst_stride;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + i * 4 + ty, c = c0 + tx;
        tile[i * 4 + ty][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : (bf16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + i * 4 + ty, r = r0 + tx;
        if (c < C && r < R) dst[(size_t)c * R + r] = tile[tx][i * 4 + ty];
    }
}

inline int grid_for(size_t n) { return (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096); }

}  // namespace

// also holds the fused form's partial rows: two per 256-row tile of the 8-phase GEMM + the rows it did not cover (kernels.hpp)
size_t colsum_scratch_floats(int M, int N) { return (size_t)(2 * ((M + 255) / 256) + 2) * N; }
// column sums of the trailing rows of a 16-bit matrix into partial row `chunk0` (+ following), then the fixed-order reduce
// over all `chunk0 + chunks` partial rows (the first chunk0 were written by the GEMM epilogue)
int launch_colsum_finish_fused(const void* tail_rows, int tail, int chunk0, float* out, float* scratch, int N, int ld,
                               hipStream_t s) {
    int chunks = chunk0;
    if (tail > 0) {
        const int tc = (tail + 255) / 256;
        hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, dim3((N + 255) / 256, tc), dim3(256), 0, s, (const bf16_t*)tail_rows,
                           scratch + (size_t)chunk0 * N, tail, N, ld);
        VITSEG_LAUNCH_CHECK("colsum_partial(tail)");
        chunks += tc;
    }
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((N + 63) / 64), dim3(1024), 0, s, scratch, out, chunks, N);
    VITSEG_LAUNCH_CHECK("colsum_finish");
    return VITSEG_OK;
}
// Few rows (the reference's batch 4 x 224x224: 788): one launch -- a block owns 16 columns, its 64 row groups walk the rows
// 64 apart (64-byte pieces of a row per group, four independent sums per group: 13 rows = 4 dependent loads at 788 rows),
// LDS combines the groups in group order -- instead of the partial + finish pair (7 + 5 us, many times per training step).
__global__ __launch_bounds__(1024) void colsum_small_kernel(const float* __restrict__ X, float* __restrict__ out, int M, int N, int ld) {
    __shared__ float red[64][17];
    const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int col = min((int)blockIdx.x * 16 + c, N - 1);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = rg;
    for (; r + 192 < M; r += 256) {
        s0 += X[(size_t)r * ld + col];
        s1 += X[(size_t)(r + 64) * ld + col];
        s2 += X[(size_t)(r + 128) * ld + col];
        s3 += X[(size_t)(r + 192) * ld + col];
    }
    if (r < M) s0 += X[(size_t)r * ld + col];
    if (r + 64 < M) s1 += X[(size_t)(r + 64) * ld + col];
    if (r + 128 < M) s2 += X[(size_t)(r + 128) * ld + col];
    red[rg][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && (int)blockIdx.x * 16 + c < N) {
        float t = red[0][c];
#pragma unroll
        for (int g = 1; g < 64; ++g) t += red[g][c];
        out[col] = t;
    }
}

int launch_colsum(const void* X, int x_is_bf16, float* out, float* scratch, int M, int N, int ld, hipStream_t s) {
    if (!x_is_bf16 && M <= 4096) {
        hipLaunchKernelGGL(colsum_small_kernel, dim3((N + 15) / 16), dim3(1024), 0, s, (const float*)X, out, M, N, ld);
        VITSEG_LAUNCH_CHECK("colsum_small");
        return VITSEG_OK;
    }
    const int chunks = (M + 255) / 256;
    if (x_is_bf16)
        hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, dim3((N + 255) / 256, chunks), dim3(256), 0, s,
                           (const bf16_t*)X, scratch, M, N, ld);
    else
        hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3((N + 255) / 256, chunks), dim3(256), 0, s,
                           (const float*)X, scratch, M, N, ld);
    VITSEG_LAUNCH_CHECK("colsum_partial");
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((N + 63) / 64), dim3(1024), 0, s, scratch, out, chunks, N);
    VITSEG_LAUNCH_CHECK("colsum_finish");
    return VITSEG_OK;
}

int launch_transpose_layers_bf16(const void* arena_lp, void* out, const size_t src0[4], const int R[4], const int C[4],
                                 size_t src_stride, int layers, hipStream_t s) {
    TransposeBatch tb{};
    size_t off = 0;
    int maxR = 0, maxC = 0;
    for (int k = 0; k < 4; ++k) {
        tb.src0[k] = src0[k];
        tb.dst0[k] = off;
        tb.R[k] = R[k];
        tb.C[k] = C[k];
        off += (size_t)R[k] * C[k];
        maxR = R[k] > maxR ? R[k] : maxR;
        maxC = C[k] > maxC ? C[k] : maxC;
    }
    tb.src_stride = src_stride;
    tb.dst_stride = off;
    hipLaunchKernelGGL(transpose_layers_bf16_kernel, dim3((maxC + 63) / 64, (maxR + 63) / 64, 4 * layers), dim3(256), 0, s,
                       (const bf16_t*)arena_lp, (bf16_t*)out, tb);
    VITSEG_LAUNCH_CHECK("transpose_layers_bf16");
    return VITSEG_OK;
}

int launch_transpose_bf16(const void* in, void* out, int R, int C, int ldin, int Rpad, hipStream_t s) {
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((C + 63) / 64, (Rpad + 63) / 64), dim3(256), 0, s, (const bf16_t*)in,
                       (bf16_t*)out, R, C, ldin, Rpad);
    VITSEG_LAUNCH_CHECK("transpose_bf16");
    return VITSEG_OK;
}

// grid of the LayerNorm backward: 64-row blocks, capped at what is resident at once (3 blocks of 4 waves per CU up to D = 768:
// <= 168 VGPRs by __launch_bounds__, 36 KiB of LDS each; 2 beyond); few rows (the reference's batch 4 x 224x224: 788) are cut
// finer, 8 rows per block, instead of 13 blocks on 256 CUs (22 -> 8 us per launch)
static int lnb_blocks(int rows, int D) {
    const int nvl = (D / 4 + 63) / 64;
    const int cap = (nvl <= 3 ? 3 : 2) * device_num_cus();
    const int per = rows < 64 * device_num_cus() ? 8 : 64;
    const int b = (rows + per - 1) / per;
    return b < cap ? b : cap;
}
size_t layernorm_bwd_scratch_floats(int rows, int D) {
    const size_t fine = (size_t)(rows + 7) / 8, coarse = (size_t)(rows + 63) / 64;   // (an upper bound of lnb_blocks for any CU count)
    return (fine < 4096 ? (fine > coarse ? fine : coarse) : coarse) * 3 * D;
}
int launch_layernorm_bwd(const float* x, const float* w, const void* g, int g_is_bf16, const float* dres_in,
                         float* dres_out, float* dw, float* db, float* scratch, int rows, int D, float eps,
                         hipStream_t s, void* br_out, DropArgs br_drop, float* br_dbias) {
    VITSEG_CHECK_ARG(D % 4 == 0 && D <= 1024, VITSEG_ESHAPE, "layernorm_bwd: D=%d must be a multiple of 4, <= 1024", D);
    VITSEG_CHECK_ARG(!br_out || (g_is_bf16 && br_dbias), VITSEG_EINVAL, "layernorm_bwd: branch output needs bf16 g + dbias");
    const int nvl = (D / 4 + 63) / 64;
    const int blocks = lnb_blocks(rows, D);
#define VITSEG_LNB(NV)                                                                                            \
    do {                                                                                                          \
        if (br_out)                                                                                               \
            hipLaunchKernelGGL((layernorm_bwd_kernel<NV, bf16_t, true>), dim3(blocks), dim3(256), 0, s, x, w,     \
                               (const bf16_t*)g, dres_in, dres_out, scratch, rows, D, eps, (bf16_t*)br_out,       \
                               br_drop);                                                                          \
        else if (g_is_bf16)                                                                                       \
            hipLaunchKernelGGL((layernorm_bwd_kernel<NV, bf16_t, false>), dim3(blocks), dim3(256), 0, s, x, w,    \
                               (const bf16_t*)g, dres_in, dres_out, scratch, rows, D, eps, (bf16_t*)nullptr,      \
                               br_drop);                                                                          \
        else                                                                                                      \
            hipLaunchKernelGGL((layernorm_bwd_kernel<NV, float, false>), dim3(blocks), dim3(256), 0, s, x, w,     \
                               (const float*)g, dres_in, dres_out, scratch, rows, D, eps, (bf16_t*)nullptr,       \
                               br_drop);                                                                          \
    } while (0)
    if (nvl <= 1) VITSEG_LNB(1);
    else if (nvl <= 2) VITSEG_LNB(2);
    else if (nvl <= 3) VITSEG_LNB(3);
    else VITSEG_LNB(4);
#undef VITSEG_LNB
    VITSEG_LAUNCH_CHECK("layernorm_bwd");
    const int sets = br_out ? 3 : 2;
    hipLaunchKernelGGL(layernorm_bwd_finish_kernel, dim3((sets * D + 63) / 64), dim3(1024), 0, s, scratch, dw, db,
                       br_dbias, blocks, D, sets);
    VITSEG_LAUNCH_CHECK("layernorm_bwd_finish");
    return VITSEG_OK;
}

int launch_layernorm_bwd_small(const float* x, const float* w, const float* g, size_t g_stride, int g_splits,
                               const float* dres_in, float* dres_out, float* dw, float* db, float* scratch, int rows, int D,
                               float eps, hipStream_t s, float* br_out, DropArgs br_drop, float* br_dbias) {
    VITSEG_CHECK_ARG(x && w && g && dres_out && dw && db && scratch && rows > 0 && g_splits >= 1, VITSEG_EINVAL, "layernorm_bwd_small: bad arguments");
    VITSEG_CHECK_ARG(D % 4 == 0 && D <= 1024, VITSEG_ESHAPE, "layernorm_bwd_small: D=%d must be a multiple of 4, <= 1024", D);
    VITSEG_CHECK_ARG(!br_drop.thresh || (br_out && br_dbias), VITSEG_EINVAL, "layernorm_bwd_small: a dropped branch needs br_out and br_dbias");
    const int nvl = (D / 4 + 63) / 64;
    const int blocks = lnb_blocks(rows, D);
    LnbSmall p{x, w, g, g_stride, g_splits, dres_in, dres_out, scratch, rows, D, br_dbias ? 3 : 2, eps, br_out, br_drop};
    if (nvl <= 1) hipLaunchKernelGGL(layernorm_bwd_small_kernel<1>, dim3(blocks), dim3(256), 0, s, p);
    else if (nvl <= 2) hipLaunchKernelGGL(layernorm_bwd_small_kernel<2>, dim3(blocks), dim3(256), 0, s, p);
    else if (nvl <= 3) hipLaunchKernelGGL(layernorm_bwd_small_kernel<3>, dim3(blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(layernorm_bwd_small_kernel<4>, dim3(blocks), dim3(256), 0, s, p);
    VITSEG_LAUNCH_CHECK("layernorm_bwd_small");
    hipLaunchKernelGGL(layernorm_bwd_finish_kernel, dim3((p.sets * D + 63) / 64), dim3(1024), 0, s, scratch, dw, db, br_dbias,
                       blocks, D, p.sets);
    VITSEG_LAUNCH_CHECK("layernorm_bwd_finish");
    return VITSEG_OK;
}

int launch_upsample_bwd(const float* G, float* dZ, int B, int C, int g, int S, hipStream_t s) {
    const size_t cells = (size_t)B * C * g * g;
    if (S <= UPB_MAX_S)
        hipLaunchKernelGGL(upsample_bwd_kernel, dim3((unsigned)((size_t)B * C * g)), dim3(256), 0, s, G, dZ, B, C, g, S);
    else
        hipLaunchKernelGGL(upsample_bwd_cell_kernel, dim3((unsigned)((cells + 3) / 4)), dim3(256), 0, s, G, dZ, B, C, g, S);
    VITSEG_LAUNCH_CHECK("upsample_bwd");
    return VITSEG_OK;
}

static int head1x1_rpb(int M) { return M < 16384 ? 8 : 64; }   // a function of the row count only (the order of the dW2 sum)
size_t head1x1_bwd_scratch_floats(int B, int Np, int C) {
    const int M = B * Np, rpb = head1x1_rpb(M);
    return (size_t)((M + rpb - 1) / rpb) * C * MID;
}
int launch_head1x1_bwd(const float* dZ, const float* F, const float* W2, float* dFpre, float* dW2, float* db2,
                       float* scratch, int B, int Np, int C, hipStream_t s) {
    VITSEG_CHECK_ARG(C <= 32, VITSEG_ESHAPE, "training supports at most 32 classes (got %d)", C);
    const int rpb = head1x1_rpb(B * Np), blocks = (B * Np + rpb - 1) / rpb;
    hipLaunchKernelGGL(head1x1_bwd_kernel, dim3(blocks), dim3(256), 0, s, dZ, F, W2, dFpre, scratch, B, Np, C, rpb);
    VITSEG_LAUNCH_CHECK("head1x1_bwd");
    // scratch is [blocks][C*256]: column sums over the blocks
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((C * MID + 63) / 64), dim3(1024), 0, s, scratch, dW2, blocks, C * MID);
    VITSEG_LAUNCH_CHECK("head1x1_bwd_finish");
    hipLaunchKernelGGL(head_bias_bwd_kernel, dim3(C), dim3(1024), 0, s, dZ, db2, B, Np, C);
    VITSEG_LAUNCH_CHECK("head_bias_bwd");
    return VITSEG_OK;
}

// bf16 -> bf16 copy of the same gather (raw 16-bit moves, 8 channels per thread): the T-form operand of the bf16
// weight-gradient GEMM of seg_head.0
__global__ __launch_bounds__(256) void im2col3x3_bf16_kernel(const bf16_t* __restrict__ H, bf16_t* __restrict__ T, int B,
                                                             int g, int D) {
    const int nv = D >> 3;
    const size_t total = (size_t)B * g * g * 9 * nv;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int dv = (int)(i % nv);
        const int tap = (int)((i / nv) % 9);
        const size_t m = i / ((size_t)nv * 9);
        const int x = (int)(m % g), y = (int)((m / g) % g);
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        uint4 v = {0u, 0u, 0u, 0u};
        if ((unsigned)yy < (unsigned)g && (unsigned)xx < (unsigned)g)
            v = *(const uint4*)(H + (m + (ptrdiff_t)(yy - y) * g + (xx - x)) * D + 8 * dv);
        ((uint4*)T)[i] = v;
    }
}

int launch_im2col3x3_bf16(const void* H, void* T, int B, int g, int D, hipStream_t s) {
    VITSEG_CHECK_ARG(D % 8 == 0, VITSEG_ESHAPE, "im2col3x3_bf16: D %% 8");
    hipLaunchKernelGGL(im2col3x3_bf16_kernel, dim3(grid_for((size_t)B * g * g * 9 * (D / 8))), dim3(256), 0, s,
                       (const bf16_t*)H, (bf16_t*)T, B, g, D);
    VITSEG_LAUNCH_CHECK("im2col3x3_bf16");
    return VITSEG_OK;
}

int launch_im2col3x3(const void* H, int h_is_bf16, float* T, int B, int g, int D, hipStream_t s) {
    const dim3 grid(grid_for((size_t)B * g * g * 9 * (D / 4)));
    if (h_is_bf16)
        hipLaunchKernelGGL(im2col3x3_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)H, T, B, g, D);
    else
        hipLaunchKernelGGL(im2col3x3_kernel<float>, grid, dim3(256), 0, s, (const float*)H, T, B, g, D);
    VITSEG_LAUNCH_CHECK("im2col3x3");
    return VITSEG_OK;
}
// the same rows rounded to bf16 (T-form operand of the bf16 patch-embedding weight gradient)
__global__ __launch_bounds__(256) void im2col_patch_bf16_kernel(const float* __restrict__ img, bf16_t* __restrict__ T, int B,
                                                                int Cin, int S, int P) {
    const int g = S / P, Kp = Cin * P * P, nv = Kp >> 2;
    const size_t total = (size_t)B * g * g * nv;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int kv = (int)(i % nv);
        const size_t m = i / nv;
        const int gx = (int)(m % g), gy = (int)((m / g) % g), b = (int)(m / ((size_t)g * g));
        const int k = kv * 4, c = k / (P * P), rem = k - c * P * P, py = rem / P, px = rem - py * P;
        const f32x4 v = *(const f32x4*)(img + (((size_t)b * Cin + c) * S + gy * P + py) * S + gx * P + px);
        uint2 h;
        h.x = pack2_bf16(v[0], v[1]);
        h.y = pack2_bf16(v[2], v[3]);
        ((uint2*)T)[i] = h;
    }
}
int launch_im2col_patch_bf16(const float* img, void* T, int B, int Cin, int S, int P, hipStream_t s) {
    hipLaunchKernelGGL(im2col_patch_bf16_kernel, dim3(grid_for((size_t)B * (S / P) * (S / P) * (Cin * P * P / 4))), dim3(256),
                       0, s, img, (bf16_t*)T, B, Cin, S, P);
    VITSEG_LAUNCH_CHECK("im2col_patch_bf16");
    return VITSEG_OK;
}
int launch_im2col_patch(const float* img, float* T, int B, int Cin, int S, int P, hipStream_t s) {
    hipLaunchKernelGGL(im2col_patch_kernel, dim3(grid_for((size_t)B * (S / P) * (S / P) * (Cin * P * P / 4))), dim3(256),
                       0, s, img, T, B, Cin, S, P);
    VITSEG_LAUNCH_CHECK("im2col_patch");
    return VITSEG_OK;
}
int launch_conv_dgrad_weight(const float* W0, float* Wd, int D, hipStream_t s) {
    hipLaunchKernelGGL(conv_dgrad_weight_kernel, dim3(grid_for((size_t)D * 9 * MID)), dim3(256), 0, s, W0, Wd, D);
    VITSEG_LAUNCH_CHECK("conv_dgrad_weight");
    return VITSEG_OK;
}
int launch_embed_bwd(const float* dX, float* dpos, float* dcls, int B, int Np, int D, hipStream_t s) {
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)(((size_t)(Np + 1) * D + 255) / 256)), dim3(256), 0, s, dX, dpos,
                       dcls, B, Np, D);
    VITSEG_LAUNCH_CHECK("embed_bwd");
    return VITSEG_OK;
}
int launch_adam(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, int step,
                float grad_scale, hipStream_t s, float weight_decay) {
    VITSEG_CHECK_ARG(n % 4 == 0 && step >= 1, VITSEG_EINVAL, "adam: n %% 4 != 0 or step < 1");
    const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, p, g, m, v, n / 4, b1, b2, eps,
                       (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), grad_scale, (float)(1.0 - (double)lr * (double)weight_decay));
    VITSEG_LAUNCH_CHECK("adam");
    return VITSEG_OK;
}

}  // namespace vitseg
