// Backward of the attention core, bf16 operands / fp32 accumulation (mixed-precision training).
// Same algorithm and two-kernel, atomic-free structure as attention_bwd_f32.hip; the five products run on
// v_mfma_f32_32x32x16_bf16:
//   S / dP   : A = token rows from LDS (one ds_read_b128 per 16-wide k-step), B = this lane's row in registers;
//   dQ^T = K^T dS^T,  dV^T = dO^T P,  dK^T = Q^T dS :
//       B = the exponentiated / differentiated accumulator registers 8s..8s+7 packed to bf16 (the k index of a
//       lane half is token 16s + 8(j>>2) + 4h + (j&3)); A = the TRANSPOSE of a row-major LDS tile, gathered with
//       two ds_read_b64_tr_b16 per k-step (same addressing as V in attention_bf16.hip).
// Inputs q|k|v, ctx, dctx are bf16; lse/delta fp32; dq|dk|dv are written as bf16.
#include "kernels.hpp"

namespace vitseg {
namespace {

constexpr int HD = 64, TB = 128, TT = 64;
constexpr float LOG2E = 1.4426950408889634f;
typedef unsigned short bf16_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int kappa(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }
__device__ __forceinline__ size_t tok_row(int b, int n, int B, int Np) {
    return n < Np ? (size_t)b * Np + n : (size_t)B * Np + b;
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// delta[b][h][n] = sum_d dO * O.  A thread owns 8 consecutive channels of a token row (one 16-byte load of each
// tensor), the 8 lanes of a head combine with three lane swaps; consecutive threads walk a row, so a wave reads 1 KiB
// contiguous.  (One element per lane -- 128-byte loads -- ran at 1.1 TB/s.)
__global__ __launch_bounds__(256) void attn_delta_bf16_kernel(const bf16_t* __restrict__ ctx,
                                                              const bf16_t* __restrict__ dctx,
                                                              float* __restrict__ delta, int B, int Np, int A) {
    const int N = Np + 1, cpr = A * 8;                     // 16-byte chunks per row
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)(B * N) * cpr;
    float v = 0.f;
    size_t row = 0;
    int h = 0;
    if (i < total) {
        row = i / cpr;
        const int c = (int)(i - row * cpr);
        h = c >> 3;
        const uint4 o = *(const uint4*)(ctx + row * (size_t)(A * HD) + c * 8);
        const uint4 d = *(const uint4*)(dctx + row * (size_t)(A * HD) + c * 8);
        v = bf_lo(o.x) * bf_lo(d.x) + bf_hi(o.x) * bf_hi(d.x) + bf_lo(o.y) * bf_lo(d.y) + bf_hi(o.y) * bf_hi(d.y) +
            bf_lo(o.z) * bf_lo(d.z) + bf_hi(o.z) * bf_hi(d.z) + bf_lo(o.w) * bf_lo(d.w) + bf_hi(o.w) * bf_hi(d.w);
    }
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    if (i < total && (threadIdx.x & 7) == 0) {
        // row index in the patches-first layout -> (image, token) with the CLS token last
        const size_t BNp = (size_t)B * Np;
        const int bimg = row < BNp ? (int)(row / Np) : (int)(row - BNp);
        const int n = row < BNp ? (int)(row - (size_t)bimg * Np) : Np;
        delta[((size_t)bimg * A + h) * N + n] = v;
    }
}

// One [64 tokens][64 d] bf16 tile: rows of 128 B, 16-byte chunk c of row r stored at c ^ ((r >> 1) & 7)
// (conflict-free ds_read_b128 row reads; the transposed reads below go through the same map).
__device__ __forceinline__ int tile_off(int row, int chunk) { return row * HD + ((chunk ^ ((row >> 1) & 7)) << 3); }

// A operand = transpose of a row-major tile: rows = 16 tokens t0 .. t0+15 of k-step s (lane half h takes
// tokens t0 + 4h + {0..3} and + 8), columns d = 32 dt + (lane & 31).
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* tile, int t0, int dt, int lane) {
    const int g = lane & 15, grp = lane >> 4, tq = g >> 2, tp = g & 3;
    const int row = t0 + 4 * (grp >> 1) + tq;
    const int chunk = 4 * dt + 2 * (grp & 1) + (tp >> 1), half = (tp & 1) * 4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(tile + tile_off(row, chunk) + half));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(tile + tile_off(row + 8, chunk) + half));
    const bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return f;
}

// ---------------------------------------------------------------------------------- dQ
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_bf16_kernel(const bf16_t* __restrict__ qkv,
                                                                  const bf16_t* __restrict__ dctx,
                                                                  const float* __restrict__ lse,
                                                                  const float* __restrict__ delta,
                                                                  bf16_t* __restrict__ dqkv, int B, int Np, int A,
                                                                  DropArgs dr) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[2][2][TT * HD];  // [buffer][K|V]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const AttnTile at = attn_tile((Np + 1 + TB - 1) / TB, A);
    const int head = at.head, b = at.b;
    const int D = A * HD, ld = 3 * D, N = Np + 1;
    const bf16_t* kbase = qkv + D + head * HD;
    const bf16_t* vbase = qkv + 2 * D + head * HD;
    const float c = 0.125f * LOG2E;

    const int nq = at.rt * TB + wave * 32 + li;
    const bool q_valid = nq < N;
    const size_t q_row = tok_row(b, q_valid ? nq : 0, B, Np);
    f32x4 qf[4], dof[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qf[s] = *(const f32x4*)(qkv + q_row * ld + head * HD + 16 * s + 8 * lh);
        dof[s] = *(const f32x4*)(dctx + q_row * (size_t)D + head * HD + 16 * s + 8 * lh);
    }
    const size_t stat = ((size_t)b * A + head) * N + (q_valid ? nq : 0);
    const float lse_q = lse[stat], delta_q = delta[stat];
    const unsigned dkey = drop_key(dr.seed, dr.stream, (unsigned)((b * A + head) * N + nq));  // same mask as forward

    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

    const int lc = tid & 7, lr = tid >> 3;
    f32x4 rk[2], rv[2];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = min(kt * TT + lr + 32 * i, N - 1);
            const size_t off = tok_row(b, key, B, Np) * ld + 8 * lc;
            rk[i] = *(const f32x4*)(kbase + off);
            rv[i] = *(const f32x4*)(vbase + off);
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = lr + 32 * i;
            *(f32x4*)&lds[buf][0][tile_off(key, lc)] = rk[i];
            *(f32x4*)&lds[buf][1][tile_off(key, lc)] = rv[i];
        }
    };
    const int nkt = (N + TT - 1) / TT;
    gload(0);
    swrite(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        gload(min(kt + 1, nkt - 1));
        __builtin_amdgcn_sched_barrier(0);
        const bf16_t* Ks = lds[buf][0];
        const bf16_t* Vs = lds[buf][1];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = dp[r] = 0.f;
            const int key = kb * 32 + li;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x4 kf = *(const f32x4*)&Ks[tile_off(key, 2 * s + lh)];
                const f32x4 vf = *(const f32x4*)&Vs[tile_off(key, 2 * s + lh)];
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf),
                                                             __builtin_bit_cast(bf16x8, qf[s]), st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf),
                                                             __builtin_bit_cast(bf16x8, dof[s]), dp, 0, 0, 0);
            }
            unsigned pk[8];  // dS^T fragments: words 4 s + w = registers 8 s + 2 w, + 1
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float d0 = 0.f, d1 = 0.f;
                const int k0 = kt * TT + kb * 32 + kappa(r, lh), k1 = kt * TT + kb * 32 + kappa(r + 1, lh);
                float g0 = dp[r], g1 = dp[r + 1];
                if (dr.thresh) {
                    g0 = drop_keep(dkey, (unsigned)k0, dr.thresh) ? g0 * dr.scale : 0.f;
                    g1 = drop_keep(dkey, (unsigned)k1, dr.thresh) ? g1 * dr.scale : 0.f;
                }
                if (k0 < N) d0 = __builtin_amdgcn_exp2f(fmaf(st[r], c, -lse_q)) * (g0 - delta_q);
                if (k1 < N) d1 = __builtin_amdgcn_exp2f(fmaf(st[r + 1], c, -lse_q)) * (g1 - delta_q);
                pk[r >> 1] = pack2_bf16(d0, d1);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 dsf = __builtin_bit_cast(bf16x8, (uint4){pk[4 * s], pk[4 * s + 1], pk[4 * s + 2], pk[4 * s + 3]});
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Ks, kb * 32 + 16 * s, dt, lane), dsf, dq[dt],
                                                                     0, 0, 0);
            }
        }
        swrite(buf ^ 1);
        __syncthreads();
    }
    if (q_valid) {
        bf16_t* out = dqkv + q_row * ld + head * HD;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                uint2 t;
                t.x = pack2_bf16(dq[dt][4 * g4] * 0.125f, dq[dt][4 * g4 + 1] * 0.125f);
                t.y = pack2_bf16(dq[dt][4 * g4 + 2] * 0.125f, dq[dt][4 * g4 + 3] * 0.125f);
                *(uint2*)(out + dt * 32 + 8 * g4 + 4 * lh) = t;
            }
    }
}

// ---------------------------------------------------------------------------------- dK, dV
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_bf16_kernel(const bf16_t* __restrict__ qkv,
                                                                   const bf16_t* __restrict__ dctx,
                                                                   const float* __restrict__ lse,
                                                                   const float* __restrict__ delta,
                                                                   bf16_t* __restrict__ dqkv, int B, int Np, int A,
                                                                   DropArgs dr) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[2][2][TT * HD];  // [buffer][Q|dO]
    __shared__ float stats[2][3][TT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const AttnTile at = attn_tile((Np + 1 + TB - 1) / TB, A);
    const int head = at.head, b = at.b;
    const int D = A * HD, ld = 3 * D, N = Np + 1;
    const float c = 0.125f * LOG2E;

    const int nk = at.rt * TB + wave * 32 + li;
    const bool k_valid = nk < N;
    const size_t k_row = tok_row(b, k_valid ? nk : 0, B, Np);
    f32x4 kf[4], vf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        kf[s] = *(const f32x4*)(qkv + k_row * ld + D + head * HD + 16 * s + 8 * lh);
        vf[s] = *(const f32x4*)(qkv + k_row * ld + 2 * D + head * HD + 16 * s + 8 * lh);
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[dt][r] = dv[dt][r] = 0.f;

    const int lc = tid & 7, lr = tid >> 3;
    f32x4 rq[2], rd[2];
    float rs = 0.f, rdl = 0.f;
    unsigned rkey = 0;
    auto gload = [&](int qt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = min(qt * TT + lr + 32 * i, N - 1);
            const size_t row = tok_row(b, q, B, Np);
            rq[i] = *(const f32x4*)(qkv + row * ld + head * HD + 8 * lc);
            rd[i] = *(const f32x4*)(dctx + row * (size_t)D + head * HD + 8 * lc);
        }
        if (tid < TT) {
            const int q = min(qt * TT + tid, N - 1);
            rs = lse[((size_t)b * A + head) * N + q];
            rdl = delta[((size_t)b * A + head) * N + q];
            // the query's dropout key, hashed ONCE per query here instead of once per (query, key) element below
            rkey = drop_key(dr.seed, dr.stream, (unsigned)((b * A + head) * N + qt * TT + tid));
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = lr + 32 * i;
            *(f32x4*)&lds[buf][0][tile_off(q, lc)] = rq[i];
            *(f32x4*)&lds[buf][1][tile_off(q, lc)] = rd[i];
        }
        if (tid < TT) {
            stats[buf][0][tid] = rs;
            stats[buf][1][tid] = rdl;
            stats[buf][2][tid] = __uint_as_float(rkey);
        }
    };
    const int nqt = (N + TT - 1) / TT;
    gload(0);
    swrite(0);
    __syncthreads();
    for (int qt = 0; qt < nqt; ++qt) {
        const int buf = qt & 1;
        gload(min(qt + 1, nqt - 1));
        __builtin_amdgcn_sched_barrier(0);
        const bf16_t* Qs = lds[buf][0];
        const bf16_t* Os = lds[buf][1];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = dp[r] = 0.f;
            const int q = qb * 32 + li;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x4 qa = *(const f32x4*)&Qs[tile_off(q, 2 * s + lh)];
                const f32x4 oa = *(const f32x4*)&Os[tile_off(q, 2 * s + lh)];
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa),
                                                             __builtin_bit_cast(bf16x8, kf[s]), st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, oa),
                                                             __builtin_bit_cast(bf16x8, vf[s]), dp, 0, 0, 0);
            }
            unsigned pp[8], pd[8];  // P and dS fragments (B operands), query = register index
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float p0 = 0.f, p1 = 0.f;
                const int q0 = qb * 32 + kappa(r, lh), q1 = qb * 32 + kappa(r + 1, lh);
                if (qt * TT + q0 < N) p0 = __builtin_amdgcn_exp2f(fmaf(st[r], c, -stats[buf][0][q0]));
                if (qt * TT + q1 < N) p1 = __builtin_amdgcn_exp2f(fmaf(st[r + 1], c, -stats[buf][0][q1]));
                float k0 = 1.f, k1 = 1.f;
                if (dr.thresh) {
                    k0 = drop_keep(__float_as_uint(stats[buf][2][q0]), (unsigned)nk, dr.thresh) ? dr.scale : 0.f;
                    k1 = drop_keep(__float_as_uint(stats[buf][2][q1]), (unsigned)nk, dr.thresh) ? dr.scale : 0.f;
                }
                pp[r >> 1] = pack2_bf16(p0 * k0, p1 * k1);  // dropped P (what multiplied V in the forward)
                pd[r >> 1] = pack2_bf16(p0 * (dp[r] * k0 - stats[buf][1][q0]), p1 * (dp[r + 1] * k1 - stats[buf][1][q1]));
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pf = __builtin_bit_cast(bf16x8, (uint4){pp[4 * s], pp[4 * s + 1], pp[4 * s + 2], pp[4 * s + 3]});
                const bf16x8 df = __builtin_bit_cast(bf16x8, (uint4){pd[4 * s], pd[4 * s + 1], pd[4 * s + 2], pd[4 * s + 3]});
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Os, qb * 32 + 16 * s, dt, lane), pf, dv[dt],
                                                                     0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Qs, qb * 32 + 16 * s, dt, lane), df, dk[dt],
                                                                     0, 0, 0);
                }
            }
        }
        swrite(buf ^ 1);
        __syncthreads();
    }
    if (k_valid) {
        bf16_t* outk = dqkv + k_row * ld + D + head * HD;
        bf16_t* outv = dqkv + k_row * ld + 2 * D + head * HD;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                uint2 tk, tv;
                tk.x = pack2_bf16(dk[dt][4 * g4] * 0.125f, dk[dt][4 * g4 + 1] * 0.125f);
                tk.y = pack2_bf16(dk[dt][4 * g4 + 2] * 0.125f, dk[dt][4 * g4 + 3] * 0.125f);
                tv.x = pack2_bf16(dv[dt][4 * g4], dv[dt][4 * g4 + 1]);
                tv.y = pack2_bf16(dv[dt][4 * g4 + 2], dv[dt][4 * g4 + 3]);
                *(uint2*)(outk + dt * 32 + 8 * g4 + 4 * lh) = tk;
                *(uint2*)(outv + dt * 32 + 8 * g4 + 4 * lh) = tv;
            }
    }
}

}  // namespace

int launch_attention_bwd_bf16(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* dvec,
                              void* dqkv, int B, int Np, int A, DropArgs dr, hipStream_t s) {
    VITSEG_CHECK_ARG(qkv && ctx && dctx && lse && dvec && dqkv, VITSEG_EINVAL, "attention_bwd_bf16: null pointer");
    const int N = Np + 1;
    hipLaunchKernelGGL(attn_delta_bf16_kernel, dim3((unsigned)(((size_t)B * (Np + 1) * A * 8 + 255) / 256)), dim3(256), 0, s, (const bf16_t*)ctx,
                       (const bf16_t*)dctx, dvec, B, Np, A);
    VITSEG_LAUNCH_CHECK("attn_delta_bf16");
    const dim3 grid((unsigned)((N + TB - 1) / TB) * A * B);  // 1-D: attn_tile() places the tiles
    hipLaunchKernelGGL(attn_bwd_dq_bf16_kernel, grid, dim3(256), 0, s, (const bf16_t*)qkv, (const bf16_t*)dctx, lse, dvec,
                       (bf16_t*)dqkv, B, Np, A, dr);
    VITSEG_LAUNCH_CHECK("attn_bwd_dq_bf16");
    hipLaunchKernelGGL(attn_bwd_dkv_bf16_kernel, grid, dim3(256), 0, s, (const bf16_t*)qkv, (const bf16_t*)dctx, lse, dvec,
                       (bf16_t*)dqkv, B, Np, A, dr);
    VITSEG_LAUNCH_CHECK("attn_bwd_dkv_bf16");
    return VITSEG_OK;
}

}  // namespace vitseg
