// Backward of the attention core (fp32):  given dctx, recompute P from q, k and the saved log-sum-exp
// and produce dq, dk, dv (what autograd derives for eager_attention_forward,
// transformers/models/vit/modeling_vit.py:164-189).  Flash-style, nothing of size N x N is stored.
//
//   S = c q k^T (log2 units, c = hd^-1/2 log2 e),  P = exp2(S - lse),  dP = dO V^T,
//   dS = P o (dP - delta),  delta_i = sum_d dO_id O_id,
//   dV = P^T dO,   dK = hd^-1/2 dS^T Q,   dQ = hd^-1/2 dS K.
//
// Two kernels, no atomics (bitwise reproducible):
//   attn_bwd_dq : one block per 128 queries, loops over key tiles (mirror of the forward kernel:
//                 S^T and dP^T put a query on each lane; dS^T registers are the B operand of dQ^T = K^T dS^T);
//   attn_bwd_dkv: one block per 128 keys, loops over query tiles (S and dP put a KEY on each lane;
//                 P / dS registers are the B operands of dV^T = dO^T P and dK^T = Q^T dS).
// Tokens are indexed generically: n < Np is patch row b*Np + n, n = Np is the CLS row B*Np + b; tiles past
// N = Np + 1 are masked.  fp32-input MFMA 32x32x2 throughout (bound: 157.3 TFLOP/s).
#include "kernels.hpp"

namespace vitseg {
namespace {

constexpr int HD = 64, TB = 128, TT = 64;  // head dim, tokens per block (lane side), tokens per LDS tile
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int kappa(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }
__device__ __forceinline__ size_t tok_row(int b, int n, int B, int Np) {
    return n < Np ? (size_t)b * Np + n : (size_t)B * Np + b;
}

// delta[b][h][n] = sum_d dO[row][h*64+d] * O[row][h*64+d].  A thread owns 4 consecutive channels (16-byte loads), the 16
// lanes of a head combine with four lane swaps; threads walk a row, so a wave reads 1 KiB contiguous.
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ ctx, const float* __restrict__ dctx,
                                                         float* __restrict__ delta, int B, int Np, int A) {
    const int N = Np + 1, cpr = A * 16;                    // 16-byte chunks per row
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)(B * N) * cpr;
    float v = 0.f;
    size_t row = 0;
    int h = 0;
    if (i < total) {
        row = i / cpr;
        const int c = (int)(i - row * cpr);
        h = c >> 4;
        const f32x4 o = *(const f32x4*)(ctx + row * (size_t)(A * HD) + c * 4);
        const f32x4 d = *(const f32x4*)(dctx + row * (size_t)(A * HD) + c * 4);
        v = (o[0] * d[0] + o[1] * d[1]) + (o[2] * d[2] + o[3] * d[3]);
    }
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    if (i < total && (threadIdx.x & 15) == 0) {
        const size_t BNp = (size_t)B * Np;
        const int bimg = row < BNp ? (int)(row / Np) : (int)(row - BNp);
        const int n = row < BNp ? (int)(row - (size_t)bimg * Np) : Np;
        delta[((size_t)bimg * A + h) * N + n] = v;
    }
}

// Shared tile staging: two [64][64] fp32 tiles (X swizzled for 16-byte row reads: chunk ^ (row & 15))
struct Stage {
    f32x4 r0[4], r1[4];
};

// ---------------------------------------------------------------------------------- dQ
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ dctx,
                                                             const float* __restrict__ lse,
                                                             const float* __restrict__ delta, float* __restrict__ dqkv,
                                                             int B, int Np, int A, DropArgs dr) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][TT * HD];  // [buffer][K|V]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const AttnTile at = attn_tile((Np + 1 + TB - 1) / TB, A);
    const int head = at.head, b = at.b;
    const int D = A * HD, ld = 3 * D, N = Np + 1;
    const float* kbase = qkv + D + head * HD;
    const float* vbase = qkv + 2 * D + head * HD;
    const float c = 0.125f * LOG2E;

    const int nq = at.rt * TB + wave * 32 + li;
    const bool q_valid = nq < N;
    const size_t q_row = tok_row(b, q_valid ? nq : 0, B, Np);
    float qreg[32], doreg[32];  // element 4c+e = X[8c + 4 lh + e]
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
        const f32x4 t = *(const f32x4*)(qkv + q_row * ld + head * HD + 8 * cc + 4 * lh);
        const f32x4 u = *(const f32x4*)(dctx + q_row * (size_t)D + head * HD + 8 * cc + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            qreg[4 * cc + e] = t[e] * c;
            doreg[4 * cc + e] = u[e];
        }
    }
    const size_t stat = ((size_t)b * A + head) * N + (q_valid ? nq : 0);
    const float lse_q = lse[stat], delta_q = delta[stat];
    const unsigned dkey = drop_key(dr.seed, dr.stream, (unsigned)((b * A + head) * N + nq));  // same mask as forward

    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

    const int lc = tid & 15, lr = tid >> 4;
    f32x4 rk[4], rv[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = min(kt * TT + lr + 16 * i, N - 1);
            const size_t off = tok_row(b, key, B, Np) * ld + 4 * lc;
            rk[i] = *(const f32x4*)(kbase + off);
            rv[i] = *(const f32x4*)(vbase + off);
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = lr + 16 * i;
            const int pos = key * HD + ((lc ^ (key & 15)) << 2);
            *(f32x4*)&lds[buf][0][pos] = rk[i];
            *(f32x4*)&lds[buf][1][pos] = rv[i];
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
        const float* Ks = lds[buf][0];
        const float* Vs = lds[buf][1];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            // S^T and dP^T for 32 keys: [key][query]
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = dp[r] = 0.f;
            const int key = kb * 32 + li;
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                const int pos = key * HD + (((2 * cc + lh) ^ (key & 15)) << 2);
                const f32x4 kf = *(const f32x4*)&Ks[pos];
                const f32x4 vf = *(const f32x4*)&Vs[pos];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qreg[4 * cc + e], st, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[e], doreg[4 * cc + e], dp, 0, 0, 0);
                }
            }
            // dS^T = P o (dP - delta); masked keys contribute nothing
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool kvalid = kt * TT + kb * 32 + kappa(r, lh) < N;
                const float pv = kvalid ? __builtin_amdgcn_exp2f(st[r] - lse_q) : 0.f;
                float dpr = dp[r];
                if (dr.thresh)
                    dpr = drop_keep(dkey, (unsigned)(kt * TT + kb * 32 + kappa(r, lh)), dr.thresh) ? dpr * dr.scale : 0.f;
                st[r] = pv * (dpr - delta_q);
            }
            // dQ^T[d][query] += K^T[d][key] dS^T[key][query]
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int kk = kb * 32 + kappa(s, lh);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int d = dt * 32 + li;
                    const float kf = Ks[kk * HD + ((((d >> 2) ^ (kk & 15)) << 2) | (d & 3))];
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf, st[s], dq[dt], 0, 0, 0);
                }
            }
        }
        swrite(buf ^ 1);
        __syncthreads();
    }
    if (q_valid) {
        float* out = dqkv + q_row * ld + head * HD;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = dq[dt][4 * g4 + e] * 0.125f;
                *(f32x4*)(out + dt * 32 + 8 * g4 + 4 * lh) = t;
            }
    }
}

// ---------------------------------------------------------------------------------- dK, dV
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_kernel(const float* __restrict__ qkv,
                                                              const float* __restrict__ dctx,
                                                              const float* __restrict__ lse,
                                                              const float* __restrict__ delta, float* __restrict__ dqkv,
                                                              int B, int Np, int A, DropArgs dr) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][TT * HD];  // [buffer][Q|dO]
    __shared__ float stats[2][3][TT];                                  // [buffer][lse|delta]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const AttnTile at = attn_tile((Np + 1 + TB - 1) / TB, A);
    const int head = at.head, b = at.b;
    const int D = A * HD, ld = 3 * D, N = Np + 1;
    const float c = 0.125f * LOG2E;

    const int nk = at.rt * TB + wave * 32 + li;
    const bool k_valid = nk < N;
    const size_t k_row = tok_row(b, k_valid ? nk : 0, B, Np);
    float kreg[32], vreg[32];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
        const f32x4 t = *(const f32x4*)(qkv + k_row * ld + D + head * HD + 8 * cc + 4 * lh);
        const f32x4 u = *(const f32x4*)(qkv + k_row * ld + 2 * D + head * HD + 8 * cc + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            kreg[4 * cc + e] = t[e] * c;
            vreg[4 * cc + e] = u[e];
        }
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[dt][r] = dv[dt][r] = 0.f;

    const int lc = tid & 15, lr = tid >> 4;
    f32x4 rq[4], rd[4];
    float rs = 0.f, rdl = 0.f;
    unsigned rkey = 0;
    auto gload = [&](int qt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = min(qt * TT + lr + 16 * i, N - 1);
            const size_t row = tok_row(b, q, B, Np);
            rq[i] = *(const f32x4*)(qkv + row * ld + head * HD + 4 * lc);
            rd[i] = *(const f32x4*)(dctx + row * (size_t)D + head * HD + 4 * lc);
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
        for (int i = 0; i < 4; ++i) {
            const int q = lr + 16 * i;
            const int pos = q * HD + ((lc ^ (q & 15)) << 2);
            *(f32x4*)&lds[buf][0][pos] = rq[i];
            *(f32x4*)&lds[buf][1][pos] = rd[i];
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
        const float* Qs = lds[buf][0];
        const float* Os = lds[buf][1];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            // S[query][key] and dP[query][key] for 32 queries x this lane's key
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = dp[r] = 0.f;
            const int q = qb * 32 + li;
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                const int pos = q * HD + (((2 * cc + lh) ^ (q & 15)) << 2);
                const f32x4 qf = *(const f32x4*)&Qs[pos];
                const f32x4 of = *(const f32x4*)&Os[pos];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[e], kreg[4 * cc + e], st, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(of[e], vreg[4 * cc + e], dp, 0, 0, 0);
                }
            }
            // P and dS per (query = register, key = lane)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qq = qb * 32 + kappa(r, lh);
                const bool qvalid = qt * TT + qq < N;
                const float pv = qvalid ? __builtin_amdgcn_exp2f(st[r] - stats[buf][0][qq]) : 0.f;
                float keep = 1.f;
                if (dr.thresh)
                    keep = drop_keep(__float_as_uint(stats[buf][2][qq]), (unsigned)nk, dr.thresh) ? dr.scale : 0.f;
                st[r] = pv * keep;                                   // dropped P (what multiplied V in the forward)
                dp[r] = pv * (dp[r] * keep - stats[buf][1][qq]);     // dS
            }
            // dV^T[d][key] += dO^T[d][query] P[query][key];  dK^T[d][key] += Q^T[d][query] dS[query][key]
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int qq = qb * 32 + kappa(s, lh);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int d = dt * 32 + li;
                    const int pos = qq * HD + ((((d >> 2) ^ (qq & 15)) << 2) | (d & 3));
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Os[pos], st[s], dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[pos], dp[s], dk[dt], 0, 0, 0);
                }
            }
        }
        swrite(buf ^ 1);
        __syncthreads();
    }
    if (k_valid) {
        float* outk = dqkv + k_row * ld + D + head * HD;
        float* outv = dqkv + k_row * ld + 2 * D + head * HD;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 tk, tv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    tk[e] = dk[dt][4 * g4 + e] * 0.125f;
                    tv[e] = dv[dt][4 * g4 + e];
                }
                *(f32x4*)(outk + dt * 32 + 8 * g4 + 4 * lh) = tk;
                *(f32x4*)(outv + dt * 32 + 8 * g4 + 4 * lh) = tv;
            }
    }
}

}  // namespace

int launch_attention_bwd_f32(const float* qkv, const float* ctx, const float* dctx, const float* lse, float* dvec,
                             float* dqkv, int B, int Np, int A, DropArgs dr, hipStream_t s) {
    VITSEG_CHECK_ARG(qkv && ctx && dctx && lse && dvec && dqkv, VITSEG_EINVAL, "attention_bwd: null pointer");
    const int N = Np + 1;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)(((size_t)B * N * A * 16 + 255) / 256)), dim3(256), 0, s, ctx, dctx,
                       dvec, B, Np, A);
    VITSEG_LAUNCH_CHECK("attn_delta");
    const dim3 grid((unsigned)((N + TB - 1) / TB) * A * B);  // 1-D: attn_tile() places the tiles
    hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), 0, s, qkv, dctx, lse, dvec, dqkv, B, Np, A, dr);
    VITSEG_LAUNCH_CHECK("attn_bwd_dq");
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), 0, s, qkv, dctx, lse, dvec, dqkv, B, Np, A, dr);
    VITSEG_LAUNCH_CHECK("attn_bwd_dkv");
    return VITSEG_OK;
}

}  // namespace vitseg
