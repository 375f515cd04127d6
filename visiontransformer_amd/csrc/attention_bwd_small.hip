// Backward of the attention core for SHORT sequences in fp32 (small.hpp): dq, dk, dv from dctx, the saved q / k / v, the
// saved output and the saved log-sum-exp -- what autograd derives for eager_attention_forward
// (transformers/models/vit/modeling_vit.py:164-189) at the reference's own training size: N = 197 tokens, batch 4
// (model/CE/trainCurrentViTmodel.py:57).
//
//   S = c q k^T (log2 units, c = hd^-1/2 log2 e),  P = exp2(S - lse),  dP = dO V^T,  delta_i = sum_d dO_id O_id,
//   dS = P o (keep dP - delta),   dV = (keep P)^T dO,   dK = hd^-1/2 dS^T Q,   dQ = hd^-1/2 dS K.
//
// Why not attention_bwd_f32.hip: its blocks own 128 tokens and walk the other side in a serial loop -- at N = 197 and batch 4
// that is 96 blocks on 256 CUs with four dependent 64-token tiles each (43 + 54 us per layer, plus the delta launch:
// profiles/r05_train_b4_224_f32_kernel_stats_after.csv).  Here the work is cut the way attention_small.hip cuts the forward:
//   * ONE launch; the first B A ceil(N / 32) blocks produce dK / dV for 32 keys each, the second half dQ for 32 queries each;
//   * the four waves of a block hold the SAME 32 lane-side tokens and split the OTHER side in 32-token pieces (wave w takes
//     pieces w, w + 4, ...: 2 of the 7 at N = 197); the four partial sums are added through LDS in wave order -- a fixed order,
//     and the split is a function of N only (batch invariant, no atomics);
//   * operands in row form (the A operand of S / dP) come out of a wave-private swizzled LDS tile or straight from global
//     memory, the transposed forms (K^T, Q^T, dO^T) out of the same wave-private tile: no block barrier inside the loop;
//   * delta is formed from the dO and O rows the block reads anyway (no separate launch).
// Same dropout mask as the forward: key (seed, stream, (b A + head) N + query), element = key index.
// head_dim is 64 in every configuration of the reference.  Exact fp32 MFMAs (v_mfma_f32_32x32x2_f32); bound: latency.
#include "small.hpp"

namespace vitseg {
namespace {

constexpr int HD = 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int WAVE_FLOATS = 2 * 32 * HD + 128;   // per wave: two [32][64] tiles + 3 x 32 per-query values

__device__ __forceinline__ int kappa(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }

struct BwdSmall {
    const float* qkv;
    const float* ctx;
    const float* dctx;
    const float* lse;
    float* dqkv;
    int B, Np, A;
    DropArgs dr;
};

// ---------------------------------------------------------------------------------- dQ: 32 queries, waves split the keys
__device__ __forceinline__ void dq_block(const BwdSmall& p, int b, int head, int rt, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int Np = p.Np, N = Np + 1, D = p.A * HD, ld = 3 * D;
    auto token_row = [&](int t) -> size_t { return t < Np ? (size_t)b * Np + t : (size_t)p.B * Np + b; };
    const float* kbase = p.qkv + D + head * HD;
    const float* vbase = p.qkv + 2 * D + head * HD;
    const float c = 0.125f * LOG2E;

    const int nq = rt * 32 + li;
    const bool q_valid = nq < N;
    const size_t q_row = token_row(q_valid ? nq : 0);
    float qreg[32], doreg[32];   // element 4 cc + e = X[8 cc + 4 lh + e]
    float dl = 0.f;
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
        const f32x4 t = *(const f32x4*)(p.qkv + q_row * ld + head * HD + 8 * cc + 4 * lh);
        const f32x4 u = *(const f32x4*)(p.dctx + q_row * (size_t)D + head * HD + 8 * cc + 4 * lh);
        const f32x4 o = *(const f32x4*)(p.ctx + q_row * (size_t)D + head * HD + 8 * cc + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            qreg[4 * cc + e] = t[e] * c;
            doreg[4 * cc + e] = u[e];
        }
        dl += (o[0] * u[0] + o[1] * u[1]) + (o[2] * u[2] + o[3] * u[3]);
    }
    const float delta_q = dl + __shfl_xor(dl, 32, 64);
    const float lse_q = p.lse[((size_t)b * p.A + head) * N + (q_valid ? nq : 0)];
    const unsigned dkey = drop_key(p.dr.seed, p.dr.stream, (unsigned)((b * p.A + head) * N + nq));

    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

    float* kw = lds + wave * WAVE_FLOATS;   // this wave's K piece [key][d], rows 68 floats apart
    constexpr int KS = HD + 4;              // (a lane writes 16-byte pieces of ITS key's row: 16 lanes x 4 banks = all 64 banks)
    const int KH = (N + 31) / 32;
    // row form straight from global memory: lane (li, lh) = key li of the piece, pieces 8 cc + 4 lh of its K and V rows
    f32x4 kf[8], vf[8];
    auto load_piece = [&](int kh) {
        const size_t krow = token_row(min(kh * 32 + li, N - 1)) * ld + 4 * lh;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            kf[cc] = *(const f32x4*)(kbase + krow + 8 * cc);
            vf[cc] = *(const f32x4*)(vbase + krow + 8 * cc);
        }
    };
    if (wave < KH) load_piece(wave);
    for (int kh = wave; kh < KH; kh += 4) {
        // the K piece once more in LDS for the transposed read of the dQ product (in order behind the previous piece's reads)
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) *(f32x4*)(kw + li * KS + 8 * cc + 4 * lh) = kf[cc];
        // S^T and dP^T: [key][query]
        f32x16 st, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = dp[r] = 0.f;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[cc][e], qreg[4 * cc + e], st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[cc][e], doreg[4 * cc + e], dp, 0, 0, 0);
            }
        const int key0 = kh * 32;
        if (kh + 4 < KH) load_piece(kh + 4);   // the next piece's rows in flight under the dS arithmetic and the dQ products
        // dS^T = P o (keep dP - delta); keys beyond N contribute nothing
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + kappa(r, lh);
            const float pv = key < N ? __builtin_amdgcn_exp2f(st[r] - lse_q) : 0.f;
            float dpr = dp[r];
            if (p.dr.thresh) dpr = drop_keep(dkey, (unsigned)key, p.dr.thresh) ? dpr * p.dr.scale : 0.f;
            st[r] = pv * (dpr - delta_q);
        }
        // dQ^T[d][query] += K^T[d][key] dS^T[key][query]
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float k0 = kw[kappa(s, lh) * KS + li], k1 = kw[kappa(s, lh) * KS + 32 + li];
            dq[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(k0, st[s], dq[0], 0, 0, 0);
            dq[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(k1, st[s], dq[1], 0, 0, 0);
        }
    }
    // the four partial sums through LDS (each wave parks its 32 registers in its own region: in order behind its last reads),
    // wave w adds registers [8 w, 8 w + 8) in wave order and stores them
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) kw[(dt * 16 + r) * 64 + lane] = dq[dt][r];
    __syncthreads();
    {
        const int dt = wave >> 1;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int g4 = 2 * (wave & 1) + g;
            f32x4 t;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int at = (dt * 16 + 4 * g4 + e) * 64 + lane;
                t[e] = (((lds[at] + lds[WAVE_FLOATS + at]) + lds[2 * WAVE_FLOATS + at]) + lds[3 * WAVE_FLOATS + at]) * 0.125f;
            }
            if (q_valid) *(f32x4*)(p.dqkv + q_row * ld + head * HD + dt * 32 + 8 * g4 + 4 * lh) = t;
        }
    }
}

// ---------------------------------------------------------------------------------- dK, dV: 32 keys, waves split the queries
__device__ __forceinline__ void dkv_block(const BwdSmall& p, int b, int head, int kt, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int Np = p.Np, N = Np + 1, D = p.A * HD, ld = 3 * D;
    auto token_row = [&](int t) -> size_t { return t < Np ? (size_t)b * Np + t : (size_t)p.B * Np + b; };
    const float c = 0.125f * LOG2E;

    const int nk = kt * 32 + li;
    const bool k_valid = nk < N;
    const size_t k_row = token_row(k_valid ? nk : 0);
    float kreg[32], vreg[32];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
        const f32x4 t = *(const f32x4*)(p.qkv + k_row * ld + D + head * HD + 8 * cc + 4 * lh);
        const f32x4 u = *(const f32x4*)(p.qkv + k_row * ld + 2 * D + head * HD + 8 * cc + 4 * lh);
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

    float* Qs = lds + wave * WAVE_FLOATS;   // [query][d], 16-byte chunk ch of row q at ch ^ (q & 15)
    float* Os = Qs + 32 * HD;               // dO, same layout
    float* stats = Os + 32 * HD;            // [lse | delta | dropout key][32]
    const int QH = (N + 31) / 32;
    for (int qh = wave; qh < QH; qh += 4) {
        // whole 256-byte rows: instruction i moves queries 4 i .. 4 i + 3 (lane l = query 4 i + (l >> 4), chunk l & 15)
        const int ch = lane & 15;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i == 4) __builtin_amdgcn_sched_barrier(0);   // two batches of twelve 16-byte loads (register budget: 256)
            const int ql = 4 * i + (lane >> 4);
            const size_t row = token_row(min(qh * 32 + ql, N - 1));
            const f32x4 q4 = *(const f32x4*)(p.qkv + row * ld + head * HD + 4 * ch);
            const f32x4 d4 = *(const f32x4*)(p.dctx + row * (size_t)D + head * HD + 4 * ch);
            const f32x4 o4 = *(const f32x4*)(p.ctx + row * (size_t)D + head * HD + 4 * ch);
            const int pos = ql * HD + ((ch ^ (ql & 15)) << 2);
            *(f32x4*)(Qs + pos) = q4;
            *(f32x4*)(Os + pos) = d4;
            float v = (o4[0] * d4[0] + o4[1] * d4[1]) + (o4[2] * d4[2] + o4[3] * d4[3]);   // delta: the 16 lanes of a row combine
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 8, 64);
            if (ch == 0) stats[32 + ql] = v;
        }
        if (lane < 32) {
            stats[lane] = p.lse[((size_t)b * p.A + head) * N + min(qh * 32 + lane, N - 1)];
            stats[64 + lane] = __uint_as_float(drop_key(p.dr.seed, p.dr.stream, (unsigned)((b * p.A + head) * N + qh * 32 + lane)));
        }
        // (the LDS addresses below are loop invariant; computed from opaque copies of the lane coordinates so that they are
        // formed where they are used -- one v_xor each -- instead of being kept in ~50 registers across the loop: 256 is the budget)
        int lio = li, lho = lh;
        asm volatile("" : "+v"(lio), "+v"(lho));
        // S[query][key] and dP[query][key] for the piece's 32 queries x this lane's key
        f32x16 st, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = dp[r] = 0.f;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            const int pos = lio * HD + (((2 * cc + lho) ^ (lio & 15)) << 2);
            const f32x4 qf = *(const f32x4*)(Qs + pos);
            const f32x4 of = *(const f32x4*)(Os + pos);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                st = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[e], kreg[4 * cc + e], st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(of[e], vreg[4 * cc + e], dp, 0, 0, 0);
            }
        }
        // P and dS per (query = register, key = lane)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = kappa(r, lh);
            const float pv = qh * 32 + qq < N ? __builtin_amdgcn_exp2f(st[r] - stats[qq]) : 0.f;
            float keep = 1.f;
            if (p.dr.thresh) keep = drop_keep(__float_as_uint(stats[64 + qq]), (unsigned)nk, p.dr.thresh) ? p.dr.scale : 0.f;
            st[r] = pv * keep;                               // dropped P (what multiplied V in the forward)
            dp[r] = pv * (dp[r] * keep - stats[32 + qq]);    // dS
        }
        // dV^T[d][key] += dO^T[d][query] P[query][key];  dK^T[d][key] += Q^T[d][query] dS[query][key]
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int qq = kappa(s, lho);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int d = dt * 32 + lio;
                const int pos = qq * HD + ((((d >> 2) ^ (qq & 15)) << 2) | (d & 3));
                dv[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Os[pos], st[s], dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[pos], dp[s], dk[dt], 0, 0, 0);
            }
        }
    }
    // partial sums through LDS: 64 registers per lane and wave (dk | dv) in the wave's own tiles; wave w adds registers
    // [16 w, 16 w + 16) in wave order: waves 0, 1 store dk (d < 32, d >= 32), waves 2, 3 dv
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            Qs[(dt * 16 + r) * 64 + lane] = dk[dt][r];
            Qs[(32 + dt * 16 + r) * 64 + lane] = dv[dt][r];
        }
    __syncthreads();
    {
        const int dt = wave & 1;
        const float scale = wave < 2 ? 0.125f : 1.f;
        float* out = p.dqkv + k_row * ld + (wave < 2 ? D : 2 * D) + head * HD + dt * 32 + 4 * lh;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            f32x4 t;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int at = (16 * wave + 4 * g4 + e) * 64 + lane;
                t[e] = (((lds[at] + lds[WAVE_FLOATS + at]) + lds[2 * WAVE_FLOATS + at]) + lds[3 * WAVE_FLOATS + at]) * scale;
            }
            if (k_valid) *(f32x4*)(out + 8 * g4) = t;
        }
    }
}

__global__ __launch_bounds__(256, 2) void attn_bwd_small_kernel(const BwdSmall p) {
    __shared__ __attribute__((aligned(16))) float lds[4 * WAVE_FLOATS];
    const int T = (p.Np + 1 + 31) / 32;           // 32-token pieces of the sequence
    const int half = p.B * p.A * T;
    const int t = (int)blockIdx.x < half ? (int)blockIdx.x : (int)blockIdx.x - half;
    const int bh = t / T, rt = t - bh * T;
    const int b = bh / p.A, head = bh - b * p.A;
    // the longer dk / dv blocks first: the dq blocks fill the slots (two blocks per CU) as they come free
    if ((int)blockIdx.x < half) dkv_block(p, b, head, rt, lds);
    else dq_block(p, b, head, rt, lds);
}

}  // namespace

int launch_attention_bwd_small(const float* qkv, const float* ctx, const float* dctx, const float* lse, float* dqkv, int B,
                               int Np, int A, DropArgs dr, hipStream_t s) {
    VITSEG_CHECK_ARG(qkv && ctx && dctx && lse && dqkv && B > 0 && Np > 0 && A > 0, VITSEG_EINVAL, "attention_bwd_small: bad arguments");
    BwdSmall p{qkv, ctx, dctx, lse, dqkv, B, Np, A, dr};
    const int T = (Np + 1 + 31) / 32;
    hipLaunchKernelGGL(attn_bwd_small_kernel, dim3((unsigned)(2 * B * A * T)), dim3(256), 0, s, p);
    VITSEG_LAUNCH_CHECK("attention_bwd_small");
    return VITSEG_OK;
}

}  // namespace vitseg
