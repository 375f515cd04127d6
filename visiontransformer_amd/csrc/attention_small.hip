// Multi-head self-attention core for SHORT sequences in fp32 (small.hpp):  ctx = softmax(q k^T * hd^-0.5) v
// (transformers/models/vit/modeling_vit.py:164-189 behind ViTAttention :207-238) at the reference's own sizes:
// N = 197 tokens (224x224, P = 16), batch 1-8.
//
// Why not attention_f32.hip: that kernel gives a block 128 queries and walks the keys in a serial loop -- at N = 197
// and batch 1 that is 24 blocks on 256 CUs, each with four dependent key tiles (27 us per layer,
// profiles/r05_before_ref_grid_b16_batch1_kernel_stats.csv, for 0.12 GFLOP).  Here the work is cut the other way:
//   * one block per 32 queries of one (image, head): B A ceil(N / 32) blocks (84 at batch 1, 336 at batch 4);
//   * the four waves hold the SAME 32 queries and split the KEYS (wave w takes key tiles w, w + 4, ...: 2 of the 7
//     tiles at N = 197), each with its own online softmax; the four partial (max, sum, output) states are merged through
//     LDS in wave order -- a fixed order, and the key split does not depend on the batch size;
//   * the CLS token is token Np of its image (query and key), addressed at row B Np + b of the patches-first layout;
//     keys beyond N are masked in the last tile;
//   * scores transposed as in attention_f32.hip (S^T = K Q^T: a lane owns one query), K fragments straight from global
//     memory into the MFMA A operand (a lane reads 16-byte pieces of its key's row), V through a wave-private 8 KiB LDS
//     tile for the transposed read; exact fp32 MFMAs (v_mfma_f32_32x32x2_f32).
// head_dim is 64 in every configuration of the reference.  Bound: latency (0.5 GFLOP per layer at batch 4).
#include <type_traits>

#include "small.hpp"

namespace vitseg {
namespace {

constexpr int HD = 64;
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int kappa(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }

// Training (lse != null and / or dr.thresh != 0): the log2-domain log-sum-exp of every query is saved for the backward and the
// probabilities are dropped AFTER the normalising sum is taken (torch: dropout(softmax(s))), with the mask of
// attention_f32.hip -- key (seed, stream, (b A + head) N + query), element = key index.
// MM: 0 = exact fp32 products (v_mfma_f32_32x32x2_f32); 1 / 2 = the 16-bit form of the route: q, k, the probabilities and v are
// rounded to bf16 / fp16 in registers and multiplied on v_mfma_f32_32x32x16_* (4 + 4 products per key tile instead of 32 + 32;
// fp32 accumulate, fp32 softmax).  The operand slots of the wide MFMA are filled from the SAME register / LDS layout as the
// fp32 form -- slot e of half lh is head-dim element 8 (2 j + (e >> 2)) + 4 lh + (e & 3) on both sides of q k^T, and key
// kappa(8 j + e, lh) on both sides of P V -- so nothing is re-laid out.
template <int MM>
__global__ __launch_bounds__(256, 2) void attn_small_kernel(const float* __restrict__ qkv, float* __restrict__ ctx,
                                                            float* __restrict__ lse, int B, int Np, int A, DropArgs dr,
                                                            int ctx_fmt) {
    __shared__ __attribute__((aligned(16))) float vt[4][32 * HD];        // per wave: V tile [key][d]
    __shared__ __attribute__((aligned(16))) float om[4][32 * HD];        // per wave: partial output [query][d]
    __shared__ float ml[4][2][32];                                       // per wave: running max / sum per query

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int N = Np + 1, QT = (N + 31) / 32;
    const AttnTile at = attn_tile(QT, A);
    const int head = at.head, b = at.b;
    const int D = A * HD, ld = 3 * D;
    auto token_row = [&](int t) -> size_t { return t < Np ? (size_t)b * Np + t : (size_t)B * Np + b; };   // t clamped by the caller
    const float* qbase = qkv + head * HD;
    const float* kbase = qkv + D + head * HD;
    const float* vbase = qkv + 2 * D + head * HD;

    // this lane's query, pre-scaled by hd^-0.5 log2(e): element 4 c + e = Q[8 c + 4 lh + e]
    const int q_tok = at.rt * 32 + li;
    const float* qrow = qbase + token_row(min(q_tok, N - 1)) * ld;
    float qreg[32];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f32x4 t = *(const f32x4*)(qrow + 8 * c + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; ++e) qreg[4 * c + e] = t[e] * (0.125f * LOG2E);
    }

    typedef typename std::conditional<MM == 2, f16_t, bf16_t>::type HT;
    auto pack8 = [](float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
        return __builtin_bit_cast(bf16x8, uint4{H16<HT>::pack2(a0, a1), H16<HT>::pack2(a2, a3), H16<HT>::pack2(a4, a5), H16<HT>::pack2(a6, a7)});
    };
    bf16x8 qh[4];
    if constexpr (MM != 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            qh[j] = pack8(qreg[8 * j], qreg[8 * j + 1], qreg[8 * j + 2], qreg[8 * j + 3], qreg[8 * j + 4], qreg[8 * j + 5], qreg[8 * j + 6],
                          qreg[8 * j + 7]);
    }
    const unsigned dkey = drop_key(dr.seed, dr.stream, (unsigned)((b * A + head) * N + q_tok));
    const int KTn = (N + 31) / 32;
    f32x4 kreg[8], vreg[8];
    auto load_tile = [&](int kt) {
        // K: lane (li, lh) = key li of the tile, pieces 8 c + 4 lh; V: instruction c moves keys 4 c .. 4 c + 3 as whole
        // 256-byte rows (lane l = key 4 c + (l >> 4), 16-byte piece l & 15): coalesced reads, conflict-free LDS writes
        const float* kr = kbase + token_row(min(kt * 32 + li, N - 1)) * ld + 4 * lh;
#pragma unroll
        for (int c = 0; c < 8; ++c) kreg[c] = *(const f32x4*)(kr + 8 * c);
#pragma unroll
        for (int c = 0; c < 8; ++c)
            vreg[c] = *(const f32x4*)(vbase + token_row(min(kt * 32 + 4 * c + (lane >> 4), N - 1)) * ld + 4 * (lane & 15));
    };

    float m_run = -INFINITY, l_run = 0.f;   // this lane's half of the keys; halves are merged at the end
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;

    float* vw = vt[wave];
    if (wave < KTn) load_tile(wave);
    for (int kt = wave; kt < KTn; kt += 4) {
        // V tile to LDS [key][d] (LDS operations of one wave execute in order: the previous tile's reads are ahead of these)
#pragma unroll
        for (int c = 0; c < 8; ++c) *(f32x4*)(vw + (4 * c + (lane >> 4)) * HD + 4 * (lane & 15)) = vreg[c];
        // S^T = K Q^T (log2 units)
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
        if constexpr (MM != 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                sacc = H16<HT>::mfma(pack8(kreg[2 * j][0], kreg[2 * j][1], kreg[2 * j][2], kreg[2 * j][3], kreg[2 * j + 1][0], kreg[2 * j + 1][1],
                                           kreg[2 * j + 1][2], kreg[2 * j + 1][3]), qh[j], sacc);
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[c][e], qreg[4 * c + e], sacc, 0, 0, 0);
        }
        if (kt + 4 < KTn) load_tile(kt + 4);   // next tile's K / V in flight under the softmax and the PV products
        if (kt * 32 + 32 > N) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kt * 32 + kappa(r, lh) >= N) sacc[r] = -INFINITY;
        }
        // the two halves of a query (lanes li, li + 32: different keys) feed ONE accumulator in the PV product, so they
        // share the running maximum; key 32 kt of every tile is valid, so the maximum is finite from the first tile on
        float mx = sacc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // 0 on the first tile (m_run = -inf)
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sacc[r] = __builtin_amdgcn_exp2f(sacc[r] - m_new);
            rs += sacc[r];
        }
        l_run = l_run * alpha + rs;
        m_run = m_new;
        if (dr.thresh) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sacc[r] = drop_keep(dkey, (unsigned)(kt * 32 + kappa(r, lh)), dr.thresh) ? sacc[r] * dr.scale : 0.f;
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        // O^T += V^T P^T: A operand = V[key kappa(r, lh)][d = 32 dt + li] from LDS, B operand = the P registers
        if constexpr (MM != 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x8 pb = pack8(sacc[8 * j], sacc[8 * j + 1], sacc[8 * j + 2], sacc[8 * j + 3], sacc[8 * j + 4], sacc[8 * j + 5],
                                        sacc[8 * j + 6], sacc[8 * j + 7]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    float t[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] = vw[kappa(8 * j + e, lh) * HD + 32 * dt + li];
                    o[dt] = H16<HT>::mfma(pack8(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]), pb, o[dt]);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v0 = vw[kappa(r, lh) * HD + li], v1 = vw[kappa(r, lh) * HD + 32 + li];
                o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, sacc[r], o[0], 0, 0, 0);
                o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, sacc[r], o[1], 0, 0, 0);
            }
        }
    }

    // ---- merge of the four waves' states (max m, sum l, output o over each wave's keys) through LDS, in wave order ----
    // o[dt][r] = O^T[d = 32 dt + kappa(r, lh)][q = li]: 4 consecutive d per register quad -> 16-byte piece 8 dt + 2 g4 + lh of
    // row q, parked at piece position ^ (q & 15) (conflict-free for the writers and for the row-wise readers below)
    {
        float* ow = om[wave];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *(f32x4*)(ow + li * HD + (((8 * dt + 2 * g4 + lh) ^ (li & 15)) << 2)) =
                    f32x4{o[dt][4 * g4], o[dt][4 * g4 + 1], o[dt][4 * g4 + 2], o[dt][4 * g4 + 3]};
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);   // half 0 + half 1 of the wave's keys
        if (lh == 0) {
            ml[wave][0][li] = m_run;    // -inf for a wave without keys
            ml[wave][1][li] = l_tot;
        }
    }
    __syncthreads();
    // thread -> query tid >> 3, d = 8 (tid & 7) .. + 7
    {
        const int q = tid >> 3, d0 = (tid & 7) * 8;
        const int tok = at.rt * 32 + q;
        float mw[4], M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            mw[w] = ml[w][0][q];
            M = fmaxf(M, mw[w]);
        }
        float L = 0.f;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = __builtin_amdgcn_exp2f(mw[w] - M);   // wave 0 always has keys: M is finite, an empty wave gives 0
            L += ml[w][1][q] * f;
            const int c0 = 2 * (tid & 7);
            const f32x4 a0 = *(const f32x4*)(om[w] + q * HD + ((c0 ^ (q & 15)) << 2));
            const f32x4 a1 = *(const f32x4*)(om[w] + q * HD + (((c0 + 1) ^ (q & 15)) << 2));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0[e] += a0[e] * f;
                acc1[e] += a1[e] * f;
            }
        }
        if (lse && tok < N && (tid & 7) == 0) lse[((size_t)b * A + head) * N + tok] = M + __builtin_amdgcn_logf(L);
        if (tok < N) {
            const float inv = 1.0f / L;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0[e] *= inv;
                acc1[e] *= inv;
            }
            if (ctx_fmt == 0) {
                float* dst = ctx + token_row(tok) * D + head * HD + d0;
                *(f32x4*)dst = acc0;
                *(f32x4*)(dst + 4) = acc1;
            } else {   // the 16-bit route: ctx is o_proj's operand
                uint4 h;
                if (ctx_fmt == 2) {
                    h.x = H16<f16_t>::pack2(acc0[0], acc0[1]); h.y = H16<f16_t>::pack2(acc0[2], acc0[3]);
                    h.z = H16<f16_t>::pack2(acc1[0], acc1[1]); h.w = H16<f16_t>::pack2(acc1[2], acc1[3]);
                } else {
                    h.x = pack2_bf16(acc0[0], acc0[1]); h.y = pack2_bf16(acc0[2], acc0[3]);
                    h.z = pack2_bf16(acc1[0], acc1[1]); h.w = pack2_bf16(acc1[2], acc1[3]);
                }
                *(uint4*)((unsigned short*)ctx + token_row(tok) * D + head * HD + d0) = h;
            }
        }
    }
}

}  // namespace

int launch_attention_small(const float* qkv, float* ctx, int B, int Np, int A, hipStream_t s, float* lse, DropArgs dr, int ctx_fmt) {
    VITSEG_CHECK_ARG(qkv && ctx && B > 0 && Np > 0 && A > 0, VITSEG_EINVAL, "attention_small: bad arguments");
    const int QT = (Np + 1 + 31) / 32;
    // (the 16-bit form multiplies in its format; the training forward -- lse / dropout -- is fp32)
    const int mm = ctx_fmt != 0 && !lse && !dr.thresh ? ctx_fmt : 0;
    const dim3 grid((unsigned)(B * A * QT));
    if (mm == 1) hipLaunchKernelGGL(attn_small_kernel<1>, grid, dim3(256), 0, s, qkv, ctx, lse, B, Np, A, dr, ctx_fmt);
    else if (mm == 2) hipLaunchKernelGGL(attn_small_kernel<2>, grid, dim3(256), 0, s, qkv, ctx, lse, B, Np, A, dr, ctx_fmt);
    else hipLaunchKernelGGL(attn_small_kernel<0>, grid, dim3(256), 0, s, qkv, ctx, lse, B, Np, A, dr, ctx_fmt);
    VITSEG_LAUNCH_CHECK("attention_small");
    return VITSEG_OK;
}

}  // namespace vitseg
