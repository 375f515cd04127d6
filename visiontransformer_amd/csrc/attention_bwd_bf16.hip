// Backward of the attention core, bf16 operands / fp32 accumulation (mixed-precision training): what autograd derives
// for softmax(q k^T hd^-1/2) (dropout) v, transformers/models/vit/modeling_vit.py:164-189.  Flash-style: P is
// recomputed from q, k and the forward's log-sum-exp; nothing of size N x N is stored.
//
//   S = c q k^T (log2 units, c = hd^-1/2 log2 e),  P = exp2(S - lse),  P~ = P o M (M = dropout keep / (1 - p)),
//   dP~ = dO V^T,  dS = P o (dP~ o M - delta),  delta_i = sum_d dO_id O_id,
//   dV = P~^T dO,   dK = hd^-1/2 dS^T Q,   dQ = hd^-1/2 dS K.
//
// Two MFMA kernels over the PATCH tokens (whole 64 / 128 token tiles at 512 x 512: no ragged tail, no per-element
// bounds tests) plus the CLS token as rank-1 vector work, no atomics (bitwise reproducible):
//   attn_bwd_dq : a block owns 128 patch queries (one per lane: S^T / dP^T tiles) and loops over the patch-key tiles;
//                 dS^T registers are directly the B operand of dQ^T += K^T dS^T; the CLS key is one vector update.
//   attn_bwd_dkv: a block owns 128 patch keys and loops over the patch-query tiles (a KEY on each lane; P / dS
//                 registers are the B operands of dV^T += dO^T P~ and dK^T += Q^T dS); the CLS query is one vector update.
//   the CLS token's own gradients -- dq of the CLS query (a sum over all keys), dk / dv of the CLS key (sums over all
//   queries) -- leave the two kernels above as per-block partial vectors (round 4: the dQ kernel has p, dS of every one of
//   its queries against the CLS key, the dK/dV kernel those of the CLS query against every one of its keys; a side
//   kernel used to stream q | k | v | dO of every token a third time, 403 MB per layer at B = 64) and attn_bwd_cls_finish
//   adds them up in block order with the CLS-CLS pair's term.  delta_i = sum_d dO_id O_id is formed by the dQ kernel in its
//   prologue (it holds dO_i anyway; the dK/dV kernel, launched behind it, reads the values it wrote) instead of a pass of
//   its own over ctx and dctx.
// The row constants enter as the INITIAL accumulators of the S and dP chains (-lse / c, -delta), so the inner element
// work is  p = exp2(c s'),  ds = p dp'  -- no subtraction, no running maximum -- and the loop bodies are branch-free
// (dropout and raggedness are template parameters): hipcc can then overlap one tile's MFMAs with the other's VALU.
// A operands that are transposes of row-major LDS tiles are gathered with ds_read_b64_tr_b16 (as V in attention_bf16.hip).
// Inputs q|k|v, ctx, dctx are bf16; lse / delta fp32; dq|dk|dv are written as bf16.  Roofline: MFMA bf16, 7 products
// of 2 N^2 hd per (image, head) (S and dP are formed in both kernels).
#include "kernels.hpp"

namespace vitseg {
namespace {

constexpr int HD = 64, TB = 128, TT = 64;
constexpr float LOG2E = 1.4426950408889634f;
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int kappa(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// One [64 tokens][64 d] bf16 tile serves BOTH kinds of read: rows of 128 B, 16-byte chunk c of row r stored at
// c ^ f(r), f = the 3 bits of r >> 1 rotated right by one.  f is a bijection of (r >> 1) & 7, so the 16 rows of a
// ds_read_b128 lane group still hit 16 distinct 16-byte slots; and rows r, r + 2 of a transposed read (4 rows x 64 B
// per half-wave) now differ in bit 2 of f, i.e. land in different 64-byte halves (with the plain (r >> 1) & 7 they
// shared one: 2-way conflicts on every ds_read_b64_tr_b16, 20 % of the LDS cycles in the counters).
__device__ __forceinline__ int tile_off(int row, int chunk) {
    const int x = (row >> 1) & 7;
    return row * HD + ((chunk ^ (((x & 1) << 2) | (x >> 1))) << 3);
}

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

// LDS-DMA staging of one [64 tokens][64 d] tile (the layout of tile_off): a wave instruction moves 8 token rows x 128 B
// straight into the tile (lane l lands at + 16 l: row l >> 3, chunk position l & 7; tile_off's XOR is applied to the per-lane
// SOURCE chunk).  One buffer descriptor per tensor, per-lane offsets are loop constants, the tile's row offset a scalar: no
// vector instruction, no staging registers, no ds_write per tile (the flat-load form spent ~30 VALU per tile on 64-bit
// addresses and 16 VGPRs on the hop, at two waves per SIMD).  Rows beyond the descriptor's range read as zeros.  Issued from
// inline asm; the caller orders it with one hand-placed vmcnt(0) before the tile's barrier.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_rsrc(const void* base, int bytes) {
    const unsigned long long b = (unsigned long long)base;
    i32x4 r;
    r[0] = (int)(unsigned)b;
    r[1] = (int)(unsigned)((b >> 32) & 0xffffu);   // stride 0
    r[2] = bytes;
    r[3] = 0x00020000;
    return r;
}
// byte offset of this lane's 16 bytes inside the source tile whose rows are `row_bytes` apart: piece pc of wave `wave`
__device__ __forceinline__ unsigned dma_voff(int wave, int pc, int lane, int row_bytes) {
    const int row = 16 * wave + 8 * pc + (lane >> 3);
    const int x = (row >> 1) & 7;
    return (unsigned)(row * row_bytes + ((((lane & 7)) ^ (((x & 1) << 2) | (x >> 1))) << 4));
}
__device__ __forceinline__ void dma_piece(unsigned lds_dst, unsigned voff, const i32x4& rsrc, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// partial dot product of two 8 x 4 bf16-pair fragments (this lane's 32 of the 64 channels), completed across the lane halves
__device__ __forceinline__ float dot_frag(const f32x4 (&a)[4], const f32x4 (&b)[4]) {
    float part = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned x = __float_as_uint(a[s][e]), y = __float_as_uint(b[s][e]);
            part = fmaf(bf_lo(x), bf_lo(y), part);
            part = fmaf(bf_hi(x), bf_hi(y), part);
        }
    return part + __shfl_xor(part, 32, 64);
}

// t summed over the 4 lanes of a quad, on every lane of it: two DPP quad permutes (lane ^ 1, lane ^ 2) folded into the adds
// -- __shfl_xor compiles to ds_bpermute_b32 here, an LDS round trip per value and step (194 of them with 150 waits in each
// kernel's tail before this)
__device__ __forceinline__ float quad_sum(float t) {
    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xb1, 0xf, 0xf, true));
    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4e, 0xf, 0xf, true));
    return t;
}

// Sum over the block's 128 tokens of per-token vectors.  Lane (li, lh) holds 32 channels {16 s + 8 lh + j} of ITS token's
// vector as scale * (the bf16 fragment registers f).  Quad sums by DPP (2 steps), the 32 quad sums of the block (8 per
// wave) as rows of `red` (32 x 64 floats per vector, in the tile buffers, which are free once the loop is over), 64
// threads per vector add the rows in a fixed order.  stage: every lane, after a block barrier that ends the tile loop;
// finish: after the barrier that follows the last stage.
__device__ __forceinline__ void cls_stage(const f32x4 (&f)[4], float scale, float* red, int tid) {
    const int lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    float* row = red + (wave * 8 + (li >> 2)) * 64 + 8 * lh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float t[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned u = __float_as_uint(f[s][e]);
            t[2 * e] = bf_lo(u) * scale;
            t[2 * e + 1] = bf_hi(u) * scale;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = quad_sum(t[j]);
        if ((li & 3) == 0) {   // channels 16 s + 8 lh + 0..7: two 16-byte stores
            *(f32x4*)(row + 16 * s) = f32x4{t[0], t[1], t[2], t[3]};
            *(f32x4*)(row + 16 * s + 4) = f32x4{t[4], t[5], t[6], t[7]};
        }
    }
}
__device__ __forceinline__ void cls_finish(const float* red, float* out, int nv, int tid) {
    if (tid < nv * 64) {
        const float* col = red + (tid >> 6) * 2048 + (tid & 63);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;   // four chains: the 32 reads are in flight together
#pragma unroll
        for (int r = 0; r < 32; r += 4) {
            a0 += col[r * 64];
            a1 += col[(r + 1) * 64];
            a2 += col[(r + 2) * 64];
            a3 += col[(r + 3) * 64];
        }
        out[tid] = (a0 + a1) + (a2 + a3);
    }
}

// The same sum for per-token vectors that live in MFMA ACCUMULATORS (dQ^T, dK^T, dV^T: lane (li, lh) holds channels
// 32 dt + 8 g4 + 4 lh + e of its token in register 4 g4 + e of acc[dt]): the block's share of a COLUMN SUM of dq | dk | dv,
// i.e. of the QKV bias gradient -- formed here from the fp32 accumulators the kernels hold anyway instead of by a pass
// over the 302 MB of dQKV they wrote (round 4).  Rows of `red` as in cls_stage; finish with cls_finish.
__device__ __forceinline__ void col_stage(const f32x16 (&acc)[2], float scale, float* red, int tid) {
    const int lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    float* row = red + (wave * 8 + (li >> 2)) * 64 + 4 * lh;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            float t[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                t[e] = quad_sum(acc[dt][4 * g4 + e] * scale);
            }
            if ((li & 3) == 0) *(f32x4*)(row + 32 * dt + 8 * g4) = f32x4{t[0], t[1], t[2], t[3]};
        }
}

// ---------------------------------------------------------------------------------- dQ (patch queries)
// MW (both MFMA kernels; with DROP, without RAGGED): keep bits from the precomputed words (common.hpp
// attn_dropmask_words) instead of one hash per element.
template <bool DROP, bool RAGGED, bool MW>
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_bf16_kernel(const bf16_t* __restrict__ qkv,
                                                                  const bf16_t* __restrict__ ctx,
                                                                  const bf16_t* __restrict__ dctx,
                                                                  const float* __restrict__ lse,
                                                                  float* __restrict__ delta, float* __restrict__ clsp,
                                                                  float* __restrict__ colp,
                                                                  bf16_t* __restrict__ dqkv, int B, int Np, int A,
                                                                  DropArgs dr, const unsigned* __restrict__ maskw) {
    static_assert(!MW || (DROP && !RAGGED), "mask words: dropout on, whole 128-token blocks");
    __shared__ __attribute__((aligned(16))) bf16_t lds[2][2][TT * HD];  // [buffer][K|V]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const AttnTile at = attn_tile((Np + TB - 1) / TB, A);
    const int head = at.head, b = at.b;
    const int D = A * HD, ld = 3 * D, N = Np + 1;
    const size_t row0 = (size_t)b * Np, cls_row = (size_t)B * Np + b;
    const bf16_t* kbase = qkv + D + head * HD;
    const bf16_t* vbase = qkv + 2 * D + head * HD;
    const float c = 0.125f * LOG2E;

    const int nq = at.rt * TB + wave * 32 + li;
    const bool q_valid = nq < Np;
    const size_t q_row = row0 + (q_valid ? nq : Np - 1);
    f32x4 qf[4], dof[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qf[s] = *(const f32x4*)(qkv + q_row * ld + head * HD + 16 * s + 8 * lh);
        dof[s] = *(const f32x4*)(dctx + q_row * (size_t)D + head * HD + 16 * s + 8 * lh);
    }
    const size_t stat = ((size_t)b * A + head) * N + (q_valid ? nq : Np - 1);
    // delta of this lane's query = dO . O over the 64 channels (the bf16 context the forward wrote); published for the dK/dV
    // kernel, which is launched behind this one
    float dl_q;
    {
        f32x4 of[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) of[s] = *(const f32x4*)(ctx + q_row * (size_t)D + head * HD + 16 * s + 8 * lh);
        dl_q = dot_frag(dof, of);
        if (q_valid && lh == 0) delta[stat] = dl_q;
        if (at.rt == 0 && wave == 0) {   // ... and the CLS query's (wave-uniform branch): one channel per lane
            const float x = bf16_to_f32(dctx[cls_row * (size_t)D + head * HD + lane]) * bf16_to_f32(ctx[cls_row * (size_t)D + head * HD + lane]);
            const float t = wave_sum(x);
            if (lane == 0) delta[((size_t)b * A + head) * N + Np] = t;
        }
    }
    const float lse_q = lse[stat], ndelta = -dl_q;
    const float nlse = -lse_q * (1.0f / c);     // initial accumulator of the S chain: p = exp2(c (q.k - lse / c))
    const unsigned dkey = drop_key(dr.seed, dr.stream, (unsigned)((b * A + head) * N + nq));  // same mask as forward

    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

    // K | V tiles by LDS-DMA (see dma_piece): one descriptor over this image's rows from its K segment on
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const i32x4 kv_rsrc = make_rsrc(kbase + row0 * ld, (Np - 1) * ld * 2 + (D + HD) * 2);
    unsigned vk[2], vv[2];
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
        vk[pc] = dma_voff(wv, pc, lane, ld * 2);
        vv[pc] = vk[pc] + (unsigned)(D * 2);
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)&lds[0][0][0];
    auto stage = [&](int kt, int buf) {
        const unsigned soff = (unsigned)(kt * TT * ld * 2);
        const unsigned dst = lds_base + (unsigned)(buf * 2 * TT * HD * 2 + wv * 2048);
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            dma_piece(dst + pc * 1024, vk[pc], kv_rsrc, soff);
            dma_piece(dst + TT * HD * 2 + pc * 1024, vv[pc], kv_rsrc, soff);
        }
    };
    const int nkt = (Np + TT - 1) / TT;
    const unsigned long* mrow = nullptr;  // this wave's lane masks: 16 per 32-key block, wave-uniform -> scalar loads
    if (MW) {
        const int nb = Np >> 5, qg = at.rt * 4 + __builtin_amdgcn_readfirstlane(wave);
        mrow = (const unsigned long*)maskw + ((size_t)((b * A + head) * nb + qg) * nb) * 16;
    }
    TileMasks lm;
    if (MW) lm.load(mrow);
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (MW) lm.wait();
        const bf16_t* Ks = lds[buf][0];
        const bf16_t* Vs = lds[buf][1];
        // one 32-key block at a time (S, dP -> dS -> dQ^T): 32 score registers live instead of 64, which is what lets three
        // waves share a SIMD -- the other waves' MFMAs fill this wave's exponentials (profiles/r03_simd_overlap_probe.txt)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = nlse;
                dp[r] = DROP ? 0.f : ndelta;
            }
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
            if (kb == 0) stage(min(kt + 1, nkt - 1), buf ^ 1);   // next tile's DMA behind these 8 MFMAs (see the dK/dV kernel)
            unsigned pk[8];  // dS^T fragments: words 4 s + w = registers 8 s + 2 w, + 1
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int k0 = kt * TT + kb * 32 + kappa(r, lh), k1 = k0 + 1;   // kappa(r + 1) = kappa(r) + 1 for even r
                const float p0 = __builtin_amdgcn_exp2f(st[r] * c), p1 = __builtin_amdgcn_exp2f(st[r + 1] * c);
                float d0, d1;
                if (DROP) {
                    float m0, m1;
                    if (MW) {
                        m0 = mask_select(lm.reg(kb, r), dr.scale);
                        m1 = mask_select(lm.reg(kb, r + 1), dr.scale);
                    } else {
                        m0 = drop_keep(dkey, (unsigned)k0, dr.thresh) ? dr.scale : 0.f;
                        m1 = drop_keep(dkey, (unsigned)k1, dr.thresh) ? dr.scale : 0.f;
                    }
                    d0 = p0 * fmaf(dp[r], m0, ndelta);
                    d1 = p1 * fmaf(dp[r + 1], m1, ndelta);
                } else {
                    d0 = p0 * dp[r];
                    d1 = p1 * dp[r + 1];
                }
                if (RAGGED) {
                    d0 = k0 < Np ? d0 : 0.f;
                    d1 = k1 < Np ? d1 : 0.f;
                }
                pk[r >> 1] = pack2_bf16(d0, d1);
            }
            // the 4 dQ MFMAs with their transposed K fragments (two ds_read_b64_tr_b16 each) read TWO MFMAs ahead: left alone,
            // hipcc issues a fragment's reads and waits for them right in front of its MFMA (dQ 422 -> 401 us per layer)
            __builtin_amdgcn_sched_barrier(0);
            {
                bf16x8 fr[4];   // i = 2 s + dt
                auto ldf = [&](int i) { fr[i] = tr_frag(Ks, kb * 32 + 16 * (i >> 1), i & 1, lane); };
                ldf(0);
                ldf(1);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i + 2 < 4) ldf(i + 2);
                    const int s = i >> 1, dt = i & 1;
                    const bf16x8 dsf = __builtin_bit_cast(bf16x8, (uint4){pk[4 * s], pk[4 * s + 1], pk[4 * s + 2], pk[4 * s + 3]});
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i], dsf, dq[dt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MW) lm.load(mrow + (size_t)min(kt + 1, nkt - 1) * 32);   // next tile's lane masks (see TileMasks)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile kt + 1 have landed
        __syncthreads();
    }
    if (MW) lm.wait();   // the last iteration's mask load is still writing its 64 SGPRs: nothing may reuse them before it lands
    // ---- the CLS key: one vector update per query (ds is a scalar per lane) ----
    float cls_ds, cls_pm;
    {
        f32x4 kc[4], vc[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kc[s] = *(const f32x4*)(kbase + cls_row * ld + 16 * s + 8 * lh);
            vc[s] = *(const f32x4*)(vbase + cls_row * ld + 16 * s + 8 * lh);
        }
        const float sc = dot_frag(qf, kc), dpc = dot_frag(dof, vc);
        const float p = __builtin_amdgcn_exp2f(fmaf(sc, c, -lse_q));
        float m = 1.f;
        if (DROP) m = drop_keep(dkey, (unsigned)Np, dr.thresh) ? dr.scale : 0.f;
        const float ds = p * fmaf(dpc, m, ndelta);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {   // accumulator register 4 g4 + e = channel 32 dt + 8 g4 + 4 lh + e
                const uint2 t = *(const uint2*)(kbase + cls_row * ld + dt * 32 + 8 * g4 + 4 * lh);
                dq[dt][4 * g4 + 0] = fmaf(ds, bf_lo(t.x), dq[dt][4 * g4 + 0]);
                dq[dt][4 * g4 + 1] = fmaf(ds, bf_hi(t.x), dq[dt][4 * g4 + 1]);
                dq[dt][4 * g4 + 2] = fmaf(ds, bf_lo(t.y), dq[dt][4 * g4 + 2]);
                dq[dt][4 * g4 + 3] = fmaf(ds, bf_hi(t.y), dq[dt][4 * g4 + 3]);
            }
        cls_ds = ds;
        cls_pm = p * m;
    }
    // the dQ third of the QKV bias gradient's partial record: from the accumulators while they are complete and still live
    // (the tile buffers are free: every wave has passed the loop's last barrier and nothing below reads them as tiles)
    if (colp) col_stage(dq, q_valid ? 0.125f : 0.f, (float*)&lds[0][0][0] + 4096, tid);   // block-uniform
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
    // (behind the stores: the 32 accumulator registers are free by now)
    {
    // ---- this block's share of the CLS KEY's gradients: dk_cls += ds q, dv_cls += p~ dO over its (valid) queries;
    // partial record of block rt = [dq_cls | dk_cls | dv_cls][64] ----
        float* rec = clsp + (((size_t)b * A + head) * ((Np + TB - 1) / TB) + at.rt) * 192;
        float* red = (float*)&lds[0][0][0];
        __syncthreads();   // every wave is done with the tiles
        cls_stage(qf, q_valid ? cls_ds : 0.f, red, tid);
        cls_stage(dof, q_valid ? cls_pm : 0.f, red + 2048, tid);
        __syncthreads();
        cls_finish(red, rec + 64, 2, tid);
        if (colp) cls_finish(red + 4096, colp + (rec - clsp), 1, tid);
    }
}

// ---------------------------------------------------------------------------------- dK, dV (patch keys)
template <bool DROP, bool RAGGED, bool MW>
__global__ __launch_bounds__(256, 3) void attn_bwd_dkv_bf16_kernel(const bf16_t* __restrict__ qkv,
                                                                   const bf16_t* __restrict__ dctx,
                                                                   const float* __restrict__ lse,
                                                                   const float* __restrict__ delta,
                                                                   float* __restrict__ clsp, float* __restrict__ colp,
                                                                   bf16_t* __restrict__ dqkv, int B, int Np, int A,
                                                                   DropArgs dr, const unsigned* __restrict__ maskw) {
    static_assert(!MW || (DROP && !RAGGED), "mask words: dropout on, whole 128-token blocks");
    __shared__ __attribute__((aligned(16))) bf16_t lds[2][2][TT * HD];  // [buffer][Q|dO]
    __shared__ __attribute__((aligned(16))) float stats[2][3][TT];      // -lse / c, -delta, dropout key of the tile's queries
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const AttnTile at = attn_tile((Np + TB - 1) / TB, A);
    const int head = at.head, b = at.b;
    const int D = A * HD, ld = 3 * D, N = Np + 1;
    const size_t row0 = (size_t)b * Np, cls_row = (size_t)B * Np + b;
    const size_t stat0 = ((size_t)b * A + head) * N;
    const float c = 0.125f * LOG2E, inv_c = 1.0f / c;

    const int nk = at.rt * TB + wave * 32 + li;
    const bool k_valid = nk < Np;
    const size_t k_row = row0 + (k_valid ? nk : Np - 1);
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

    float rs = 0.f, rdl = 0.f, raw_l = 0.f, raw_d = 0.f;
    int raw_q = 0;
    unsigned rkey = 0;
    // mask words of this lane's KEY (both lane halves the same key): one word per 32-query group, bit = query.
    // Word position inside the key's 32-key block: 2 r + h with key = kappa(r, h).
    unsigned nw[2] = {0u, 0u};
    // (one descriptor over this (image, head) pair's words, a 32-bit per-lane offset and a scalar tile offset: a 64-bit
    // per-lane pointer here costs two registers the kernel does not have at three waves per SIMD)
    const int mnb = Np >> 5;
    auto mw_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)maskw, 0, 0, 0x00020000);
    unsigned mw_voff = 0;
    if (MW) {
        const int pos = 2 * ((li & 3) + 4 * (li >> 3)) + ((li >> 2) & 1);
        mw_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(maskw + (size_t)((b * A + head) * mnb) * mnb * 32), 0,
                                                    mnb * mnb * 32 * 4, 0x00020000);
        mw_voff = (unsigned)(((nk >> 5) * 32 + pos) * 4);   // + qg * nb * 32 words
    }
    // Q and dO tiles by LDS-DMA (see dma_piece): Q rows are 3 D apart (the q | k | v rows), dO rows D apart
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const i32x4 q_rsrc = make_rsrc(qkv + row0 * ld + head * HD, (Np - 1) * ld * 2 + HD * 2);
    const i32x4 o_rsrc = make_rsrc(dctx + row0 * (size_t)D + head * HD, (Np - 1) * D * 2 + HD * 2);
    unsigned vq[2], vo[2];
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
        vq[pc] = dma_voff(wv, pc, lane, ld * 2);
        vo[pc] = dma_voff(wv, pc, lane, D * 2);
    }
    const auto lse_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(lse + stat0), 0, Np * 4, 0x00020000);
    const auto dl_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(delta + stat0), 0, Np * 4, 0x00020000);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)&lds[0][0][0];
    // the tile's row constants and (MW) this key's mask words travel through registers as before: gload fetches them,
    // swrite publishes the row constants next to the tile
    // gload_half(.., 0): the Q tile + the row constants / mask words; (.., 1): the dO tile.  In the loop the two halves are
    // issued right behind the S / dP MFMAs of the two query blocks -- the wave would otherwise only wait for those MFMAs
    // there, while at the top of the tile the ~300 cycles of DMA issue had nothing to hide behind.
    auto gload_half = [&](int qt, int buf, int half) {
        const unsigned dst = lds_base + (unsigned)(buf * 2 * TT * HD * 2 + wv * 2048);
        if (half == 1) {
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) dma_piece(dst + TT * HD * 2 + pc * 1024, vo[pc], o_rsrc, (unsigned)(qt * TT * D * 2));
            return;
        }
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) dma_piece(dst + pc * 1024, vq[pc], q_rsrc, (unsigned)(qt * TT * ld * 2));
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (MW) nw[i] = __builtin_amdgcn_raw_buffer_load_b32(mw_rsrc, mw_voff, (unsigned)((qt * 2 + i) * mnb * 32 * 4), 0);
        if (tid < TT) {
            // (only REQUESTED here: the arithmetic on them waits in swrite, at the tile's end -- done here, wave 0 sat out
            // the loads' latency at the start of every tile while waves 1-3 went ahead and then waited for it at the
            // barrier: 350-450 of a tile's ~5 400 cycles, tools/probes/attn_bwd_trace.sh)
            const unsigned so = (unsigned)(qt * TT * 4);
            raw_l = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lse_rsrc, (unsigned)(tid * 4), so, 0));
            raw_d = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dl_rsrc, (unsigned)(tid * 4), so, 0));
            raw_q = qt * TT + tid;
        }
    };
    auto swrite = [&](int buf) {
        if (tid < TT) {
            const bool ok = !RAGGED || raw_q < Np;
            // a query beyond the last patch gets p = exp2(c * -inf) = 0 and contributes nothing (its row constants read as
            // zeros: out of the descriptors' range)
            rs = ok ? -raw_l * inv_c : -INFINITY;
            rdl = ok ? -raw_d : 0.f;
            // the query's dropout key, hashed ONCE per query here instead of once per (query, key) element below
            if (DROP && !MW) rkey = drop_key(dr.seed, dr.stream, (unsigned)((b * A + head) * N + raw_q));
            stats[buf][0][tid] = rs;
            stats[buf][1][tid] = rdl;
            stats[buf][2][tid] = __uint_as_float(rkey);
        }
    };
    auto gload = [&](int qt, int buf) {
        gload_half(qt, buf, 0);
        gload_half(qt, buf, 1);
    };
    const int nqt = (Np + TT - 1) / TT;
    gload(0, 0);
    swrite(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int qt = 0; qt < nqt; ++qt) {
        const int buf = qt & 1;
        const unsigned cw[2] = {nw[0] >> (4 * lh), nw[1] >> (4 * lh)};   // bit 8 g4 + e = register 4 g4 + e of this lane half
        const bf16_t* Qs = lds[buf][0];
        const bf16_t* Os = lds[buf][1];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            // accumulator register 4 g4 + e <-> query qb * 32 + 8 g4 + 4 lh + e: four consecutive row constants per 16-byte read
            f32x16 st, dp;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int q0 = qb * 32 + 8 * g4 + 4 * lh;
                const f32x4 a = *(const f32x4*)&stats[buf][0][q0];
                f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
                if (!DROP) d4 = *(const f32x4*)&stats[buf][1][q0];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st[4 * g4 + e] = a[e];
                    dp[4 * g4 + e] = d4[e];      // DROP: the mask comes between dP and -delta, so dP starts at zero
                }
            }
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
            gload_half(min(qt + 1, nqt - 1), buf ^ 1, qb);   // next tile: half of its staging behind these 8 MFMAs
            unsigned pp[8], pd[8];  // P~ and dS fragments (B operands), query = register index
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                // DROP: -delta and the hashed dropout key of four queries are read from the tile's row constants HERE (one
                // 16-byte broadcast read each) instead of living in 16 (+16) registers across the MFMAs above
                f32x4 dl4 = {0.f, 0.f, 0.f, 0.f}, k4 = {0.f, 0.f, 0.f, 0.f};
                if (DROP) {
                    const int q0 = qb * 32 + 8 * g4 + 4 * lh;
                    dl4 = *(const f32x4*)&stats[buf][1][q0];
                    if (!MW) k4 = *(const f32x4*)&stats[buf][2][q0];
                }
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const int r = 4 * g4 + 2 * w;
                    const float p0 = __builtin_amdgcn_exp2f(st[r] * c), p1 = __builtin_amdgcn_exp2f(st[r + 1] * c);
                    if (DROP) {
                        float m0, m1;
                        if (MW) {   // 0 / -1 from the sign-extended bit, ANDed with the bits of 1 / (1 - p): two VALU ops
                            // (asm: hipcc rewrites the intrinsic form into and + compare + select)
                            const unsigned sb = __float_as_uint(dr.scale);
                            int b0, b1;
                            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(b0) : "v"(cw[qb]), "n"(8 * (r >> 2) + (r & 3)));
                            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(b1) : "v"(cw[qb]), "n"(8 * (r >> 2) + (r & 3) + 1));
                            m0 = __uint_as_float((unsigned)b0 & sb);
                            m1 = __uint_as_float((unsigned)b1 & sb);
                        } else {
                            m0 = drop_keep(__float_as_uint(k4[2 * w]), (unsigned)nk, dr.thresh) ? dr.scale : 0.f;
                            m1 = drop_keep(__float_as_uint(k4[2 * w + 1]), (unsigned)nk, dr.thresh) ? dr.scale : 0.f;
                        }
                        pp[r >> 1] = pack2_bf16(p0 * m0, p1 * m1);  // dropped P (what multiplied V in the forward)
                        pd[r >> 1] = pack2_bf16(p0 * fmaf(dp[r], m0, dl4[2 * w]), p1 * fmaf(dp[r + 1], m1, dl4[2 * w + 1]));
                    } else {
                        pp[r >> 1] = pack2_bf16(p0, p1);
                        pd[r >> 1] = pack2_bf16(p0 * dp[r], p1 * dp[r + 1]);
                    }
                }
            }
            // (reading these fragments ahead of their MFMAs, as the dQ kernel does, needs 8 more registers than the 168 of three
            // waves per SIMD leave: measured 597 -> 766 us per layer with the spills, so each fragment is read where it is used)
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile qt + 1 have landed
        __syncthreads();
    }
    // ---- the CLS query: one vector update per key (p and ds are scalars per lane) ----
    float cls_ds;
    {
        f32x4 qc[4], oc[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qc[s] = *(const f32x4*)(qkv + cls_row * ld + head * HD + 16 * s + 8 * lh);
            oc[s] = *(const f32x4*)(dctx + cls_row * (size_t)D + head * HD + 16 * s + 8 * lh);
        }
        const float sc = dot_frag(kf, qc), dpc = dot_frag(vf, oc);
        const float p = __builtin_amdgcn_exp2f(fmaf(sc, c, -lse[stat0 + Np]));
        float m = 1.f;
        if (DROP) m = drop_keep(drop_key(dr.seed, dr.stream, (unsigned)((b * A + head) * N + Np)), (unsigned)nk, dr.thresh) ? dr.scale : 0.f;
        const float pm = p * m, ds = p * fmaf(dpc, m, -delta[stat0 + Np]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const uint2 tq = *(const uint2*)(qkv + cls_row * ld + head * HD + dt * 32 + 8 * g4 + 4 * lh);
                const uint2 to = *(const uint2*)(dctx + cls_row * (size_t)D + head * HD + dt * 32 + 8 * g4 + 4 * lh);
                dk[dt][4 * g4 + 0] = fmaf(ds, bf_lo(tq.x), dk[dt][4 * g4 + 0]);
                dk[dt][4 * g4 + 1] = fmaf(ds, bf_hi(tq.x), dk[dt][4 * g4 + 1]);
                dk[dt][4 * g4 + 2] = fmaf(ds, bf_lo(tq.y), dk[dt][4 * g4 + 2]);
                dk[dt][4 * g4 + 3] = fmaf(ds, bf_hi(tq.y), dk[dt][4 * g4 + 3]);
                dv[dt][4 * g4 + 0] = fmaf(pm, bf_lo(to.x), dv[dt][4 * g4 + 0]);
                dv[dt][4 * g4 + 1] = fmaf(pm, bf_hi(to.x), dv[dt][4 * g4 + 1]);
                dv[dt][4 * g4 + 2] = fmaf(pm, bf_lo(to.y), dv[dt][4 * g4 + 2]);
                dv[dt][4 * g4 + 3] = fmaf(pm, bf_hi(to.y), dv[dt][4 * g4 + 3]);
            }
        cls_ds = ds;
    }
    if (colp) {   // block-uniform: the dK and dV thirds of the bias gradient's partial record (see the dQ kernel)
        col_stage(dk, k_valid ? 0.125f : 0.f, (float*)&lds[0][0][0] + 2048, tid);
        col_stage(dv, k_valid ? 1.f : 0.f, (float*)&lds[0][0][0] + 4096, tid);
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
    // (behind the stores: the 64 accumulator registers are free by now)
    // ---- this block's share of the CLS QUERY's gradient: dq_cls += ds k over its (valid) keys ----
    {
        float* red = (float*)&lds[0][0][0];
        __syncthreads();   // every wave is done with the tiles
        const size_t rec = (((size_t)b * A + head) * ((Np + TB - 1) / TB) + at.rt) * 192;
        cls_stage(kf, k_valid ? cls_ds : 0.f, red, tid);
        __syncthreads();
        cls_finish(red, clsp + rec, 1, tid);
        if (colp) cls_finish(red + 2048, colp + rec + 64, 2, tid);
    }
}

// ---------------------------------------------------------------------------------- the CLS token
// dq of the CLS query, dk / dv of the CLS key of one (head, image): the per-block partial vectors of the two kernels above,
// summed in block order, plus the CLS-CLS pair's term.  One wave per (head, image), lane = channel.
template <bool DROP>
__global__ __launch_bounds__(64) void attn_bwd_cls_finish_kernel(const bf16_t* __restrict__ qkv,
                                                                 const bf16_t* __restrict__ dctx,
                                                                 const float* __restrict__ lse,
                                                                 const float* __restrict__ delta,
                                                                 const float* __restrict__ clsp,
                                                                 bf16_t* __restrict__ dqkv, int B, int Np, int A,
                                                                 DropArgs dr) {
    const int d = threadIdx.x, head = blockIdx.x, b = blockIdx.y;
    const int D = A * HD, ld = 3 * D, N = Np + 1, nrt = (Np + TB - 1) / TB;
    const size_t cls_row = (size_t)B * Np + b, stat_c = ((size_t)b * A + head) * N + Np;
    const float c = 0.125f * LOG2E;
    float aq = 0.f, ak = 0.f, av = 0.f;
    const float* rec = clsp + ((size_t)b * A + head) * nrt * 192;
    for (int rt = 0; rt < nrt; ++rt) {   // fixed order: deterministic
        aq += rec[rt * 192 + d];
        ak += rec[rt * 192 + 64 + d];
        av += rec[rt * 192 + 128 + d];
    }
    const float qc = bf16_to_f32(qkv[cls_row * ld + head * HD + d]);
    const float kc = bf16_to_f32(qkv[cls_row * ld + D + head * HD + d]);
    const float vc = bf16_to_f32(qkv[cls_row * ld + 2 * D + head * HD + d]);
    const float oc = bf16_to_f32(dctx[cls_row * (size_t)D + head * HD + d]);
    const float s = wave_sum(qc * kc), dp = wave_sum(oc * vc);
    const float p = __builtin_amdgcn_exp2f(fmaf(s, c, -lse[stat_c]));
    float m = 1.f;
    if (DROP) m = drop_keep(drop_key(dr.seed, dr.stream, (unsigned)((b * A + head) * N + Np)), (unsigned)Np, dr.thresh) ? dr.scale : 0.f;
    const float ds = p * fmaf(dp, m, -delta[stat_c]);
    aq = fmaf(ds, kc, aq);
    ak = fmaf(ds, qc, ak);
    av = fmaf(p * m, oc, av);
    dqkv[cls_row * ld + head * HD + d] = f32_to_bf16(aq * 0.125f);
    dqkv[cls_row * ld + D + head * HD + d] = f32_to_bf16(ak * 0.125f);
    dqkv[cls_row * ld + 2 * D + head * HD + d] = f32_to_bf16(av);
}

// d(qkv bias)[part D + 64 head + d] = the column sum of dq | dk | dv: the per-block partial records of the two MFMA kernels
// (patch tokens, fp32) plus the B CLS rows as the finish kernel stored them, in a fixed order.  Block = (part, head), 1024
// threads = 16 groups x 64 channels (group g takes images g, g + 16, ...: with 4 groups a thread walked 144 dependent
// loads, 36 us for 4.7 MB).
__global__ __launch_bounds__(1024) void attn_bwd_bias_kernel(const float* __restrict__ colp, const bf16_t* __restrict__ dqkv,
                                                             float* __restrict__ dbias, int B, int Np, int A) {
    __shared__ float red[16][64];
    const int d = threadIdx.x & 63, grp = threadIdx.x >> 6, head = blockIdx.x % A, part = blockIdx.x / A;
    const int D = A * HD, ld = 3 * D, nrt = (Np + TB - 1) / TB;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int b = grp; b < B; b += 16) {
        const float* rec = colp + ((size_t)b * A + head) * nrt * 192 + part * 64 + d;
        int rt = 0;
        for (; rt + 3 < nrt; rt += 4) {   // four loads in flight
            a0 += rec[rt * 192];
            a1 += rec[(rt + 1) * 192];
            a2 += rec[(rt + 2) * 192];
            a3 += rec[(rt + 3) * 192];
        }
        for (; rt < nrt; ++rt) a0 += rec[rt * 192];
        a1 += bf16_to_f32(dqkv[((size_t)B * Np + b) * ld + part * D + head * HD + d]);   // the CLS row of image b
    }
    red[grp][d] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (grp == 0) {
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) v += red[g][d];
        dbias[part * D + head * HD + d] = v;
    }
}

template <bool DROP>
int launch_bwd(const bf16_t* qkv, const bf16_t* ctx, const bf16_t* dctx, const float* lse, float* dvec, bf16_t* dqkv, int B,
               int Np, int A, DropArgs dr, hipStream_t s, const unsigned* maskw, float* dbias) {
    const dim3 grid((unsigned)((Np + TB - 1) / TB) * A * B);  // 1-D: attn_tile() places the tiles
    float* clsp = dvec + (size_t)B * A * (Np + 1);            // per-block CLS partials behind the delta vector
    float* colp = dbias ? clsp + (size_t)B * A * ((Np + TB - 1) / TB) * 192 : nullptr;   // ... column-sum partials behind them
#define VITSEG_BWD(RG, MWORDS)                                                                                         \
    do {                                                                                                               \
        hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<DROP, RG, MWORDS>), grid, dim3(256), 0, s, qkv, ctx, dctx, lse,    \
                           dvec, clsp, colp, dqkv, B, Np, A, dr, maskw);                                               \
        VITSEG_LAUNCH_CHECK("attn_bwd_dq_bf16");                                                                       \
        hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<DROP, RG, MWORDS>), grid, dim3(256), 0, s, qkv, dctx, lse, dvec,  \
                           clsp, colp, dqkv, B, Np, A, dr, maskw);                                                     \
        VITSEG_LAUNCH_CHECK("attn_bwd_dkv_bf16");                                                                      \
    } while (0)
    if (Np % TB != 0) VITSEG_BWD(true, false);
    else if (DROP && maskw) VITSEG_BWD(false, DROP);
    else VITSEG_BWD(false, false);
#undef VITSEG_BWD
    hipLaunchKernelGGL(attn_bwd_cls_finish_kernel<DROP>, dim3(A, B), dim3(64), 0, s, qkv, dctx, lse, dvec, clsp, dqkv, B, Np,
                       A, dr);
    VITSEG_LAUNCH_CHECK("attn_bwd_cls_finish");
    if (dbias) {
        hipLaunchKernelGGL(attn_bwd_bias_kernel, dim3(3 * A), dim3(1024), 0, s, colp, dqkv, dbias, B, Np, A);
        VITSEG_LAUNCH_CHECK("attn_bwd_bias");
    }
    return VITSEG_OK;
}

}  // namespace

// floats of launch_attention_bwd_bf16's scratch: delta [B, A, Np + 1] + one [dq_cls | dk_cls | dv_cls][64] record and one
// [colsum dq | dk | dv][64] record (the QKV bias gradient's partials) per 128-token block of every (image, head)
size_t attention_bwd_bf16_scratch_floats(int B, int Np, int A) {
    return (size_t)B * A * ((size_t)(Np + 1) + (size_t)((Np + TB - 1) / TB) * 384);
}

int launch_attention_bwd_bf16(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* dvec,
                              void* dqkv, int B, int Np, int A, DropArgs dr, hipStream_t s, const unsigned* maskw,
                              float* dbias) {
    VITSEG_CHECK_ARG(qkv && ctx && dctx && lse && dvec && dqkv, VITSEG_EINVAL, "attention_bwd_bf16: null pointer");
    return dr.thresh ? launch_bwd<true>((const bf16_t*)qkv, (const bf16_t*)ctx, (const bf16_t*)dctx, lse, dvec, (bf16_t*)dqkv, B, Np, A, dr, s, maskw, dbias)
                     : launch_bwd<false>((const bf16_t*)qkv, (const bf16_t*)ctx, (const bf16_t*)dctx, lse, dvec, (bf16_t*)dqkv, B, Np, A, dr, s, nullptr, dbias);
}

}  // namespace vitseg
