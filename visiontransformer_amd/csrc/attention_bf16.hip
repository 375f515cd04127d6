// Multi-head self-attention core, bf16 operands / fp32 softmax and accumulation:
//   ctx = softmax(q k^T * hd^-0.5) v
//
// bf16 counterpart of attention_f32.hip (same reference lines:
// transformers/models/vit/modeling_vit.py:164-189, :207-238; same "patches first" row layout,
// CLS key folded into the initial online-softmax state, CLS queries in a side kernel).
//
// v_mfma_f32_32x32x16_bf16 for both products (bound: 2.5 PFLOP/s dense bf16; at head_dim 64 the
// softmax VALU work per tile is comparable to the MFMA time, so the kernel is co-bound by VALU).
//   S^T = K . Q^T : A = K rows from LDS (one ds_read_b128 per k-step), B = Q in registers.
//       The accumulator puts a QUERY on each lane: max/sum/rescale are lane-local + one lane^32 swap.
//   O^T = V^T . P^T : B = the exponentiated S^T registers 8s..8s+7 packed to bf16 (element j of lane
//       half h is key 16s + 8(j>>2) + 4h + (j&3)), A = V^T gathered from the ROW-major V tile by two
//       ds_read_b64_tr_b16 per k-step (hardware transpose; lane 4q+p of a 16-lane group supplies
//       &V[key0+q][d0+4p] and receives V[key0..key0+3][d0 + lane] -- verified by tools/probes/tr16_probe.hip).
// Scores are scaled inside the exponent: p = exp2(fma(s, c, -m*c)), c = hd^-0.5 * log2(e).
#include <type_traits>

#include "kernels.hpp"

namespace vitseg {
namespace {

constexpr int HD = 64;   // head dim
constexpr int QB = 128;  // queries per block (4 waves x 32)
constexpr int KB = 64;   // keys per LDS tile
constexpr float LOG2E = 1.4426950408889634f;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int kappa(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }

// H: format tag of the 16-bit storage (bf16_t or f16_t, common.hpp H16<>); pointers are raw 16-bit words.
// DROP: attention-probability dropout compiled in (training); RAGGED: Np is not a multiple of the 128-query block.
// Online softmax with a DEFERRED maximum: p = exp2(fma(s, c, -c m)) with the running maximum m (raw score units) in the
// fma's addend -- one VALU op per score, accumulators started from the MFMA's zero constant; m only moves -- and O, l are
// only rescaled -- when some score exceeds it by more than 8 / c, i.e. p > 2^8 (wave-uniform, rare after the first tiles):
// P stays <= 256, exact in the fp32 sums and harmless in the 16-bit P.  (MFMA and fp32 VALU cycles add up on a SIMD --
// DESIGN.md section 3 -- so every op per score counts.)  (Prescaling q by c instead would save the multiply too, but a second rounding of q to the 16-bit
// format costs accuracy on peaked rows: measured 0.041 vs 0.03 max error in the op test; it belongs into the QKV
// projection's epilogue, before the first rounding.)
// MW (with DROP, without RAGGED): the keep bits come as precomputed words (common.hpp attn_dropmask_words): the lane mask
// of accumulator register r is one scalar 64-bit load, applied with one v_cndmask; the 1 / (1 - p) factor moves into the
// final normalisation.
template <bool DROP, bool RAGGED, bool MW, typename H>
__global__ __launch_bounds__(256, 3) void attn_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                           float* __restrict__ lse, int B, int Np, int A,
                                                           DropArgs dr, const unsigned* __restrict__ maskw) {
    static_assert(!MW || (DROP && !RAGGED), "mask words: dropout on, whole 128-query blocks");
    // [buffer][K|V][key * 64 + d] bf16, rows of 128 B with XOR-swizzled 16-B chunks: 32 KiB
    __shared__ __attribute__((aligned(16))) bf16_t lds[2][2][KB * HD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const AttnTile at = attn_tile((Np + QB - 1) / QB, A);
    const int head = at.head, b = at.b;
    const int D = A * HD, ld = 3 * D;
    const size_t row0 = (size_t)b * Np;
    const size_t cls_row = (size_t)B * Np + b;
    const bf16_t* qbase = qkv + head * HD;
    const bf16_t* kbase = qkv + D + head * HD;
    const bf16_t* vbase = qkv + 2 * D + head * HD;
    const float c = 0.125f * LOG2E;

    // ---- this lane's query row: k-step s holds Q[16 s + 8 lh .. +7] (B operand of S^T) ----
    const int q_local = at.rt * QB + wave * 32 + li;
    const bool q_valid = !RAGGED || q_local < Np;
    const size_t q_row = row0 + (q_valid ? q_local : Np - 1);
    f32x4 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const f32x4*)(qbase + q_row * ld + 16 * s + 8 * lh);

    // attention-probability dropout: the row sum keeps every term, only the P that multiplies V is masked
    const unsigned dkey = drop_key(dr.seed, dr.stream, (unsigned)((b * A + head) * (Np + 1) + q_local));
    // ---- online-softmax state initialised with the CLS key (raw-score units) ----
    float m_run, l_run;
    f32x16 o[2];
    {
        float part = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 kc = *(const f32x4*)(kbase + cls_row * ld + 16 * s + 8 * lh);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned qu = __float_as_uint(qf[s][e]), ku = __float_as_uint(kc[e]);
                part = fmaf(H16<H>::lo(qu), H16<H>::lo(ku), part);
                part = fmaf(H16<H>::hi(qu), H16<H>::hi(ku), part);
            }
        }
        m_run = part + __shfl_xor(part, 32, 64);
        l_run = lh == 0 ? 1.f : 0.f;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const uint2 t = *(const uint2*)(vbase + cls_row * ld + dt * 32 + 8 * g4 + 4 * lh);
                o[dt][4 * g4 + 0] = H16<H>::lo(t.x);
                o[dt][4 * g4 + 1] = H16<H>::hi(t.x);
                o[dt][4 * g4 + 2] = H16<H>::lo(t.y);
                o[dt][4 * g4 + 3] = H16<H>::hi(t.y);
            }
        if (DROP) {
            const float kc = drop_keep(dkey, (unsigned)Np, dr.thresh) ? (MW ? 1.f : dr.scale) : 0.f;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= kc;
        }
    }

    // ---- K/V staging by LDS-DMA: a wave instruction moves 8 keys x 128 B straight into the tile (lane l lands at
    // + 16 l: key l >> 3, chunk position l & 7; the XOR swizzles of the two images are applied to the per-lane SOURCE
    // chunk).  ONE buffer descriptor over this image's K|V rows; the per-lane offset is a loop constant and the tile's
    // row offset a scalar, so staging costs no vector instruction, no staging registers and no ds_write (the flat-load
    // form spent ~45 VALU per tile on addresses and 16 VGPRs on the hop; the loop is VALU-bound:
    // profiles/r03_pmcw_attn_fwd_serial.json, 0.73 VALU active).  Rows beyond the image's last patch are out of the
    // descriptor's range and read as zeros (RAGGED masks their scores).  Issued from inline asm and ordered by one
    // hand-placed vmcnt(0) before the tile's barrier (the loop has no other vector-memory instruction).
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 kv_rsrc;
    {
        const unsigned long long base = (unsigned long long)(kbase + row0 * ld);
        kv_rsrc[0] = (int)(unsigned)base;
        kv_rsrc[1] = (int)(unsigned)((base >> 32) & 0xffffu);   // stride 0
        kv_rsrc[2] = Np * ld * 2;                                // bytes from the first patch row to the end of the image
        kv_rsrc[3] = 0x00020000;
    }
    const int drow = 16 * __builtin_amdgcn_readfirstlane(wave) + (lane >> 3);   // key of this lane in piece 0; piece 1: + 8
    unsigned voff_k[2], voff_v[2];
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
        const int key = drow + 8 * pc;
        voff_k[pc] = (unsigned)(key * ld * 2 + (((lane & 7) ^ ((key >> 1) & 7)) << 4));
        voff_v[pc] = (unsigned)(key * ld * 2 + D * 2 + (((lane & 7) ^ (((key >> 1) & 1) << 2)) << 4));
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)&lds[0][0][0];
    auto stage = [&](int kt, int buf) {   // tile kt -> ring slot buf
        const unsigned soff = (unsigned)(kt * KB * ld * 2);
        const unsigned dst = lds_base + (unsigned)(buf * 2 * KB * HD * 2) + (unsigned)(__builtin_amdgcn_readfirstlane(wave) * 2048);
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(dst + pc * 1024), "v"(voff_k[pc]), "s"(kv_rsrc), "s"(soff) : "memory");
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(dst + KB * HD * 2 + pc * 1024), "v"(voff_v[pc]), "s"(kv_rsrc), "s"(soff) : "memory");
        }
    };
    // per-lane LDS element offsets
    const int k_off = li * HD;                 // + kb*32*HD, chunk (2s+lh) ^ ((li>>1)&7)
    const int k_sw = (li >> 1) & 7;
    const int g = lane & 15, grp = lane >> 4, tq = g >> 2, tp = g & 3;
    // V^T fragment of (kb, s, dt): keys kb*32 + 16 s + 4 (grp>>1) + tq (+8), d = 32 dt + 16 (grp&1) + 4 tp
    const int v_row = 4 * (grp >> 1) + tq;
    const int v_sw = ((tq >> 1) & 1) << 2;
    const int v_dchunk = 2 * (grp & 1) + (tp >> 1), v_half = (tp & 1) * 4;

    const int nkt = (Np + KB - 1) / KB;
    // this wave's row of mask words: 16 lane masks (64-bit) per 32-key block, wave-uniform address -> scalar loads
    const unsigned long* mrow = nullptr;
    if (MW) {
        const int nb = Np >> 5, qg = at.rt * 4 + __builtin_amdgcn_readfirstlane(wave);
        mrow = (const unsigned long*)maskw + ((size_t)((b * A + head) * nb + qg) * nb) * 16;
    }
    TileMasks lm;
    if (MW) lm.load(mrow);
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // One 64-key tile.
    // No running-maximum pass: the tile is exponentiated against the CURRENT m (p = exp2(c s - c m)) and only its
    // per-lane sum is looked at -- when some lane's sum exceeds 2^16 (or is not finite) m was too low for this tile, and
    // the rare path below raises it to the tile's maximum, rescales O and l and exponentiates again.  Otherwise every
    // p <= 2^16: exact in the fp32 sums, harmless in the 16-bit P (same exponent range as fp32 for bf16; IEEE half
    // holds 65 504, so the fp16 build uses 2^12).  That removes 16 v_max3, an LDS round trip and a compare per tile.
    constexpr float PSUM_LIMIT = sizeof(H) == 2 && std::is_same<H, f16_t>::value ? 4096.f : 65536.f;
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (MW) lm.wait();
        const bf16_t* Ks = lds[buf][0];
        const bf16_t* Vs = lds[buf][1];

        // S^T[key][query], two blocks of 32 keys
        f32x16 st[2];
        auto scores = [&]() {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;   // inline-constant C operand of the first MFMA: no moves
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const f32x4 kf = *(const f32x4*)&Ks[kb * 32 * HD + k_off + (((2 * s + lh) ^ k_sw) << 3)];
                    st[kb] = H16<H>::mfma(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[s]), st[kb]);
                }
            }
            if (RAGGED) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kt * KB + kb * 32 + kappa(r, lh) >= Np) st[kb][r] = -INFINITY;
            }
        };
        scores();
        // the next tile's DMA is issued HERE, behind the 8 QK^T MFMAs the wave would otherwise only wait for (at the top of
        // the tile the ~250 cycles of issue had nothing to hide behind); the last tile re-stages itself: branch-free body
        stage(min(kt + 1, nkt - 1), buf ^ 1);
        float nmc = -m_run * c;   // p = exp2(c s - c m): the running maximum enters through the fma's addend
        float psum;
        unsigned pk[2][8];  // P^T fragments: pk[kb][4 s + w] = registers 8 s + 2 w, 8 s + 2 w + 1
        auto exponentiate = [&]() {
            float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    float p0 = __builtin_amdgcn_exp2f(fmaf(st[kb][r], c, nmc));
                    float p1 = __builtin_amdgcn_exp2f(fmaf(st[kb][r + 1], c, nmc));
                    ps0 += p0;
                    ps1 += p1;
                    if (MW) {
                        p0 = mask_select(lm.reg(kb, r), p0);
                        p1 = mask_select(lm.reg(kb, r + 1), p1);
                    } else if (DROP) {
                        const unsigned k0 = (unsigned)(kt * KB + kb * 32);
                        p0 = drop_keep(dkey, k0 + kappa(r, lh), dr.thresh) ? p0 * dr.scale : 0.f;
                        p1 = drop_keep(dkey, k0 + kappa(r + 1, lh), dr.thresh) ? p1 * dr.scale : 0.f;
                    }
                    pk[kb][r >> 1] = H16<H>::pack2(p0, p1);
                    // pin the pack HERE: otherwise hipcc sinks all 16 of them below the guard branch and keeps the 32
                    // fp32 probabilities alive across it (+16 registers = one wave per SIMD less)
                    asm volatile("" : "+v"(pk[kb][r >> 1]));
                }
            psum = ps0 + ps1;
        };
        exponentiate();
        if (__builtin_amdgcn_ballot_w64(!(psum <= PSUM_LIMIT)) != 0) {   // wave-uniform, rare after the first tiles
            scores();            // again, rather than keeping 32 score registers alive across the common path
            float mx = fmaxf(st[0][0], st[1][0]);
#pragma unroll
            for (int r = 1; r < 16; r += 1) mx = fmaxf(fmaxf(mx, st[0][r]), st[1][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float delta = fmaxf(mx - m_run, 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-delta * c);
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            m_run += delta;
            nmc = -m_run * c;
            exponentiate();      // now every p <= 1
        }
        l_run += psum;

        // O^T[d][query] += V^T[d][key] . P^T[key][query]
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pf = __builtin_bit_cast(
                    bf16x8, (uint4){pk[kb][4 * s], pk[kb][4 * s + 1], pk[kb][4 * s + 2], pk[kb][4 * s + 3]});
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int row_a = kb * 32 + 16 * s + v_row;
                    const int ch = (4 * dt + v_dchunk) ^ v_sw;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(Vs + row_a * HD + (ch << 3) + v_half));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(Vs + (row_a + 8) * HD + (ch << 3) + v_half));
                    const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o[dt] = H16<H>::mfma(vf, pf, o[dt]);
                }
            }

        if (MW) lm.load(mrow + (size_t)min(kt + 1, nkt - 1) * 32);   // next tile's lane masks (see TileMasks)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile kt + 1 have landed
        __syncthreads();
    }
    if (MW) lm.wait();   // the last iteration's mask load is still writing its 64 SGPRs: nothing may reuse them before it lands

    // ---- normalise and store: lane holds d = 32 dt + 8 g4 + 4 lh + e of its query ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = (MW ? dr.scale : 1.0f) / l_tot;
    // log-sum-exp of the scaled scores in log2 units, [b][head][token] with the CLS token last (backward)
    if (lse && q_valid && lh == 0)
        lse[((size_t)b * A + head) * (Np + 1) + q_local] = m_run * c + __builtin_amdgcn_logf(l_tot);
    if (q_valid) {
        bf16_t* out = ctx + q_row * (size_t)D + head * HD;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                uint2 t;
                t.x = H16<H>::pack2(o[dt][4 * g4] * inv, o[dt][4 * g4 + 1] * inv);
                t.y = H16<H>::pack2(o[dt][4 * g4 + 2] * inv, o[dt][4 * g4 + 3] * inv);
                *(uint2*)(out + dt * 32 + 8 * g4 + 4 * lh) = t;
            }
    }
}

// The B*A CLS queries: one block per (head, image); plain VALU in fp32 on bf16 inputs.
template <typename H>
__global__ __launch_bounds__(1024) void attn_cls_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                            float* __restrict__ lse, int B, int Np, int A,
                                                            DropArgs dr) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int N = Np + 1;
    float* sc = sm;
    float* red = sm + ((N + 63) & ~63);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int head = blockIdx.x, b = blockIdx.y;
    const int D = A * HD, ld = 3 * D;
    const size_t row0 = (size_t)b * Np, cls_row = (size_t)B * Np + b;
    const bf16_t* qp = qkv + cls_row * ld + head * HD;
    const bf16_t* kbase = qkv + D + head * HD;
    const bf16_t* vbase = qkv + 2 * D + head * HD;
    const int sub = lane & 15, grp = lane >> 4;  // 16 lanes x 8 B cover one 64-element row

    const uint2 qu = *(const uint2*)(qp + 4 * sub);
    const float qs = 0.125f * LOG2E;
    const float q0 = H16<H>::lo(qu.x) * qs, q1 = H16<H>::hi(qu.x) * qs, q2 = H16<H>::lo(qu.y) * qs, q3 = H16<H>::hi(qu.y) * qs;
    // 4 keys per 16-lane group and iteration: four independent row loads in flight (the loop is latency-bound)
    for (int base = wave * 4; base < N; base += 256) {
        float part[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = base + 64 * u + grp;
            part[u] = 0.f;
            if (key < N) {
                const size_t row = key < Np ? row0 + key : cls_row;
                const uint2 ku = *(const uint2*)(kbase + row * ld + 4 * sub);
                part[u] = q0 * H16<H>::lo(ku.x) + q1 * H16<H>::hi(ku.x) + q2 * H16<H>::lo(ku.y) + q3 * H16<H>::hi(ku.y);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float pu = part[u];
            pu = row16_sum(pu);   // (DPP: __shfl_xor compiles to ds_bpermute_b32)
            const int key = base + 64 * u + grp;
            if (sub == 0 && key < N) sc[key] = pu;
        }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int i = tid; i < N; i += 1024) mx = fmaxf(mx, sc[i]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);
    __syncthreads();
    float sum = 0.f;
    for (int i = tid; i < N; i += 1024) {
        const float pv = __builtin_amdgcn_exp2f(sc[i] - mx);
        sc[i] = pv;
        sum += pv;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    float ltot = 0.f;
    for (int w = 0; w < 16; ++w) ltot += red[w];
    const float inv = 1.0f / ltot;
    if (lse && tid == 0) lse[((size_t)b * A + head) * N + Np] = mx + __builtin_amdgcn_logf(ltot);
    __syncthreads();
    const int kg = tid >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int key0 = kg; key0 < N; key0 += 256) {
        uint2 vu[4];
        float pv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = min(key0 + 64 * u, N - 1);
            const size_t row = key < Np ? row0 + key : cls_row;
            vu[u] = *(const uint2*)(vbase + row * ld + 4 * sub);
            pv[u] = key0 + 64 * u < N ? sc[key] : 0.f;
            if (dr.thresh)
                pv[u] = drop_keep(drop_key(dr.seed, dr.stream, (unsigned)((b * A + head) * N + Np)), (unsigned)key,
                                  dr.thresh) ? pv[u] * dr.scale : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[0] = fmaf(pv[u], H16<H>::lo(vu[u].x), acc[0]);
            acc[1] = fmaf(pv[u], H16<H>::hi(vu[u].x), acc[1]);
            acc[2] = fmaf(pv[u], H16<H>::lo(vu[u].y), acc[2]);
            acc[3] = fmaf(pv[u], H16<H>::hi(vu[u].y), acc[3]);
        }
    }
    *(f32x4*)&red[kg * 64 + 4 * sub] = acc;
    __syncthreads();
    if (tid < 64) {
        float s = 0.f;
        for (int gI = 0; gI < 64; ++gI) s += red[gI * 64 + tid];
        ctx[cls_row * (size_t)D + head * HD + tid] = H16<H>::bits(s * inv);
    }
}

}  // namespace

template <typename H>
static int launch_attention_h16(const void* qkv, void* ctx, float* lse, int B, int Np, int A, DropArgs dr, hipStream_t s,
                                const unsigned* maskw) {
    VITSEG_CHECK_ARG(qkv && ctx && B > 0 && Np > 0 && A > 0, VITSEG_EINVAL, "attention_bf16: bad arguments");
    const dim3 grid((unsigned)((Np + QB - 1) / QB) * A * B);  // 1-D: attn_tile() places the tiles
    const bool ragged = Np % QB != 0, drop = dr.thresh != 0;
#define VITSEG_ATTN_LAUNCH(DR, RG, MWORDS)                                                                           \
    hipLaunchKernelGGL((attn_bf16_kernel<DR, RG, MWORDS, H>), grid, dim3(256), 0, s, (const bf16_t*)qkv, (bf16_t*)ctx, \
                       lse, B, Np, A, dr, maskw)
    if (drop) {
        if (ragged) VITSEG_ATTN_LAUNCH(true, true, false);
        else if (maskw) VITSEG_ATTN_LAUNCH(true, false, true);
        else VITSEG_ATTN_LAUNCH(true, false, false);
    } else {
        if (ragged) VITSEG_ATTN_LAUNCH(false, true, false); else VITSEG_ATTN_LAUNCH(false, false, false);
    }
#undef VITSEG_ATTN_LAUNCH
    VITSEG_LAUNCH_CHECK("attn_bf16");
    const size_t smem = (size_t)(((Np + 1 + 63) & ~63) + 64 * 64) * sizeof(float);
    VITSEG_CHECK_ARG(smem <= 64 * 1024, VITSEG_ESHAPE, "attention_bf16: sequence too long for the CLS kernel");
    hipLaunchKernelGGL(attn_cls_bf16_kernel<H>, dim3(A, B), dim3(1024), smem, s, (const bf16_t*)qkv, (bf16_t*)ctx, lse, B,
                       Np, A, dr);
    VITSEG_LAUNCH_CHECK("attn_cls_bf16");
    return VITSEG_OK;
}

int launch_attention_bf16(const void* qkv, void* ctx, float* lse, int B, int Np, int A, DropArgs dr, hipStream_t s,
                          bool f16, const unsigned* maskw) {
    return f16 ? launch_attention_h16<f16_t>(qkv, ctx, lse, B, Np, A, dr, s, maskw)
               : launch_attention_h16<bf16_t>(qkv, ctx, lse, B, Np, A, dr, s, maskw);
}

}  // namespace vitseg
