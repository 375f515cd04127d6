// Multi-head self-attention core in fp32:  ctx = softmax(q k^T * hd^-0.5) v
//
// Replaces eager_attention_forward / torch SDPA behind ViTAttention
// (transformers/models/vit/modeling_vit.py:164-189, called from :207-238); 7-23 % of the
// reference's CPU profile.  No mask, eval mode, fp32 softmax.
//
// Flash-style (the [N,N] score matrix is never materialised), on the fp32-input matrix
// cores (v_mfma_f32_32x32x2_f32, exact fmaf chains; bound: 157.3 TFLOP/s fp32 matrix peak).
//
// Row layout ("patches first"): q/k/v of patch token t of image b live in row b*Np + t of
// qkv[.., 3D]; the CLS token of image b in row B*Np + b.  That keeps the Np patch keys of an
// image a whole number of 64-key tiles at 512x512 (Np = 1024) -- the CLS key is folded in as
// the INITIAL state of the online softmax (m = s_cls, l = 1, O = v_cls), so the key loop has
// no ragged tail and no mask.  The B*A CLS *queries* are one extra row each and run in a
// small second kernel.
//
// Per block: 128 queries of one (image, head); 4 waves x 32 queries.  Scores are computed
// transposed, S^T = K . Q^T (A operand = K tile from LDS, B operand = Q held in registers), so a
// lane owns ONE query column: row max / row sum / rescale are lane-local plus one lane^32
// exchange, and the exponentiated S^T accumulator registers are directly the B operand of
// O^T = V^T . P^T (register s <-> key (s&3) + 8(s>>2) + 4(lane>>5) on both sides).
// head_dim is fixed at 64 (every configuration of the reference: 192/3, 512/8, 768/12, 1024/16).
#include "kernels.hpp"

namespace vitseg {
namespace {

constexpr int HD = 64;      // head dim
constexpr int QB = 128;     // queries per block
constexpr int KB = 64;      // keys per LDS tile
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int kappa(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }

template <bool RAGGED>
__global__ __launch_bounds__(256, 2) void attn_f32_kernel(const float* __restrict__ qkv, float* __restrict__ ctx,
                                                          float* __restrict__ lse, int B, int Np, int A,
                                                          DropArgs dr) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][KB * HD];  // [buffer][K|V][key*64 + d], 64 KiB

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const AttnTile at = attn_tile((Np + QB - 1) / QB, A);
    const int head = at.head, b = at.b;
    const int D = A * HD, ld = 3 * D;
    const size_t row0 = (size_t)b * Np;           // first patch row of this image
    const size_t cls_row = (size_t)B * Np + b;    // CLS row of this image
    const float* qbase = qkv + head * HD;
    const float* kbase = qkv + D + head * HD;
    const float* vbase = qkv + 2 * D + head * HD;

    // ---- this lane's query row, pre-scaled by hd^-0.5 * log2(e) (scores live in log2 units) ----
    const int q_local = at.rt * QB + wave * 32 + li;
    const bool q_valid = q_local < Np;
    const size_t q_row = row0 + (q_valid ? q_local : Np - 1);
    const float qscale = 0.125f * LOG2E;
    float qreg[32];  // element 4c+e = Q[8c + 4 lh + e]
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f32x4 t = *(const f32x4*)(qbase + q_row * ld + 8 * c + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; ++e) qreg[4 * c + e] = t[e] * qscale;
    }

    // attention-probability dropout (modeling_vit.py:184): P is dropped AFTER normalisation, so the row sum l
    // keeps every term and only the P that multiplies V is masked / rescaled
    const unsigned dkey = drop_key(dr.seed, dr.stream, (unsigned)((b * A + head) * (Np + 1) + q_local));
    // ---- softmax state, initialised with the CLS key ----
    // No running maximum: every exponential of the row is taken against ONE reference m0 (the CLS score; it only moves on
    // the guarded rare path below), carried as the MFMA C operand -m0, so the loop's vector work per score is exp2 + one
    // add.  On this chip the fp32 matrix instruction and the vector ALU do not overlap (profiles/
    // r03_simd_overlap_probe_fp32_mfma.txt): the max / subtract / rescale instructions of the classic online softmax
    // (about 130 per 64-key tile and wave) were time added to the MFMAs, not hidden behind them.
    float m_run, l_run;
    f32x16 negm;
    f32x16 o[2];
    {
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const f32x4 t = *(const f32x4*)(kbase + cls_row * ld + 8 * c + 4 * lh);
#pragma unroll
            for (int e = 0; e < 4; ++e) part = fmaf(qreg[4 * c + e], t[e], part);
        }
        m_run = part + __shfl_xor(part, 32, 64);
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = -m_run;
        l_run = lh == 0 ? 1.f : 0.f;  // halves are summed at the end
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 t = *(const f32x4*)(vbase + cls_row * ld + dt * 32 + 8 * g4 + 4 * lh);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[dt][4 * g4 + e] = t[e];
            }
        if (dr.thresh) {
            const float kc = drop_keep(dkey, (unsigned)Np, dr.thresh) ? dr.scale : 0.f;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= kc;
        }
    }

    // ---- K/V tile staging by LDS-DMA (buffer_load_dwordx4 ... lds): a wave instruction moves 4 key rows x 256 B straight
    // into the tile (lane l lands at + 16 l: row l >> 4, chunk position l & 15; K's XOR swizzle is applied to the per-lane
    // SOURCE chunk); wave w fills keys [16 w, 16 w + 16) of K and of V: 8 instructions per tile.  No staging registers, no
    // ds_write, no lgkmcnt wait in front of the tile barrier (the register-staged form spent ~4 % of the kernel there,
    // tools/probes/attn_f32_where.sh); keys beyond Np are out of the descriptor's range and read as zeros (RAGGED masks
    // them below).
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    auto make_rsrc = [](const void* base, long long bytes) {
        const unsigned long long bb = (unsigned long long)base;
        i32x4 r;
        r[0] = (int)(unsigned)bb;
        r[1] = (int)(unsigned)((bb >> 32) & 0xffffu);   // stride 0
        r[2] = (int)(bytes < 0x7fffffffll ? bytes : 0x7fffffffll);
        r[3] = 0x00020000;
        return r;
    };
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const i32x4 k_rsrc = make_rsrc(kbase + row0 * ld, ((long long)(Np - 1) * ld + HD) * 4);
    const i32x4 v_rsrc = make_rsrc(vbase + row0 * ld, ((long long)(Np - 1) * ld + HD) * 4);
    unsigned voffk[4], voffv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int key = 16 * wv + 4 * i + (lane >> 4);
        voffk[i] = (unsigned)(key * ld * 4 + (((lane & 15) ^ (key & 15)) << 4));
        voffv[i] = (unsigned)(key * ld * 4 + ((lane & 15) << 4));
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)&lds[0][0][0];
    auto stage = [&](int kt, int buf) {
        const unsigned soff = (unsigned)(kt * KB * ld * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned dk = lds_base + (unsigned)(((buf * 2 + 0) * KB * HD + (16 * wv + 4 * i) * HD) * 4);
            const unsigned dv = lds_base + (unsigned)(((buf * 2 + 1) * KB * HD + (16 * wv + 4 * i) * HD) * 4);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(dk), "v"(voffk[i]), "s"(k_rsrc), "s"(soff) : "memory");
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(dv), "v"(voffv[i]), "s"(v_rsrc), "s"(soff) : "memory");
        }
    };

    // ---- the CLS QUERY of this (image, head) rides in the block of row tile 0 (it used to be a second kernel that streamed
    // all K / V again: 36 us per layer at batch 32).  Thread = (16-byte chunk cq of the 64 channels, key subset cks): per
    // key tile it scores its 4 keys against the CLS query on its 4 channels (completed across the 16 lanes of the subset),
    // keeps an online softmax per subset and accumulates its chunk of P.V; the 16 subsets are merged at the end.  About
    // 70 vector instructions per tile and wave, in one block out of eight.
    const bool cls_blk = at.rt == 0;
    const int cq = lane & 15, cks = tid >> 4;
    f32x4 cq4 = {0.f, 0.f, 0.f, 0.f}, co = {0.f, 0.f, 0.f, 0.f};
    float cm = -1e30f, cl = 0.f;
    unsigned ckey = 0;
    auto dot16 = [&](const f32x4& a, const f32x4& bb) {   // 4-channel partial, completed across the subset's 16 lanes
        float t = a[0] * bb[0] + a[1] * bb[1] + a[2] * bb[2] + a[3] * bb[3];
        // DPP inside the 16-lane row (vector-ALU operand modifiers: row_ror 8, row_ror 4, then the two quad swaps), not
        // four trips through the LDS crossbar
        t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x128, 0xf, 0xf, true));
        t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x124, 0xf, 0xf, true));
        t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4e, 0xf, 0xf, true));
        t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xb1, 0xf, 0xf, true));
        return t;
    };
    if (cls_blk) {
        cq4 = *(const f32x4*)(qbase + cls_row * ld + 4 * cq);
#pragma unroll
        for (int e = 0; e < 4; ++e) cq4[e] *= qscale;
        ckey = drop_key(dr.seed, dr.stream, (unsigned)((b * A + head) * (Np + 1) + Np));
        const float scc = dot16(cq4, *(const f32x4*)(kbase + cls_row * ld + 4 * cq));   // the CLS key: subset 0 starts from it
        if (cks == 0) {
            cm = scc;
            cl = 1.f;
            co = *(const f32x4*)(vbase + cls_row * ld + 4 * cq);
            if (dr.thresh) {
                const float kc = drop_keep(ckey, (unsigned)Np, dr.thresh) ? dr.scale : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) co[e] *= kc;
            }
        }
    }
    auto cls_tile = [&](int kt, const float* Ks, const float* Vs) {
        float sc[4];
        f32x4 v4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = 4 * cks + u;
            const f32x4 kf = *(const f32x4*)&Ks[key * HD + ((cq ^ (key & 15)) << 2)];
            v4[u] = *(const f32x4*)&Vs[key * HD + (cq << 2)];
            sc[u] = dot16(cq4, kf);
            if (RAGGED && kt * KB + key >= Np) sc[u] = -INFINITY;
        }
        const float m_new = fmaxf(fmaxf(fmaxf(cm, sc[0]), fmaxf(sc[1], sc[2])), sc[3]);
        const float alpha = __builtin_amdgcn_exp2f(cm - m_new);
        cm = m_new;
        cl *= alpha;
#pragma unroll
        for (int e = 0; e < 4; ++e) co[e] *= alpha;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float pv = __builtin_amdgcn_exp2f(sc[u] - m_new);
            cl += pv;
            if (dr.thresh) pv = drop_keep(ckey, (unsigned)(kt * KB + 4 * cks + u), dr.thresh) ? pv * dr.scale : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) co[e] = fmaf(pv, v4[u][e], co[e]);
        }
    };

    const int nkt = (Np + KB - 1) / KB;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        stage(min(kt + 1, nkt - 1), buf ^ 1);  // the last tile re-stages itself: keeps the body branch-free
        __builtin_amdgcn_sched_barrier(0);  // pin the issue point of the prefetch
        const float* Ks = lds[buf][0];
        const float* Vs = lds[buf][1];
        if (cls_blk) cls_tile(kt, Ks, Vs);

        // S^T[key][query] - m0 for 2 blocks of 32 keys (the accumulators start at -m0)
        f32x16 st[2];
        auto scores = [&](const f32x16& c0) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                st[kb] = c0;
                const int key = kb * 32 + li;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const f32x4 kf = *(const f32x4*)&Ks[key * HD + (((2 * c + lh) ^ (key & 15)) << 2)];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        st[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qreg[4 * c + e], st[kb], 0, 0, 0);
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
        scores(negm);
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(st[kb][r]);
                st[kb][r] = pv;
                psum += pv;
            }
        // guard: a score more than ~100 log2 units above the reference (or a non-finite one).  Rare: the tile is redone
        // against a new reference = the larger of the old one and this tile's maximum, the classic rescale of l and O.
        if (__builtin_amdgcn_ballot_w64(!(psum <= 0x1p100f)) != 0ull) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            scores(zero);
            float mx = st[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kb][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[r] = -m_new;
            psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(st[kb][r] - m_new);
                    st[kb][r] = pv;
                    psum += pv;
                }
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
        l_run += psum;
        if (dr.thresh) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    st[kb][r] = drop_keep(dkey, (unsigned)(kt * KB + kb * 32 + kappa(r, lh)), dr.thresh)
                                    ? st[kb][r] * dr.scale : 0.f;
        }
        // O^T[d][query] += V^T[d][key] . P^T[key][query]
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int key = kb * 32 + kappa(s, lh);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const float vf = Vs[key * HD + dt * 32 + li];
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, st[kb][s], o[dt], 0, 0, 0);
                }
            }

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile kt + 1 have landed
        __syncthreads();
    }

    // ---- normalise and store: lane holds d = dt*32 + 8 g4 + 4 lh + e of its query ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    // log-sum-exp of the scaled scores in log2 units, [b][head][token] with the CLS token last (backward)
    if (lse && q_valid && lh == 0) lse[((size_t)b * A + head) * (Np + 1) + q_local] = m_run + __builtin_amdgcn_logf(l_tot);
    if (q_valid) {
        float* out = ctx + q_row * (size_t)D + head * HD;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = o[dt][4 * g4 + e] * inv;
                *(f32x4*)(out + dt * 32 + 8 * g4 + 4 * lh) = t;
            }
    }
    // ---- the CLS query: merge the 16 key subsets (the tile buffers are free: the loop ended with a barrier and no DMA
    // is in flight) ----
    if (cls_blk) {
        float* red = &lds[0][0][0];   // [16][64] partial outputs, then 16 maxima, 16 sums
        *(f32x4*)&red[cks * 64 + 4 * cq] = co;
        if (cq == 0) {
            red[1024 + cks] = cm;
            red[1040 + cks] = cl;
        }
        __syncthreads();
        if (tid < 64) {
            float M = red[1024];
#pragma unroll
            for (int g = 1; g < 16; ++g) M = fmaxf(M, red[1024 + g]);
            float L = 0.f, O = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float w = __builtin_amdgcn_exp2f(red[1024 + g] - M);
                L = fmaf(red[1040 + g], w, L);
                O = fmaf(red[g * 64 + tid], w, O);
            }
            ctx[cls_row * (size_t)D + head * HD + tid] = O / L;
            if (lse && tid == 0) lse[((size_t)b * A + head) * (Np + 1) + Np] = M + __builtin_amdgcn_logf(L);
        }
    }
}

// The B*A CLS queries for the split-operand (x3) path, whose patch kernel does not carry them: one block per (head, image);
// plain VALU (1 x N x 64 per block).
__global__ __launch_bounds__(1024) void attn_cls_f32_kernel(const float* __restrict__ qkv, float* __restrict__ ctx,
                                                           float* __restrict__ lse, int B, int Np, int A,
                                                           DropArgs dr) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // scores[N] then reduce scratch
    const int N = Np + 1;
    float* sc = sm;
    float* red = sm + ((N + 63) & ~63);  // 64 key groups x 64 floats
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int head = blockIdx.x, b = blockIdx.y;
    const int D = A * HD, ld = 3 * D;
    const size_t row0 = (size_t)b * Np, cls_row = (size_t)B * Np + b;
    const float* qp = qkv + cls_row * ld + head * HD;
    const float* kbase = qkv + D + head * HD;
    const float* vbase = qkv + 2 * D + head * HD;
    const int sub = lane & 15, grp = lane >> 4;  // 16 lanes x 16 B cover one 64-float row

    f32x4 q4 = *(const f32x4*)(qp + 4 * sub);
    const float qs = 0.125f * LOG2E;
    for (int e = 0; e < 4; ++e) q4[e] *= qs;
    // 4 keys per 16-lane group and iteration: four independent row loads in flight (the loop is latency-bound)
    for (int base = wave * 4; base < N; base += 256) {
        float part[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = base + 64 * u + grp;
            part[u] = 0.f;
            if (key < N) {
                const size_t row = key < Np ? row0 + key : cls_row;
                const f32x4 k4 = *(const f32x4*)(kbase + row * ld + 4 * sub);
                part[u] = q4[0] * k4[0] + q4[1] * k4[1] + q4[2] * k4[2] + q4[3] * k4[3];
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
    // out[d] = sum_key p[key] V[key][d]: thread = (key group kg of 16, 4-float chunk sub)
    const int kg = tid >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int key0 = kg; key0 < N; key0 += 256) {
        f32x4 v4[4];
        float pv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = min(key0 + 64 * u, N - 1);
            const size_t row = key < Np ? row0 + key : cls_row;
            v4[u] = *(const f32x4*)(vbase + row * ld + 4 * sub);
            pv[u] = key0 + 64 * u < N ? sc[key] : 0.f;
            if (dr.thresh)
                pv[u] = drop_keep(drop_key(dr.seed, dr.stream, (unsigned)((b * A + head) * N + Np)), (unsigned)key,
                                  dr.thresh) ? pv[u] * dr.scale : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(pv[u], v4[u][e], acc[e]);
    }
    *(f32x4*)&red[kg * 64 + 4 * sub] = acc;
    __syncthreads();
    if (tid < 64) {
        float s = 0.f;
        for (int g = 0; g < 64; ++g) s += red[g * 64 + tid];
        ctx[cls_row * (size_t)D + head * HD + tid] = s * inv;
    }
}

}  // namespace

int launch_attention_f32(const float* qkv, float* ctx, float* lse, int B, int Np, int A, DropArgs dr, hipStream_t s,
                         bool x3) {
    VITSEG_CHECK_ARG(qkv && ctx && B > 0 && Np > 0 && A > 0, VITSEG_EINVAL, "attention_f32: bad arguments");
    const dim3 grid((unsigned)((Np + QB - 1) / QB) * A * B);  // 1-D: attn_tile() places the tiles
    if (x3) {  // VITSEG_F32X3 (inference): patch queries on the fp16 pipe with split operands, attention_x3.hip
        VITSEG_CHECK_ARG(!lse && !dr.thresh, VITSEG_EINVAL, "attention (x3) is an inference path: no lse / dropout");
        if (int rc = launch_attention_x3_main(qkv, ctx, B, Np, A, s)) return rc;
    } else if (Np % QB == 0) {
        hipLaunchKernelGGL(attn_f32_kernel<false>, grid, dim3(256), 0, s, qkv, ctx, lse, B, Np, A, dr);
    } else {
        hipLaunchKernelGGL(attn_f32_kernel<true>, grid, dim3(256), 0, s, qkv, ctx, lse, B, Np, A, dr);
    }
    VITSEG_LAUNCH_CHECK("attn_f32");
    if (!x3) return VITSEG_OK;   // the fp32 kernel carries the CLS queries itself (row tile 0 of every image and head)
    const size_t smem = (size_t)(((Np + 1 + 63) & ~63) + 64 * 64) * sizeof(float);
    VITSEG_CHECK_ARG(smem <= 64 * 1024, VITSEG_ESHAPE, "attention_f32: sequence too long for the CLS kernel");
    hipLaunchKernelGGL(attn_cls_f32_kernel, dim3(A, B), dim3(1024), smem, s, qkv, ctx, lse, B, Np, A, dr);
    VITSEG_LAUNCH_CHECK("attn_cls_f32");
    return VITSEG_OK;
}

}  // namespace vitseg
