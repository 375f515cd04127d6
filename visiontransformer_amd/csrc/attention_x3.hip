// Multi-head self-attention core for VITSEG_F32X3: fp32 q | k | v in, fp32 context out, both products on the fp16 matrix
// pipe with split operands (transformers/models/vit/modeling_vit.py:164-189; same row layout and CLS handling as
// attention_f32.hip, whose side kernel still serves the CLS queries).
//
// Every operand value a is carried as hi = half(a), lo = half((a - hi) * 2^11)  (a = hi + lo * 2^-11 to 22 bits):
//   S^T = K . Q^T :  s0 += Khi.Qhi,  s1 += Klo.Qhi + Khi.Qlo,  S = s0 + s1 * 2^-11       (v_mfma_f32_32x32x16_f16)
//   softmax in fp32, lane-local (a query per lane), exactly as in the fp32 / 16-bit kernels
//   O^T = V^T . P^T: o0 += Vhi.Phi,  o1 += Vlo.Phi + Vhi.Plo,  O = (o0 + o1 * 2^-11) / l
// K and V are split while they are staged into LDS (hi and lo planes, 128-byte rows, the 16-bit kernel's swizzles),
// Q once per block in registers, P in registers right after the exponentials.  3 half MFMAs per 16 k against 8
// fp32 ones that are 16x slower each: the fp32 kernel's 10 ms per forward (its MFMA pipe at ~90 %) become ~4 ms.
#include "kernels.hpp"

namespace vitseg {
namespace {

constexpr int HD = 64;   // head dim
constexpr int QB = 128;  // queries per block (4 waves x 32)
constexpr int KB = 64;   // keys per LDS tile
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LO_SCALE = 2048.f, LO_INV = 1.0f / 2048.f;
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int kappa(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }

// 8 floats -> 8 hi halves + 8 scaled lo halves (16 bytes each)
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, f32x4& hi, f32x4& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const _Float16 a0 = (_Float16)a[2 * j], a1 = (_Float16)a[2 * j + 1];
        const _Float16 b0 = (_Float16)b[2 * j], b1 = (_Float16)b[2 * j + 1];
        h[j] = __builtin_bit_cast(unsigned, f16x2{a0, a1});
        h[2 + j] = __builtin_bit_cast(unsigned, f16x2{b0, b1});
        l[j] = H16<f16_t>::pack2((a[2 * j] - (float)a0) * LO_SCALE, (a[2 * j + 1] - (float)a1) * LO_SCALE);
        l[2 + j] = H16<f16_t>::pack2((b[2 * j] - (float)b0) * LO_SCALE, (b[2 * j + 1] - (float)b1) * LO_SCALE);
    }
    hi = __builtin_bit_cast(f32x4, (uint4){h[0], h[1], h[2], h[3]});
    lo = __builtin_bit_cast(f32x4, (uint4){l[0], l[1], l[2], l[3]});
}

template <bool RAGGED>
__global__ __launch_bounds__(256, 2) void attn_x3_kernel(const float* __restrict__ qkv, float* __restrict__ ctx, int B, int Np,
                                                         int A) {
    // [buffer][K|V][hi|lo][key * 64 + d] halves, rows of 128 B with XOR-swizzled 16-B chunks: 64 KiB
    __shared__ __attribute__((aligned(16))) unsigned short lds[2][2][2][KB * HD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const AttnTile at = attn_tile((Np + QB - 1) / QB, A);
    const int head = at.head, b = at.b;
    const int D = A * HD, ld = 3 * D;
    const size_t row0 = (size_t)b * Np;
    const size_t cls_row = (size_t)B * Np + b;
    const float* qbase = qkv + head * HD;
    const float* kbase = qkv + D + head * HD;
    const float* vbase = qkv + 2 * D + head * HD;
    const float c = 0.125f * LOG2E;

    // ---- this lane's query row: k-step s holds Q[16 s + 8 lh .. +7] (B operand of S^T), split once ----
    const int q_local = at.rt * QB + wave * 32 + li;
    const bool q_valid = q_local < Np;
    const size_t q_row = row0 + (q_valid ? q_local : Np - 1);
    f32x4 qh[4], ql[4];
    float m_run, l_run;
    f32x16 o0[2], o1[2];
    {
        // CLS key folded into the initial online-softmax state, in exact fp32 (its score and its V row)
        float part = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float* qp = qbase + q_row * ld + 16 * s + 8 * lh;
            const float* kp = kbase + cls_row * ld + 16 * s + 8 * lh;
            const f32x4 qa = *(const f32x4*)qp, qb = *(const f32x4*)(qp + 4);
            const f32x4 ka = *(const f32x4*)kp, kb = *(const f32x4*)(kp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) part = fmaf(qb[e], kb[e], fmaf(qa[e], ka[e], part));
            split8(qa, qb, qh[s], ql[s]);
        }
        m_run = part + __shfl_xor(part, 32, 64);
        l_run = lh == 0 ? 1.f : 0.f;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 t = *(const f32x4*)(vbase + cls_row * ld + dt * 32 + 8 * g4 + 4 * lh);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o0[dt][4 * g4 + e] = t[e];
                    o1[dt][4 * g4 + e] = 0.f;
                }
            }
    }

    // ---- K/V staging: thread owns 8 floats (one 16-B half chunk lc) of keys lr + 32 i ----
    const int lc = tid & 7, lr = tid >> 3;
    f32x4 rk[2][2], rv[2][2];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int key = kt * KB + lr + 32 * i;
            if (RAGGED) key = min(key, Np - 1);  // duplicates are masked below
            const size_t off = (row0 + key) * ld + 8 * lc;
            rk[i][0] = *(const f32x4*)(kbase + off);
            rk[i][1] = *(const f32x4*)(kbase + off + 4);
            rv[i][0] = *(const f32x4*)(vbase + off);
            rv[i][1] = *(const f32x4*)(vbase + off + 4);
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = lr + 32 * i;
            f32x4 hi, lo;
            split8(rk[i][0], rk[i][1], hi, lo);
            const int kpos = key * HD + ((lc ^ ((key >> 1) & 7)) << 3);                // K: row reads
            *(f32x4*)&lds[buf][0][0][kpos] = hi;
            *(f32x4*)&lds[buf][0][1][kpos] = lo;
            split8(rv[i][0], rv[i][1], hi, lo);
            const int vpos = key * HD + ((lc ^ (((key >> 1) & 1) << 2)) << 3);         // V: transposed reads
            *(f32x4*)&lds[buf][1][0][vpos] = hi;
            *(f32x4*)&lds[buf][1][1][vpos] = lo;
        }
    };
    const int k_off = li * HD;
    const int k_sw = (li >> 1) & 7;
    const int g = lane & 15, grp = lane >> 4, tq = g >> 2, tp = g & 3;
    const int v_row = 4 * (grp >> 1) + tq;
    const int v_sw = ((tq >> 1) & 1) << 2;
    const int v_dchunk = 2 * (grp & 1) + (tp >> 1), v_half = (tp & 1) * 4;

    const int nkt = (Np + KB - 1) / KB;
    gload(0);
    swrite(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        gload(min(kt + 1, nkt - 1));  // the last tile re-stages itself: keeps the body branch-free
        __builtin_amdgcn_sched_barrier(0);
        const unsigned short* Kh = lds[buf][0][0];
        const unsigned short* Kl = lds[buf][0][1];
        const unsigned short* Vh = lds[buf][1][0];
        const unsigned short* Vl = lds[buf][1][1];

        // S^T[key][query], two blocks of 32 keys
        f32x16 st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s1;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = s1[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int pos = kb * 32 * HD + k_off + (((2 * s + lh) ^ k_sw) << 3);
                const bf16x8 kh = __builtin_bit_cast(bf16x8, *(const f32x4*)&Kh[pos]);
                const bf16x8 kl = __builtin_bit_cast(bf16x8, *(const f32x4*)&Kl[pos]);
                st[kb] = H16<f16_t>::mfma(kh, __builtin_bit_cast(bf16x8, qh[s]), st[kb]);
                s1 = H16<f16_t>::mfma(kl, __builtin_bit_cast(bf16x8, qh[s]), s1);
                s1 = H16<f16_t>::mfma(kh, __builtin_bit_cast(bf16x8, ql[s]), s1);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = fmaf(s1[r], LO_INV, st[kb][r]);
        }
        if (RAGGED) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt * KB + kb * 32 + kappa(r, lh) >= Np) st[kb][r] = -INFINITY;
        }
        float mx = fmaxf(st[0][0], st[1][0]);
#pragma unroll
        for (int r = 1; r < 16; r += 1) mx = fmaxf(fmaxf(mx, st[0][r]), st[1][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o0[dt][r] *= alpha;
                    o1[dt][r] *= alpha;
                }
            m_run = m_new;
        }
        const float mc = m_run * c;
        float psum = 0.f;
        unsigned ph[2][8], pl[2][8];  // P^T fragments (hi / scaled lo): [kb][4 s + w] = registers 8 s + 2 w, + 1
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = __builtin_amdgcn_exp2f(fmaf(st[kb][r], c, -mc));
                const float p1 = __builtin_amdgcn_exp2f(fmaf(st[kb][r + 1], c, -mc));
                psum += p0 + p1;
                const _Float16 h0 = (_Float16)p0, h1 = (_Float16)p1;
                ph[kb][r >> 1] = __builtin_bit_cast(unsigned, f16x2{h0, h1});
                pl[kb][r >> 1] = H16<f16_t>::pack2((p0 - (float)h0) * LO_SCALE, (p1 - (float)h1) * LO_SCALE);
            }
        l_run += psum;

        // O^T[d][query] += V^T[d][key] . P^T[key][query]
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pfh = __builtin_bit_cast(
                    bf16x8, (uint4){ph[kb][4 * s], ph[kb][4 * s + 1], ph[kb][4 * s + 2], ph[kb][4 * s + 3]});
                const bf16x8 pfl = __builtin_bit_cast(
                    bf16x8, (uint4){pl[kb][4 * s], pl[kb][4 * s + 1], pl[kb][4 * s + 2], pl[kb][4 * s + 3]});
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int row_a = kb * 32 + 16 * s + v_row;
                    const int ch = (4 * dt + v_dchunk) ^ v_sw;
                    const int p0 = row_a * HD + (ch << 3) + v_half, p8 = (row_a + 8) * HD + (ch << 3) + v_half;
                    const s16x4 hlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vh + p0));
                    const s16x4 hhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vh + p8));
                    const s16x4 llo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vl + p0));
                    const s16x4 lhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vl + p8));
                    const bf16x8 vh = {hlo[0], hlo[1], hlo[2], hlo[3], hhi[0], hhi[1], hhi[2], hhi[3]};
                    const bf16x8 vl = {llo[0], llo[1], llo[2], llo[3], lhi[0], lhi[1], lhi[2], lhi[3]};
                    o0[dt] = H16<f16_t>::mfma(vh, pfh, o0[dt]);
                    o1[dt] = H16<f16_t>::mfma(vl, pfh, o1[dt]);
                    o1[dt] = H16<f16_t>::mfma(vh, pfl, o1[dt]);
                }
            }

        swrite(buf ^ 1);
        __syncthreads();
    }

    // ---- normalise and store: lane holds d = 32 dt + 8 g4 + 4 lh + e of its query ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_valid) {
        float* out = ctx + q_row * (size_t)D + head * HD;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = fmaf(o1[dt][4 * g4 + e], LO_INV, o0[dt][4 * g4 + e]) * inv;
                *(f32x4*)(out + dt * 32 + 8 * g4 + 4 * lh) = t;
            }
    }
}

}  // namespace

// patch queries only; the B*A CLS queries are served by attention_f32.hip's side kernel (launch_attention_f32, x3 = true)
int launch_attention_x3_main(const float* qkv, float* ctx, int B, int Np, int A, hipStream_t s) {
    const dim3 grid((unsigned)((Np + QB - 1) / QB) * A * B);  // 1-D: attn_tile() places the tiles
    if (Np % QB == 0)
        hipLaunchKernelGGL(attn_x3_kernel<false>, grid, dim3(256), 0, s, qkv, ctx, B, Np, A);
    else
        hipLaunchKernelGGL(attn_x3_kernel<true>, grid, dim3(256), 0, s, qkv, ctx, B, Np, A);
    VITSEG_LAUNCH_CHECK("attn_x3");
    return VITSEG_OK;
}

}  // namespace vitseg
