// GEMM on the matrix cores:  C[M,N] = epi(A[M,K] . W[N,K]^T + bias),  fp32 or bf16 operands.
//
// Replaces aten::addmm / mkldnn_convolution behind nn.Linear / Conv2d in the reference
// (56-65 % of the CPU profile): q/k/v/o projections and MLP
// (transformers/models/vit/modeling_vit.py:207-254), patch embedding (:62-69) and
// seg_head.0 (model/CE/classes.py:241) through the gathering A loaders.
//
// T = float : v_mfma_f32_32x32x2_f32, an exact fp32 fmaf chain (no TF32 on gfx950), 64 cycles per
//             issue per SIMD -> matrix-pipe bound by a wide margin (roofline 157.3 TFLOP/s).
// T = bf16  : v_mfma_f32_32x32x16_bf16, fp32 accumulate, 32 cycles per issue (roofline 2.5 PFLOP/s);
//             8x the FLOPs per staged byte, so this tile shape leans on L2 bandwidth.
//
// Both element types share one structure because a staged operand row is 128 bytes either way:
// block 128x128, BK = 128 B of K per row (32 floats / 64 bf16), 4 waves as 2(M) x 2(N), each wave
// 64x64 = 2x2 MFMA tiles of 32x32 (64 accumulator registers).  Lane half h consumes the 16-byte
// chunks 2j+h (j = 0..3) of a row: for bf16 that chunk IS the 32x32x16 operand (k = 8h..8h+7 of
// k-step j); for fp32 the k index inside a 32x32x2 MFMA is arbitrary as long as A and B agree, so
// the chunk's four floats feed four consecutive MFMAs.  Either way a fragment is ONE ds_read_b128
// from a row-major tile whose chunk index is XOR-swizzled with (row >> 1) & 7 (conflict-free).
// Global->LDS goes through registers (the gathering loaders need per-chunk zero-fill),
// double-buffered in LDS with one barrier per K step; fragments are double-buffered in registers.
#include <stdlib.h>

#include <type_traits>

#include "kernels.hpp"

namespace vitseg {
namespace {

constexpr int BM = 128, BN = 128;
constexpr int BKF = 32;    // row length in 4-byte LDS words

template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int CE = 4, BKE = 32; };           // elements per chunk / per row
template <> struct Elem<bf16_t> { static constexpr int CE = 8, BKE = 64; };          // bf16 bits
template <> struct Elem<f16_t> { static constexpr int CE = 8, BKE = 64; };           // IEEE half

template <typename T>
struct ARow {
    // per-row state of the A loader, computed once (row is fixed for a thread)
    const T* base;  // row base pointer (A_PLAIN / A_PATCH: image base of (b, gy, gx))
    int y, x;           // A_CONV3: pixel coordinates
    bool valid;
};

template <typename T, int AMODE>
__device__ __forceinline__ ARow<T> make_arow(const GemmArgs& p, int m) {
    ARow<T> r;
    r.valid = m < p.M;
    r.y = r.x = 0;
    const T* A = (const T*)p.A;
    if (!r.valid) {
        r.base = A;
        return r;
    }
    if (AMODE == A_PLAIN) {
        r.base = A + (size_t)m * p.lda;
    } else if (AMODE == A_PATCH) {
        const int b = m / p.Np, t = m - b * p.Np;
        const int gy = t / p.g, gx = t - gy * p.g;
        r.base = A + ((size_t)b * p.Cin * p.S + (size_t)gy * p.P) * p.S + (size_t)gx * p.P;
    } else {
        const int b = m / p.Np, t = m - b * p.Np;
        r.y = t / p.g;
        r.x = t - r.y * p.g;
        r.base = A + (size_t)m * p.D;  // centre pixel's token row
    }
    return r;
}

// Branch-free: out-of-range chunks read a safe in-bounds address and are zeroed by a select, so
// the whole K step stays one basic block and the scheduler can spread the loads between MFMAs.
template <typename T, int AMODE>
__device__ __forceinline__ f32x4 load_a(const GemmArgs& p, const ARow<T>& r, int k, bool& ok) {
    ok = r.valid && k < p.K;
    const int kc = min(k, p.K - Elem<T>::CE);
    const T* ptr;
    if (AMODE == A_PLAIN) {
        ptr = r.base + kc;
    } else if (AMODE == A_PATCH) {
        const int pp = p.P * p.P;
        const int c = kc / pp, rem = kc - c * pp;
        const int py = rem / p.P, px = rem - py * p.P;
        ptr = r.base + ((size_t)c * p.S + py) * p.S + px;
    } else {
        const int tap = kc / p.D, d = kc - tap * p.D;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int yy = r.y + ky - 1, xx = r.x + kx - 1;
        const bool in = (unsigned)yy < (unsigned)p.g && (unsigned)xx < (unsigned)p.g;
        ok = ok && in;
        ptr = r.base + (in ? ((ptrdiff_t)(ky - 1) * p.g + (kx - 1)) * p.D : 0) + d;
    }
    return *(const f32x4*)ptr;  // zeroed by the caller when !ok, at LDS-write time (keeps the load in flight)
}

// MI x NI MFMA tiles (32x32) per wave; rows start at a_row0, columns at b_col0 inside the block tile.
//   <2,2>: the regular 2(M) x 2(N) wave grid, 64x64 per wave.
//   <1,1>/<2,1>: "thin" row tiles (<= 32 / <= 64 valid rows: the CLS rows that follow the B*Np patch
//   rows); the 4 waves split the 128 columns so such a tile costs 1/4 (1/2) of a regular one and is
//   scheduled first, instead of adding a whole extra round of blocks to the launch.
// TA / TB: operand storage form.  0 ("N-form"): [row][k], the reduction index is contiguous (activations,
// nn.Linear weights).  1 ("T-form", fp32 only): [k][row], the reduction index is the slow one -- what the
// backward GEMMs meet (dgrad reads W as [n][k] with n the reduction; wgrad reads dY and X with the token
// index as the reduction).  T-form tiles are staged as [32 k][128 rows] and read one float per lane per
// MFMA (conflict-free: consecutive lanes = consecutive rows), so no transposed copies are ever made.
//
// X3 (fp32 operands only): "fp32 on the fp16 matrix pipe".  Every operand value is split while it is staged,
// a = hi + lo * 2^-11 with hi = half(a), lo = half((a - hi) * 2^11) (22 significand bits, the scaled low part stays a
// normal half), and the product is accumulated as  acc0 += hi.hi',  acc1 += lo.hi' + hi.lo'  with
// v_mfma_f32_32x32x16_f16 (fp32 accumulate); C = acc0 + acc1 * 2^-11.  Dropped: lo.lo' (2^-22 relative) and the
// rounding of the low parts (2^-22): products are good to ~2^-21 instead of exact, at 3 half-precision MFMAs per
// 16 k instead of 8 fp32 ones (16x slower each).  The hi / lo planes share the 16 KiB an fp32 operand tile uses
// (64-byte rows, chunk index XOR-swizzled with (row >> 1) & 3), so buffering, loaders and epilogue are unchanged.
template <typename T, typename OutT, int AMODE, int EPI, int MI, int NI, int TA = 0, int TB = 0, int X3 = 0>
__device__ __forceinline__ void gemm_tile(const GemmArgs& p, float (*lds)[2][BM * BKF], int m0, int n0, int a_row0,
                                          int b_col0) {
    constexpr int CE = Elem<T>::CE, BKE = Elem<T>::BKE;
    constexpr int BK = BKF;  // LDS words per row
    const int tid = threadIdx.x, lane = tid & 63;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    static_assert(sizeof(T) == 4 || (TA == 0 && TB == 0), "T-form operands are implemented for fp32 only");
    static_assert(!X3 || (sizeof(T) == 4 && TA == 0 && TB == 0), "X3 splits N-form fp32 operands");
    f32x16 acc1[X3 ? MI : 1][X3 ? NI : 1];  // X3: the 2^-11-weighted cross terms
    if constexpr (X3) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[mi][ni][r] = 0.f;
    }
    const int li = lane & 31, lh = lane >> 5;
    const int sw = (li >> 1) & 7;
    const int a_off = TA ? a_row0 + li : (a_row0 + li) * BK, b_off = TB ? b_col0 + li : (b_col0 + li) * BK;

    // Fragment registers are double-buffered one MFMA group (16 MFMAs = 1024 matrix-pipe cycles)
    // ahead, so no LDS latency is exposed: group j+1's ds_reads are issued before group j's MFMAs.
    // The loop is rotated around the barrier: group 3 of tile kt runs AFTER the barrier that
    // publishes tile kt+1, with tile kt+1's group-0 fragments already being read.
    f32x4 a[2][MI], b[2][NI];
    auto lfrag = [&](int buf, int j, int slot) {
        const int ch = (((2 * j + lh) ^ sw) << 2);
        const int kq = (8 * j + 4 * lh) * BM;  // T-form: element e of the group is k = 8j + 4h + e
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            if constexpr (TA) {
#pragma unroll
                for (int e = 0; e < 4; ++e) a[slot][mi][e] = lds[buf][0][kq + e * BM + a_off + mi * 32];
            } else {
                a[slot][mi] = *(const f32x4*)&lds[buf][0][a_off + mi * 32 * BK + ch];
            }
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            if constexpr (TB) {
#pragma unroll
                for (int e = 0; e < 4; ++e) b[slot][ni][e] = lds[buf][1][kq + e * BN + b_off + ni * 32];
            } else {
                b[slot][ni] = *(const f32x4*)&lds[buf][1][b_off + ni * 32 * BK + ch];
            }
        }
    };
    // one group = the MFMAs fed by one 16-byte chunk per operand: 4 k-steps of 32x32x2 (fp32, quarter
    // q = one float of the chunk) or 1 k-step of 32x32x16 (bf16, issued with quarter 0)
    auto mfma_group = [&](int slot, int e0, int e1) {
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int e = e0; e < e1; ++e)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[slot][mi][e], b[slot][ni][e],
                                                                           acc[mi][ni], 0, 0, 0);
        } else {
            if (e0 == 0) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = H16<T>::mfma(__builtin_bit_cast(bf16x8, a[slot][mi]),
                                                   __builtin_bit_cast(bf16x8, b[slot][ni]), acc[mi][ni]);
            }
        }
    };

    // split-K: blockIdx.y owns K steps [kt0, kt0 + KT) and writes its own partial C (p.C + y * split_stride)
    const int KT_all = (p.K + BKE - 1) / BKE;
    const int nsplit = gridDim.y, split = blockIdx.y;
    const int kt0 = (int)((long long)KT_all * split / nsplit);
    const int KT = (int)((long long)KT_all * (split + 1) / nsplit) - kt0;
    if constexpr (sizeof(T) == 4) {
        // ---- global -> register staging ----
        // N-form: thread owns 16-B chunk lc (of 8) of rows lr + 32 i.  T-form: chunk tc (of 32) of k rows tr + 8 i.
        const int lc = tid & 7, lr = tid >> 3;
        const int tc = tid & 31, tr = tid >> 5;
        ARow<T> arow[4];
        const T* wrow[4];
        bool wvalid[4];
        if constexpr (!TA) {
#pragma unroll
            for (int i = 0; i < 4; ++i) arow[i] = make_arow<T, AMODE>(p, m0 + lr + 32 * i);
        }
        if constexpr (!TB) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + lr + 32 * i;
                wvalid[i] = n < p.N;
                wrow[i] = (const T*)p.W + (size_t)(wvalid[i] ? n : 0) * p.ldw;
            }
        }
        const bool ta_col_ok = m0 + tc * 4 < p.M, tb_col_ok = n0 + tc * 4 < p.N;
        f32x4 ra[4], rb[4];
        bool oka[4], okb[4];
        auto gload = [&](int kt) {
            const int k = (kt + kt0) * BKE + lc * CE;
            const int kc = min(k, p.K - CE);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (TA) {
                    const int kr = (kt + kt0) * BKE + tr + 8 * i;
                    oka[i] = ta_col_ok && kr < p.K;
                    ra[i] = *(const f32x4*)((const T*)p.A + (size_t)(oka[i] ? kr : 0) * p.lda +
                                            (oka[i] ? m0 + tc * 4 : 0));
                } else {
                    ra[i] = load_a<T, AMODE>(p, arow[i], k, oka[i]);
                }
                if constexpr (TB) {
                    const int kr = (kt + kt0) * BKE + tr + 8 * i;
                    okb[i] = tb_col_ok && kr < p.K;
                    rb[i] = *(const f32x4*)((const T*)p.W + (size_t)(okb[i] ? kr : 0) * p.ldw +
                                            (okb[i] ? n0 + tc * 4 : 0));
                } else {
                    rb[i] = *(const f32x4*)(wrow[i] + kc);
                    okb[i] = wvalid[i] && k < p.K;
                }
            }
        };
        const int wpos = lr * BK + ((lc ^ ((lr >> 1) & 7)) << 2);  // N-form; + 32*i rows -> same swizzle term
        const int tpos = tr * BM + tc * 4;                          // T-form: [k][128], + 8*i k-rows
        // X3: row r of an operand tile = 64 B of hi halves (plane 0, first 8 KiB) and 64 B of lo halves (plane 1);
        // this thread's 4 floats are half `lc & 1` of 16-byte chunk `lc >> 1`, stored at chunk ^ ((r >> 1) & 3)
        const int xpos = lr * 16 + ((((lc >> 1) ^ ((lr >> 1) & 3)) << 2) | ((lc & 1) << 1));  // in 4-byte words
        auto split = [&](const f32x4& v, uint2& hi, uint2& lo) {
            const _Float16 h0 = (_Float16)v[0], h1 = (_Float16)v[1], h2 = (_Float16)v[2], h3 = (_Float16)v[3];
            hi.x = __builtin_bit_cast(unsigned, f16x2{h0, h1});
            hi.y = __builtin_bit_cast(unsigned, f16x2{h2, h3});
            lo.x = H16<f16_t>::pack2((v[0] - (float)h0) * 2048.f, (v[1] - (float)h1) * 2048.f);
            lo.y = H16<f16_t>::pack2((v[2] - (float)h2) * 2048.f, (v[3] - (float)h3) * 2048.f);
        };
        auto swrite = [&](int buf) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (X3) {
                    uint2 hi, lo;
                    split(oka[i] ? ra[i] : z, hi, lo);
                    *(uint2*)&lds[buf][0][xpos + 32 * i * 16] = hi;
                    *(uint2*)&lds[buf][0][2048 + xpos + 32 * i * 16] = lo;
                    if constexpr (X3 == 2) {  // weights pre-split (vitseg_cast_params_split): 16 B = 4 hi halves | 4 lo halves
                        const f32x4 w = okb[i] ? rb[i] : z;
                        hi = uint2{__float_as_uint(w[0]), __float_as_uint(w[1])};
                        lo = uint2{__float_as_uint(w[2]), __float_as_uint(w[3])};
                    } else {
                        split(okb[i] ? rb[i] : z, hi, lo);
                    }
                    *(uint2*)&lds[buf][1][xpos + 32 * i * 16] = hi;
                    *(uint2*)&lds[buf][1][2048 + xpos + 32 * i * 16] = lo;
                } else {
                    *(f32x4*)&lds[buf][0][TA ? tpos + 8 * i * BM : wpos + 32 * i * BK] = oka[i] ? ra[i] : z;
                    *(f32x4*)&lds[buf][1][TB ? tpos + 8 * i * BN : wpos + 32 * i * BK] = okb[i] ? rb[i] : z;
                }
            }
        };
        if constexpr (X3) {
            // two 16-k steps per staged tile; per step and operand one hi and one lo fragment (16 B = 8 halves)
            f32x4 ah[2][MI], al[2][MI], bh[2][NI], bl[2][NI];
            const int xsw = (li >> 1) & 3;
            auto lfragx = [&](int buf, int st, int slot) {
                const int ch = ((2 * st + lh) ^ xsw) << 2;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const float* r = &lds[buf][0][(a_row0 + li + mi * 32) * 16 + ch];
                    ah[slot][mi] = *(const f32x4*)r;
                    al[slot][mi] = *(const f32x4*)(r + 2048);
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const float* r = &lds[buf][1][(b_col0 + li + ni * 32) * 16 + ch];
                    bh[slot][ni] = *(const f32x4*)r;
                    bl[slot][ni] = *(const f32x4*)(r + 2048);
                }
            };
            auto mfmax = [&](int slot) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[slot][mi]), xl = __builtin_bit_cast(bf16x8, al[slot][mi]);
                        const bf16x8 yh = __builtin_bit_cast(bf16x8, bh[slot][ni]), yl = __builtin_bit_cast(bf16x8, bl[slot][ni]);
                        acc[mi][ni] = H16<f16_t>::mfma(xh, yh, acc[mi][ni]);
                        acc1[mi][ni] = H16<f16_t>::mfma(xl, yh, acc1[mi][ni]);
                        acc1[mi][ni] = H16<f16_t>::mfma(xh, yl, acc1[mi][ni]);
                    }
            };
            gload(0);
            swrite(0);
            __syncthreads();
            lfragx(0, 0, 0);
            for (int kt = 0; kt < KT; ++kt) {
                const int buf = kt & 1;
                const int kn = min(kt + 1, KT - 1);
                gload(kn);
                lfragx(buf, 1, 1);
                __builtin_amdgcn_sched_barrier(0);
                mfmax(0);
                __builtin_amdgcn_sched_barrier(0);
                swrite(buf ^ 1);   // split + store of the next tile (VALU) in the shadow of the MFMAs around it
                __syncthreads();
                lfragx(buf ^ 1, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                mfmax(1);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = fmaf(acc1[mi][ni][r], 1.0f / 2048.0f, acc[mi][ni][r]);
        } else {
        gload(0);
        swrite(0);
        __syncthreads();
        lfrag(0, 0, 0);
        for (int kt = 0; kt < KT; ++kt) {
            const int buf = kt & 1;
            const int kn = min(kt + 1, KT - 1);  // the last step re-stages its own tile: keeps the body branch-free
            // group 0: next tile's global loads are issued here and stay in flight for ~2 groups
            gload(kn);
            lfrag(buf, 1, 1);
            __builtin_amdgcn_sched_barrier(0);  // pin: hipcc otherwise sinks the loads down to their use
            mfma_group(0, 0, 4);
            // group 1
            lfrag(buf, 2, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(1, 0, 4);
            // group 2: the staged tile is zero-masked and written to the idle LDS buffer mid-group
            lfrag(buf, 3, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(0, 0, 2);
            __builtin_amdgcn_sched_barrier(0);
            swrite(buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(0, 2, 4);
            // group 3: one barrier hands the buffers over, then the next tile's first fragments are read
            __syncthreads();
            lfrag(buf ^ 1, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(1, 0, 4);
        }
        }
    } else {
        // ---- bf16: global -> LDS directly (global_load_lds_dwordx4), no staging registers, no VALU ----
        // One wave instruction fills 1 KiB = 8 staged rows, lane l -> row l>>3, LDS chunk position l&7.
        // The LDS image is lane-linear, so the XOR swizzle is applied to the per-lane SOURCE chunk
        // (position p of row r holds logical chunk p ^ ((r>>1)&7)) and again on the fragment reads.
        // Rows beyond M / N are clamped (their results are never stored); 3x3 taps outside the image
        // read a zero page.  Requires K % 64 == 0 (checked by the launcher).
        const int wave = tid >> 6;
        const T* asrc[4];
        const T* wsrc[4];
        int ay[4], ax[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave * 4 + i) * 8 + (lane >> 3);
            const int cpos = (lane & 7) ^ ((row >> 1) & 7);
            const int m = min(m0 + row, p.M - 1), n = min(n0 + row, p.N - 1);
            wsrc[i] = (const T*)p.W + (size_t)n * p.ldw + cpos * CE;
            if (AMODE == A_PLAIN) {
                asrc[i] = (const T*)p.A + (size_t)m * p.lda + cpos * CE;
                ay[i] = ax[i] = 0;
            } else {  // A_CONV3
                const int bimg = m / p.Np, t = m - bimg * p.Np;
                ay[i] = t / p.g;
                ax[i] = t - ay[i] * p.g;
                asrc[i] = (const T*)p.A + (size_t)m * p.D + cpos * CE;
            }
        }
        auto issue = [&](int kt, int buf) {
            const int k0 = (kt + kt0) * BKE;
            int tap = 0, d0 = k0, ky = 1, kx = 1;
            if (AMODE == A_CONV3) {
                tap = k0 / p.D;  // a 64-wide K step lies inside one tap (D % 64 == 0)
                d0 = k0 - tap * p.D;
                ky = tap / 3;
                kx = tap - ky * 3;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const T* ga;
                if (AMODE == A_PLAIN) {
                    ga = asrc[i] + k0;
                } else {
                    const int yy = ay[i] + ky - 1, xx = ax[i] + kx - 1;
                    const bool in = (unsigned)yy < (unsigned)p.g && (unsigned)xx < (unsigned)p.g;
                    ga = in ? asrc[i] + ((ptrdiff_t)(ky - 1) * p.g + (kx - 1)) * p.D + d0
                            : (const T*)p.zeros + ((lane & 7) ^ 0) * CE;
                }
                const int lrow = (wave * 4 + i) * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga,
                                                 (__attribute__((address_space(3))) void*)&lds[buf][0][lrow * BK], 16,
                                                 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + k0),
                                                 (__attribute__((address_space(3))) void*)&lds[buf][1][lrow * BK], 16,
                                                 0, 0);
            }
        };
        issue(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (KT > 1) issue(1, 1);
        lfrag(0, 0, 0);
        for (int kt = 0; kt < KT; ++kt) {
            const int buf = kt & 1;
            lfrag(buf, 1, 1);
            mfma_group(0, 0, 4);
            lfrag(buf, 2, 0);
            mfma_group(1, 0, 4);
            lfrag(buf, 3, 1);
            mfma_group(0, 0, 4);
            // tile kt+1 (issued one K step ago) must have landed for every wave, and every wave's
            // reads of this buffer must be complete before it is refilled
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < KT) issue(kt + 2, buf);
            lfrag(buf ^ 1, 0, 0);
            mfma_group(1, 0, 4);
        }
    }

    // ---- epilogue, staged through LDS so that global traffic is whole rows ----
    // acc reg r of lane (li, lh) = C[row (r&3) + 8 (r>>2) + 4 lh][col li]: one column per lane, which
    // would mean 64 scattered 2/4-byte stores per lane.  Each wave instead parks its sub-tile in its own
    // 16 KiB of the (now idle) operand buffers and re-reads it row-wise: 4 consecutive columns per
    // lane, so bias / residual / output move as 16-byte (fp32) or 8-byte (bf16) vectors along rows.
    __syncthreads();  // every wave is done with the operand tiles
    {
        const int wave = tid >> 6;
        float* wl = &lds[0][0][0] + wave * 4096;  // [64 rows][64 cols] fp32, wave-private
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    wl[(mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + ni * 32 + li] = acc[mi][ni][r];
        constexpr int LPR = NI * 8;        // lanes per row (4 columns each)
        constexpr int RPP = 64 / LPR;      // rows per pass
        const int rr = lane / LPR, c4 = (lane % LPR) * 4;
        const int gcol = n0 + b_col0 + c4;
        if (gcol < p.N) {
            f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) bias4 = *(const f32x4*)(p.bias + gcol);
            OutT* C = (OutT*)p.C + (size_t)blockIdx.y * p.split_stride;
            // all LDS reads (and residual / position loads) first, the stores last: in a kernel that contains
            // LDS-DMA hipcc waits vmcnt(0) before every use of a ds_read result, which would otherwise
            // serialise the 16 row stores one memory round trip at a time
            constexpr int NPS = MI * 32 / RPP;
            f32x4 v[NPS], extra[NPS];
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                const int row = ps * RPP + rr;
                const int grow = min(m0 + a_row0 + row, p.M - 1);
                v[ps] = *(const f32x4*)&wl[row * 64 + c4];
                if (EPI == EPI_RESADD || (EPI == EPI_DGELU && sizeof(T) == 4))
                    extra[ps] = *(const f32x4*)(p.R + (size_t)grow * p.ldc + gcol);
                if constexpr (EPI == EPI_DGELU && sizeof(T) == 2) {  // 16-bit training: R = the saved gelu'(u), 16-bit
                    const uint2 u = *(const uint2*)((const T*)p.R + (size_t)grow * p.ldc + gcol);
                    extra[ps][0] = H16<T>::lo(u.x);
                    extra[ps][1] = H16<T>::hi(u.x);
                    extra[ps][2] = H16<T>::lo(u.y);
                    extra[ps][3] = H16<T>::hi(u.y);
                }
                if (EPI == EPI_POS) extra[ps] = *(const f32x4*)(p.R + (size_t)(1 + grow % p.Np) * p.N + gcol);
            }
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                const int grow = m0 + a_row0 + ps * RPP + rr;
                if (grow >= p.M) continue;
                const size_t o = (size_t)grow * p.ldc + gcol;
                f32x4 aux4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = v[ps][e] + bias4[e];
                    // saved for the backward: fp32 keeps the pre-activation u, the 16-bit path keeps gelu'(u) itself
                    // (the backward epilogue is then one multiply instead of an erf + exp per element)
                    if (EPI == EPI_GELU) {
                        if (sizeof(T) == 4) {
                            if (p.aux) aux4[e] = x;
                            x = gelu_erf(x);
                        } else if (p.aux) {
                            { const GeluPair gp = gelu_erf_pair_fast(x); x = gp.g; aux4[e] = gp.d; }
                        } else {
                            x = gelu_erf_fast(x);
                        }
                    }
                    if (EPI == EPI_RELU) x = fmaxf(x, 0.f);
                    if (EPI == EPI_RESADD && p.drop.thresh)
                        x = drop_keep(drop_key(p.drop.seed, p.drop.stream, grow + p.row_base), gcol + e, p.drop.thresh)
                                ? x * p.drop.scale : 0.f;
                    if (EPI == EPI_RESADD || EPI == EPI_POS) x = extra[ps][e] + x;
                    if (EPI == EPI_DGELU) x *= sizeof(T) == 4 ? gelu_erf_grad(extra[ps][e]) : extra[ps][e];
                    v[ps][e] = x;
                }
                if (EPI == EPI_GELU && p.aux) {  // pre-activation, saved for the backward pass
                    if constexpr (sizeof(OutT) == 4) {
                        *(f32x4*)((float*)p.aux + o) = aux4;
                    } else {
                        uint2 h;
                        h.x = H16<OutT>::pack2(aux4[0], aux4[1]);
                        h.y = H16<OutT>::pack2(aux4[2], aux4[3]);
                        *(uint2*)((OutT*)p.aux + o) = h;
                    }
                }
                if constexpr (sizeof(OutT) == 4) {
                    *(f32x4*)(C + o) = v[ps];
                } else {
                    uint2 h;
                    h.x = H16<OutT>::pack2(v[ps][0], v[ps][1]);
                    h.y = H16<OutT>::pack2(v[ps][2], v[ps][3]);
                    *(uint2*)(C + o) = h;
                }
            }
        }
    }
}

template <typename T, typename OutT, int AMODE, int EPI, int TA = 0, int TB = 0, int X3 = 0>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][BM * BKF];  // [buffer][A|W][row*32 + swizzled chunk]

    const int wave = threadIdx.x >> 6;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    // Logical tile order (each XCD runs a contiguous piece of it, xcd_remap):
    //  1. a thin last row tile (the CLS rows) goes first;
    //  2. the rest is walked in column groups of GN tiles, row panels marching inside a group.  The ~64
    //     blocks resident on an XCD then form an 8x8 patch of tiles (each operand slice shared 8x) and a
    //     group's W panel (GN*128 rows of K) stays in the 4 MiB L2 while the A panels stream past it.
    //     (n-fastest order measured 62 % L2 hit rate / 16x over-fetch on the N = 3072 GEMM.)
    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int GN = p.gn ? p.gn : 8;  // 4/8/16 time within 1.5 % of each other (tools/gn_sweep.sh); 8 fetches least
    const bool thin_last = p.M - (tiles_m - 1) * BM <= 64 && tiles_m > 1;
    int tile_m, tile_n;
    if (thin_last && t < tiles_n) {
        tile_m = tiles_m - 1;
        tile_n = t;
    } else {
        const int rows = thin_last ? tiles_m - 1 : tiles_m;
        if (thin_last) t -= tiles_n;
        const int gsz = rows * GN, ngroups = (tiles_n + GN - 1) / GN;
        const int grp = min(t / gsz, ngroups - 1);
        const int rem = t - grp * gsz;
        const int gcols = min(GN, tiles_n - grp * GN);
        tile_m = rem / gcols;
        tile_n = grp * GN + rem - tile_m * gcols;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int rows_valid = p.M - m0;
    if (rows_valid <= 32)
        gemm_tile<T, OutT, AMODE, EPI, 1, 1, TA, TB, X3>(p, lds, m0, n0, 0, wave * 32);
    else if (rows_valid <= 64)
        gemm_tile<T, OutT, AMODE, EPI, 2, 1, TA, TB, X3>(p, lds, m0, n0, 0, wave * 32);
    else
        gemm_tile<T, OutT, AMODE, EPI, 2, 2, TA, TB, X3>(p, lds, m0, n0, (wave >> 1) * 64, (wave & 1) * 64);
}

inline int env_gn() {  // experiments only: VITSEG_GN=<n> forces the column-group width of the tile order
    return (int)opt(OPT_GN);
}

template <typename T, typename OutT, int AMODE, int EPI, int TA = 0, int TB = 0, int X3 = 0>
int launch_one(GemmArgs a, hipStream_t s) {
    if (a.ldw == 0) a.ldw = TB ? a.N : a.K;
    if (!a.gn) a.gn = env_gn();
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const int splits = a.splitk > 1 ? a.splitk : 1;
    hipLaunchKernelGGL((gemm_kernel<T, OutT, AMODE, EPI, TA, TB, X3>), dim3(tiles, splits), dim3(256), 0, s, a);
    VITSEG_LAUNCH_CHECK("gemm");
    return VITSEG_OK;
}

}  // namespace

namespace {

// out[r][c4] = epi(sum_s partial[s][r][c4] + bias): the K slices of the CLS rows, summed in slice order (deterministic).
// The epilogue is the tile kernels' one, including the training forms: hidden dropout on the residual branch (RESADD),
// the saved GELU derivative (GELU with aux, 16-bit) and the multiplication by it (DGELU, R = 16-bit derivative rows).
template <int EPI, typename OutT>
__global__ __launch_bounds__(256) void thin_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                                          const float* __restrict__ R, OutT* __restrict__ C, int rows, int N,
                                                          int ldc, int splits, OutT* __restrict__ aux, DropArgs drop,
                                                          unsigned row0) {
    const int n4 = N >> 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * n4) return;
    const int r = i / n4, c = (i - r * n4) * 4;
    const size_t slab = (size_t)rows * N;
    f32x4 acc = *(const f32x4*)(partial + (size_t)r * N + c);
    for (int sIdx = 1; sIdx < splits; ++sIdx) {
        const f32x4 v = *(const f32x4*)(partial + sIdx * slab + (size_t)r * N + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += v[e];
    }
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (bias) b4 = *(const f32x4*)(bias + c);
    f32x4 res = {0.f, 0.f, 0.f, 0.f};
    if (EPI == EPI_RESADD) res = *(const f32x4*)(R + (size_t)r * ldc + c);
    if constexpr (EPI == EPI_DGELU) {
        const uint2 d = *(const uint2*)((const OutT*)R + (size_t)r * ldc + c);
        res[0] = H16<OutT>::lo(d.x);
        res[1] = H16<OutT>::hi(d.x);
        res[2] = H16<OutT>::lo(d.y);
        res[3] = H16<OutT>::hi(d.y);
    }
    f32x4 der = {0.f, 0.f, 0.f, 0.f};
    const unsigned key = drop.thresh ? drop_key(drop.seed, drop.stream, row0 + (unsigned)r) : 0u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = acc[e] + b4[e];
        if (EPI == EPI_GELU) {
            // as the tile kernels of that format
            if (sizeof(OutT) == 4) x = gelu_erf(x);
            else if (aux) { const GeluPair gp = gelu_erf_pair_fast(x); x = gp.g; der[e] = gp.d; }
            else x = gelu_erf_fast(x);
        }
        if (EPI == EPI_RESADD) {
            if (drop.thresh) x = drop_keep(key, (unsigned)(c + e), drop.thresh) ? x * drop.scale : 0.f;
            x = res[e] + x;
        }
        if (EPI == EPI_DGELU) x *= res[e];
        acc[e] = x;
    }
    if constexpr (sizeof(OutT) == 4) {
        *(f32x4*)(C + (size_t)r * ldc + c) = acc;
    } else {
        uint2 h;
        h.x = H16<OutT>::pack2(acc[0], acc[1]);
        h.y = H16<OutT>::pack2(acc[2], acc[3]);
        *(uint2*)(C + (size_t)r * ldc + c) = h;
        if (EPI == EPI_GELU && aux) {
            h.x = H16<OutT>::pack2(der[0], der[1]);
            h.y = H16<OutT>::pack2(der[2], der[3]);
            *(uint2*)(aux + (size_t)r * ldc + c) = h;
        }
    }
}

template <typename OutT>
int launch_thin_reduce(const GemmArgs& a, int epi, int splits, hipStream_t s) {
    const int rows = a.thin_rows, body = a.M - rows;
    const int blocks = (rows * (a.N / 4) + 255) / 256;
    const float* R = nullptr;
    if (a.R) R = epi == EPI_DGELU ? (const float*)((const OutT*)a.R + (size_t)body * a.ldc) : a.R + (size_t)body * a.ldc;
    OutT* C = (OutT*)a.C + (size_t)body * a.ldc;
    OutT* aux = a.aux ? (OutT*)a.aux + (size_t)body * a.ldc : nullptr;
    const unsigned row0 = (unsigned)(a.row_base + body);
#define VITSEG_THIN(E)                                                                                                 \
    hipLaunchKernelGGL((thin_reduce_kernel<E, OutT>), dim3(blocks), dim3(256), 0, s, a.thin_scratch, a.bias, R, C, rows, \
                       a.N, a.ldc, splits, aux, a.drop, row0)
    switch (epi) {
        case EPI_BIAS: VITSEG_THIN(EPI_BIAS); break;
        case EPI_GELU: VITSEG_THIN(EPI_GELU); break;
        case EPI_DGELU:
            if constexpr (sizeof(OutT) == 4) {
                set_error("thin_reduce: dGELU epilogue is 16-bit only");
                return VITSEG_EINVAL;
            } else {
                VITSEG_THIN(EPI_DGELU);
            }
            break;
        default: VITSEG_THIN(EPI_RESADD);
    }
#undef VITSEG_THIN
    VITSEG_LAUNCH_CHECK("thin_reduce");
    return VITSEG_OK;
}

// true when the trailing rows of `a` should go through the split-K side launch (see GemmArgs::thin_scratch)
// > 0: the whole (small) GEMM goes through K slices + the reducing epilogue kernel (kernels.hpp whole_split)
int whole_split_applies(const GemmArgs& a, int epi, int kstep) {
    const int sp = whole_split(a.M, a.N, a.K, kstep);
    const bool ok = sp && a.thin_scratch && (size_t)sp * a.M * a.N <= a.thin_capacity && a.K % kstep == 0 && !a.drop.thresh &&
                    !a.aux && a.splitk <= 1 && a.ldc % 4 == 0 && (epi == EPI_BIAS || epi == EPI_GELU || epi == EPI_RESADD);
    return ok ? sp : 0;
}

// h16: the 16-bit path's reducing epilogue also covers dropout, the saved GELU derivative and dGELU (training)
bool thin_split_applies(const GemmArgs& a, int epi, bool h16 = false) {
    return a.thin_scratch && a.thin_rows > 0 && a.thin_rows <= THIN_MAX_ROWS && a.M > a.thin_rows &&
           (a.M - a.thin_rows) % BM == 0 && a.K >= 256 && a.K % 32 == 0 && (h16 || (!a.drop.thresh && !a.aux)) &&
           a.splitk <= 1 && a.ldc % 4 == 0 &&
           (epi == EPI_BIAS || epi == EPI_GELU || epi == EPI_RESADD || (h16 && epi == EPI_DGELU));
}

template <int X3>
int launch_thin_rows(const GemmArgs& a, int epi, hipStream_t s, int splits = 0) {
    const int rows = a.thin_rows, body = a.M - rows;
    if (!splits) {
        splits = a.K / 32 / 4;  // >= 4 K steps per slice
        if (splits > THIN_MAX_SPLITS) splits = THIN_MAX_SPLITS;
    }
    GemmArgs t = a;
    t.A = (const float*)a.A + (size_t)body * a.lda;
    t.M = rows;
    t.bias = nullptr;
    t.R = nullptr;
    t.C = a.thin_scratch;
    t.ldc = a.N;
    t.splitk = splits;
    t.split_stride = (size_t)rows * a.N;
    t.thin_scratch = nullptr;
    if (int rc = launch_one<float, float, A_PLAIN, EPI_BIAS, 0, 0, X3>(t, s)) return rc;
    return launch_thin_reduce<float>(a, epi, splits, s);
}

// 16-bit operands: K slices of 64-element steps, fp32 partials, output in the consumer's format
template <typename T>
int launch_thin_rows_h16(const GemmArgs& a, int epi, hipStream_t s, int splits = 0) {
    const int rows = a.thin_rows, body = a.M - rows;
    if (!splits) {
        splits = a.K / 64 / 4;
        if (splits > THIN_MAX_SPLITS) splits = THIN_MAX_SPLITS;
        if (splits < 1) splits = 1;
    }
    GemmArgs t = a;
    t.A = (const T*)a.A + (size_t)body * a.lda;
    t.M = rows;
    t.bias = nullptr;
    t.R = nullptr;
    t.C = a.thin_scratch;
    t.ldc = a.N;
    t.splitk = splits;
    t.split_stride = (size_t)rows * a.N;
    t.thin_scratch = nullptr;
    t.aux = nullptr;      // the training epilogues run in the reducing kernel
    t.drop = DropArgs{};
    if (int rc = launch_one<T, float, A_PLAIN, EPI_BIAS>(t, s)) return rc;
    return epi == EPI_RESADD ? launch_thin_reduce<float>(a, epi, splits, s) : launch_thin_reduce<T>(a, epi, splits, s);
}

}  // namespace

int launch_gemm_f32(const GemmArgs& a_in, int amode, int epi, hipStream_t s, int x3) {
    GemmArgs a = a_in;
    if (amode == A_PLAIN) {
        if (const int sp = whole_split_applies(a_in, epi, 32)) {  // small batch: every row through K slices
            GemmArgs w = a_in;
            w.thin_rows = a_in.M;
            return x3 == 2 ? launch_thin_rows<2>(w, epi, s, sp) : x3 == 1 ? launch_thin_rows<1>(w, epi, s, sp)
                                                                        : launch_thin_rows<0>(w, epi, s, sp);
        }
    }
    if (amode == A_PLAIN && thin_split_applies(a_in, epi)) {
        // the CLS rows first (tiny, split over K), then the whole-tile body: an exact number of rounds of blocks
        if (int rc = x3 == 2 ? launch_thin_rows<2>(a_in, epi, s) : x3 == 1 ? launch_thin_rows<1>(a_in, epi, s)
                                                                           : launch_thin_rows<0>(a_in, epi, s))
            return rc;
        a.M = a_in.M - a_in.thin_rows;
    }
    a.thin_scratch = nullptr;
    VITSEG_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0 && a.K % 4 == 0, VITSEG_EINVAL, "gemm_f32: bad M/N/K %d %d %d", a.M,
                     a.N, a.K);
    VITSEG_CHECK_ARG(a.N % 4 == 0 && a.ldc % 4 == 0, VITSEG_ESHAPE, "gemm: N=%d and ldc=%d must be multiples of 4", a.N,
                     a.ldc);
    // x3 = 1: both fp32 operands are split into half pairs on the fly; 2: W is the pre-split shadow arena
    // (vitseg_cast_params_split), only A is split in the kernel.  3 fp16 MFMAs per product (see gemm_tile, X3)
    if (x3 == 2) {
        if (amode == A_PLAIN) {
            VITSEG_CHECK_ARG(a.lda % 4 == 0, VITSEG_EINVAL, "gemm_f32: lda %% 4");
            switch (epi) {
                case EPI_BIAS: return launch_one<float, float, A_PLAIN, EPI_BIAS, 0, 0, 2>(a, s);
                case EPI_GELU: return launch_one<float, float, A_PLAIN, EPI_GELU, 0, 0, 2>(a, s);
                case EPI_RESADD: return launch_one<float, float, A_PLAIN, EPI_RESADD, 0, 0, 2>(a, s);
            }
        } else if (amode == A_PATCH && epi == EPI_POS) {
            VITSEG_CHECK_ARG(a.P % 4 == 0, VITSEG_ESHAPE, "patch size must be a multiple of 4");
            return launch_one<float, float, A_PATCH, EPI_POS, 0, 0, 2>(a, s);
        } else if (amode == A_CONV3 && epi == EPI_RELU) {
            VITSEG_CHECK_ARG(a.D % 4 == 0, VITSEG_ESHAPE, "hidden size must be a multiple of 4");
            return launch_one<float, float, A_CONV3, EPI_RELU, 0, 0, 2>(a, s);
        }
        set_error("gemm_f32 (x3, split W): unsupported amode/epilogue %d/%d", amode, epi);
        return VITSEG_EINVAL;
    }
    if (x3 == 1) {  // fp32 operands split into half pairs on the fly, 3 fp16 MFMAs per product (see gemm_tile, X3)
        if (amode == A_PLAIN) {
            VITSEG_CHECK_ARG(a.lda % 4 == 0, VITSEG_EINVAL, "gemm_f32: lda %% 4");
            switch (epi) {
                case EPI_BIAS: return launch_one<float, float, A_PLAIN, EPI_BIAS, 0, 0, 1>(a, s);
                case EPI_GELU: return launch_one<float, float, A_PLAIN, EPI_GELU, 0, 0, 1>(a, s);
                case EPI_RESADD: return launch_one<float, float, A_PLAIN, EPI_RESADD, 0, 0, 1>(a, s);
            }
        } else if (amode == A_PATCH && epi == EPI_POS) {
            VITSEG_CHECK_ARG(a.P % 4 == 0, VITSEG_ESHAPE, "patch size must be a multiple of 4");
            return launch_one<float, float, A_PATCH, EPI_POS, 0, 0, 1>(a, s);
        } else if (amode == A_CONV3 && epi == EPI_RELU) {
            VITSEG_CHECK_ARG(a.D % 4 == 0, VITSEG_ESHAPE, "hidden size must be a multiple of 4");
            return launch_one<float, float, A_CONV3, EPI_RELU, 0, 0, 1>(a, s);
        }
        set_error("gemm_f32 (x3): unsupported amode/epilogue %d/%d", amode, epi);
        return VITSEG_EINVAL;
    }
    if (amode == A_PLAIN) {
        VITSEG_CHECK_ARG(a.lda % 4 == 0, VITSEG_EINVAL, "gemm_f32: lda %% 4");
        if (gemm_f32p_applies(a, epi)) return launch_gemm_f32p(a, epi, s);   // large shapes: persistent 256x128 kernel
        switch (epi) {
            case EPI_BIAS: return launch_one<float, float, A_PLAIN, EPI_BIAS>(a, s);
            case EPI_GELU: return launch_one<float, float, A_PLAIN, EPI_GELU>(a, s);
            case EPI_RESADD: return launch_one<float, float, A_PLAIN, EPI_RESADD>(a, s);
            case EPI_RELU: return launch_one<float, float, A_PLAIN, EPI_RELU>(a, s);
        }
    } else if (amode == A_PATCH && epi == EPI_POS) {
        VITSEG_CHECK_ARG(a.P % 4 == 0, VITSEG_ESHAPE, "patch size must be a multiple of 4");
        return launch_one<float, float, A_PATCH, EPI_POS>(a, s);
    } else if (amode == A_CONV3 && epi == EPI_RELU) {
        VITSEG_CHECK_ARG(a.D % 4 == 0, VITSEG_ESHAPE, "hidden size must be a multiple of 4");
        return launch_one<float, float, A_CONV3, EPI_RELU>(a, s);
    }
    set_error("gemm_f32: unsupported amode/epilogue %d/%d", amode, epi);
    return VITSEG_EINVAL;
}

// Backward GEMMs in fp32 (no transposed copies, see gemm_tile):
//   dgrad  dX[M,K]  = dY[M,N] . W[N,K]        -> A N-form, B T-form;  epi: plain or * gelu'(R)
//   wgrad  dW[N,K]  = dY[M,N]^T . X[M,K]      -> A T-form, B T-form;  plain
// In GemmArgs terms M/N are always the OUTPUT rows/cols and K the reduction length.
// =====================================================================================================
// bf16 GEMM, large-M variant: block 256(M) x 128(N), BK = 64, 8 waves as 4(M) x 2(N) (64x64 per wave,
// the same per-wave work as the 128x128 kernel), ONE block per CU, 3-stage LDS ring (3 x 48 KiB).
//
// Why a second shape: at bf16 rates the 128x128 / 2-blocks-per-CU kernel stages 64 B/clk/CU, i.e. it needs
// the whole L2 bandwidth of the chip (34 TB/s) at full MFMA rate, and its single-tile prefetch exposes the
// L2/MALL latency (measured: 30 % MFMA busy, 790 TF/s asymptote, 1260 TF/s with the loads removed).  This
// tile stages 25 % fewer bytes per FLOP and keeps TWO K steps in flight: the global_load_lds of step kt+2
// are issued right after the barrier that publishes step kt and are only waited for (counted
// `s_waitcnt vmcnt(6)`: the 6 DMA pieces of step kt+1 may stay outstanding) two compute phases later.
// M = B*Np + B: the B*Np patch rows are whole 256-row tiles at 512x512; the CLS rows make one thin tile
// that is scheduled first and in which only the first wave row computes.
constexpr int LBM = 256;

// LBN = 128: waves 4(M) x 2(N), 64x64 per wave, 3-stage ring (3 x 48 KiB), 85 FLOP per staged byte.
// LBN = 256: waves 2(M) x 4(N), 128x64 per wave (128 accumulator registers), 2-stage ring (2 x 64 KiB),
//            128 FLOP per staged byte -- half the L2->LDS traffic of the 128x128 kernel, which is what bounds it.
template <typename T, typename OutT, int AMODE, int EPI, int LBN>
__global__ __launch_bounds__(512) void gemm_bf16_large_kernel(const GemmArgs p) {
    constexpr int CE = 8, BKE = 64, BK = BKF;
    constexpr int STAGES = LBN == 128 ? 3 : 2;
    constexpr int WAVES = 8;
    constexpr int MI = LBN == 128 ? 2 : 4, NI = 2;      // 32x32 MFMA tiles per wave
    constexpr int WROWS = MI * 32;                       // rows per wave
    constexpr int APW = LBM / 8 / WAVES;                 // A DMA pieces per wave (8 rows each)
    constexpr int WPW = LBN / 8 / WAVES;                 // W DMA pieces per wave
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];  // [stage][A 256 rows | W LBN rows][32 words]
    auto stageA = [&](int st) { return lds_raw + st * (LBM + LBN) * BK; };
    auto stageW = [&](int st) { return lds_raw + st * (LBM + LBN) * BK + LBM * BK; };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar branches below
    const int wm = LBN == 128 ? wave >> 1 : wave >> 2;
    const int wn = LBN == 128 ? wave & 1 : wave & 3;
    const int tiles_n = (p.N + LBN - 1) / LBN, tiles_m = (p.M + LBM - 1) / LBM;
    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int GN = p.gn ? p.gn : (LBN == 128 ? ((size_t)p.K * sizeof(T) <= 2048 ? 8 : 4) : 4);
    const bool thin_last = p.M - (tiles_m - 1) * LBM <= 64 && tiles_m > 1;
    int tile_m, tile_n;
    if (thin_last && t < tiles_n) {
        tile_m = tiles_m - 1;
        tile_n = t;
    } else {
        const int rows = thin_last ? tiles_m - 1 : tiles_m;
        if (thin_last) t -= tiles_n;
        const int gsz = rows * GN, ngroups = (tiles_n + GN - 1) / GN;
        const int grp = min(t / gsz, ngroups - 1);
        const int rem = t - grp * gsz;
        const int gcols = min(GN, tiles_n - grp * GN);
        tile_m = rem / gcols;
        tile_n = grp * GN + rem - tile_m * gcols;
    }
    const int m0 = tile_m * LBM, n0 = tile_n * LBN;
    const bool computes = (p.M - m0 > 64) || wm == 0;  // thin tile: only the first 64 rows exist

    // ---- DMA assignment: per K step 32 A pieces + LBN/8 W pieces of 1 KiB (8 rows each) ----
    const T* asrc[APW];
    const T* wsrc[WPW];
    int ay[APW], ax[APW];
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int row = (wave * APW + i) * 8 + (lane >> 3);
        const int cpos = (lane & 7) ^ ((row >> 1) & 7);
        const int m = min(m0 + row, p.M - 1);
        if (AMODE == A_PLAIN) {
            asrc[i] = (const T*)p.A + (size_t)m * p.lda + cpos * CE;
            ay[i] = ax[i] = 0;
        } else {
            const int bimg = m / p.Np, tt = m - bimg * p.Np;
            ay[i] = tt / p.g;
            ax[i] = tt - ay[i] * p.g;
            asrc[i] = (const T*)p.A + (size_t)m * p.D + cpos * CE;
        }
    }
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int row = (wave * WPW + i) * 8 + (lane >> 3);
        const int cpos = (lane & 7) ^ ((row >> 1) & 7);
        wsrc[i] = (const T*)p.W + (size_t)min(n0 + row, p.N - 1) * p.ldw + cpos * CE;
    }
    auto issue = [&](int kt, int st) {
        const int k0 = kt * BKE;
        int d0 = k0, ky = 1, kx = 1;
        if (AMODE == A_CONV3) {
            const int tap = k0 / p.D;
            d0 = k0 - tap * p.D;
            ky = tap / 3;
            kx = tap - ky * 3;
        }
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const T* ga;
            if (AMODE == A_PLAIN) {
                ga = asrc[i] + k0;
            } else {
                const int yy = ay[i] + ky - 1, xx = ax[i] + kx - 1;
                const bool in = (unsigned)yy < (unsigned)p.g && (unsigned)xx < (unsigned)p.g;
                ga = in ? asrc[i] + ((ptrdiff_t)(ky - 1) * p.g + (kx - 1)) * p.D + d0 : (const T*)p.zeros + (lane & 7) * CE;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga,
                                             (__attribute__((address_space(3))) void*)(stageA(st) + (wave * APW + i) * 8 * BK),
                                             16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WPW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + k0),
                                             (__attribute__((address_space(3))) void*)(stageW(st) + (wave * WPW + i) * 8 * BK),
                                             16, 0, 0);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const int li = lane & 31, lh = lane >> 5;
    const int sw = (li >> 1) & 7;
    const int a_off = (wm * WROWS + li) * BK, b_off = (wn * NI * 32 + li) * BK;
    f32x4 a[2][MI], b[2][NI];
    auto lfrag = [&](int st, int j, int slot) {
        const int ch = (((2 * j + lh) ^ sw) << 2);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[slot][mi] = *(const f32x4*)&stageA(st)[a_off + mi * 32 * BK + ch];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[slot][ni] = *(const f32x4*)&stageW(st)[b_off + ni * 32 * BK + ch];
    };
    auto mfmas = [&](int slot) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = H16<T>::mfma(__builtin_bit_cast(bf16x8, a[slot][mi]),
                                           __builtin_bit_cast(bf16x8, b[slot][ni]), acc[mi][ni]);
    };

    const int KT = p.K / BKE;
    // Ring of STAGES buffers: while step kt is computed, steps kt+1 .. kt+STAGES-2 are in flight.  The loop is
    // rotated around the barrier (group 3 of step kt runs after the barrier that publishes step kt+1, under the
    // first fragment reads of step kt+1); sched_barrier pins "next group's ds_reads, then this group's MFMAs".
    issue(0, 0);
    if (STAGES == 3 && KT > 1) issue(1, 1);
    if (STAGES == 3 && KT > 1)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // 6 = DMA pieces per wave per step at LBN = 128
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (STAGES == 2 && KT > 1) issue(1, 1);
    if (STAGES == 3 && KT > 2) issue(2, 2);
    int st = 0;
    if (computes) lfrag(0, 0, 0);
    for (int kt = 0; kt < KT; ++kt) {
        const int stn = st == STAGES - 1 ? 0 : st + 1;
        if (computes) {
            lfrag(st, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(0);
            __builtin_amdgcn_sched_barrier(0);
            lfrag(st, 2, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(1);
            __builtin_amdgcn_sched_barrier(0);
            lfrag(st, 3, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kt + 1 < KT) {
            // step kt+1 has landed once only the pieces of later steps are outstanding (in-order retire);
            // lgkmcnt(0): this wave's reads of stage st are complete before anyone refills it
            if (STAGES == 3 && kt + 2 < KT)
                asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + STAGES < KT) issue(kt + STAGES, st);
            if (computes) {
                lfrag(stn, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (computes) mfmas(1);
        st = stn;
    }

    // ---- epilogue: per-wave LDS staging (16 KiB = 64 rows x 64 cols fp32 at a time), row-vector stores ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!computes) return;
    float* wl = lds_raw + wave * 4096;
    const int rr = lane >> 4, c4 = (lane & 15) * 4;
    OutT* C = (OutT*)p.C;
#pragma unroll
    for (int nh = 0; nh < NI / 2; ++nh) {   // 64-column halves of the wave tile
    const int gcol = n0 + wn * NI * 32 + nh * 64 + c4;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && gcol < p.N) bias4 = *(const f32x4*)(p.bias + gcol);
#pragma unroll
    for (int half = 0; half < MI / 2; ++half) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    wl[(mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + ni * 32 + li] = acc[half * 2 + mi][nh * 2 + ni][r];
        // all LDS reads (and residual loads) first, stores last: in a kernel that contains LDS-DMA hipcc waits
        // vmcnt(0) before every use of a ds_read result, which would serialise the stores one by one
        f32x4 v[16], extra[16];
#pragma unroll
        for (int ps = 0; ps < 16; ++ps) {
            const int row = ps * 4 + rr;
            v[ps] = *(const f32x4*)&wl[row * 64 + c4];
            if (EPI == EPI_RESADD) {
                const int grow = min(m0 + wm * WROWS + half * 64 + row, p.M - 1);
                extra[ps] = *(const f32x4*)(p.R + (size_t)grow * p.ldc + min(gcol, p.N - 4));
            }
            if (EPI == EPI_DGELU) {  // training: R = the saved 16-bit gelu'(pre-activation)
                const int grow = min(m0 + wm * WROWS + half * 64 + row, p.M - 1);
                const uint2 u = *(const uint2*)((const T*)p.R + (size_t)grow * p.ldc + min(gcol, p.N - 4));
                extra[ps][0] = H16<T>::lo(u.x);
                extra[ps][1] = H16<T>::hi(u.x);
                extra[ps][2] = H16<T>::lo(u.y);
                extra[ps][3] = H16<T>::hi(u.y);
            }
        }
        if (gcol < p.N) {
#pragma unroll
            for (int ps = 0; ps < 16; ++ps) {
                const int grow = m0 + wm * WROWS + half * 64 + ps * 4 + rr;
                const size_t o = (size_t)grow * p.ldc + gcol;
                f32x4 pre = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = v[ps][e] + bias4[e];
                    if (EPI == EPI_GELU) {
                        if (p.aux) { const GeluPair gp = gelu_erf_pair_fast(x); x = gp.g; pre[e] = gp.d; }   // pre: gelu'(u), saved for the backward
                        else x = gelu_erf_fast(x);
                    }
                    if (EPI == EPI_RELU) x = fmaxf(x, 0.f);
                    if (EPI == EPI_RESADD && p.drop.thresh)
                        x = drop_keep(drop_key(p.drop.seed, p.drop.stream, grow + p.row_base), gcol + e, p.drop.thresh)
                                ? x * p.drop.scale : 0.f;
                    if (EPI == EPI_RESADD) x = extra[ps][e] + x;
                    if (EPI == EPI_DGELU) x *= extra[ps][e];
                    v[ps][e] = x;
                }
                if (EPI == EPI_GELU && p.aux && grow < p.M) {  // training: keep gelu'(pre-activation) for the backward
                    uint2 h;
                    h.x = H16<T>::pack2(pre[0], pre[1]);
                    h.y = H16<T>::pack2(pre[2], pre[3]);
                    *(uint2*)((T*)p.aux + o) = h;
                }
                if (grow < p.M) {
                    if constexpr (sizeof(OutT) == 4) {
                        *(f32x4*)(C + o) = v[ps];
                    } else {
                        uint2 h;
                        h.x = H16<OutT>::pack2(v[ps][0], v[ps][1]);
                        h.y = H16<OutT>::pack2(v[ps][2], v[ps][3]);
                        *(uint2*)(C + o) = h;
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private buffer is reused by the next half
    }
    }
}

template <typename T, typename OutT, int AMODE, int EPI, int LBN = 128>
int launch_large(GemmArgs a, hipStream_t s) {
    if (a.ldw == 0) a.ldw = a.K;
    if (!a.gn) a.gn = env_gn();
    const int tiles = ((a.M + LBM - 1) / LBM) * ((a.N + LBN - 1) / LBN);
    const size_t smem = (size_t)(LBN == 128 ? 3 : 2) * (LBM + LBN) * BKF * sizeof(float);  // 144 / 128 KiB
    int dev = 0;
    static bool attr_set[64] = {};   // the attribute is per device
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_large_kernel<T, OutT, AMODE, EPI, LBN>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(gemm_bf16_large)");
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((gemm_bf16_large_kernel<T, OutT, AMODE, EPI, LBN>), dim3(tiles), dim3(512), smem, s, a);
    VITSEG_LAUNCH_CHECK("gemm_bf16_large");
    return VITSEG_OK;
}

// Split-K for the weight gradients: the output is only a weight matrix (36-144 tiles) while the reduction
// runs over every token row, so the K range is cut into `splits` slabs (one grid.y slice each, plain
// stores into partial[split][M][N]) that splitk_reduce_kernel sums in a fixed order (deterministic).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                            size_t n4, int splits) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 acc = ((const f32x4*)partial)[i];
        for (int sIdx = 1; sIdx < splits; ++sIdx) {
            const f32x4 v = ((const f32x4*)partial)[(size_t)sIdx * n4 + i];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v[e];
        }
        ((f32x4*)out)[i] = acc;
    }
}

int launch_splitk_reduce(const float* partial, float* out, size_t n4, int splits, hipStream_t s) {
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, partial, out, n4, splits);
    VITSEG_LAUNCH_CHECK("splitk_reduce");
    return VITSEG_OK;
}

int wgrad_splits(int M, int N, int K) {
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    int splits = 1024 / tiles;                       // at most 2 whole rounds of the 512 resident blocks (rounding
                                                     // UP gave 2.04-2.25 rounds: a third, nearly empty one)
    const int ksteps = (K + 31) / 32;
    if (splits > ksteps / 4) splits = ksteps / 4;    // keep >= 4 K steps per slab
    return splits < 1 ? 1 : splits;
}
size_t wgrad_scratch_floats(int M, int N, int K) { return (size_t)wgrad_splits(M, N, K) * M * N; }

// dW[M,N] = A^T . W (both T-form) with split-K through `scratch` (>= wgrad_scratch_floats floats; ldc == N)
int launch_wgrad_f32(GemmArgs a, float* scratch, hipStream_t s) {
    const int splits = wgrad_splits(a.M, a.N, a.K);
    if (splits <= 1) return launch_gemm_f32_bwd(a, A_PLAIN, 1, 1, EPI_BIAS, s);
    VITSEG_CHECK_ARG(a.ldc == a.N && scratch, VITSEG_EINVAL, "wgrad: split-K needs a dense output and scratch");
    float* out = (float*)a.C;
    a.C = scratch;
    a.splitk = splits;
    a.split_stride = (size_t)a.M * a.N;
    if (int rc = launch_gemm_f32_bwd(a, A_PLAIN, 1, 1, EPI_BIAS, s)) return rc;
    const size_t n4 = (size_t)a.M * a.N / 4;
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, scratch, out, n4, splits);
    VITSEG_LAUNCH_CHECK("splitk_reduce");
    return VITSEG_OK;
}

int launch_gemm_f32_bwd(const GemmArgs& a, int amode, int ta, int tb, int epi, hipStream_t s) {
    VITSEG_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, VITSEG_EINVAL, "gemm_bwd: bad M/N/K %d %d %d", a.M, a.N, a.K);
    VITSEG_CHECK_ARG(a.N % 4 == 0 && a.ldc % 4 == 0 && a.lda % 4 == 0 && a.ldw % 4 == 0, VITSEG_ESHAPE,
                     "gemm_bwd: leading dimensions must be multiples of 4");
    VITSEG_CHECK_ARG(ta || a.K % 4 == 0, VITSEG_ESHAPE, "gemm_bwd: K %% 4");
    VITSEG_CHECK_ARG(!ta || a.M % 4 == 0, VITSEG_ESHAPE, "gemm_bwd: M %% 4 for a T-form A");
    if (amode == A_PLAIN && !ta && tb && epi == EPI_BIAS) return launch_one<float, float, A_PLAIN, EPI_BIAS, 0, 1>(a, s);
    if (amode == A_PLAIN && !ta && tb && epi == EPI_DGELU) return launch_one<float, float, A_PLAIN, EPI_DGELU, 0, 1>(a, s);
    if (amode == A_PLAIN && ta && tb && epi == EPI_BIAS) return launch_one<float, float, A_PLAIN, EPI_BIAS, 1, 1>(a, s);
    if (amode == A_CONV3 && !ta && !tb && epi == EPI_BIAS) return launch_one<float, float, A_CONV3, EPI_BIAS, 0, 0>(a, s);
    set_error("gemm_bwd: unsupported combination amode %d ta %d tb %d epi %d", amode, ta, tb, epi);
    return VITSEG_EINVAL;
}

// bf16 operands (A and W), fp32 accumulate.  Output type follows the consumer: bf16 for tensors
// that feed the next MFMA (q|k|v, MLP hidden), fp32 for the residual stream and the head features.
template <typename T>
// p8_rows: set to the number of leading rows whose column-sum partials the 8-phase kernel wrote (GemmArgs::colsum_*)
int launch_gemm_h16_impl(const GemmArgs& a_in, int amode, int epi, hipStream_t s, int* p8_rows) {
    GemmArgs a = a_in;
    const bool want_cs = a_in.colsum_out && a_in.colsum_scratch && epi == EPI_DGELU;
    if (!want_cs) a.colsum_scratch = nullptr;
    if (amode == A_PLAIN) {
        if (const int sp = whole_split_applies(a_in, epi, 64)) {  // small batch: every row through K slices
            GemmArgs w = a_in;
            w.thin_rows = a_in.M;
            return launch_thin_rows_h16<T>(w, epi, s, sp);
        }
    }
    // A ragged last row tile (the CLS rows) is free when it fits into the persistent kernel's last, partly empty round
    // (batch 32: 384 + 3 tiles of the N = 768 linears over 256 CUs): no side launch, no reducing kernel.
    if (amode == A_PLAIN && a_in.M % 256 != 0 && a_in.M % 256 <= 128 && gemm_p8_applies(a_in, epi) &&
        gemm_p8_rounds(a_in.M, a_in.N) == gemm_p8_rounds(a_in.M - a_in.M % 256, a_in.N) && !opt(OPT_NO_RAGGED_P8)) {
        a.thin_scratch = nullptr;
        if (want_cs) *p8_rows = a.M;
        return launch_gemm_p8(a, epi, s, std::is_same<T, f16_t>::value);
    }
    if (amode == A_PLAIN && thin_split_applies(a_in, epi, true) && (a_in.M - a_in.thin_rows) % LBM == 0 && a_in.K % 64 == 0) {
        if (int rc = launch_thin_rows_h16<T>(a_in, epi, s)) return rc;  // CLS rows: split-K side launch (GemmArgs)
        a.M = a_in.M - a_in.thin_rows;
    }
    a.thin_scratch = nullptr;
    VITSEG_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, VITSEG_EINVAL, "gemm_bf16: bad M/N/K %d %d %d", a.M, a.N, a.K);
    VITSEG_CHECK_ARG(a.K % 64 == 0, VITSEG_ESHAPE, "gemm_bf16: K=%d must be a multiple of 64", a.K);
    VITSEG_CHECK_ARG(a.N % 4 == 0 && a.ldc % 4 == 0, VITSEG_ESHAPE, "gemm: N=%d and ldc=%d must be multiples of 4", a.N,
                     a.ldc);
    // the 256x128 / 3-stage kernel and the 128x128 / 2-stage kernel measure within a few % of each other on the
    // model's shapes (both ~790 TF/s asymptote); the large one is used where its deeper prefetch helps: long K
    if (amode == A_PLAIN && gemm_p8_applies(a, epi)) {
        // persistent 256x256 kernel (gemm_p8.hip).  A ragged last row tile would cost every CU a whole extra round
        // (M = 65 600: 257 x 12 tiles over 256 CUs = 13 rounds for 12.05 rounds of work), so up to 128 trailing rows
        // (the CLS rows) go through the 128x128 kernel as a second, tiny launch with the same epilogue.
        const int tail = a.M % 256;
        if (tail == 0 || tail > 128) {
            if (want_cs) *p8_rows = a.M;
            return launch_gemm_p8(a, epi, s, std::is_same<T, f16_t>::value);
        }
        GemmArgs body = a, t = a;
        body.M = a.M - tail;
        if (want_cs) *p8_rows = body.M;
        if (int rc = launch_gemm_p8(body, epi, s, std::is_same<T, f16_t>::value)) return rc;
        const size_t ro = (size_t)body.M;
        t.M = tail;
        t.row_base = a.row_base + body.M;
        t.A = (const T*)a.A + ro * a.lda;
        if (a.aux) t.aux = (T*)a.aux + ro * a.ldc;
        if (epi == EPI_RESADD) {
            t.R = a.R + ro * a.ldc;
            t.C = (float*)a.C + ro * a.ldc;
        } else {
            if (a.R) t.R = (const float*)((const T*)a.R + ro * a.ldc);   // EPI_DGELU: 16-bit operand
            t.C = (T*)a.C + ro * a.ldc;
        }
        switch (epi) {
            case EPI_BIAS: return launch_one<T, T, A_PLAIN, EPI_BIAS>(t, s);
            case EPI_GELU: return launch_one<T, T, A_PLAIN, EPI_GELU>(t, s);
            case EPI_DGELU: return launch_one<T, T, A_PLAIN, EPI_DGELU>(t, s);
            default: return launch_one<T, float, A_PLAIN, EPI_RESADD>(t, s);
        }
    }
    const long force = opt(OPT_BF16_TILES);  // 1 small / 2 large / 3 xl for experiments
    const bool xl = force ? force == 3 : (a.M >= 8192 && a.N >= 2048);
    const bool large = force ? force == 2 : (!xl && a.M >= 4096 && a.K >= 2048);
    if (amode == A_PLAIN) {
        VITSEG_CHECK_ARG(a.lda % 8 == 0, VITSEG_EINVAL, "gemm_bf16: lda %% 8");
        switch (epi) {
            case EPI_BIAS: return xl ? launch_large<T, T, A_PLAIN, EPI_BIAS, 256>(a, s)
                                  : large ? launch_large<T, T, A_PLAIN, EPI_BIAS>(a, s)
                                          : launch_one<T, T, A_PLAIN, EPI_BIAS>(a, s);
            case EPI_GELU: return xl ? launch_large<T, T, A_PLAIN, EPI_GELU, 256>(a, s)
                                  : large ? launch_large<T, T, A_PLAIN, EPI_GELU>(a, s)
                                          : launch_one<T, T, A_PLAIN, EPI_GELU>(a, s);
            case EPI_DGELU: return xl ? launch_large<T, T, A_PLAIN, EPI_DGELU, 256>(a, s)
                                   : large ? launch_large<T, T, A_PLAIN, EPI_DGELU>(a, s)
                                           : launch_one<T, T, A_PLAIN, EPI_DGELU>(a, s);
            case EPI_RESADD: return xl ? launch_large<T, float, A_PLAIN, EPI_RESADD, 256>(a, s)
                                    : large ? launch_large<T, float, A_PLAIN, EPI_RESADD>(a, s)
                                            : launch_one<T, float, A_PLAIN, EPI_RESADD>(a, s);
        }
    } else if (amode == A_CONV3 && epi == EPI_RELU) {
        VITSEG_CHECK_ARG(a.D % 64 == 0 && a.zeros, VITSEG_ESHAPE, "hidden size must be a multiple of 64");
        return large ? launch_large<T, float, A_CONV3, EPI_RELU>(a, s) : launch_one<T, float, A_CONV3, EPI_RELU>(a, s);
    } else if (amode == A_CONV3 && epi == EPI_BIAS) {  // training: dgrad of the 3x3 conv (correlation with the flipped taps)
        VITSEG_CHECK_ARG(a.D % 64 == 0 && a.zeros, VITSEG_ESHAPE, "channel count must be a multiple of 64");
        return large ? launch_large<T, float, A_CONV3, EPI_BIAS>(a, s) : launch_one<T, float, A_CONV3, EPI_BIAS>(a, s);
    }
    set_error("gemm_bf16: unsupported amode/epilogue %d/%d", amode, epi);
    return VITSEG_EINVAL;
}

template <typename T>
int launch_gemm_h16(const GemmArgs& a, int amode, int epi, hipStream_t s) {
    int covered = 0;
    if (int rc = launch_gemm_h16_impl<T>(a, amode, epi, s, &covered)) return rc;
    if (!a.colsum_out) return VITSEG_OK;
    VITSEG_CHECK_ARG(a.colsum_scratch && (std::is_same<T, bf16_t>::value), VITSEG_EINVAL, "gemm: column sums need bf16 + scratch");
    if (covered > 0)   // per-tile partials are in the scratch; add the rows the persistent kernel did not cover and reduce
        return launch_colsum_finish_fused((const T*)a.C + (size_t)covered * a.ldc, a.M - covered, 2 * ((covered + 255) / 256),
                                          a.colsum_out, a.colsum_scratch, a.N, a.ldc, s);
    return launch_colsum(a.C, 1, a.colsum_out, a.colsum_scratch, a.M, a.N, a.ldc, s);
}

int launch_gemm_bf16(const GemmArgs& a, int amode, int epi, hipStream_t s, bool f16) {
    return f16 ? launch_gemm_h16<f16_t>(a, amode, epi, s) : launch_gemm_h16<bf16_t>(a, amode, epi, s);
}

// =====================================================================================================
// bf16 weight-gradient GEMM with BOTH operands in T-form:  C[M,N] = sum_k A[k][M-index] * W[k][N-index]
// (dW = dY^T X: A = dY [tokens][M], W = X [tokens][N], the reduction runs over the token rows).  No
// transposed copies: a K step stages 64 token rows x 128 columns of each operand as they lie in memory
// ([64][128] bf16, 256-B rows, global_load_lds) and the MFMA operands are gathered DOWN the columns with
// ds_read_b64_tr_b16 (4 tokens x 16 columns per 16-lane group).  Both operands use the same k permutation
// (element j of lane half h = token 16s + 8(j>>2) + 4h + (j&3)), so the products pair up correctly.
// 16-byte chunk c of row r is stored at c ^ ((r & 3) << 2): the 4 rows x 2 column blocks of one transposed
// read then hit 16 distinct 16-byte slots of the 256-B bank row.  Split-K over grid.y, fp32 partial output.
template <int DUMMY = 0>
__global__ __launch_bounds__(256, 2) void gemm_bf16_tt_kernel(const GemmArgs p) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) bf16_t lds[2][2][64 * 128];  // [buffer][A|W][token][column]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = t / tiles_n, tile_n = t - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int KT_all = (p.K + 63) / 64;
    const int nsplit = gridDim.y, split = blockIdx.y;
    const int kt0 = (int)((long long)KT_all * split / nsplit);
    const int KT = (int)((long long)KT_all * (split + 1) / nsplit) - kt0;

    // DMA: per operand 16 pieces of 1 KiB (4 token rows x 256 B) per K step; wave w issues pieces 4w .. 4w+3.
    // lane l -> row l >> 4, chunk position l & 15 holding logical chunk (l & 15) ^ ((row & 3) << 2).
    const int drow = lane >> 4;
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (wave * 4 + i) * 4 + drow;  // token row inside the tile, 0..63
            const int ch = (lane & 15) ^ ((r & 3) << 2);
            const int tok = (kt + kt0) * 64 + r;
            const bool ok = tok < p.K;
            const int ca = min(m0 + ch * 8, p.M - 8), cw = min(n0 + ch * 8, p.N - 8);  // clamped: never stored
            const bf16_t* ga = ok ? (const bf16_t*)p.A + (size_t)tok * p.lda + ca : (const bf16_t*)p.zeros;
            const bf16_t* gw = ok ? (const bf16_t*)p.W + (size_t)tok * p.ldw + cw : (const bf16_t*)p.zeros;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga,
                                             (__attribute__((address_space(3))) void*)&lds[buf][0][(wave * 4 + i) * 4 * 128],
                                             16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gw,
                                             (__attribute__((address_space(3))) void*)&lds[buf][1][(wave * 4 + i) * 4 * 128],
                                             16, 0, 0);
        }
    };
    // transposed fragment: tokens t0 .. t0+15 (lane half h: t0 + 4h + {0..3} and + 8), column col0 + (lane & 31)
    const int g = lane & 15, grp = lane >> 4, tq = g >> 2, tp = g & 3;
    auto tr_frag = [&](const bf16_t* tile, int t0, int col0) {
        const int row = t0 + 4 * (grp >> 1) + tq;
        const int col = col0 + 16 * (grp & 1) + 4 * tp;
        const int off0 = row * 128 + ((((col >> 3) ^ ((row & 3) << 2)) << 3) | (col & 7));
        const int off1 = (row + 8) * 128 + ((((col >> 3) ^ (((row + 8) & 3) << 2)) << 3) | (col & 7));
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + off0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + off1));
        const bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return f;
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    if (KT > 0) issue(0, 0);
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // step kt is in LDS for every wave; buffer buf^1 is no longer read
        if (kt + 1 < KT) issue(kt + 1, buf ^ 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = tr_frag(lds[buf][0], 16 * s, wm * 64 + mi * 32);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[ni] = tr_frag(lds[buf][1], 16 * s, wn * 64 + ni * 32);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
    }
    // ---- epilogue: fp32 partial tile, staged through LDS for row-vector stores ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int li = lane & 31, lh = lane >> 5;
    float* wl = (float*)&lds[0][0][0] + wave * 4096;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                wl[(mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + ni * 32 + li] = acc[mi][ni][r];
    const int rr = lane >> 4, c4 = (lane & 15) * 4;
    const int gcol = n0 + wn * 64 + c4;
    float* C = (float*)p.C + (size_t)blockIdx.y * p.split_stride;
    f32x4 v[16];
#pragma unroll
    for (int ps = 0; ps < 16; ++ps) v[ps] = *(const f32x4*)&wl[(ps * 4 + rr) * 64 + c4];
    if (gcol >= p.N) return;
#pragma unroll
    for (int ps = 0; ps < 16; ++ps) {
        const int grow = m0 + wm * 64 + ps * 4 + rr;
        if (grow < p.M) *(f32x4*)(C + (size_t)grow * p.ldc + gcol) = v[ps];
    }
}

// dW[M,N] (fp32, dense) = A^T . W with A = [K][M], W = [K][N] bf16 row-major; split-K through `scratch`.
int launch_wgrad_bf16_tt(GemmArgs a, float* scratch, hipStream_t s) {
    VITSEG_CHECK_ARG(a.M % 8 == 0 && a.N % 8 == 0 && a.lda % 8 == 0 && a.ldw % 8 == 0 && a.ldc == a.N && a.zeros,
                     VITSEG_ESHAPE, "wgrad_bf16_tt: M, N and the leading dimensions must be multiples of 8");
    if (wgrad_p8_applies(a)) return launch_wgrad_p8(a, scratch, s);   // 256x256 tiles, 8-phase stream (gemm_p8.hip)
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    int splits = 1024 / tiles;  // <= 2 whole rounds of 512 resident blocks
    const int ksteps = (a.K + 63) / 64;
    if (splits > ksteps / 4) splits = ksteps / 4;
    if (splits < 1) splits = 1;
    float* out = (float*)a.C;
    if (splits > 1) {
        VITSEG_CHECK_ARG(scratch, VITSEG_EINVAL, "wgrad_bf16_tt: split-K needs scratch");
        a.C = scratch;
    }
    a.split_stride = (size_t)a.M * a.N;
    hipLaunchKernelGGL(gemm_bf16_tt_kernel<0>, dim3(tiles, splits), dim3(256), 0, s, a);
    VITSEG_LAUNCH_CHECK("gemm_bf16_tt");
    if (splits > 1) {
        const size_t n4 = (size_t)a.M * a.N / 4;
        const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, scratch, out, n4, splits);
        VITSEG_LAUNCH_CHECK("splitk_reduce");
    }
    return VITSEG_OK;
}

// bf16 training GEMMs (all N-form: dgrad multiplies by a transposed bf16 copy of the weight, wgrad by
// transposed copies of dY and X whose reduction length is zero-padded to a multiple of 64).
//   out_f32 = 0: C bf16, epi EPI_BIAS (plain dgrad), EPI_GELU (forward, `aux` = bf16 pre-activation) or
//                EPI_DGELU (R = bf16 pre-activation);
//   out_f32 = 1: C fp32, EPI_BIAS, optional split-K into `scratch` (weight gradients).
int launch_gemm_bf16_train(GemmArgs a, int epi, int out_f32, float* scratch, hipStream_t s) {
    VITSEG_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0 && a.K % 64 == 0, VITSEG_ESHAPE, "gemm_bf16_train: K=%d %% 64", a.K);
    VITSEG_CHECK_ARG(a.N % 4 == 0 && a.ldc % 4 == 0 && a.lda % 8 == 0, VITSEG_ESHAPE, "gemm_bf16_train: alignment");
    if (!out_f32) {  // same tile selection as inference (256x256 for wide outputs, 256x128 for long K, else 128x128)
        if (epi == EPI_BIAS || epi == EPI_GELU || epi == EPI_DGELU) return launch_gemm_h16<bf16_t>(a, A_PLAIN, epi, s);
    } else if (epi == EPI_BIAS) {
        const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
        int splits = 1024 / tiles;  // <= 2 whole rounds of 512 resident blocks
        if (splits > a.K / 64 / 4) splits = a.K / 64 / 4;
        if (splits <= 1 || !scratch) return launch_one<bf16_t, float, A_PLAIN, EPI_BIAS>(a, s);
        VITSEG_CHECK_ARG(a.ldc == a.N, VITSEG_EINVAL, "gemm_bf16_train: split-K needs a dense output");
        float* out = (float*)a.C;
        a.C = scratch;
        a.splitk = splits;
        a.split_stride = (size_t)a.M * a.N;
        if (int rc = launch_one<bf16_t, float, A_PLAIN, EPI_BIAS>(a, s)) return rc;
        const size_t n4 = (size_t)a.M * a.N / 4;
        const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, scratch, out, n4, splits);
        VITSEG_LAUNCH_CHECK("splitk_reduce");
        return VITSEG_OK;
    }
    set_error("gemm_bf16_train: unsupported epilogue %d / out_f32 %d", epi, out_f32);
    return VITSEG_EINVAL;
}
size_t wgrad_bf16_scratch_floats(int M, int N, int K) {
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    int splits = 1024 / tiles;  // <= 2 whole rounds of 512 resident blocks
    if (splits > K / 64 / 4) splits = K / 64 / 4;
    if (splits < 1) splits = 1;
    if (M % 256 == 0 && N % 256 == 0) {   // the 8-phase kernel's slicing (wgrad_p8_splits), whichever is larger
        const int sp8 = wgrad_p8_splits(M, N, K);
        if (sp8 > splits) splits = sp8;
    }
    return (size_t)splits * M * N;
}

}  // namespace vitseg
