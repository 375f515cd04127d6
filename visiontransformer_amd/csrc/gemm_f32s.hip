// fp32 GEMM of the SMALL-BATCH regime (small.hpp):  C[M,N] = epi(A[M,K] . W[N,K]^T)  for M < 2048 token rows --
// the q/k/v/o projections, the MLP, the patch embedding and the 3x3 head conv of ViTSegmentationModel.forward
// (/root/reference/model/CE/classes.py:246-257, transformers/models/vit/modeling_vit.py:62-69,207-254) at the batch
// sizes the reference runs (4 x 224x224) and serves (1 image).  Exact fp32 products (v_mfma_f32_32x32x2_f32).
// Roofline: fp32 matrix pipe, 157.3 TFLOP/s; algorithmic work 2 M N K per launch.  (16-bit operand form, template parameter H:
// the same byte streams on v_mfma_f32_32x32x16_bf16 / _f16 -- bound by the loader wave's DMA issue rate, not the matrix pipe.)
//
// Why a third fp32 kernel: at 197-1576 rows the persistent 256x128 kernel (gemm_f32p.hip) has 1-7 row tiles and the
// 128x128 tile kernel cut into K slices (whole_split) ran 6 slices of FOUR K steps each behind a 2 us prologue
// (profiles/r05_before_ref_grid_*: 36 launches x 41 us per forward at batch 4 = 0.38 of the fp32 peak on the GEMMs).  Here:
//   * one 256-thread block per (row tile, column tile, K chunk); the tile is a template parameter (32..128 rows x
//     64..192 columns, 4 waves, 1-4 MFMA tiles of 32x32 per wave) and small_plan() picks, per launch, the variant
//     with the least work on the busiest CU -- the chunk count is NOT its choice (small_splits: shape-only), so the
//     bits of every output are the same at every batch size;
//   * LDS = a ring of 3-4 K steps (32 floats = 128-byte rows, XOR-swizzled 16-byte chunks) filled by LDS-DMA
//     (buffer_load_dwordx4 ... lds) 2-3 steps ahead behind counted vmcnt waits; one barrier per K step, rotated in
//     front of the last MFMA group as in gemm_f32p.hip;
//   * operands addressed through buffer descriptors: rows beyond M / N read as zeros, the 3x3 conv's taps outside the
//     image too (an out-of-range offset), the patch gather (SA_PATCH) is a per-lane offset plus a scalar per K step;
//   * C transposed in the accumulators (MFMA A operand = W rows), parked through a wave-private 4 KiB slab in the
//     (finished) ring and written as whole 128-byte row pieces.
#include <stdlib.h>

#include <type_traits>

#include "small.hpp"

namespace vitseg {
namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

struct Variant {
    int MT, NT, WGM, WGN, NSTAGE;
    constexpr int bm() const { return 32 * MT * WGM; }
    constexpr int bn() const { return 32 * NT * WGN; }
    constexpr int lds() const { return NSTAGE * (bm() + bn()) * 128; }
};
// 0: 64x128  1: 32x128  2: 64x192  3: 128x128  4: 64x64
constexpr Variant VARIANTS[] = {{1, 2, 2, 2, 3}, {1, 1, 1, 4, 4}, {1, 3, 2, 2, 3}, {2, 2, 2, 2, 3}, {1, 1, 2, 2, 4}};
constexpr int NVARIANTS = sizeof(VARIANTS) / sizeof(VARIANTS[0]);

// H: 0 = fp32 operands (v_mfma_f32_32x32x2_f32); 1 / 2 = bf16 / fp16 operands (v_mfma_f32_32x32x16_*): the SAME bytes move --
// the launcher hands K / lda / ldw in 4-byte units, a 128-byte row piece is 64 values, and the 16-byte chunk (2 j + lh) a lane
// reads is exactly the eight values (k = 16 j + 8 lh ..) the wide MFMA takes from it -- so only the products and the store of
// the GELU epilogue (the next GEMM's 16-bit operand) differ.  SA_PLAIN only.
template <int V, int EPI, int AMODE, int H = 0>
__global__ __launch_bounds__(320) void gemm_f32s_kernel(const SGemm p) {
    static_assert(H == 0 || (AMODE == SA_PLAIN && (EPI == SE_PARTIAL || EPI == SE_BIAS || EPI == SE_GELU)), "16-bit operands: the forward linears");
    typedef typename std::conditional<H == 2, f16_t, bf16_t>::type HT;
    constexpr Variant CV = VARIANTS[V];
    constexpr int MT = CV.MT, NT = CV.NT, WGN = CV.WGN, NSTAGE = CV.NSTAGE;
    constexpr int BM = CV.bm(), BN = CV.bn();
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int PIECES = (BM + BN) / 8;   // 1 KiB DMA pieces per K step
    static_assert((NSTAGE - 2) * PIECES <= 63, "the loader's counted waits must fit the 6-bit vmcnt");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..3 compute, 4 = the loader
    unsigned long long* stamp = p.stamps && tid == 0 ? p.stamps + (size_t)blockIdx.x * 8 : nullptr;   // diagnostics only
    if (stamp) {
        stamp[0] = __builtin_amdgcn_s_memrealtime();
        stamp[2] = __builtin_amdgcn_s_memtime();
        stamp[6] = __builtin_amdgcn_s_getreg(63492);   // HW_REG_HW_ID
        stamp[7] = __builtin_amdgcn_s_getreg(63508);   // HW_REG_XCC_ID
    }

    // block -> (row tile, column tile, chunk): row tiles fastest, so the blocks that share a W panel are neighbours in
    // the list and (xcd_remap) land on one XCD's L2; the whole A operand (< 8 MiB) lives in every L2 anyway
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = t % p.tiles_m, t1 = t / p.tiles_m;
    const int tn = t1 % p.tiles_n, sp = t1 / p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kc = AMODE == SA_CONV3 ? p.K : AMODE == SA_CONV3_ALL ? 9 * p.K : p.K / p.splits;   // conv3: one tap (K = D values) per chunk
    const int KT = kc / 32;
    const int k0w = sp * kc;                                    // first W column of this chunk

    auto make_rsrc = [](const void* base, long long bytes) {
        const unsigned long long b = (unsigned long long)base;
        i32x4 r;
        r[0] = (int)(unsigned)b;
        r[1] = (int)(unsigned)((b >> 32) & 0xffffu);   // stride 0
        r[2] = (int)(unsigned)(bytes <= 0 ? 0 : (bytes < 0x7fffffffll ? bytes : 0x7fffffffll));
        r[3] = 0x00020000;
        return r;
    };
    constexpr bool WT = AMODE == SA_PLAIN_WT || AMODE == SA_TT;   // W[k][n] (n contiguous): the stage's W region is [32 k][BN n]
    constexpr bool AT = AMODE == SA_TT;                           // A[k][m] likewise: [32 k][BM m]
    const long long krows = AT && p.kvalid > 0 ? p.kvalid : p.K;  // rows of the T-form operands that exist (the rest reads as zeros)
    const i32x4 ra = AMODE == SA_TT ? make_rsrc(p.A + m0, (krows * p.lda - m0) * 4)
                     : AMODE == SA_PLAIN || AMODE == SA_PLAIN_WT ? make_rsrc(p.A + (size_t)m0 * p.lda, (long long)(p.M - m0) * p.lda * 4)
                     : AMODE == SA_CONV3 || AMODE == SA_CONV3_ALL ? make_rsrc(p.A, (long long)p.M * p.lda * 4)
                                         : make_rsrc(p.A, (long long)(p.M / p.Np) * p.Cin * p.S * p.S * 4);
    const i32x4 rw = WT ? make_rsrc(p.W + n0, (krows * p.ldw - n0) * 4)
                        : make_rsrc(p.W + (size_t)n0 * p.ldw, (long long)(p.N - n0) * p.ldw * 4);

    // Barrier protocol (all five waves): B0 = K step 0 has landed; B(kt + 1), kt = 0 .. KT - 2 = step kt + 1 has landed AND
    // every compute wave holds its last fragments of step kt (its ring stage is free); one more after the loop (the ring is
    // free for the epilogue's slabs).  The loader waits on its own vmcnt in front of each barrier; nobody else issues DMA.
    if (wave == 4) {
        // ---- the loader wave: the whole LDS-DMA stream of the block.  Issued from the compute waves (the first version of
        // this kernel), a 1 KiB piece cost ~100 cycles of issue between two 64-cycle MFMAs and 5-8 pieces per step left the
        // matrix pipe idle for 500-650 cycles of every K step (profiles/r05_small_stamps_1.txt); a wave that does nothing else
        // issues a piece in ~25 ----
        // piece q = rows [8 q, 8 q + 8) of the stage (A rows first); lane l -> row 8 q + (l >> 3), chunk position l & 7, which
        // holds logical chunk (l & 7) ^ ((row >> 1) & 7)
        constexpr unsigned OOB = 0x7ffffff0u;   // beyond every descriptor's range: reads as zeros
        int dy = 0, dx = 0;
        if (AMODE == SA_CONV3) {
            dy = sp / 3 - 1;
            dx = sp % 3 - 1;
        }
        unsigned voff[PIECES];
        auto set_tap = [&](int tap) {   // SA_CONV3_ALL: the A rows of tap (ky, kx) = the map shifted by the tap, zero outside the image
            const int ty = tap / 3 - 1, tx = tap % 3 - 1;
#pragma unroll
            for (int q = 0; q < BM / 8; ++q) {
                const int row = 8 * q + (lane >> 3), chunk = (lane & 7) ^ ((row >> 1) & 7);
                const int r = m0 + row;
                const int rem = r % p.Np, y = rem / p.g + ty, x = rem % p.g + tx;
                const bool ok = r < p.M && y >= 0 && y < p.g && x >= 0 && x < p.g;
                voff[q] = ok ? (unsigned)(r + ty * p.g + tx) * (unsigned)p.lda * 4u + (unsigned)(chunk * 16) : OOB;
            }
        };
#pragma unroll
        for (int q = 0; q < PIECES; ++q) {
            const int row = 8 * q + (lane >> 3), pos = lane & 7;
            if (q < BM / 8 && AMODE == SA_CONV3_ALL) {
                voff[q] = OOB;   // (set_tap fills these per tap)
            } else if (q < BM / 8 && AT) {
                const unsigned off = (unsigned)(q * 1024 + 16 * lane);   // the [32 k][BM m] region in order, as the T-form W below
                voff[q] = (off / (BM * 4)) * (unsigned)p.lda * 4u + off % (BM * 4);
            } else if (q < BM / 8) {
                const int chunk = pos ^ ((row >> 1) & 7);
                if (AMODE == SA_PLAIN || AMODE == SA_PLAIN_WT) {
                    voff[q] = (unsigned)row * (unsigned)p.lda * 4u + (unsigned)(chunk * 16);
                } else if (AMODE == SA_CONV3) {
                    const int r = m0 + row;
                    const int rem = r % p.Np, y = rem / p.g + dy, x = rem % p.g + dx;
                    const bool ok = r < p.M && y >= 0 && y < p.g && x >= 0 && x < p.g;
                    voff[q] = ok ? (unsigned)(r + dy * p.g + dx) * (unsigned)p.lda * 4u + (unsigned)(chunk * 16) : OOB;
                } else {
                    const int r = m0 + row;
                    const int b = r / p.Np, rem = r % p.Np, gy = rem / p.g, gx = rem % p.g;
                    const int e4 = 4 * chunk;
                    voff[q] = r < p.M ? (unsigned)(((b * p.Cin) * p.S + gy * p.P + e4 / p.P) * p.S + gx * p.P + e4 % p.P) * 4u : OOB;
                }
            } else if (WT) {
                // T-form: piece = 1 KiB of the [32 k][BN n] region in order (k row = BN * 4 bytes; a column group beyond N reads the
                // next k row's values: those columns are computed and never stored)
                const unsigned off = (unsigned)((q - BM / 8) * 1024 + 16 * lane);
                voff[q] = (off / (BN * 4)) * (unsigned)p.ldw * 4u + off % (BN * 4);
            } else {
                const int wrow = row - BM;
                voff[q] = (unsigned)wrow * (unsigned)p.ldw * 4u + (unsigned)((pos ^ ((wrow >> 1) & 7)) * 16);
            }
        }
        const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
        // scalar byte offsets of K step kt: W columns k0w + 32 kt; A by mode
        auto soff_a = [&](int kt) -> unsigned {
            if (AT) return (unsigned)(k0w + 32 * kt) * (unsigned)p.lda * 4u;
            if (AMODE == SA_PLAIN || AMODE == SA_PLAIN_WT) return (unsigned)(k0w + 32 * kt) * 4u;
            if (AMODE == SA_CONV3) return (unsigned)(32 * kt) * 4u;
            if (AMODE == SA_CONV3_ALL) return (unsigned)(32 * (kt % (p.K / 32))) * 4u;
            const int k = k0w + 32 * kt, pp = p.P * p.P;      // patch: k = (c, py, px); 32 | P^2 and P | 32 kt (P = 8, 16, 32)
            return (unsigned)((k / pp) * p.S * p.S + ((k % pp) / p.P) * p.S) * 4u;
        };
        auto dma_step = [&](int kt) {   // K step kt into ring stage kt % NSTAGE
            if (AMODE == SA_CONV3_ALL && kt % (p.K / 32) == 0) set_tap(kt / (p.K / 32));
            const unsigned sb = lds_base + (unsigned)((kt % NSTAGE) * STAGE);
            const unsigned sa = soff_a(kt), sw_ = WT ? (unsigned)(k0w + 32 * kt) * (unsigned)p.ldw * 4u : (unsigned)(k0w + 32 * kt) * 4u;
#pragma unroll
            for (int q = 0; q < PIECES; ++q) {
                if (q < BM / 8)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                                 :: "s"(sb + (unsigned)(q * 1024)), "v"(voff[q]), "s"(ra), "s"(sa) : "memory");
                else
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                                 :: "s"(sb + (unsigned)(q * 1024)), "v"(voff[q]), "s"(rw), "s"(sw_) : "memory");
            }
        };
        auto wait_landed = [&](int ahead) {   // all but the youngest `ahead` K steps have landed
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NSTAGE - 2) * PIECES >= 2 * PIECES ? 2 * PIECES : PIECES) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        int issued = 0;   // K steps issued so far
        for (; issued < NSTAGE - 1 && issued < KT; ++issued) dma_step(issued);
        wait_landed(issued - 1);
        __builtin_amdgcn_s_barrier();   // B0
        for (int kt = 0; kt + 1 < KT; ++kt) {
            // behind B(kt) every compute wave has left stage (kt - 1) % NSTAGE: refill it, NSTAGE - 1 steps ahead
            if (issued < KT) dma_step(issued++);
            wait_landed(issued - 1 - (kt + 1));
            __builtin_amdgcn_s_barrier();   // B(kt + 1)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // the ring is free
        return;
    }

    // ---- compute waves: lane (li, lh) reads 16-byte chunk (2 j + lh) ^ sw of row li of each 32-row MFMA tile ----
    const int wr = wave / WGN, wc = wave % WGN;
    const int li = lane & 31, lh = lane >> 5;
    const int sw = (li >> 1) & 7;
    int offj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) offj[j] = li * 128 + (((2 * j + lh) ^ sw) << 4);
    f32x4 fa[2][MT], fw[2][NT];
    auto read_frags = [&](int stage, int j, int slot) {
        const int ab = stage * STAGE + wr * MT * 4096 + offj[j];
        const int wb = stage * STAGE + BM * 128 + wc * NT * 4096 + offj[j];
        if (AT) {
            const float* ar_ = (const float*)(lds + stage * STAGE) + (8 * j + 4 * lh) * BM + wr * MT * 32 + li;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                fa[slot][mt] = f32x4{ar_[mt * 32], ar_[BM + mt * 32], ar_[2 * BM + mt * 32], ar_[3 * BM + mt * 32]};
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fa[slot][mt] = *(const f32x4*)(lds + ab + mt * 4096);
        }
        if (WT) {   // [k][n] region: lane (li, lh) gathers k = 8 j + 4 lh + e, n = its column -- four conflict-free 4-byte reads
            const float* wr_ = (const float*)(lds + stage * STAGE + BM * 128) + (8 * j + 4 * lh) * BN + wc * NT * 32 + li;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                fw[slot][nt] = f32x4{wr_[nt * 32], wr_[BN + nt * 32], wr_[2 * BN + nt * 32], wr_[3 * BN + nt * 32]};
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) fw[slot][nt] = *(const f32x4*)(lds + wb + nt * 4096);
        }
    };
    f32x16 acc[MT][NT];   // acc[mt][nt][r] = C[m = 32 mt + li][n = 32 nt + (r & 3) + 8 (r >> 2) + 4 lh]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    auto mfma_group = [&](int slot) {   // chunk pair j: k = 8 j + 4 lh + e, e = 0..3 (16-bit: k = 16 j + 8 lh + 0..7 in one product)
        if constexpr (H != 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = H16<HT>::mfma(__builtin_bit_cast(bf16x8, fw[slot][nt]), __builtin_bit_cast(bf16x8, fa[slot][mt]), acc[mt][nt]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[slot][nt][e], fa[slot][mt][e], acc[mt][nt], 0, 0, 0);
        }
    };

    // Direct epilogues (one K chunk per launch) sum the reduction in p.kh equal pieces -- acc is folded into `sum` and
    // restarted from zero every KT / kh steps: result = ((h0 + h1) + h2) + h3 -- because the one-image kernel below
    // (gemm_f32s_kw_kernel) computes the pieces on four waves at once, and a row's bits must not depend on which kernel ran.
    constexpr bool FOLD = EPI != SE_PARTIAL;
    f32x16 sum[FOLD ? MT : 1][FOLD ? NT : 1];
    const int KC = FOLD && p.kh > 1 ? KT / p.kh : KT;
    int kc_left = KC;
    bool folded = false;
    __builtin_amdgcn_s_barrier();   // B0
    __builtin_amdgcn_sched_barrier(0);
    read_frags(0, 0, 0);
    if (stamp) stamp[3] = __builtin_amdgcn_s_memtime();
    int st = 0;
    for (int kt = 0; kt < KT; ++kt) {
        const int nst = st + 1 == NSTAGE ? 0 : st + 1;
        read_frags(st, 1, 1);
        mfma_group(0);
        read_frags(st, 2, 0);
        mfma_group(1);
        read_frags(st, 3, 1);
        mfma_group(0);
        if (kt + 1 < KT) {
            // the rotated barrier: all of this wave's reads of stage st are in registers, the next step's first fragments
            // are read behind it under the last group's MFMAs
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // B(kt + 1)
            __builtin_amdgcn_sched_barrier(0);
            read_frags(nst, 0, 0);
        }
        mfma_group(1);
        st = nst;
        if constexpr (FOLD) {
            if (--kc_left == 0 && KC != KT) {   // a piece is complete
                kc_left = KC;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            sum[mt][nt][r] = folded ? sum[mt][nt][r] + acc[mt][nt][r] : acc[mt][nt][r];
                            acc[mt][nt][r] = 0.f;
                        }
                folded = true;
            }
        }
    }
    if constexpr (FOLD) {
        if (folded) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = sum[mt][nt];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (stamp) stamp[4] = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();   // the ring is free: every wave parks its tiles in its own 4 KiB of it

    // ---- epilogue ----
    float* slab = (float*)(lds + wave * 4096);
    const int rrow = lane >> 3, c8 = lane & 7;
    float* cbase = p.C + (EPI == SE_PARTIAL ? (size_t)sp * p.split_stride : 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x16& tl = acc[mt][nt];
            // park: row li, chunk 2 q + lh at position ^ (li & 7); re-read row-wise: lane -> rows 8 ps + rrow, columns 4 c8 .. + 3
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(f32x4*)(slab + li * 32 + (((2 * q + lh) ^ (li & 7)) << 2)) = f32x4{tl[4 * q], tl[4 * q + 1], tl[4 * q + 2], tl[4 * q + 3]};
            const int gcol = n0 + (wc * NT + nt) * 32 + c8 * 4;
            f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
            if ((EPI == SE_BIAS || EPI == SE_GELU || EPI == SE_RELU) && gcol < p.N) bias4 = *(const f32x4*)(p.bias + gcol);
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = 8 * ps + rrow;
                f32x4 v = *(const f32x4*)(slab + row * 32 + ((c8 ^ (row & 7)) << 2));
                const int grow = m0 + (wr * MT + mt) * 32 + row;
                const bool live = grow < p.M && gcol < p.N;
                const size_t o = (size_t)grow * p.ldc + gcol;
                if (EPI == SE_BIAS || EPI == SE_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] + bias4[e];
                    if (EPI == SE_GELU) {
                        if (p.aux && live) *(f32x4*)(p.aux + o) = v;   // the pre-activation, saved for the backward
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                    }
                }
                if (EPI == SE_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] + bias4[e], 0.f);
                }
                if (EPI == SE_DGELU) {
                    f32x4 u = {0.f, 0.f, 0.f, 0.f};
                    if (live) u = *(const f32x4*)(p.R + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] * gelu_erf_grad(u[e]);
                }
                if (H != 0 && EPI == SE_GELU) {   // the MLP hidden as the next GEMM reads it: 16-bit
                    if (live) *(uint2*)((unsigned short*)cbase + o) = uint2{H16<HT>::pack2(v[0], v[1]), H16<HT>::pack2(v[2], v[3])};
                } else if (live) {
                    *(f32x4*)(cbase + o) = v;
                }
            }
        }
    if (stamp) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp[5] = __builtin_amdgcn_s_memtime();
        stamp[1] = __builtin_amdgcn_s_memrealtime();
    }
}


// ---- one image: the four K pieces of a direct-epilogue GEMM on four waves --------------------------------------------
// At 197 rows a wave of gemm_f32s_kernel owns one 32 x 32 tile over the WHOLE reduction: 24 dependent K steps = 12 us, with
// half the CUs idle (7 x 18 tiles of 32 x 128).  Only cutting K shortens that chain.  Here a block is one 32 x (32 NT) output
// tile and its four compute waves each run ONE piece of the reduction (K / 4: 6 steps at K = 768) on all NT sub-tiles; the
// loader wave streams the four pieces side by side (a ring stage = 4 x (32 + 32 NT) operand rows); the pieces are parked in
// LDS and added in piece order ((h0 + h1) + h2) + h3 -- exactly what gemm_f32s_kernel's FOLD computes one piece after the
// other, so the two kernels give the same bits and small_plan may pick either by M.
template <int NT, int EPI, int H = 0>
__global__ __launch_bounds__(320) void gemm_f32s_kw_kernel(const SGemm p) {
    typedef typename std::conditional<H == 2, f16_t, bf16_t>::type HT;
    constexpr int BN = 32 * NT, CHROWS = 32 + BN;
    constexpr int STAGE = 4 * CHROWS * 128;
    constexpr int NSTAGE = 163840 / STAGE >= 3 ? 3 : 2;
    constexpr int PPC = CHROWS / 8, PIECES = 4 * PPC;   // 1 KiB DMA pieces per K piece / per ring stage
    constexpr int WAITN = PIECES < 63 ? PIECES : 63;     // (vmcnt is 6 bits; in-order retirement makes 63 a safe stand-in for 64)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..3 = K piece, 4 = the loader
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = t % p.tiles_m, tn = t / p.tiles_m;
    const int m0 = tm * 32, n0 = tn * BN;
    const int KC = p.K / 128;                                    // K steps per piece
    auto make_rsrc = [](const void* base, long long bytes) {
        const unsigned long long b = (unsigned long long)base;
        i32x4 r;
        r[0] = (int)(unsigned)b;
        r[1] = (int)(unsigned)((b >> 32) & 0xffffu);
        r[2] = (int)(unsigned)(bytes <= 0 ? 0 : (bytes < 0x7fffffffll ? bytes : 0x7fffffffll));
        r[3] = 0x00020000;
        return r;
    };
    const i32x4 ra = make_rsrc(p.A + (size_t)m0 * p.lda, (long long)(p.M - m0) * p.lda * 4);
    const i32x4 rw = make_rsrc(p.W + (size_t)n0 * p.ldw, (long long)(p.N - n0) * p.ldw * 4);
    // barriers (all five waves): B0, B(kt + 1) for kt = 0 .. KC - 2, "ring free", "pieces parked"
    if (wave == 4) {
        unsigned voff[PPC];
#pragma unroll
        for (int q = 0; q < PPC; ++q) {
            const int row = 8 * q + (lane >> 3), pos = lane & 7;
            if (q < 4) voff[q] = (unsigned)row * (unsigned)p.lda * 4u + (unsigned)((pos ^ ((row >> 1) & 7)) * 16);
            else voff[q] = (unsigned)(row - 32) * (unsigned)p.ldw * 4u + (unsigned)((pos ^ (((row - 32) >> 1) & 7)) * 16);
        }
        const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
        auto dma_step = [&](int kt) {
            const unsigned sb = lds_base + (unsigned)((kt % NSTAGE) * STAGE);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned so = (unsigned)((c * KC + kt) * 32) * 4u;
#pragma unroll
                for (int q = 0; q < PPC; ++q) {
                    if (q < 4)
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                                     :: "s"(sb + (unsigned)((c * PPC + q) * 1024)), "v"(voff[q]), "s"(ra), "s"(so) : "memory");
                    else
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                                     :: "s"(sb + (unsigned)((c * PPC + q) * 1024)), "v"(voff[q]), "s"(rw), "s"(so) : "memory");
                }
            }
        };
        auto wait_landed = [&](int ahead) {
            if (ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WAITN) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        int issued = 0;
        for (; issued < NSTAGE - 1 && issued < KC; ++issued) dma_step(issued);
        wait_landed(issued - 1);
        __builtin_amdgcn_s_barrier();   // B0
        for (int kt = 0; kt + 1 < KC; ++kt) {
            if (issued < KC) dma_step(issued++);
            wait_landed(issued - 1 - (kt + 1));
            __builtin_amdgcn_s_barrier();   // B(kt + 1)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // the ring is free
        __builtin_amdgcn_s_barrier();   // the pieces are parked
        return;
    }
    const int li = lane & 31, lh = lane >> 5;
    const int sw = (li >> 1) & 7;
    int offj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) offj[j] = wave * CHROWS * 128 + li * 128 + (((2 * j + lh) ^ sw) << 4);
    f32x4 fa[2], fw[2][NT];
    auto read_frags = [&](int stage, int j, int slot) {
        const int ab = stage * STAGE + offj[j];
        fa[slot] = *(const f32x4*)(lds + ab);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) fw[slot][nt] = *(const f32x4*)(lds + ab + 4096 + nt * 4096);
    };
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    auto mfma_group = [&](int slot) {
        if constexpr (H != 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = H16<HT>::mfma(__builtin_bit_cast(bf16x8, fw[slot][nt]), __builtin_bit_cast(bf16x8, fa[slot]), acc[nt]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[slot][nt][e], fa[slot][e], acc[nt], 0, 0, 0);
        }
    };
    __builtin_amdgcn_s_barrier();   // B0
    __builtin_amdgcn_sched_barrier(0);
    read_frags(0, 0, 0);
    int st = 0;
    for (int kt = 0; kt < KC; ++kt) {
        const int nst = st + 1 == NSTAGE ? 0 : st + 1;
        read_frags(st, 1, 1);
        mfma_group(0);
        read_frags(st, 2, 0);
        mfma_group(1);
        read_frags(st, 3, 1);
        mfma_group(0);
        if (kt + 1 < KC) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // B(kt + 1)
            __builtin_amdgcn_sched_barrier(0);
            read_frags(nst, 0, 0);
        }
        mfma_group(1);
        st = nst;
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // the ring is free: piece w parks sub-tile nt in slab 4 nt + w
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        float* slab = (float*)(lds + (4 * nt + wave) * 4096);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *(f32x4*)(slab + li * 32 + (((2 * q + lh) ^ (li & 7)) << 2)) = f32x4{acc[nt][4 * q], acc[nt][4 * q + 1], acc[nt][4 * q + 2], acc[nt][4 * q + 3]};
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // the pieces are parked
    if (wave >= NT) return;
    // wave nt finishes sub-tile nt: rows 8 ps + rrow, columns 4 c8 .. + 3 of the four pieces, added in piece order
    const int nt = wave, rrow = lane >> 3, c8 = lane & 7;
    const int gcol = n0 + nt * 32 + c8 * 4;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (gcol < p.N) bias4 = *(const f32x4*)(p.bias + gcol);
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int row = 8 * ps + rrow;
        const int o = row * 32 + ((c8 ^ (row & 7)) << 2);
        f32x4 v = *(const f32x4*)((const float*)(lds + (4 * nt) * 4096) + o);
#pragma unroll
        for (int c = 1; c < 4; ++c) {
            const f32x4 h = *(const f32x4*)((const float*)(lds + (4 * nt + c) * 4096) + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] + h[e];
        }
        const int grow = m0 + row;
        const bool live = grow < p.M && gcol < p.N;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] + bias4[e];
        if (EPI == SE_GELU) {
            if (p.aux && live) *(f32x4*)(p.aux + (size_t)grow * p.ldc + gcol) = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        }
        if (H != 0 && EPI == SE_GELU) {
            if (live) *(uint2*)((unsigned short*)p.C + (size_t)grow * p.ldc + gcol) = uint2{H16<HT>::pack2(v[0], v[1]), H16<HT>::pack2(v[2], v[3])};
        } else if (live) {
            *(f32x4*)(p.C + (size_t)grow * p.ldc + gcol) = v;
        }
    }
}

// Least work on the busiest CU: a block's time is its MFMA stream (KT x MT x NT x 16 MFMAs of 64 cycles on each wave's
// SIMD) plus a fixed prologue / epilogue; co-resident blocks share the matrix pipes, so their streams add up.
constexpr int KW_FIRST = NVARIANTS;   // variant ids NVARIANTS (32 x 64) and NVARIANTS + 1 (32 x 96): gemm_f32s_kw_kernel
bool kw_applies(const SGemm& a, int epi, int amode, int nt) {
    return epi != SE_PARTIAL && amode == SA_PLAIN && a.kh == 4 && a.K % 128 == 0 && a.N % (32 * nt) == 0;
}
int small_plan(const SGemm& a, int kc, int epi, int amode) {
    const int ncu = device_num_cus();
    double best = 1e30;
    int bv = 0;
    for (int v = 0; v < NVARIANTS; ++v) {
        const Variant& V = VARIANTS[v];
        const int tm = (a.M + V.bm() - 1) / V.bm(), tn = (a.N + V.bn() - 1) / V.bn();
        const long blocks = (long)tm * tn * a.splits;
        const long per_cu = (blocks + ncu - 1) / ncu;
        const int occ = 163840 / V.lds() < 1 ? 1 : 163840 / V.lds();
        // measured with in-kernel stamps (tools/small_stamps.py, profiles/r05_small_stamps_*.txt): a K step costs its MFMAs
        // (1024 cycles per tile) + ~200 for the barrier; prologue ~2700, epilogue ~1600 per tile, ~3000 of launch ramp
        // (16-bit operands: 4 products of 32 cycles per tile and step -- the step is the loader's: ~25 cycles per 1 KiB piece)
        const int pieces = (V.bm() + V.bn()) / 8;
        const double step = a.h16 ? (V.MT * V.NT * 128 > 25 * pieces ? V.MT * V.NT * 128 : 25 * pieces) : V.MT * V.NT * 1024;
        const double stream = (double)(kc / 32) * (step + 200);
        const double fixed = 5700.0 + 1600.0 * V.MT * V.NT;
        const double time = per_cu * stream + ((per_cu + occ - 1) / occ) * fixed;
        if (time < best) {
            best = time;
            bv = v;
        }
    }
    for (int nt = 2; nt <= 3; ++nt) {   // the one-image kernel: a quarter of the K steps per wave, one block per CU
        if (!kw_applies(a, epi, amode, nt)) continue;
        const long blocks = (long)((a.M + 31) / 32) * (a.N / (32 * nt));
        const long per_cu = (blocks + ncu - 1) / ncu;
        const double kstep = a.h16 ? (nt * 128 > 25 * (4 + 4 * nt) * 4 ? nt * 128 : 25 * (4 + 4 * nt) * 4) : nt * 1024;
        const double time = per_cu * ((double)(a.K / 128) * (kstep + 200) + 7200.0 + 1600.0 * nt);
        if (time < best) {
            best = time;
            bv = KW_FIRST + nt - 2;
        }
    }
    return bv;
}

template <int NT, int EPI, int H = 0>
int launch_kw(const SGemm& a, hipStream_t s) {
    int dev = 0;
    static bool attr_set[64] = {};
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_f32s_kw_kernel<NT, EPI, H>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(gemm_f32s_kw)");
        attr_set[dev] = true;
    }
    constexpr int STAGE = 4 * (32 + 32 * NT) * 128;
    constexpr int LDS = (163840 / STAGE >= 3 ? 3 : 2) * STAGE;
    hipLaunchKernelGGL((gemm_f32s_kw_kernel<NT, EPI, H>), dim3(a.tiles_m * a.tiles_n), dim3(320), LDS, s, a);
    VITSEG_LAUNCH_CHECK("gemm_f32s_kw");
    return VITSEG_OK;
}

template <int V, int EPI, int AMODE, int H = 0>
int launch_one(const SGemm& a, hipStream_t s) {
    constexpr int LDS0 = VARIANTS[V].lds();
    const int LDS = LDS0 + (a.lds_pad > 0 && LDS0 + a.lds_pad <= 163840 ? a.lds_pad : 0);
    int dev = 0;
    static bool attr_set[64] = {};
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_f32s_kernel<V, EPI, AMODE, H>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(gemm_f32s)");
        attr_set[dev] = true;
    }
    const int blocks = a.tiles_m * a.tiles_n * a.splits;
    hipLaunchKernelGGL((gemm_f32s_kernel<V, EPI, AMODE, H>), dim3(blocks), dim3(320), LDS, s, a);
    VITSEG_LAUNCH_CHECK("gemm_f32s");
    return VITSEG_OK;
}

template <int EPI, int AMODE, int H = 0>
int launch_variant(const SGemm& a, hipStream_t s) {
    switch (a.variant) {
        case 0: return launch_one<0, EPI, AMODE, H>(a, s);
        case 1: return launch_one<1, EPI, AMODE, H>(a, s);
        case 2: return launch_one<2, EPI, AMODE, H>(a, s);
        case 3: return launch_one<3, EPI, AMODE, H>(a, s);
        default: return launch_one<4, EPI, AMODE, H>(a, s);
    }
}
// 16-bit operands (SGemm::h16 = 1 bf16, 2 fp16): the forward linears
template <int H>
int launch_h16(const SGemm& a, int epi, hipStream_t s) {
    if (a.variant >= NVARIANTS) {
        const int nt = a.variant - NVARIANTS + 2;
        if (nt == 2) return epi == SE_GELU ? launch_kw<2, SE_GELU, H>(a, s) : launch_kw<2, SE_BIAS, H>(a, s);
        return epi == SE_GELU ? launch_kw<3, SE_GELU, H>(a, s) : launch_kw<3, SE_BIAS, H>(a, s);
    }
    switch (epi) {
        case SE_PARTIAL: return launch_variant<SE_PARTIAL, SA_PLAIN, H>(a, s);
        case SE_BIAS: return launch_variant<SE_BIAS, SA_PLAIN, H>(a, s);
        case SE_GELU: return launch_variant<SE_GELU, SA_PLAIN, H>(a, s);
    }
    set_error("gemm_f32s: epilogue %d with 16-bit operands", epi);
    return VITSEG_EINVAL;
}

}  // namespace

int launch_gemm_f32s(SGemm a, int epi, int amode, hipStream_t s) {
    VITSEG_CHECK_ARG(a.A && a.W && a.C && a.M > 0 && a.N > 0 && a.K > 0 && a.splits >= 1, VITSEG_EINVAL, "gemm_f32s: bad arguments");
    VITSEG_CHECK_ARG(epi == SE_PARTIAL || (a.splits == 1 && (epi == SE_DGELU ? a.R != nullptr : a.bias != nullptr)), VITSEG_EINVAL,
                     "gemm_f32s: a direct epilogue takes one chunk and its operand (bias / saved pre-activation)");
    VITSEG_CHECK_ARG(amode != SA_PLAIN_WT || epi == SE_PARTIAL || epi == SE_DGELU, VITSEG_EINVAL, "gemm_f32s: the T-form serves the activation gradients");
    VITSEG_CHECK_ARG(amode != SA_TT || (epi == SE_PARTIAL && a.splits == 1), VITSEG_EINVAL, "gemm_f32s: the TT form writes one plain chunk");
    if (a.h16) {   // 16-bit operands: K / lda / ldw arrive in values; the kernels count 4-byte units (the same bytes move)
        VITSEG_CHECK_ARG(a.h16 <= 2 && amode == SA_PLAIN && (epi == SE_PARTIAL || epi == SE_BIAS || epi == SE_GELU) && !a.aux, VITSEG_EINVAL,
                         "gemm_f32s: 16-bit operands serve the forward linears (plain operands; partial / bias / GELU epilogue)");
        VITSEG_CHECK_ARG(a.K % 2 == 0 && a.lda % 8 == 0 && a.ldw % 8 == 0, VITSEG_ESHAPE, "gemm_f32s: 16-bit K / leading dimensions");
        a.K /= 2; a.lda /= 2; a.ldw /= 2;
    }
    const int kc = amode == SA_CONV3 ? a.K : amode == SA_CONV3_ALL ? 9 * a.K : a.K / a.splits;
    VITSEG_CHECK_ARG(kc % 32 == 0 && a.K % 32 == 0 && (amode == SA_CONV3 || amode == SA_CONV3_ALL || kc * a.splits == a.K), VITSEG_ESHAPE, "gemm_f32s: K chunk %d is not a multiple of 32", kc);
    VITSEG_CHECK_ARG(a.N % 4 == 0 && a.lda % 4 == 0 && a.ldw % 4 == 0 && a.ldc % 4 == 0, VITSEG_ESHAPE, "gemm_f32s: N / leading dimensions must be multiples of 4");
    VITSEG_CHECK_ARG(amode != SA_PATCH || ((a.P == 8 || a.P == 16 || a.P == 32) && a.S % 4 == 0), VITSEG_ESHAPE, "gemm_f32s: patch size %d", a.P);
    VITSEG_CHECK_ARG((size_t)(amode == SA_TT ? a.K : a.M + 128) * a.lda * 4 < 0x7fffffffull && (size_t)(amode == SA_PLAIN_WT || amode == SA_TT ? a.K : a.N) * a.ldw * 4 < 0x7fffffffull, VITSEG_ESHAPE,
                     "gemm_f32s: operand beyond one buffer descriptor");
    a.kh = epi == SE_PARTIAL || amode == SA_CONV3_ALL ? 1 : small_pieces(a.K);   // (the whole conv is ONE chain, as gemm.hip computes it)
    VITSEG_CHECK_ARG((amode == SA_CONV3_ALL) == (epi == SE_RELU), VITSEG_EINVAL, "gemm_f32s: SE_RELU is the whole-conv epilogue");
    const long env = opt(OPT_SMALL_VARIANT);   // 1..5: a tile variant of gemm_f32s_kernel; 6, 7: the one-image kernel (where it applies)
    a.variant = env > 0 && env <= NVARIANTS + 2 ? (int)env - 1 : small_plan(a, kc, epi, amode);
    if (a.variant >= KW_FIRST && !kw_applies(a, epi, amode, a.variant - KW_FIRST + 2)) a.variant = small_plan(a, kc, epi, amode);
    if (a.variant >= KW_FIRST) {
        const int nt = a.variant - KW_FIRST + 2;
        a.tiles_m = (a.M + 31) / 32;
        a.tiles_n = a.N / (32 * nt);
        if (a.h16) return a.h16 == 2 ? launch_h16<2>(a, epi, s) : launch_h16<1>(a, epi, s);
        if (nt == 2) return epi == SE_GELU ? launch_kw<2, SE_GELU>(a, s) : launch_kw<2, SE_BIAS>(a, s);
        return epi == SE_GELU ? launch_kw<3, SE_GELU>(a, s) : launch_kw<3, SE_BIAS>(a, s);
    }
    a.tiles_m = (a.M + VARIANTS[a.variant].bm() - 1) / VARIANTS[a.variant].bm();
    a.tiles_n = (a.N + VARIANTS[a.variant].bn() - 1) / VARIANTS[a.variant].bn();
    if (a.h16) return a.h16 == 2 ? launch_h16<2>(a, epi, s) : launch_h16<1>(a, epi, s);
    if (amode == SA_CONV3) return launch_variant<SE_PARTIAL, SA_CONV3>(a, s);
    if (amode == SA_PATCH) return launch_variant<SE_PARTIAL, SA_PATCH>(a, s);
    if (amode == SA_CONV3_ALL) return launch_variant<SE_RELU, SA_CONV3_ALL>(a, s);
    if (amode == SA_TT) return launch_variant<SE_PARTIAL, SA_TT>(a, s);
    if (amode == SA_PLAIN_WT) return epi == SE_DGELU ? launch_variant<SE_DGELU, SA_PLAIN_WT>(a, s) : launch_variant<SE_PARTIAL, SA_PLAIN_WT>(a, s);
    switch (epi) {
        case SE_PARTIAL: return launch_variant<SE_PARTIAL, SA_PLAIN>(a, s);
        case SE_BIAS: return launch_variant<SE_BIAS, SA_PLAIN>(a, s);
        case SE_GELU: return launch_variant<SE_GELU, SA_PLAIN>(a, s);
    }
    set_error("gemm_f32s: epilogue %d", epi);
    return VITSEG_EINVAL;
}

}  // namespace vitseg
