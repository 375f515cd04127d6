// fp32 GEMM of the SMALL-BATCH regime (small.hpp):  C[M,N] = epi(A[M,K] . W[N,K]^T)  for M < 2048 token rows --
// the q/k/v/o projections, the MLP, the patch embedding and the 3x3 head conv of ViTSegmentationModel.forward
// (/root/reference/model/CE/classes.py:246-257, transformers/models/vit/modeling_vit.py:62-69,207-254) at the batch
// sizes the reference runs (4 x 224x224) and serves (1 image).  Exact fp32 products (v_mfma_f32_32x32x2_f32).
// Roofline: fp32 matrix pipe, 157.3 TFLOP/s; algorithmic work 2 M N K per launch.
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

template <int V, int EPI, int AMODE>
__global__ __launch_bounds__(256) void gemm_f32s_kernel(const SGemm p) {
    constexpr Variant CV = VARIANTS[V];
    constexpr int MT = CV.MT, NT = CV.NT, WGN = CV.WGN, NSTAGE = CV.NSTAGE;
    constexpr int BM = CV.bm(), BN = CV.bn();
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int PIECES = (BM + BN) / 8, PW = PIECES / 4;   // 1 KiB DMA pieces per stage / per wave
    static_assert(PIECES % 4 == 0 && (NSTAGE - 2) * PW <= 63, "piece count");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WGN, wc = wave % WGN;
    const int li = lane & 31, lh = lane >> 5;

    // block -> (row tile, column tile, chunk): row tiles fastest, so the blocks that share a W panel are neighbours in
    // the list and (xcd_remap) land on one XCD's L2; the whole A operand (< 8 MiB) lives in every L2 anyway
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = t % p.tiles_m, t1 = t / p.tiles_m;
    const int tn = t1 % p.tiles_n, sp = t1 / p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kc = AMODE == SA_CONV3 ? p.K : p.K / p.splits;   // conv3: one tap (K = D values) per chunk
    const int KT = kc / 32;
    const int k0w = sp * kc;                                    // first W column of this chunk

    // ---- DMA side: piece q = rows [8 q, 8 q + 8) of the stage (A rows first), lane l -> row 8 q + (l >> 3), chunk position
    // l & 7, which holds logical chunk (l & 7) ^ ((row >> 1) & 7) ----
    auto make_rsrc = [](const void* base, long long bytes) {
        const unsigned long long b = (unsigned long long)base;
        i32x4 r;
        r[0] = (int)(unsigned)b;
        r[1] = (int)(unsigned)((b >> 32) & 0xffffu);   // stride 0
        r[2] = (int)(unsigned)(bytes <= 0 ? 0 : (bytes < 0x7fffffffll ? bytes : 0x7fffffffll));
        r[3] = 0x00020000;
        return r;
    };
    constexpr unsigned OOB = 0x7ffffff0u;   // beyond every descriptor's range: reads as zeros
    int dy = 0, dx = 0;
    if (AMODE == SA_CONV3) {
        dy = sp / 3 - 1;
        dx = sp % 3 - 1;
    }
    unsigned voff[PW];
    bool is_a[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int q = wave * PW + i;
        const int row = 8 * q + (lane >> 3), pos = lane & 7;
        is_a[i] = q < BM / 8;   // wave-uniform
        if (is_a[i]) {
            const int chunk = pos ^ ((row >> 1) & 7);
            if (AMODE == SA_PLAIN) {
                voff[i] = (unsigned)row * (unsigned)p.lda * 4u + (unsigned)(chunk * 16);
            } else if (AMODE == SA_CONV3) {
                const int r = m0 + row;
                const int rem = r % p.Np, y = rem / p.g + dy, x = rem % p.g + dx;
                const bool ok = r < p.M && y >= 0 && y < p.g && x >= 0 && x < p.g;
                voff[i] = ok ? (unsigned)(r + dy * p.g + dx) * (unsigned)p.lda * 4u + (unsigned)(chunk * 16) : OOB;
            } else {
                const int r = m0 + row;
                const int b = r / p.Np, rem = r % p.Np, gy = rem / p.g, gx = rem % p.g;
                const int e4 = 4 * chunk;
                voff[i] = r < p.M ? (unsigned)(((b * p.Cin) * p.S + gy * p.P + e4 / p.P) * p.S + gx * p.P + e4 % p.P) * 4u : OOB;
            }
        } else {
            const int wrow = row - BM;
            voff[i] = (unsigned)wrow * (unsigned)p.ldw * 4u + (unsigned)((pos ^ ((wrow >> 1) & 7)) * 16);
        }
    }
    const i32x4 ra = AMODE == SA_PLAIN ? make_rsrc(p.A + (size_t)m0 * p.lda, (long long)(p.M - m0) * p.lda * 4)
                     : AMODE == SA_CONV3 ? make_rsrc(p.A, (long long)p.M * p.lda * 4)
                                         : make_rsrc(p.A, (long long)(p.M / p.Np) * p.Cin * p.S * p.S * 4);
    const i32x4 rw = make_rsrc(p.W + (size_t)n0 * p.ldw, (long long)(p.N - n0) * p.ldw * 4);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    // scalar byte offsets of K step kt: W columns k0w + 32 kt; A by mode
    auto soff_a = [&](int kt) -> unsigned {
        if (AMODE == SA_PLAIN) return (unsigned)(k0w + 32 * kt) * 4u;
        if (AMODE == SA_CONV3) return (unsigned)(32 * kt) * 4u;
        const int k = k0w + 32 * kt, pp = p.P * p.P;      // patch: k = (c, py, px); 32 | P^2 and P | 32 kt (P = 8, 16, 32)
        return (unsigned)((k / pp) * p.S * p.S + ((k % pp) / p.P) * p.S) * 4u;
    };
#define F32S_DMA(dst, voff_, rsrc, soff)                                                               \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"            \
                 :: "s"(dst), "v"(voff_), "s"(rsrc), "s"(soff) : "memory")
    // piece i of this wave for K step kt into ring stage kt % NSTAGE
    auto dma_piece = [&](int kt, int i) {
        const unsigned sb = lds_base + (unsigned)((kt % NSTAGE) * STAGE) + (unsigned)((wave * PW + i) * 1024);
        if (is_a[i]) F32S_DMA(sb, voff[i], ra, soff_a(kt));
        else F32S_DMA(sb, voff[i], rw, (unsigned)(k0w + 32 * kt) * 4u);
    };
    auto dma_step = [&](int kt) {
#pragma unroll
        for (int i = 0; i < PW; ++i) dma_piece(kt, i);
    };

    // ---- fragment side: lane (li, lh) reads 16-byte chunk (2 j + lh) ^ sw of row li of each 32-row MFMA tile ----
    const int sw = (li >> 1) & 7;
    int offj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) offj[j] = li * 128 + (((2 * j + lh) ^ sw) << 4);
    f32x4 fa[2][MT], fw[2][NT];
    auto read_frags = [&](int stage, int j, int slot) {
        const int ab = stage * STAGE + wr * MT * 4096 + offj[j];
        const int wb = stage * STAGE + BM * 128 + wc * NT * 4096 + offj[j];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fa[slot][mt] = *(const f32x4*)(lds + ab + mt * 4096);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) fw[slot][nt] = *(const f32x4*)(lds + wb + nt * 4096);
    };
    f32x16 acc[MT][NT];   // acc[mt][nt][r] = C[m = 32 mt + li][n = 32 nt + (r & 3) + 8 (r >> 2) + 4 lh]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    // chunk pair j (k = 8 j + 4 lh + e, e = 0..3) = four clusters of MT x NT MFMAs.  The DMA pieces of the step being
    // prefetched go BETWEEN the clusters of groups 0..2, one at a time: issuing a piece takes about what one 64-cycle MFMA
    // covers -- issued in a row at the top of the step (the first version of this loop) they left the matrix pipe idle for
    // ~300 of a 1024-cycle step.  Piece i follows cluster (12 i) / PW of the step.
    auto mfma_group = [&](int slot, int j, int kt_dma) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[slot][nt][e], fa[slot][mt][e], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < PW; ++i)
                if ((12 * i) / PW == 4 * j + e && kt_dma >= 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    dma_piece(kt_dma, i);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    };
    auto wait_landed = [&](int ahead) {   // all but the youngest `ahead` K steps of this wave's DMA have landed
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(2 * PW) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    };

    // ---- prologue: steps 0 .. NSTAGE - 2 in flight, first fragments of step 0 ----
#pragma unroll
    for (int k = 0; k < NSTAGE - 1; ++k)
        if (k < KT) dma_step(k);
    wait_landed(min(NSTAGE - 2, KT - 1));
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_frags(0, 0, 0);

    for (int kt = 0; kt < KT; ++kt) {
        const int st = kt % NSTAGE;
        // every wave has passed the barrier of step kt - 1, i.e. finished reading stage (kt - 1) % NSTAGE: it is refilled
        // with step kt + NSTAGE - 1 during groups 0..2
        const int kd = kt + NSTAGE - 1 < KT ? kt + NSTAGE - 1 : -1;
        read_frags(st, 1, 1);
        mfma_group(0, 0, kd);
        read_frags(st, 2, 0);
        mfma_group(1, 1, kd);
        read_frags(st, 3, 1);
        mfma_group(0, 2, kd);
        if (kt + 1 < KT) {
            // step kt + 1 has landed for this wave and, behind the barrier, for every wave; all reads of stage st precede it
            __builtin_amdgcn_sched_barrier(0);
            wait_landed(min(kt + NSTAGE - 1, KT - 1) - (kt + 1));
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            read_frags((kt + 1) % NSTAGE, 0, 0);
        }
        mfma_group(1, 3, -1);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // the ring is free: every wave parks its tiles in its own 4 KiB of it
#undef F32S_DMA

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
            if (EPI != SE_PARTIAL && gcol < p.N) bias4 = *(const f32x4*)(p.bias + gcol);
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = 8 * ps + rrow;
                f32x4 v = *(const f32x4*)(slab + row * 32 + ((c8 ^ (row & 7)) << 2));
                const int grow = m0 + (wr * MT + mt) * 32 + row;
                if (EPI != SE_PARTIAL) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = v[e] + bias4[e];
                        if (EPI == SE_GELU) x = gelu_erf(x);
                        v[e] = x;
                    }
                }
                if (grow < p.M && gcol < p.N) *(f32x4*)(cbase + (size_t)grow * p.ldc + gcol) = v;
            }
        }
}

// Least work on the busiest CU: a block's time is its MFMA stream (KT x MT x NT x 16 MFMAs of 64 cycles on each wave's
// SIMD) plus a fixed prologue / epilogue; co-resident blocks share the matrix pipes, so their streams add up.
int small_plan(const SGemm& a, int kc, int amode) {
    const int ncu = device_num_cus();
    double best = 1e30;
    int bv = 0;
    for (int v = 0; v < NVARIANTS; ++v) {
        const Variant& V = VARIANTS[v];
        const int tm = (a.M + V.bm() - 1) / V.bm(), tn = (a.N + V.bn() - 1) / V.bn();
        const long blocks = (long)tm * tn * a.splits;
        const long per_cu = (blocks + ncu - 1) / ncu;
        const int occ = 163840 / V.lds() < 1 ? 1 : 163840 / V.lds();
        const double stream = (double)(kc / 32) * V.MT * V.NT * 16 * 64;
        const double fixed = 7000.0 + 1500.0 * V.MT * V.NT;   // cycles: launch ramp, first loads, parked epilogue
        const double time = per_cu * stream + ((per_cu + occ - 1) / occ) * fixed;
        if (time < best) {
            best = time;
            bv = v;
        }
    }
    (void)amode;
    return bv;
}

template <int V, int EPI, int AMODE>
int launch_one(const SGemm& a, hipStream_t s) {
    constexpr int LDS = VARIANTS[V].lds();
    int dev = 0;
    static bool attr_set[64] = {};
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_f32s_kernel<V, EPI, AMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(gemm_f32s)");
        attr_set[dev] = true;
    }
    const int blocks = a.tiles_m * a.tiles_n * a.splits;
    hipLaunchKernelGGL((gemm_f32s_kernel<V, EPI, AMODE>), dim3(blocks), dim3(256), LDS, s, a);
    VITSEG_LAUNCH_CHECK("gemm_f32s");
    return VITSEG_OK;
}

template <int EPI, int AMODE>
int launch_variant(const SGemm& a, hipStream_t s) {
    switch (a.variant) {
        case 0: return launch_one<0, EPI, AMODE>(a, s);
        case 1: return launch_one<1, EPI, AMODE>(a, s);
        case 2: return launch_one<2, EPI, AMODE>(a, s);
        case 3: return launch_one<3, EPI, AMODE>(a, s);
        default: return launch_one<4, EPI, AMODE>(a, s);
    }
}

}  // namespace

int launch_gemm_f32s(SGemm a, int epi, int amode, hipStream_t s) {
    VITSEG_CHECK_ARG(a.A && a.W && a.C && a.M > 0 && a.N > 0 && a.K > 0 && a.splits >= 1, VITSEG_EINVAL, "gemm_f32s: bad arguments");
    VITSEG_CHECK_ARG(epi == SE_PARTIAL || (a.splits == 1 && a.bias), VITSEG_EINVAL, "gemm_f32s: a direct epilogue takes one chunk and a bias");
    const int kc = amode == SA_CONV3 ? a.K : a.K / a.splits;
    VITSEG_CHECK_ARG(kc % 32 == 0 && (amode == SA_CONV3 || kc * a.splits == a.K), VITSEG_ESHAPE, "gemm_f32s: K chunk %d is not a multiple of 32", kc);
    VITSEG_CHECK_ARG(a.N % 4 == 0 && a.lda % 4 == 0 && a.ldw % 4 == 0 && a.ldc % 4 == 0, VITSEG_ESHAPE, "gemm_f32s: N / leading dimensions must be multiples of 4");
    VITSEG_CHECK_ARG(amode != SA_PATCH || ((a.P == 8 || a.P == 16 || a.P == 32) && a.S % 4 == 0), VITSEG_ESHAPE, "gemm_f32s: patch size %d", a.P);
    VITSEG_CHECK_ARG((size_t)(a.M + 128) * a.lda * 4 < 0x7fffffffull && (size_t)a.N * a.ldw * 4 < 0x7fffffffull, VITSEG_ESHAPE,
                     "gemm_f32s: operand beyond one buffer descriptor");
    const long env = opt(OPT_SMALL_VARIANT);
    a.variant = env > 0 && env <= NVARIANTS ? (int)env - 1 : small_plan(a, kc, amode);
    a.tiles_m = (a.M + VARIANTS[a.variant].bm() - 1) / VARIANTS[a.variant].bm();
    a.tiles_n = (a.N + VARIANTS[a.variant].bn() - 1) / VARIANTS[a.variant].bn();
    if (amode == SA_CONV3) return launch_variant<SE_PARTIAL, SA_CONV3>(a, s);
    if (amode == SA_PATCH) return launch_variant<SE_PARTIAL, SA_PATCH>(a, s);
    switch (epi) {
        case SE_PARTIAL: return launch_variant<SE_PARTIAL, SA_PLAIN>(a, s);
        case SE_BIAS: return launch_variant<SE_BIAS, SA_PLAIN>(a, s);
        case SE_GELU: return launch_variant<SE_GELU, SA_PLAIN>(a, s);
    }
    set_error("gemm_f32s: epilogue %d", epi);
    return VITSEG_EINVAL;
}

}  // namespace vitseg
