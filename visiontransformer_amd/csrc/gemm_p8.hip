// 16-bit GEMM for the large linear layers:  C[M,N] = epi(A[M,K] . W[N,K]^T + bias)   (bf16 or IEEE-half operands,
// fp32 accumulate) -- the QKV / MLP projections of modeling_vit.py:207-254 at the BASELINE batches and the
// activation-gradient ("dgrad") GEMMs of the training step.
//
// Structure (gfx950): persistent 256x256 output tiles, BK = 64, 8 waves as 2 (M) x 4 (N), 128x64 per wave
// (128 accumulator registers), v_mfma_f32_16x16x32.  One block per CU owns the whole 160 KiB LDS:
//   * a ring of 8 "half-tiles" (16 KiB = 128 operand rows x 64 k): per K step A0 | B0 | B1 | A1, where A-half h holds
//     rows {wr * 128 + h * 64 + i} of both wave rows and B-half h columns {wc * 64 + h * 32 + i} of all four wave
//     columns, so that quadrant (ha, hb) of every wave's sub-tile needs exactly A-half ha and B-half hb;
//   * filled by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB = 8 rows per wave instruction, XOR swizzle applied to the
//     per-lane SOURCE chunk), issued SIX half-tiles ahead of their first read and waited for with counted
//     s_waitcnt vmcnt(6|8): the stream never drains inside the loop and runs on across tile boundaries, so
//     a tile's prologue is hidden behind its predecessor;
//   * a K step is 4 phases = the 4 quadrants (a0,b0) (a0,b1) (a1,b1) (a1,b0): each phase reads only the fragments
//     that change (8 A or 4 B ds_read_b128) and issues 16 MFMAs.  The two wave rows run the same program one
//     barrier interval apart, so on every SIMD one wave is in its MFMA segment while its partner reads LDS and
//     issues DMA ("ping-pong"); s_setprio pins the MFMA clusters between the raw s_barriers.
//   * The accumulators hold C TRANSPOSED (MFMA A operand = W rows, B operand = activation rows): a lane then owns 4
//     consecutive output columns of one row, which it parks as one 16-byte LDS write in a wave-private 4 KiB
//     staging slab (outside the ring, so the operand stream keeps flowing); the slab is read back row-wise for
//     the bias / GELU / residual / dropout arithmetic and whole-line global stores.  (Storing straight from the
//     accumulator layout -- 8 / 16 bytes per lane, 32 / 64 contiguous bytes per row and instruction, no LDS round
//     trip -- was measured 4-25 % slower on every training shape: partial-line writes.)
// TT = 1 is the weight-gradient form  C[M,N] = sum_k A[k][m] W[k][n]  (dW = dY^T X: both operands lie [token][column],
// the reduction runs over the token rows, split over `splitk` slices that write fp32 partial tiles): the same ring,
// phases and stream, but a half-tile is [64 tokens][128 columns] (256-byte rows, 16-byte chunk c of row r stored at
// c ^ (((r & 3) << 2) | ((r >> 2) & 3))) and the MFMA operands are gathered DOWN the columns with ds_read_b64_tr_b16
// (two per fragment; both operands see the same token order, so the products pair up); no transposed copies exist.
// Roofline: MFMA bf16/f16 dense 2.5 PFLOP/s; algorithmic work 2 M N K per launch.
#include <stdlib.h>

#include "kernels.hpp"

namespace vitseg {
namespace {

constexpr int PT = 256;                 // tile edge (M and N)
constexpr int HALF_BYTES = 16384;       // 128 rows x 128 B
constexpr int RING_BYTES = 8 * HALF_BYTES;
constexpr int STAGE_BYTES = 4096;       // per wave: [16 rows][64 cols] fp32
constexpr int P8_LDS = RING_BYTES + 8 * STAGE_BYTES;   // 163 840 B = the whole LDS of a CU

typedef __attribute__((address_space(3))) void lds_ptr_t;

template <typename T> struct Mfma16;
template <> struct Mfma16<bf16_t> {
    static __device__ __forceinline__ f32x4 run(bf16x8 a, bf16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma16<f16_t> {
    static __device__ __forceinline__ f32x4 run(bf16x8 a, bf16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

struct TileCoord {
    int m0, n0, split;
};
// Logical order of this block's `round`-th work item: the 32 blocks of one XCD (blockIdx % 8 equal) take 32 consecutive
// logical items of every round, so that they share operand panels through that XCD's L2.
__device__ __forceinline__ int logical_item(int round, int nitems) {
    const int first = round * (int)gridDim.x;
    const int live = min((int)gridDim.x, nitems - first);   // blocks that still have an item in this round
    return first + xcd_remap(min((int)blockIdx.x, live - 1), live);
}
// NT: column groups of GN tiles, row panels marching inside a group (32 consecutive = an (8 x 4)-ish patch of tiles).
__device__ __forceinline__ TileCoord tile_coord(int round, int tiles_m, int tiles_n, int gn) {
    const int t = logical_item(round, tiles_m * tiles_n);
    const int gsz = tiles_m * gn, ngroups = (tiles_n + gn - 1) / gn;
    const int grp = min(t / gsz, ngroups - 1);
    const int rem = t - grp * gsz;
    const int gcols = min(gn, tiles_n - grp * gn);
    TileCoord c;
    const int tm = rem / gcols;
    c.m0 = tm * PT;
    c.n0 = (grp * gn + rem - tm * gcols) * PT;
    c.split = 0;
    return c;
}
// TT: split-major -- the blocks that run together reduce the same token range into different output tiles
__device__ __forceinline__ TileCoord item_coord_tt(int round, int tiles_m, int tiles_n, int splits) {
    const int nt = tiles_m * tiles_n;
    const int t = logical_item(round, nt * splits);
    TileCoord c;
    c.split = t / nt;
    const int rem = t - c.split * nt;
    const int tm = rem / tiles_n;
    c.m0 = tm * PT;
    c.n0 = (rem - tm * tiles_n) * PT;
    return c;
}

template <typename T, typename OutT, int EPI, int TT>
__global__ __launch_bounds__(512) void gemm_p8_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];   // ring | per-wave staging (ONE array: see the guide)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, lq = lane >> 4;

    const int tiles_m = (p.M + PT - 1) / PT, tiles_n = p.N / PT;
    const int splits = TT ? max(p.splitk, 1) : 1;
    const int ntiles = tiles_m * tiles_n * splits;          // work items
    const int gn = (tiles_n % 4 == 0) ? 4 : (tiles_n % 3 == 0 ? 3 : (tiles_n >= 4 ? 4 : tiles_n));
    // K steps (64 deep) per item, always even: NT K / 64 (K % 128 == 0); TT an even share of the token rows per slice
    // (rows beyond K are out of the buffer range and read as zeros)
    const int KT = TT ? ((((p.K + 63) / 64 + splits - 1) / splits + 1) & ~1) : p.K / 64;
    // this block's items: blockIdx.x, blockIdx.x + gridDim.x, ...
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total_k = my_tiles * KT;            // K steps this block walks
    auto coord = [&](int round) {
        return TT ? item_coord_tt(round, tiles_m, tiles_n, splits) : tile_coord(round, tiles_m, tiles_n, gn);
    };

    // ---- DMA side: per-lane byte offsets inside a tile (constant for the whole kernel) ----
    unsigned voffA[2][2], voffW[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            if (TT) {   // piece = 4 token rows x 256 B; half h = columns [128 h, 128 h + 128) of the tile
                const int r = 4 * (2 * wave + pc) + (lane >> 4);
                const int ch = (lane & 15) ^ (((r & 3) << 2) | ((r >> 2) & 3));
                voffA[h][pc] = (unsigned)r * (unsigned)p.lda * 2u + (h * 128 + ch * 8) * 2;
                voffW[h][pc] = (unsigned)r * (unsigned)p.ldw * 2u + (h * 128 + ch * 8) * 2;
            } else {
                const int i = 16 * wave + 8 * pc + (lane >> 3);          // row inside the half-tile
                const int cpos = (lane & 7) ^ ((i >> 1) & 7);            // logical chunk stored at position lane & 7
                const int ra = (i >> 6) * 128 + h * 64 + (i & 63);
                const int rw = (i >> 5) * 64 + h * 32 + (i & 31);
                voffA[h][pc] = (unsigned)ra * (unsigned)p.lda * 2u + cpos * 16;
                voffW[h][pc] = (unsigned)rw * (unsigned)p.ldw * 2u + cpos * 16;
            }
        }
    // issue side: two cursors, because one K step G issues halves B1 / A1 of step G + 1 (cursor 0, phases 0 and 1) and
    // A0 / B0 of step G + 2 (cursor 1, phases 2 and 3); both advance once per K step.  A cursor is a pair of buffer
    // descriptors rebased to the tile (NT: rows beyond M; TT: token rows beyond K are out of range and read as zeros)
    // plus the K byte offset.
    // The DMA is issued from inline asm on purpose: hipcc then keeps its ordinary exact vmcnt bookkeeping for the
    // epilogue's loads and stores (with an LDS-DMA builtin in the kernel it waits vmcnt(0) before every use of a ds_read
    // or global-load result, which serialises the epilogue store by store); the ring is ordered by hand-counted waits.
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    struct Cursor {
        i32x4 a, w;       // raw buffer descriptors {base lo, base hi, num_records, flags}
        unsigned soff;    // NT: K byte offset inside the tile's rows (both operands); TT: byte offset of the step's token rows in A
        unsigned soffw;   // TT: ... in W (the two row strides differ)
        int ts, kt;       // item sequence number of this block, K step inside the item
        int m0, n0, tok0; // TT: tile origin and first token row of the item
    };
    Cursor cur0, cur1;
    auto make_rsrc = [](const void* base, long long bytes) {   // (once per tile; the per-step path is 32-bit, below)
        const unsigned long long b = (unsigned long long)base;
        i32x4 r;
        r[0] = (int)(unsigned)b;
        r[1] = (int)(unsigned)((b >> 32) & 0xffffu);                       // stride 0
        r[2] = (int)(unsigned)(bytes <= 0 ? 0 : (bytes < 0x7fffffffll ? bytes : 0x7fffffffll));
        r[3] = 0x00020000;
        return r;
    };
    // TT: the descriptors are rebased once per item (tile columns, first token row of the K slice) and cover everything from
    // there to the end of the matrix; the K step's token rows are the SCALAR offset of the load.  The range check of a raw
    // buffer load includes that offset (voffset + soffset + 16 > num_records reads as zeros, also with soffset beyond
    // num_records; tools/probes/soffset_range.hip, both the VGPR and the LDS-DMA form), so token rows beyond K -- the tail of
    // the last slice -- arrive as zeros and a K step costs one scalar add per operand.
    const unsigned stepA = 64u * (unsigned)p.lda * 2u, stepW = 64u * (unsigned)p.ldw * 2u;
    const int capA = (int)(0x7fffffffu / ((unsigned)p.lda * 2u)), capW = (int)(0x7fffffffu / ((unsigned)p.ldw * 2u));
    auto set_records_tt = [&](Cursor& c) {
        const int left = c.ts < my_tiles ? p.K - c.tok0 : 0;     // token rows from the item's first one to the end
        c.a[2] = left <= 0 ? 0 : (left >= capA ? 0x7fffffff : left * p.lda * 2 - c.m0 * 2);
        c.w[2] = left <= 0 ? 0 : (left >= capW ? 0x7fffffff : left * p.ldw * 2 - c.n0 * 2);
    };
    auto set_tile = [&](Cursor& c) {
        if (c.ts < my_tiles) {
            const TileCoord tc = coord(c.ts);
            if (TT) {
                c.m0 = tc.m0; c.n0 = tc.n0; c.tok0 = tc.split * KT * 64;
                c.a = make_rsrc((const T*)p.A + (long long)c.tok0 * p.lda + c.m0, 0);
                c.w = make_rsrc((const T*)p.W + (long long)c.tok0 * p.ldw + c.n0, 0);
            } else {
                c.a = make_rsrc((const T*)p.A + (size_t)tc.m0 * p.lda, (long long)(p.M - tc.m0) * p.lda * 2);
                c.w = make_rsrc((const T*)p.W + (size_t)tc.n0 * p.ldw, (long long)PT * p.ldw * 2);
            }
        } else {                               // past the end: zero-record descriptors, the DMA moves nothing
            c.m0 = c.n0 = c.tok0 = 0;
            c.a = make_rsrc(p.A, 0);
            c.w = make_rsrc(p.W, 0);
        }
        if (TT) set_records_tt(c);
    };
    auto set_cursor = [&](Cursor& c, int gstep) {   // gstep: index of the K step in this block's stream
        c.ts = gstep / KT;
        c.kt = gstep - c.ts * KT;
        c.soff = TT ? (unsigned)c.kt * stepA : (unsigned)c.kt * 128u;
        c.soffw = TT ? (unsigned)c.kt * stepW : c.soff;
        set_tile(c);
    };
    auto advance = [&](Cursor& c) {                 // next K step of the stream
        ++c.kt;
        c.soff += TT ? stepA : 128u;
        if (TT) c.soffw += stepW;
        if (c.kt == KT) {
            c.kt = 0;
            c.soff = c.soffw = 0;
            ++c.ts;
            set_tile(c);
        }
    };
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    auto issue_half = [&](const Cursor& c, int j, int region) {   // j: 0 A0, 1 B0, 2 B1, 3 A1
        const bool isA = (j == 0 || j == 3);
        const int h = (j == 0 || j == 1) ? 0 : 1;
        const unsigned dst = lds_base + region * HALF_BYTES + (16 * wave) * 128;
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            // M0 = LDS destination of the wave's 1 KiB piece (lane l lands at + 16 l); written in the same statement
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(dst + pc * 1024), "v"(isA ? voffA[h][pc] : voffW[h][pc]), "s"(isA ? c.a : c.w),
                            "s"(TT && !isA ? c.soffw : c.soff)
                         : "memory");
        }
    };

    // ---- fragment side ----
    const int sw = l15 >> 1;
    int a_rd[2], b_rd[2];
    if (TT) {
        // transposed gather: lane = 16 g + 4 q + p reads 4 tokens (8 g + 4 s + q is ITS address row) x 16 columns per
        // ds_read_b64_tr_b16; a_rd[s] / b_rd[s] = byte offset of (m|n tile 0, k step 0), tile t at ^ (t << 5)
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
        for (int sI = 0; sI < 2; ++sI) {
            const int row = 8 * g + 4 * sI + q;
            const int f = (q << 2) | ((2 * g + sI) & 3);
            a_rd[sI] = row * 256 + ((((wr << 3) | (pp >> 1)) ^ f) << 4) + (pp & 1) * 8;
            b_rd[sI] = row * 256 + ((((wc << 2) | (pp >> 1)) ^ f) << 4) + (pp & 1) * 8;
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            a_rd[ks] = (wr * 64 + l15) * 128 + (((4 * ks + lq) ^ sw) << 4);
            b_rd[ks] = (wc * 32 + l15) * 128 + (((4 * ks + lq) ^ sw) << 4);
        }
    }
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    auto tr_frag = [&](int region, int off0, int off1, int tile, int ks) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(lds + region * HALF_BYTES + ks * 8192 + (off0 ^ (tile << 5))));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(lds + region * HALF_BYTES + ks * 8192 + (off1 ^ (tile << 5))));
        const bf16x8 fr = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return fr;
    };
    bf16x8 xa[4][2];          // activation fragments of the current A-half: [m tile][k step]
    bf16x8 wx[2][2], wy[2][2];  // weight fragments of the two B-halves (roles alternate per K step)
    auto read_a = [&](int region) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (TT)
                    xa[mt][ks] = tr_frag(region, a_rd[0], a_rd[1], mt, ks);   // m tile t = chunks 2 t, 2 t + 1: ^ (t << 5)
                else
                    xa[mt][ks] = *(const bf16x8*)(lds + region * HALF_BYTES + mt * 2048 + a_rd[ks]);
            }
    };
    auto read_b = [&](bf16x8 (&w)[2][2], int region) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (TT)
                    w[nt][ks] = tr_frag(region, b_rd[0], b_rd[1], nt, ks);
                else
                    w[nt][ks] = *(const bf16x8*)(lds + region * HALF_BYTES + nt * 2048 + b_rd[ks]);
            }
    };
    f32x4 acc[8][4];          // [m tile][n tile]: rows n = 4 lq + r, column m = l15 (C transposed)
    auto zero_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto mfma_quad = [&](int ha, int hb, bf16x8 (&w)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[ha * 4 + mt][hb * 2 + nt] = Mfma16<T>::run(w[nt][ks], xa[mt][ks], acc[ha * 4 + mt][hb * 2 + nt]);
        __builtin_amdgcn_s_setprio(0);
    };
#define P8_BAR()                              \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)
#define P8_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

    // ---- epilogue of one tile (both wave rows run it in the same barrier interval) ----
    float* stage = (float*)(lds + RING_BYTES + wave * STAGE_BYTES);
    auto epilogue = [&](const TileCoord tc) {
        constexpr bool OUT16 = sizeof(OutT) == 2;
        // row side: OUT16: lane -> rows (lane >> 3) + 8 i, 8 columns 8 (lane & 7); fp32: rows (lane >> 4) + 4 i, 4 columns
        constexpr int CPL = OUT16 ? 8 : 4;                 // columns per lane
        constexpr int LPR = 64 / CPL;                      // lanes per row
        constexpr int RPI = 64 / LPR;                      // rows per instruction
        const int rrow = lane / LPR, rcol = (lane % LPR) * CPL;   // rcol: column inside the wave's 64-column slab
        // global column of slab column rcol: NT the wave owns 64 consecutive columns; TT two runs of 32 (one per B-half)
        const int gcol = TT ? tc.n0 + (rcol >> 5) * 128 + wc * 32 + (rcol & 31) : tc.n0 + wc * 64 + rcol;
        OutT* Cbase = (OutT*)p.C + (TT ? (size_t)tc.split * p.split_stride : 0);
        float bias_r[CPL];
#pragma unroll
        for (int c4 = 0; c4 < CPL / 4; ++c4) {
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) b4 = *(const f32x4*)(p.bias + gcol + 4 * c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bias_r[c4 * 4 + e] = b4[e];
                asm volatile("" : "+v"(bias_r[c4 * 4 + e]));   // consume here: the only wait for this load sits here
            }
        }
        constexpr int NI = 16 / RPI;                       // row groups per 16-row slab
        constexpr bool HAS_EXTRA = EPI == EPI_RESADD || EPI == EPI_DGELU;
        // operand of the epilogue arithmetic (residual rows / saved GELU derivative), fetched ONE slab ahead: vmcnt retires
        // in issue order, so a load issued after the previous slab's stores could only be waited for together with them
        float extra[2][NI][CPL];
        // m tile mt = (A-half mt >> 2, tile mt & 3 inside it): NT halves interleave per wave row, TT halves are 128-row runs
        auto row_of = [&](int mt, int i) {
            return tc.m0 + (TT ? (mt >> 2) * 128 + wr * 64 + (mt & 3) * 16 : wr * 128 + mt * 16) + rrow + RPI * i;
        };
        auto off_of = [&](int mt, int i) {
            const int g = row_of(mt, i);
            return (size_t)(g < p.M ? g : 0) * p.ldc + gcol;
        };
        auto load_extra = [&](int buf, int mt) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (EPI == EPI_RESADD) {
                    const f32x4 r4 = *(const f32x4*)(p.R + off_of(mt, i));
#pragma unroll
                    for (int e = 0; e < 4; ++e) extra[buf][i][e] = r4[e];
                }
                if (EPI == EPI_DGELU) {   // R = the saved 16-bit gelu'(pre-activation)
                    const uint4 u = *(const uint4*)((const T*)p.R + off_of(mt, i));
                    const unsigned uu[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        extra[buf][i][2 * e] = H16<T>::lo(uu[e]);
                        extra[buf][i][2 * e + 1] = H16<T>::hi(uu[e]);
                    }
                }
            }
        };
        constexpr bool COLSUM = EPI == EPI_DGELU && OUT16 && !TT;   // per-tile column sums of the stored values (bias gradient)
        float cs[COLSUM ? CPL : 1] = {};
        if (HAS_EXTRA) load_extra(0, 0);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            // park the 16 x 64 slab: lane writes 4 consecutive columns (n = nt * 16 + 4 lq + r) of row l15
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                *(f32x4*)(stage + l15 * 64 + (((nt * 4 + lq) ^ l15) << 2)) = acc[mt][nt];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private: no barrier needed
            float v[NI][CPL];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int row = rrow + RPI * i;
#pragma unroll
                for (int c4 = 0; c4 < CPL / 4; ++c4) {
                    const f32x4 t = *(const f32x4*)(stage + row * 64 + ((((rcol >> 2) + c4) ^ row) << 2));
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[i][c4 * 4 + e] = t[e];
                }
            }
            if (HAS_EXTRA && mt + 1 < 8) load_extra((mt + 1) & 1, mt + 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slab is free for the next m tile
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int grow = row_of(mt, i);
                const size_t o = off_of(mt, i);
                float pre[CPL] = {};
#pragma unroll
                for (int e = 0; e < CPL; ++e) {
                    float x = v[i][e] + bias_r[e];
                    if (EPI == EPI_GELU) {
                        if (p.aux) { const GeluPair gp = gelu_erf_pair_fast(x); x = gp.g; pre[e] = gp.d; }   // pre: gelu'(u), saved for the backward
                        else x = gelu_erf_fast(x);
                    }
                    if (EPI == EPI_RESADD && p.drop.thresh)
                        x = drop_keep(drop_key(p.drop.seed, p.drop.stream, grow + p.row_base), gcol + e, p.drop.thresh)
                                ? x * p.drop.scale : 0.f;
                    if (EPI == EPI_RESADD) x = extra[mt & 1][i][e] + x;
                    if (EPI == EPI_DGELU) x *= extra[mt & 1][i][e];
                    v[i][e] = x;
                    if constexpr (COLSUM) cs[e] += grow < p.M ? x : 0.f;
                }
                if constexpr (OUT16) {
                    uint4 h, ha;
                    h.x = H16<OutT>::pack2(v[i][0], v[i][1]); h.y = H16<OutT>::pack2(v[i][2], v[i][3]);
                    h.z = H16<OutT>::pack2(v[i][4], v[i][5]); h.w = H16<OutT>::pack2(v[i][6], v[i][7]);
                    ha.x = H16<T>::pack2(pre[0], pre[1]); ha.y = H16<T>::pack2(pre[2], pre[3]);
                    ha.z = H16<T>::pack2(pre[4], pre[5]); ha.w = H16<T>::pack2(pre[6], pre[7]);
                    if (grow < p.M) {
                        if (EPI == EPI_GELU && p.aux) *(uint4*)((T*)p.aux + o) = ha;
                        *(uint4*)(Cbase + o) = h;
                    }
                } else {
                    if (grow < p.M) *(f32x4*)((float*)Cbase + o) = f32x4{v[i][0], v[i][1], v[i][2], v[i][3]};
                }
            }
        }
        if constexpr (COLSUM) {
            if (p.colsum_scratch) {   // the 8 row lanes of a column group (lane >> 3) combine; partial row = (row tile, wave row)
#pragma unroll
                for (int e = 0; e < CPL; ++e) {
                    float t = cs[e];
                    t += __shfl_xor(t, 8, 64);
                    t += __shfl_xor(t, 16, 64);
                    t += __shfl_xor(t, 32, 64);
                    cs[e] = t;
                }
                if (rrow == 0) {
                    float* dst = p.colsum_scratch + (size_t)((tc.m0 / PT) * 2 + wr) * p.N + gcol;
                    *(f32x4*)dst = f32x4{cs[0], cs[1], cs[2], cs[3]};
                    *(f32x4*)(dst + 4) = f32x4{cs[CPL - 4], cs[CPL - 3], cs[CPL - 2], cs[CPL - 1]};
                }
            }
        }
    };

    // ---- prologue: fill the stream (halves 0 .. 5 of the block's sequence), first B0 fragments ----
    // stream order per K step G: A0 (region 4s), B0 (4s+1), B1 (4s+2), A1 (4s+3), s = G & 1
    set_cursor(cur0, 0);
    set_cursor(cur1, 1);
    issue_half(cur0, 0, 0); issue_half(cur0, 1, 1); issue_half(cur0, 2, 2); issue_half(cur0, 3, 3);   // step 0
    issue_half(cur1, 0, 4); issue_half(cur1, 1, 5);                                                    // step 1: A0, B0
    P8_VMCNT(8);                              // step 0's A0, B0 landed (this wave's pieces)
    P8_BAR();
    zero_acc();
    read_b(wx, 1);                            // B0 of step 0
    // cursor 0 now follows step G+1 (B1, A1), cursor 1 step G+2 (A0, B0)
    set_cursor(cur0, 1);
    set_cursor(cur1, 2);
    if (wr == 1) P8_BAR();                    // wave row 1 runs one barrier interval behind wave row 0

    TileCoord tc_cur = coord(0);
    int kt_in_tile = 0, tile_seq = 0;
    // one K step; `s` = parity of the step, (w0, w1) = (fragments holding this step's B0, the other buffer)
#define P8_KSTEP(s, w0, w1, G)                                                                       \
    do {                                                                                             \
        /* phase 0: quadrant (a0, b0) */                                                             \
        read_a(4 * (s) + 0);                                                                         \
        issue_half(cur0, 2, 4 * ((s) ^ 1) + 2);                                                         \
        P8_VMCNT(8);                                                                                 \
        P8_BAR();                                                                                    \
        mfma_quad(0, 0, w0);                                                                         \
        P8_BAR();                                                                                    \
        /* phase 1: (a0, b1) */                                                                      \
        read_b(w1, 4 * (s) + 2);                                                                     \
        issue_half(cur0, 3, 4 * ((s) ^ 1) + 3);                                                         \
        P8_VMCNT(8);                                                                                 \
        P8_BAR();                                                                                    \
        mfma_quad(0, 1, w1);                                                                         \
        P8_BAR();                                                                                    \
        /* phase 2: (a1, b1) */                                                                      \
        read_a(4 * (s) + 3);                                                                         \
        issue_half(cur1, 0, 4 * (s) + 0);                                                               \
        P8_VMCNT(6);                                                                                 \
        P8_BAR();                                                                                    \
        mfma_quad(1, 1, w1);                                                                         \
        P8_BAR();                                                                                    \
        /* phase 3: (a1, b0); the next step's B0 goes into the buffer b1 just vacated */             \
        read_b(w1, 4 * ((s) ^ 1) + 1);                                                               \
        issue_half(cur1, 1, 4 * (s) + 1);                                                               \
        advance(cur0);                                                                               \
        advance(cur1);                                                                               \
        P8_VMCNT(8);                                                                                 \
        P8_BAR();                                                                                    \
        mfma_quad(1, 0, w0);                                                                         \
        P8_BAR();                                                                                    \
    } while (0)

    for (int G = 0; G < total_k; G += 2) {
        P8_KSTEP(0, wx, wy, G);
        P8_KSTEP(1, wy, wx, G + 1);
        kt_in_tile += 2;
        if (kt_in_tile == KT) {   // tile complete (KT is even): both wave rows run the epilogue in the SAME interval
            if (wr == 0) P8_BAR();
            epilogue(tc_cur);
            if (wr == 1) P8_BAR();
            zero_acc();
            kt_in_tile = 0;
            ++tile_seq;
            if (tile_seq < my_tiles) tc_cur = coord(tile_seq);
        }
    }
    if (wr == 0) P8_BAR();
    P8_VMCNT(0);
#undef P8_KSTEP
}

}  // namespace

// true when the 8-phase kernel takes this GEMM (plain A, whole 256-column tiles, an even number of 64-deep K steps)
bool gemm_p8_applies(const GemmArgs& a, int epi) {
    if (opt(OPT_NO_P8)) return false;   // A/B against the previous kernels (tools/gemm_probe.py)
    const size_t a_bytes = ((size_t)a.M + PT) * a.lda * 2, w_bytes = (size_t)a.N * (a.ldw ? a.ldw : a.K) * 2;
    return a.M >= 2048 && a.N % PT == 0 && a.K % 128 == 0 && a.K >= 256 && a.lda % 8 == 0 && a.ldc % 8 == 0 &&
           (a.ldw == 0 || a.ldw % 8 == 0) && a_bytes < 0x7fffffffull && w_bytes < 0x7fffffffull && a.splitk <= 1 &&
           (epi == EPI_BIAS || epi == EPI_GELU || epi == EPI_RESADD || epi == EPI_DGELU);
}

int device_num_cus() {   // per device: a process may drive several
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    static int ncu[64] = {};
    if (!ncu[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        ncu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return ncu[dev];
}

// rounds of the persistent grid an NT launch with M rows takes (one 256 x 256 tile per CU and round)
int gemm_p8_rounds(int M, int N) {
    const int ncu = device_num_cus();
    return (((M + PT - 1) / PT) * (N / PT) + ncu - 1) / ncu;
}

template <typename T, typename OutT, int EPI, int TT>
static int launch_p8_one(GemmArgs a, int items, hipStream_t s) {
    int dev = 0;
    static bool attr_set[64] = {};   // the attribute is per device
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_p8_kernel<T, OutT, EPI, TT>, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(gemm_p8)");
        attr_set[dev] = true;
    }
    const int ncu = device_num_cus();
    const int grid = items < ncu ? items : ncu;
    hipLaunchKernelGGL((gemm_p8_kernel<T, OutT, EPI, TT>), dim3(grid), dim3(512), P8_LDS, s, a);
    VITSEG_LAUNCH_CHECK("gemm_p8");
    return VITSEG_OK;
}

template <typename T>
static int launch_p8_t(GemmArgs a, int epi, hipStream_t s) {
    if (a.ldw == 0) a.ldw = a.K;
    const int tiles = ((a.M + PT - 1) / PT) * (a.N / PT);
    switch (epi) {
        case EPI_BIAS: return launch_p8_one<T, T, EPI_BIAS, 0>(a, tiles, s);
        case EPI_GELU: return launch_p8_one<T, T, EPI_GELU, 0>(a, tiles, s);
        case EPI_DGELU: return launch_p8_one<T, T, EPI_DGELU, 0>(a, tiles, s);
        case EPI_RESADD: return launch_p8_one<T, float, EPI_RESADD, 0>(a, tiles, s);
    }
    set_error("gemm_p8: unsupported epilogue %d", epi);
    return VITSEG_EINVAL;
}

int launch_gemm_p8(const GemmArgs& a, int epi, hipStream_t s, bool f16) {
    if (gemm_h16p_applies(a, epi)) return launch_gemm_h16p(a, s, f16);   // bias epilogue, short K (VITSEG_NO_H16P=1: this file's kernel)
    return f16 ? launch_p8_t<f16_t>(a, epi, s) : launch_p8_t<bf16_t>(a, epi, s);
}

// ---- weight gradients:  dW[M,N] (fp32, dense) = A^T . W,  A = [K tokens][M], W = [K tokens][N] bf16 row-major ----
// Work items = output tiles x K slices, as close to one per CU as an even number of 64-token steps per slice allows;
// each slice stores its fp32 partial tile and splitk_reduce sums the slices in a fixed order (deterministic).
bool wgrad_p8_applies(const GemmArgs& a) {
    if (opt(OPT_NO_P8)) return false;
    return a.M % PT == 0 && a.N % PT == 0 && a.K >= 1024 && a.lda % 8 == 0 && a.ldw % 8 == 0 && a.ldc == a.N &&
           (size_t)a.K * a.lda * 2 < 0x7fffffffull && (size_t)a.K * a.ldw * 2 < 0x7fffffffull;
}
int wgrad_p8_splits(int M, int N, int K) {
    const int tiles = (M / PT) * (N / PT);
    int splits = device_num_cus() / tiles;
    const int ksteps = (K + 63) / 64;
    if (splits > ksteps / 8) splits = ksteps / 8;   // >= 8 K steps per slice
    return splits < 1 ? 1 : splits;
}
int launch_wgrad_p8(GemmArgs a, float* scratch, hipStream_t s) {
    const int tiles = (a.M / PT) * (a.N / PT);
    const int splits = wgrad_p8_splits(a.M, a.N, a.K);
    VITSEG_CHECK_ARG(splits == 1 || scratch, VITSEG_EINVAL, "wgrad_p8: split-K needs scratch");
    a.splitk = splits;
    a.split_stride = (size_t)a.M * a.N;
    a.bias = nullptr;
    float* out = (float*)a.C;
    if (splits > 1) a.C = scratch;
    if (int rc = launch_p8_one<bf16_t, float, EPI_BIAS, 1>(a, tiles * splits, s)) return rc;
    if (splits > 1) return launch_splitk_reduce(scratch, out, (size_t)a.M * a.N / 4, splits, s);
    return VITSEG_OK;
}

}  // namespace vitseg
