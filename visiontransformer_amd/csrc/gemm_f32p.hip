// fp32 GEMM for the large linear layers:  C[M,N] = epi(A[M,K] . W[N,K]^T + bias)   (exact fp32 products,
// v_mfma_f32_32x32x2_f32) -- the q/k/v/o projections and the MLP of modeling_vit.py:207-254 on the parity path
// (BASELINE configs[1], VITSEG_F32).  Roofline: fp32 matrix pipe, 157.3 TFLOP/s; algorithmic work 2 M N K per launch.
//
// Why a second fp32 kernel (gemm.hip's 128x128 one stays for the gathering loaders, the T-forms and small shapes):
// that kernel keeps the pipe 0.74-0.77 busy (profiles/r02_pmc_bench_f32.json).  It runs two independent 4-wave blocks
// per CU that stay in lockstep (identical tiles), so both are in their prologue / epilogue / staging waits at the same
// time, its operand prefetch is one K step deep (an HBM-fed operand such as fc2's 403 MB activation is late), and the
// two blocks' barriers couple across SIMDs.  Here:
//   * persistent: one 256-thread block per CU walks its tiles (grid = CUs); 4 waves as 2 (M) x 2 (N), ONE wave per
//     SIMD, 128 x 64 of the 256 x 128 block tile per wave = 4 x 2 MFMA tiles = 128 accumulator registers.  A 64-cycle
//     fp32 MFMA leaves ~12 issue slots per gap; the loop needs < 1 (6 ds_read_b128 + 4 DMA pieces per 32 MFMAs), so a
//     single wave keeps its pipe busy without a partner;
//   * LDS = a ring of 3 K steps (32 floats = 128-byte rows, 48 KiB per step: A 256 rows | W 128 rows) filled by LDS-DMA
//     (buffer_load_dwordx4 ... lds, XOR swizzle on the per-lane SOURCE chunk) TWO steps ahead, waited for with a
//     counted vmcnt(12); the stream runs on across tile boundaries, so a tile has no prologue;
//   * one barrier per K step, placed before the last MFMA group (rotated loop: the next step's first fragments are read
//     behind the barrier under 32 MFMAs);
//   * the accumulators hold C TRANSPOSED (MFMA A operand = W rows): a lane owns 4 consecutive columns of a row, parks
//     them with one ds_write_b128 in a wave-private 4 KiB slab (outside the ring) and the slab is re-read row-wise, so
//     global traffic is whole 128-byte lines.
// The K order of every dot product is the one of gemm.hip's kernel (k = 32 kt + 8 j + 4 h + e), so results are
// bit-identical to it -- and independent of the batch size.
#include <stdlib.h>

#include <type_traits>

#include "kernels.hpp"

namespace vitseg {
namespace {

constexpr int FM = 256, FN = 128, FK = 32;
constexpr int A_BYTES = FM * 128, W_BYTES = FN * 128, STAGE_BYTES = A_BYTES + W_BYTES;   // 48 KiB
constexpr int NSTAGE = 3;
constexpr int RING_BYTES = NSTAGE * STAGE_BYTES;                                          // 144 KiB
constexpr int SLAB_BYTES = 4096;                                                          // per wave: [32][32] fp32
constexpr int F32P_LDS = RING_BYTES + 4 * SLAB_BYTES;                                     // 160 KiB

typedef int i32x4 __attribute__((ext_vector_type(4)));

#define F32P_SB() __builtin_amdgcn_sched_barrier(0)

// DROP: hidden dropout on the residual branch compiled in (fp32 training forward)
// AUX: the GELU epilogue also stores the pre-activation (fp32 training forward)
// INL: a finished tile is written under the first K step of the block's next tile (kstep_first; GELU's ~18 VALU per
//      value are twice what that step's MFMAs hide, so the step is VALU-bound there)
template <int EPI, bool DROP, bool AUX, bool INL>
__global__ __launch_bounds__(256) void gemm_f32p_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];   // ring | per-wave slabs
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    const int tiles_m = (p.M + FM - 1) / FM, tiles_n = p.N / FN;
    const int ntiles = tiles_m * tiles_n;
    const int KT = p.K / FK;
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    // Tile order: column groups of `gn` tiles, row panels marching inside a group; the 32 blocks of one XCD
    // (blockIdx % 8 equal) take 32 consecutive items of every round, so an A panel slice is fetched into that L2 once
    // per group and the group's W panels stay there.
    const int gn = p.gn > 0 ? min(p.gn, tiles_n) : (tiles_n % 6 == 0 ? 6 : (tiles_n % 4 == 0 ? 4 : min(tiles_n, 6)));
    auto coord = [&](int seq, int& m0, int& n0) {
        const int first = seq * (int)gridDim.x;
        const int live = min((int)gridDim.x, ntiles - first);
        const int t = first + xcd_remap(min((int)blockIdx.x, live - 1), live);
        const int gsz = tiles_m * gn, ngroups = (tiles_n + gn - 1) / gn;
        const int grp = min(t / gsz, ngroups - 1);
        const int rem = t - grp * gsz;
        const int gcols = min(gn, tiles_n - grp * gn);
        const int tm = rem / gcols;
        m0 = tm * FM;
        n0 = (grp * gn + rem - tm * gcols) * FN;
    };

    // ---- DMA side ----
    // piece = 1 KiB = 8 rows x 128 B per wave instruction; lane l lands at + 16 l: row l >> 3, chunk position l & 7, which
    // holds logical chunk (l & 7) ^ ((row >> 1) & 7).  Wave w fills A rows [64 w, 64 w + 64) and W rows [32 w, 32 w + 32).
    unsigned voffA[8], voffW[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 64 * wave + 8 * i + (lane >> 3);
        voffA[i] = (unsigned)row * (unsigned)p.lda * 4u + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 32 * wave + 8 * i + (lane >> 3);
        voffW[i] = (unsigned)row * (unsigned)p.ldw * 4u + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
    }
    auto make_rsrc = [](const void* base, long long bytes) {
        const unsigned long long b = (unsigned long long)base;
        i32x4 r;
        r[0] = (int)(unsigned)b;
        r[1] = (int)(unsigned)((b >> 32) & 0xffffu);   // stride 0
        r[2] = (int)(unsigned)(bytes <= 0 ? 0 : (bytes < 0x7fffffffll ? bytes : 0x7fffffffll));
        r[3] = 0x00020000;
        return r;
    };
    // cursor of the DMA stream: K step `kt` of this block's `ts`-th tile; rows beyond M / N are out of the
    // descriptor's range and read as zeros, steps beyond the block's last tile move nothing
    i32x4 ca, cw;
    unsigned csoff = 0;
    int cts = 0, ckt = 0;
    auto set_tile = [&]() {
        if (cts < my_tiles) {
            int m0, n0;
            coord(cts, m0, n0);
            ca = make_rsrc((const float*)p.A + (size_t)m0 * p.lda, (long long)(p.M - m0) * p.lda * 4);
            cw = make_rsrc((const float*)p.W + (size_t)n0 * p.ldw, (long long)(p.N - n0) * p.ldw * 4);
        } else {
            ca = make_rsrc(p.A, 0);
            cw = make_rsrc(p.W, 0);
        }
    };
    auto advance = [&]() {
        ++ckt;
        csoff += 128u;
        if (ckt == KT) {
            ckt = 0;
            csoff = 0;
            ++cts;
            set_tile();
        }
    };
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    // the DMA is issued from inline asm and ordered by hand-counted waits: hipcc then keeps its exact vmcnt
    // bookkeeping for the epilogue's own loads and stores (see gemm_p8.hip)
#define F32P_DMA(dst, voff, rsrc)                                                                       \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"             \
                 :: "s"(dst), "v"(voff), "s"(rsrc), "s"(csoff) : "memory")
    auto dma_a = [&](unsigned stage_base, int i) { F32P_DMA(stage_base + (unsigned)((wave * 8 + i) * 1024), voffA[i], ca); };
    auto dma_w = [&](unsigned stage_base, int i) {
        F32P_DMA(stage_base + (unsigned)(A_BYTES + (wave * 4 + i) * 1024), voffW[i], cw);
    };

    // ---- fragment side: lane (li, lh) reads 16-byte chunk (2 j + lh) ^ sw of row li of each 32-row MFMA tile ----
    const int sw = (li >> 1) & 7;
    int offj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) offj[j] = li * 128 + (((2 * j + lh) ^ sw) << 4);
    f32x4 fa[2][4], fw[2][2];   // [slot][tile]: activation rows (MFMA B operand), weight rows (MFMA A operand)
    auto read_frags = [&](int roff, int j, int slot) {
        const int ab = roff + wr * 16384 + offj[j];
        const int wb = roff + A_BYTES + wc * 8192 + offj[j];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) fa[slot][mt] = *(const f32x4*)(lds + ab + mt * 4096);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) fw[slot][nt] = *(const f32x4*)(lds + wb + nt * 4096);
    };
    f32x16 acc[4][2];           // acc[mt][nt][r] = C[m = 32 mt + li][n = 32 nt + (r & 3) + 8 (r >> 2) + 4 lh]
    auto zero_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    };
    auto mfma8 = [&](int slot, int e) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[slot][nt][e], fa[slot][mt][e], acc[mt][nt], 0, 0, 0);
    };

    // ---- epilogue ----
    float* slab = (float*)(lds + RING_BYTES + wave * SLAB_BYTES);
    const int rrow = lane >> 3, c8 = lane & 7;
    // park a 32 x 32 sub-tile: row li, chunk 2 q + lh at position ^ (li & 7) (conflict-free both ways), re-read row-wise:
    // lane -> rows 8 ps + rrow, columns 4 c8 .. + 3.  LDS operations of one wave execute in order, so the slab needs
    // no waits beyond the data dependences.
    auto park = [&](const f32x16& t, f32x4 (&v)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *(f32x4*)(slab + li * 32 + (((2 * q + lh) ^ (li & 7)) << 2)) = f32x4{t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = 8 * ps + rrow;
            v[ps] = *(const f32x4*)(slab + row * 32 + ((c8 ^ (row & 7)) << 2));
        }
    };
    // one output value: x = accumulator + bias -> the epilogue's function of it (the residual is added by the caller)
    auto epi_value = [&](float x, float& pre, int grow, int gcol) {
        if (EPI == EPI_GELU) {
            pre = x;                            // saved pre-activation (fp32 training)
            x = gelu_erf(x);                   // (straight-line code: common.hpp)
        }
        if (EPI == EPI_RELU) x = fmaxf(x, 0.f);
        if (EPI == EPI_RESADD && DROP)
            x = drop_keep(drop_key(p.drop.seed, p.drop.stream, grow + p.row_base), gcol, p.drop.thresh) ? x * p.drop.scale : 0.f;
        return x;
    };
    // Immediate form (whole tile at once, nothing overlapped): the block's last tile, tiles that hang over the last row
    // of C (GUARD: stores masked per row), and every tile of the GELU kernels.
    auto epilogue = [&](int m0, int n0, auto guard_tag) {
        constexpr bool GUARD = decltype(guard_tag)::value;
        const int gcol0 = n0 + wc * 64 + c8 * 4;
        f32x4 bias4[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            bias4[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias) bias4[nt] = *(const f32x4*)(p.bias + gcol0 + nt * 32);
        }
        constexpr bool HAS_R = EPI == EPI_RESADD;
        f32x4 extra[8][4];   // the whole residual tile in flight at once: one memory latency per tile, not one per sub-tile
        auto row_of = [&](int mt, int ps) { return m0 + wr * 128 + mt * 32 + 8 * ps + rrow; };
        auto off_of = [&](int mt, int nt, int ps) {
            const int g = row_of(mt, ps);
            return (size_t)(!GUARD || g < p.M ? g : 0) * p.ldc + gcol0 + nt * 32;
        };
        if (HAS_R) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) extra[t][ps] = *(const f32x4*)(p.R + off_of(t >> 1, t & 1, ps));
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int mt = t >> 1, nt = t & 1;
            f32x4 v[4];
            park(acc[mt][nt], v);
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int grow = row_of(mt, ps);
                const size_t o = off_of(mt, nt, ps);
                f32x4 pre = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pre_e = 0.f;
                    float x = epi_value(v[ps][e] + bias4[nt][e], pre_e, grow, gcol0 + nt * 32 + e);
                    pre[e] = pre_e;
                    if (EPI == EPI_RESADD) x = extra[t][ps][e] + x;
                    v[ps][e] = x;
                }
                if (!GUARD || grow < p.M) {
                    if (EPI == EPI_GELU && AUX) *(f32x4*)((float*)p.aux + o) = pre;
                    *(f32x4*)((float*)p.C + o) = v[ps];
                }
            }
        }
    };

    // ---- prologue: K steps 0 and 1 of the stream in flight, first fragments of step 0 ----
    set_tile();
    {
#pragma unroll
        for (int i = 0; i < 8; ++i) dma_a(lds_base, i);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_w(lds_base, i);
        advance();
#pragma unroll
        for (int i = 0; i < 8; ++i) dma_a(lds_base + STAGE_BYTES, i);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_w(lds_base + STAGE_BYTES, i);
        advance();
    }
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // this wave's pieces of step 0
    F32P_SB();
    __builtin_amdgcn_s_barrier();
    F32P_SB();
    zero_acc();
    read_frags(0, 0, 0);

    int roff = 0;                          // ring offset of the step being computed
    unsigned doff = 2 * STAGE_BYTES;       // ring offset the DMA of step G + 2 goes to
    auto rotate_ring = [&]() {
        roff = roff + STAGE_BYTES == RING_BYTES ? 0 : roff + STAGE_BYTES;
        doff = doff + STAGE_BYTES == RING_BYTES ? 0 : doff + STAGE_BYTES;
    };
    // One K step = 4 groups of 32 MFMAs (one 16-byte chunk pair each).  NB: the step opens with the bias loads of the
    // tile it belongs to (inline asm, covered by this step's own vmcnt(12) wait; see kstep_first).
    f32x4 nb0 = {0.f, 0.f, 0.f, 0.f}, nb1 = {0.f, 0.f, 0.f, 0.f};   // bias of the tile being computed (this lane's columns)
    auto kstep = [&](auto nb_tag, int n0) {
        constexpr bool NB = decltype(nb_tag)::value;
        const int nroff = roff + STAGE_BYTES == RING_BYTES ? 0 : roff + STAGE_BYTES;
        const unsigned dbase = lds_base + doff;
        if (NB) {
            const float* bp = p.bias + n0 + wc * 64;
            const unsigned bo = (unsigned)(c8 * 16);
            asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:128"
                         : "=&v"(nb0), "=&v"(nb1) : "v"(bo), "s"(bp) : "memory");
        }
// the group's 6 fragment reads (for the NEXT group) go one per MFMA into the first cluster: issued in a row in front of
// it they outlast the one MFMA that is in flight (-3 % measured, tools/probes/f32p_where.sh)
#define F32P_READS_UNDER_MFMAS()                                                \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0)
// (one DMA piece at a time between clusters of 8 MFMAs: issuing a piece costs ~60 cycles, what ONE 64-cycle MFMA covers;
// two in a row leave the matrix pipe idle for the second)
#define F32P_GROUP(slot, jn, D0, D1, D2, D3)                                    \
        read_frags(roff, jn, (slot) ^ 1);                                       \
        mfma8(slot, 0);                                                         \
        F32P_READS_UNDER_MFMAS();                                               \
        F32P_SB();                                                              \
        D0;                                                                     \
        F32P_SB();                                                              \
        mfma8(slot, 1);                                                         \
        F32P_SB();                                                              \
        D1;                                                                     \
        F32P_SB();                                                              \
        mfma8(slot, 2);                                                         \
        F32P_SB();                                                              \
        D2;                                                                     \
        F32P_SB();                                                              \
        mfma8(slot, 3);                                                         \
        F32P_SB();                                                              \
        D3;                                                                     \
        F32P_SB();
        F32P_GROUP(0, 1, dma_a(dbase, 0), dma_a(dbase, 1), dma_a(dbase, 2), dma_a(dbase, 3))
        F32P_GROUP(1, 2, dma_a(dbase, 4), dma_a(dbase, 5), dma_a(dbase, 6), dma_a(dbase, 7))
        F32P_GROUP(0, 3, dma_w(dbase, 0), dma_w(dbase, 1), dma_w(dbase, 2), dma_w(dbase, 3))
#undef F32P_GROUP
        advance();
        F32P_SB();
        // group 3: step G + 1 has landed for this wave (its 12 pieces are older than the 12 just issued) and, behind
        // the barrier, for every wave; every wave's reads of stage `roff` precede the barrier too
        if (NB)
            asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" : "+v"(nb0), "+v"(nb1) :: "memory");
        else
            asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        F32P_SB();
        read_frags(nroff, 0, 0);
        mfma8(1, 0);
        F32P_READS_UNDER_MFMAS();
        F32P_SB();
        mfma8(1, 1);
        mfma8(1, 2);
        mfma8(1, 3);
        F32P_SB();
        rotate_ring();
    };

    // First K step of a tile when the block's PREVIOUS tile is still in the accumulators (INL kernels: every epilogue
    // but GELU).  The MFMAs run sub-tile by sub-tile -- 16 in a row on one accumulator (a dependent fp32 MFMA chain
    // issues at full rate), the first with C = 0 -- so sub-tile t of the finished tile can be parked, finished and
    // stored right before its accumulator is reused: the epilogue costs no registers, no zeroing pass and hides behind
    // the step's 128 MFMAs.  The k order of every accumulator is that of the plain step, so results do not change.
    // Residual rows come through inline-asm loads one sub-tile ahead with hand-counted waits (a compiler-tracked load is
    // waited for with a count that ignores the DMA pieces issued behind it: thousands of cycles).
    f32x4 pb0 = {0.f, 0.f, 0.f, 0.f}, pb1 = {0.f, 0.f, 0.f, 0.f};   // bias of the pending tile
    int pm0 = 0, pn0 = 0;
    const unsigned rl0 = (unsigned)((rrow * p.ldc + c8 * 4) * 4);     // this lane's piece of row group 0 of a sub-tile
    const unsigned rstep = (unsigned)(8 * p.ldc * 4);                 // + row group
    auto kstep_first = [&]() {
        const int nroff = roff + STAGE_BYTES == RING_BYTES ? 0 : roff + STAGE_BYTES;
        const unsigned dbase = lds_base + doff;
        f32x4 ga[2][4], gw[2][4];   // [buffer][chunk pair]: the sub-tile's activation rows / weight rows
        auto read_sub = [&](int t, int buf) {
            const int ab = roff + wr * 16384 + (t >> 1) * 4096, wb = roff + A_BYTES + wc * 8192 + (t & 1) * 4096;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ga[buf][j] = *(const f32x4*)(lds + ab + offj[j]);
                gw[buf][j] = *(const f32x4*)(lds + wb + offj[j]);
            }
        };
        f32x4 rv[2][4];
        auto load_r = [&](int t, int buf) {
            const float* rb = p.R + (size_t)(pm0 + wr * 128 + (t >> 1) * 32) * p.ldc + pn0 + wc * 64 + (t & 1) * 32;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps)
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(rv[buf][ps]) : "v"(rl0 + ps * rstep), "s"(rb) : "memory");
        };
        read_sub(0, 0);
        if (EPI == EPI_RESADD) load_r(0, 0);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int mt = t >> 1, nt = t & 1, buf = t & 1;
            if (EPI == EPI_RESADD && t < 7) load_r(t + 1, buf ^ 1);
            f32x4 sv[4];
            park(acc[mt][nt], sv);
            if (t == 7) {
                // the rotated barrier of the plain step, before the last sub-tile: every wave has read stage `roff` (the
                // fragments of sub-tile 7 are in registers) and step G + 1 has landed (RESADD: the vmcnt(4) below ran)
                if (EPI == EPI_RESADD)
                    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" : "+v"(rv[1][0]), "+v"(rv[1][1]), "+v"(rv[1][2]), "+v"(rv[1][3]) :: "memory");
                else
                    // 12 DMA + 28 stores issued by this step (GELU with the saved pre-activation: 56 stores -- the count,
                    // capped by the 6-bit counter, then also waits for this step's oldest 28, which are long done)
                    asm volatile("s_waitcnt vmcnt(40) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                F32P_SB();
                read_frags(nroff, 0, 0);
            }
            F32P_SB();
            // the sub-tile's 16 MFMAs with the DMA pieces of step G + 2 between them, one at a time (2 per sub-tile,
            // sub-tiles 0..5)
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            f32x16 c = zero;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // the next sub-tile's 8 fragment reads go one per MFMA under chunk pairs 1 and 2
                if (j == 1 && t < 7) read_sub(t + 1, buf ^ 1);
#pragma unroll
                for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(gw[buf][j][e], ga[buf][j][e], c, 0, 0, 0);
                if (EPI == EPI_GELU) {
                    // GELU: ~18 VALU per value, twice the issue slots one MFMA leaves -- row group j of the parked sub-tile
                    // is finished beside chunk pair j's 4 MFMAs (~20 VALU behind each)
                    const int grow = pm0 + wr * 128 + mt * 32 + 8 * j + rrow, gcol = pn0 + wc * 64 + nt * 32 + c8 * 4;
                    f32x4 pre4, out4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float pre = 0.f;
                        out4[e] = epi_value(sv[j][e] + (nt ? pb1[e] : pb0[e]), pre, grow, gcol + e);
                        pre4[e] = pre;
                    }
                    const size_t o = (size_t)grow * p.ldc + gcol;
                    if (AUX) *(f32x4*)((float*)p.aux + o) = pre4;
                    *(f32x4*)((float*)p.C + o) = out4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (j == 2 && t < 7) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 20, 0);
                    }
                } else if (j == 2 && t < 7) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                if (j == 0 || j == 2) {
                    const int pc = 2 * t + (j >> 1);   // pieces 0..7 of A, then 0..3 of W
                    F32P_SB();
                    if (t < 4) dma_a(dbase, pc); else if (t < 6) dma_w(dbase, pc - 8);
                    F32P_SB();
                }
            }
            acc[mt][nt] = c;
            // finish and store the parked sub-tile of the previous tile (VALU + 4 stores under the MFMAs above)
            const int row0 = pm0 + wr * 128 + mt * 32, col0 = pn0 + wc * 64 + nt * 32;
            if (EPI == EPI_RESADD && t < 7) {
                // R(t) is older than: DMA of sub-tile t - 1 (2), stores of t - 1 (4), R(t + 1) (4), DMA of t (2)
                if (t == 0)
                    asm volatile("s_waitcnt vmcnt(6)" : "+v"(rv[buf][0]), "+v"(rv[buf][1]), "+v"(rv[buf][2]), "+v"(rv[buf][3]) :: "memory");
                else if (t < 6)
                    asm volatile("s_waitcnt vmcnt(12)" : "+v"(rv[buf][0]), "+v"(rv[buf][1]), "+v"(rv[buf][2]), "+v"(rv[buf][3]) :: "memory");
                else
                    asm volatile("s_waitcnt vmcnt(10)" : "+v"(rv[buf][0]), "+v"(rv[buf][1]), "+v"(rv[buf][2]), "+v"(rv[buf][3]) :: "memory");
            }
            char* cb = (char*)((float*)p.C + (size_t)row0 * p.ldc + col0);
#pragma unroll
            for (int ps = 0; ps < 4 && EPI != EPI_GELU; ++ps) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pre = 0.f;
                    float x = epi_value(sv[ps][e] + (nt ? pb1[e] : pb0[e]), pre, row0 + 8 * ps + rrow, col0 + c8 * 4 + e);
                    if (EPI == EPI_RESADD) x = rv[buf][ps][e] + x;
                    sv[ps][e] = x;
                }
                *(f32x4*)(cb + rl0 + ps * rstep) = sv[ps];
            }
            F32P_SB();
        }
        advance();
        rotate_ring();
    };

    bool pending = false;
    for (int ts = 0; ts < my_tiles; ++ts) {
        int m0, n0;
        coord(ts, m0, n0);
        if (pending) kstep_first(); else kstep(std::false_type{}, 0);
        if (INL && p.bias) kstep(std::true_type{}, n0); else kstep(std::false_type{}, 0);
        for (int kt = 2; kt < KT; ++kt) kstep(std::false_type{}, 0);
        if (INL && m0 + FM <= p.M && ts + 1 < my_tiles) {
            pending = true;          // written under the first K step of the block's next tile
            pm0 = m0;
            pn0 = n0;
            pb0 = nb0;
            pb1 = nb1;
        } else {
            pending = false;
            if (m0 + FM <= p.M) epilogue(m0, n0, std::false_type{}); else epilogue(m0, n0, std::true_type{});
            zero_acc();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may land after the block has left the CU
#undef F32P_DMA
}

template <int EPI, bool DROP, bool AUX, bool INL>
int launch_f32p_one_i(const GemmArgs& a, hipStream_t s) {
    int dev = 0;
    static bool attr_set[64] = {};
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_f32p_kernel<EPI, DROP, AUX, INL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           F32P_LDS);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(gemm_f32p)");
        attr_set[dev] = true;
    }
    const int tiles = ((a.M + FM - 1) / FM) * (a.N / FN);
    const int ncu = device_num_cus();
    hipLaunchKernelGGL((gemm_f32p_kernel<EPI, DROP, AUX, INL>), dim3(tiles < ncu ? tiles : ncu), dim3(256), F32P_LDS, s, a);
    VITSEG_LAUNCH_CHECK("gemm_f32p");
    return VITSEG_OK;
}

template <int EPI, bool DROP = false, bool AUX = false>
int launch_f32p_one(const GemmArgs& a, hipStream_t s) {
    if (opt(OPT_F32P_NOINL) || a.K < 4 * FK)   // (option: experiments, every epilogue at its tile's end)
        return launch_f32p_one_i<EPI, DROP, AUX, false>(a, s);
    return launch_f32p_one_i<EPI, DROP, AUX, true>(a, s);
}

}  // namespace

// true when the persistent kernel takes this fp32 GEMM: plain A, whole 128-column tiles, whole 32-float K steps,
// enough tiles to fill the chip, operands addressable through one 2 GiB buffer descriptor
bool gemm_f32p_applies(const GemmArgs& a, int epi) {
    if (opt(OPT_NO_F32P)) return false;   // A/B against gemm.hip's kernel (tools/gemm_probe.py)
    const int ldw = a.ldw ? a.ldw : a.K;
    const size_t a_bytes = ((size_t)a.M + FM) * a.lda * 4, w_bytes = (size_t)a.N * ldw * 4;
    const int tiles = ((a.M + FM - 1) / FM) * (a.N / FN);
    return tiles >= 128 && a.N % FN == 0 && a.K % FK == 0 && a.K >= 3 * FK && a.lda % 4 == 0 && a.ldc % 4 == 0 &&
           ldw % 4 == 0 && a_bytes < 0x7fffffffull && w_bytes < 0x7fffffffull && a.splitk <= 1 &&
           (epi == EPI_BIAS || epi == EPI_GELU || epi == EPI_RESADD || epi == EPI_RELU);
}

int launch_gemm_f32p(const GemmArgs& a_in, int epi, hipStream_t s) {
    GemmArgs a = a_in;
    if (a.ldw == 0) a.ldw = a.K;
    switch (epi) {
        case EPI_BIAS: return launch_f32p_one<EPI_BIAS>(a, s);
        case EPI_GELU: return a.aux ? launch_f32p_one<EPI_GELU, false, true>(a, s) : launch_f32p_one<EPI_GELU>(a, s);
        case EPI_RESADD:
            return a.drop.thresh ? launch_f32p_one<EPI_RESADD, true>(a, s) : launch_f32p_one<EPI_RESADD>(a, s);
        case EPI_RELU: return launch_f32p_one<EPI_RELU>(a, s);
    }
    set_error("gemm_f32p: unsupported epilogue %d", epi);
    return VITSEG_EINVAL;
}

}  // namespace vitseg
