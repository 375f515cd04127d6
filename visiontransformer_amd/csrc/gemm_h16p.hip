// 16-bit GEMM with a bias epilogue and 16-bit output, second generation:  C[M,N] = A[M,K] . W[N,K]^T + bias   (bf16 or
// IEEE-half operands, fp32 accumulate) -- the fused QKV projection of modeling_vit.py:207-222 and the activation-gradient
// GEMMs with a short reduction.  Roofline: MFMA bf16/f16 dense 2.5 PFLOP/s; algorithmic work 2 M N K per launch.
//
// Why a second kernel next to gemm_p8.hip: tools/probes/p8_where.sh splits that kernel's time into a main loop at
// 0.51-0.60 of the matrix peak and an epilogue that adds 25-90 % on top (QKV at batch 64: 183 us + 47 us).
//   * The epilogue is expensive because all 256 CUs reach it together (same tile schedule): their stores arrive as one
//     burst (33 MB in ~3.5 us), and every wave then sits in the next counted vmcnt until its stores are acknowledged
//     (loads and stores retire through one in-order counter).  Here a finished tile leaves UNDER the first K step of the
//     block's next tile: that step runs sub-tile pair by sub-tile pair, each pair is parked in LDS right before its
//     accumulators restart from C = 0, the bias add / rounding / stores sit between the MFMAs (bf16 MFMAs and VALU
//     overlap, profiles/r03_simd_overlap_probe.txt), and the counted waits of that step and the next one leave room for
//     the stores to stay in flight.  QKV at batch 64: 228 -> 201 us, batch 32: 115 -> 104 us (tools/probes/p8_where.sh).
//   * That needs the accumulators of a sub-tile to be free when its restart comes, i.e. ONE wave per SIMD owning
//     128 x 128 (256 accumulator registers = the AGPR file, v_mfma_f32_32x32x16) instead of two waves of 128 x 64.
//     The main loop is not faster for it (both are bound by the LDS: fragment reads + DMA writes of a 256 x 256 x 64
//     step take 0.75-1.0 of its 2048 matrix cycles), and VALU-heavy epilogues lose: one wave issues a vector instruction
//     every 4 cycles, two waves every 2 -- GELU (+ saved derivative), the residual and dGELU forms were built on this
//     structure, measured 3-8 % slower than gemm_p8.hip and removed; they stay there, as does the TT weight-gradient form.
// Structure shared with gemm_p8.hip: persistent 256 x 256 tiles in XCD-aware order, a ring of 8 half-tiles (16 KiB = 128
// rows x 64 k; per K step A0 | B0 | B1 | A1) filled by LDS-DMA with the XOR swizzle on the per-lane source chunk, a K
// step of 4 phases = the quadrants (a0,b0) (a0,b1) (a1,b1) (a1,b0), each reading only the half that changes.
// One barrier per phase, in front of its last MFMA.  Phase P reads the fragments of half P + 2 (for phase P + 1 or later),
// issues the DMA of half P + 8 into the slot of half P, and ends with vmcnt(20): half P + 3 has landed.
#include <stdlib.h>

#include <type_traits>

#include "kernels.hpp"

namespace vitseg {
namespace {

constexpr int QT = 256;                  // tile edge
constexpr int QHALF = 16384;             // half-tile: 128 rows x 128 B
constexpr int QRING = 8 * QHALF;         // 128 KiB
constexpr int QSLAB = 8192;              // per wave: two [32][32] fp32 sub-tiles
constexpr int Q_LDS = QRING + 4 * QSLAB; // 160 KiB

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mfma32;
template <> struct Mfma32<bf16_t> {
    static __device__ __forceinline__ f32x16 run(bf16x8 a, bf16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma32<f16_t> {
    static __device__ __forceinline__ f32x16 run(bf16x8 a, bf16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
// first MFMA of an accumulator's new tile: C = 0, written IN PLACE ("+a": the old and the new accumulator are one
// register tuple for the allocator -- a builtin call with a zero C defines a new value, which it placed elsewhere and
// then spilled accumulators around the 256-register AGPR file)
template <typename T>
__device__ __forceinline__ void mfma32_restart(f32x16& acc, bf16x8 a, bf16x8 b) {
    if (std::is_same<T, bf16_t>::value)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "+a"(acc) : "v"(a), "v"(b));
    else
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "+a"(acc) : "v"(a), "v"(b));
}

#define Q_SB() __builtin_amdgcn_sched_barrier(0)
#define Q_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)" ::: "memory")

// vmcnt budget of a phase-end wait: 20 = the DMA pieces of the 5 newest halves, + the stores of an in-loop epilogue that
// are younger than the half that must have landed (the counter holds 63)
constexpr int q_allow(int extra) { return 20 + extra > 63 ? 63 : 20 + extra; }

template <typename T>
__global__ __launch_bounds__(256) void gemm_h16p_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];   // ring | per-wave slabs
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    const int tiles_m = (p.M + QT - 1) / QT, tiles_n = p.N / QT;
    const int ntiles = tiles_m * tiles_n;
    const int KT = p.K / 64;   // even (K % 128 == 0)
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int gn = (tiles_n % 4 == 0) ? 4 : (tiles_n % 3 == 0 ? 3 : (tiles_n >= 4 ? 4 : tiles_n));
    // tile order: column groups of gn tiles, row panels marching inside a group; the 32 blocks of one XCD take 32
    // consecutive items of every round (gemm_p8.hip)
    auto coord = [&](int seq, int& m0, int& n0) {
        const int first = seq * (int)gridDim.x;
        const int live = min((int)gridDim.x, ntiles - first);
        const int t = first + xcd_remap(min((int)blockIdx.x, live - 1), live);
        const int gsz = tiles_m * gn, ngroups = (tiles_n + gn - 1) / gn;
        const int grp = min(t / gsz, ngroups - 1);
        const int rem = t - grp * gsz;
        const int gcols = min(gn, tiles_n - grp * gn);
        const int tm = rem / gcols;
        m0 = tm * QT;
        n0 = (grp * gn + rem - tm * gcols) * QT;
    };

    // ---- DMA side ----
    // half-tile row i (0..127): A-half h = tile row (i >> 6) * 128 + h * 64 + (i & 63) (both wave rows), B-half h the same
    // with wave columns.  Piece = 8 rows x 128 B per wave instruction; wave w fills rows [32 w, 32 w + 32) of every half.
    unsigned voffA[2][4], voffW[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            const int i = 32 * wave + 8 * pc + (lane >> 3);
            const int cpos = (lane & 7) ^ ((i >> 1) & 7);   // logical chunk stored at position lane & 7
            const int row = (i >> 6) * 128 + h * 64 + (i & 63);
            voffA[h][pc] = (unsigned)row * (unsigned)p.lda * 2u + cpos * 16;
            voffW[h][pc] = (unsigned)row * (unsigned)p.ldw * 2u + cpos * 16;
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
    // cursor of the DMA stream: K step ckt of this block's cts-th tile (rows beyond M read as zeros, steps beyond the
    // block's last tile move nothing)
    i32x4 ca, cw;
    unsigned csoff = 0;
    int cts = 0, ckt = 0;
    auto set_tile = [&]() {
        if (cts < my_tiles) {
            int m0, n0;
            coord(cts, m0, n0);
            ca = make_rsrc((const T*)p.A + (size_t)m0 * p.lda, (long long)(p.M - m0) * p.lda * 2);
            cw = make_rsrc((const T*)p.W + (size_t)n0 * p.ldw, (long long)QT * p.ldw * 2);
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
    // (inline asm + hand-counted waits: see gemm_p8.hip)
#define Q_DMA(dst, voff, rsrc)                                                                          \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"             \
                 :: "s"(dst), "v"(voff), "s"(rsrc), "s"(csoff) : "memory")
    // piece pc of the half of kind J (0 A0, 1 B0, 2 B1, 3 A1) of the cursor's K step into ring slot `slot`
    auto dma_piece = [&](int J, int slot, int pc) {   // (J, slot, pc are literals at every call site)
        const bool isA = J == 0 || J == 3;
        const int h = J >= 2 ? 1 : 0;
        const unsigned dst = lds_base + (unsigned)(slot * QHALF + (32 * wave + 8 * pc) * 128);
        if (isA) Q_DMA(dst, voffA[h][pc], ca); else Q_DMA(dst, voffW[h][pc], cw);
    };

    // ---- fragment side: lane (li, lh) reads 16-byte chunk (2 j + lh) ^ sw of row li of each 32-row MFMA tile ----
    const int sw = (li >> 1) & 7;
    int offj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) offj[j] = li * 128 + (((2 * j + lh) ^ sw) << 4);
    bf16x8 FA[2][2][4], FB[2][2][4];   // [buffer][32-row tile of the half][k16 step]
    auto read_half = [&](bf16x8 (&F)[2][4], int slot, int wsel) {
        const int base = slot * QHALF + wsel * 8192;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) F[t][j] = *(const bf16x8*)(lds + base + t * 4096 + offj[j]);
    };
    f32x16 acc[4][4];   // acc[mt][nt][r] = C[m = 32 mt + li][n = 32 nt + (r & 3) + 8 (r >> 2) + 4 lh]   (C transposed)
    auto zero_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    };

    // ---- epilogue pieces ----
    float* slab = (float*)(lds + QRING + wave * QSLAB);
    // row side of the re-read: lane -> rows 8 ps + (lane >> 3), 8 columns 8 (lane & 7) of the pair's 64
    const int rrow = lane >> 3, cg = lane & 7;
    const int sub = cg >> 2, cch = (cg & 3) * 2;   // sub-tile of the pair, first 16-byte chunk inside it
    // park sub-tile pair (mt, 2 hb), (mt, 2 hb + 1): row li, chunk 2 q + lh at position ^ (li & 7); the second sub-tile
    // lies 4 KiB behind with its chunk positions ^ 1, so that the 8 lanes of an output row (4 per sub-tile) re-read both
    // halves of its 128 bytes in one instruction without bank conflicts
    auto park2 = [&](const f32x16& t0, const f32x16& t1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *(f32x4*)(slab + li * 32 + (((2 * q + lh) ^ (li & 7)) << 2)) = f32x4{t0[4 * q], t0[4 * q + 1], t0[4 * q + 2], t0[4 * q + 3]};
            *(f32x4*)(slab + 1024 + li * 32 + (((2 * q + lh) ^ (li & 7) ^ 1) << 2)) = f32x4{t1[4 * q], t1[4 * q + 1], t1[4 * q + 2], t1[4 * q + 3]};
        }
    };
    auto reread = [&](float (&v)[4][8], int ps0, int ps1) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            if (ps < ps0 || ps >= ps1) continue;
            const int row = 8 * ps + rrow;
            const int sz = (row & 7) ^ sub;
#pragma unroll
            for (int c4 = 0; c4 < 2; ++c4) {
                const f32x4 t = *(const f32x4*)(slab + sub * 1024 + row * 32 + (((cch + c4) ^ sz) << 2));
#pragma unroll
                for (int e = 0; e < 4; ++e) v[ps][c4 * 4 + e] = t[e];
            }
        }
    };
    // the tile an epilogue works on
    unsigned ptile_off = 0;   // byte offset of the wave's 128 x 128 corner in C
    // bias of that tile at this lane's columns, per quadrant column half hb.  The loads are inline asm WRITING THESE
    // registers (no temporaries: the data arrives long after the instruction, and a temporary's registers would be handed
    // to something else in the meantime -- seen as a fault at the address the bias held); pin_bias() marks the first point
    // where they are read.
    f32x4 pbias4[2][2];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
        for (int c4 = 0; c4 < 2; ++c4) pbias4[hb][c4] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load_bias = [&](int n0) {
        if (p.bias) {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int c4 = 0; c4 < 2; ++c4) {
                    const float* bp = p.bias + n0 + wc * 128 + hb * 64 + cg * 8 + 4 * c4;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(pbias4[hb][c4]) : "v"(bp) : "memory");
                }
        }
    };
    // First point where the bias registers are read.  The loads were issued at the start of the tile's SECOND K step: at
    // least that step's 16 DMA pieces are younger than the last of them on every path, so vmcnt(16) covers them whatever K
    // is (with K >= 256, all this dispatcher admits, they landed K steps ago and the wait only asks for the ring's four
    // oldest pieces one phase early).  The wait names the registers: nothing that reads them is scheduled above it
    // (tests/test_isa_cpu.py checks the emitted code path by path).
    auto pin_bias = [&]() {
        asm volatile("s_waitcnt vmcnt(16)"
                     : "+v"(pbias4[0][0]), "+v"(pbias4[0][1]), "+v"(pbias4[1][0]), "+v"(pbias4[1][1]) :: "memory");
    };
    // C is addressed through a buffer descriptor over the whole matrix: rows beyond M are out of range and their stores
    // are dropped, so a ragged tile needs no masks.  The descriptor is EMPTY until the block has finished its first tile:
    // the first K step of every tile is the epilogue-carrying form (one loop body, one life of the accumulators), and for
    // the first tile its stores go nowhere.
    const long long c_bytes = (long long)p.M * p.ldc * 2;
    __amdgpu_buffer_rsrc_t rsrcC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0, 0x00020000);
    const unsigned lane_off = (unsigned)((rrow * p.ldc + cg * 8) * 2);
    const unsigned ldcB = (unsigned)(p.ldc * 2);
    auto set_ptile = [&](int m0, int n0) {
        ptile_off = ((unsigned)(m0 + wr * 128) * (unsigned)p.ldc + (unsigned)(n0 + wc * 128)) * 2u;
    };
    // bias add, rounding and stores of rows [ps0, ps1) of one round: v = re-read accumulators, columns of quadrant half hb
    auto finish_rows = [&](float (&v)[4][8], int mt, int hb, int ps0, int ps1) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            if (ps < ps0 || ps >= ps1) continue;
            // the whole offset goes into the VGPR operand: with an SGPR soffset hipcc places no wait state between a
            // 16-byte buffer store and a VALU write of its data registers (its hazard table exempts that form), and on
            // gfx950 the store then picked up the NEXT row group's values (seen on hardware: dwords 2-3 of every other
            // row group of a round's last stores)
            const unsigned vo = lane_off + ptile_off + (unsigned)(mt * 32 + 8 * ps) * ldcB + (unsigned)(hb * 128);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[ps][e] += pbias4[hb][e >> 2][e & 3];
            const u32x4 h = {H16<T>::pack2(v[ps][0], v[ps][1]), H16<T>::pack2(v[ps][2], v[ps][3]),
                             H16<T>::pack2(v[ps][4], v[ps][5]), H16<T>::pack2(v[ps][6], v[ps][7])};
            __builtin_amdgcn_raw_buffer_store_b128(h, rsrcC, vo, 0, 0);
        }
    };
    constexpr int ST_ROUND = 4, E_PHASE = 2 * ST_ROUND;   // stores of one round / of one phase of a first step

    // Immediate form (whole tile at once, nothing overlapped): the block's last tile.  Round = (mt, hb): 32 rows x 64
    // columns of the wave's 128 x 128.
    auto epilogue = [&]() {
        pin_bias();
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int mt = rr >> 1, hb = rr & 1;
            park2(acc[mt][2 * hb], acc[mt][2 * hb + 1]);
            float v[4][8];
            reread(v, 0, 4);
            finish_rows(v, mt, hb, 0, 4);
            Q_SB();
        }
    };

    // ---- prologue: K steps 0 and 1 of the stream in flight (halves 0..7), fragments of halves 0 (A0) and 1 (B0) ----
    set_tile();
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) dma_piece(0, 0, pc);
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) dma_piece(1, 1, pc);
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) dma_piece(2, 2, pc);
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) dma_piece(3, 3, pc);
    advance();
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) dma_piece(0, 4, pc);
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) dma_piece(1, 5, pc);
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) dma_piece(2, 6, pc);
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) dma_piece(3, 7, pc);
    advance();
    Q_WAIT(20);   // halves 0, 1, 2 (this wave's pieces)
    Q_SB();
    __builtin_amdgcn_s_barrier();
    Q_SB();
    zero_acc();
    read_half(FA[0], 0, wr);
    read_half(FB[0], 1, wc);
    Q_WAIT(20);
    Q_SB();
    __builtin_amdgcn_s_barrier();   // every wave's prologue reads are done before slot 0 is refilled in phase 0
    Q_SB();

    auto read_tile = [&](bf16x8 (&F)[2][4], int slot, int wsel, int t) {   // one 32-row tile of a half: 4 reads
        const int base = slot * QHALF + wsel * 8192 + t * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) F[t][j] = *(const bf16x8*)(lds + base + offj[j]);
    };

    // ---- one phase of a plain K step ----
    // PH: 0 (a0,b0) 1 (a0,b1) 2 (a1,b1) 3 (a1,b0); S: parity of the K step (ring slots, roles of the B buffers);
    // ALLOW: vmcnt budget of the phase-end wait.  Order: 8 MFMAs (k16 steps 0, 1) with the 8 fragment reads of half
    // P + 2 between them, then k16 steps 2, 3 as  M M D  M M D  M M D  M D W B M   (D = DMA piece of half P + 8,
    // W = counted wait, B = barrier: in front of the last MFMA, which covers the barrier's latency)
    auto phase = [&](auto ph_tag, auto s_tag, auto allow_tag) {
        constexpr int PH = decltype(ph_tag)::value, S = decltype(s_tag)::value, ALLOW = decltype(allow_tag)::value;
        constexpr int ha = PH >= 2 ? 1 : 0, hb = (PH == 1 || PH == 2) ? 1 : 0;
        constexpr int rslot = (4 * S + PH + 2) & 7, dslot = 4 * S + PH;
        bf16x8 (&Fa)[2][4] = FA[ha];
        bf16x8 (&Fb)[2][4] = FB[hb == 0 ? S : (S ^ 1)];
        // fragments of half P + 2: B1 of this step / A1 of this step / A0 of the next / B0 of the next
        if (PH == 0) read_half(FB[S ^ 1], rslot, wc);
        if (PH == 1) read_half(FA[1], rslot, wr);
        if (PH == 2) read_half(FA[0], rslot, wr);
        if (PH == 3) read_half(FB[S ^ 1], rslot, wc);
        auto mm = [&](int j, int t, int u) {
            acc[2 * ha + t][2 * hb + u] = Mfma32<T>::run(Fb[u][j], Fa[t][j], acc[2 * ha + t][2 * hb + u]);
        };
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) mm(j, t, u);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        Q_SB();
        mm(2, 0, 0); mm(2, 0, 1);
        Q_SB(); dma_piece(PH, dslot, 0); Q_SB();
        mm(2, 1, 0); mm(2, 1, 1);
        Q_SB(); dma_piece(PH, dslot, 1); Q_SB();
        mm(3, 0, 0); mm(3, 0, 1);
        Q_SB(); dma_piece(PH, dslot, 2); Q_SB();
        mm(3, 1, 0);
        Q_SB(); dma_piece(PH, dslot, 3);
        if (PH == 3) advance();
        Q_SB();
        // half P + 3 has landed for this wave (its pieces are older than the ALLOW newest vector-memory instructions) and,
        // behind the barrier, for every wave; every wave's reads of half P + 2 are complete (lgkmcnt)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(ALLOW) : "memory");
        __builtin_amdgcn_s_barrier();
        Q_SB();
        mm(3, 1, 1);
        Q_SB();
    };

    // ---- one phase of the FIRST K step of a tile: the block's previous tile is still in the accumulators ----
    // Two rounds = the two 32-row blocks of the quadrant.  A round parks its sub-tile pair, restarts the pair's
    // accumulators from C = 0 (8 MFMAs: k16 steps 0..3, the k order of the plain step) and finishes + stores the parked
    // values between those MFMAs.  Vector-memory order of a round: DMA | stores of rows 0-15 | DMA | stores of rows 16-31.
    auto phase_first = [&](auto ph_tag) {
        constexpr int PH = decltype(ph_tag)::value, S = 0;
        constexpr int ha = PH >= 2 ? 1 : 0, hb = (PH == 1 || PH == 2) ? 1 : 0;
        constexpr int rslot = (4 * S + PH + 2) & 7, dslot = 4 * S + PH;
        bf16x8 (&Fa)[2][4] = FA[ha];
        bf16x8 (&Fb)[2][4] = FB[hb == 0 ? S : (S ^ 1)];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int mt = 2 * ha + t;
            park2(acc[mt][2 * hb], acc[mt][2 * hb + 1]);
            float v[4][8];
            reread(v, 0, 2);   // (rows 16-31 are re-read behind the second DMA piece: 16 registers less)
            Q_SB();            // the parking writes read the old accumulators: they are issued before the restart
            mfma32_restart<T>(acc[mt][2 * hb], Fb[0][0], Fa[t][0]);
            mfma32_restart<T>(acc[mt][2 * hb + 1], Fb[1][0], Fa[t][0]);
            Q_SB();
            dma_piece(PH, dslot, 2 * t);
            Q_SB();
            // the phase's fragment reads: tile t of half P + 2, behind these MFMAs
            if (PH == 0) read_tile(FB[S ^ 1], rslot, wc, t);
            if (PH == 1) read_tile(FA[1], rslot, wr, t);
            if (PH == 2) read_tile(FA[0], rslot, wr, t);
            if (PH == 3) read_tile(FB[S ^ 1], rslot, wc, t);
#pragma unroll
            for (int j = 1; j < 3; ++j) {
                acc[mt][2 * hb] = Mfma32<T>::run(Fb[0][j], Fa[t][j], acc[mt][2 * hb]);
                acc[mt][2 * hb + 1] = Mfma32<T>::run(Fb[1][j], Fa[t][j], acc[mt][2 * hb + 1]);
            }
            finish_rows(v, mt, hb, 0, 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
            }
            Q_SB();
            dma_piece(PH, dslot, 2 * t + 1);
            Q_SB();
            reread(v, 2, 4);
            acc[mt][2 * hb] = Mfma32<T>::run(Fb[0][3], Fa[t][3], acc[mt][2 * hb]);
            acc[mt][2 * hb + 1] = Mfma32<T>::run(Fb[1][3], Fa[t][3], acc[mt][2 * hb + 1]);
            finish_rows(v, mt, hb, 2, 4);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
            }
            Q_SB();
        }
        if (PH == 3) advance();
        // as in the plain phase; the budget also lets this step's stores stay in flight
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(q_allow(E_PHASE * (PH + 1))) : "memory");
        __builtin_amdgcn_s_barrier();
        Q_SB();
    };

    auto kstep_plain = [&](auto s_tag) {
        using A20 = std::integral_constant<int, 20>;
        phase(std::integral_constant<int, 0>{}, s_tag, A20{});
        phase(std::integral_constant<int, 1>{}, s_tag, A20{});
        phase(std::integral_constant<int, 2>{}, s_tag, A20{});
        phase(std::integral_constant<int, 3>{}, s_tag, A20{});
    };
    auto kstep_first = [&]() {
        pin_bias();   // fetched in the finished tile's second K step: at least two K steps of counted waits ago
        phase_first(std::integral_constant<int, 0>{});
        phase_first(std::integral_constant<int, 1>{});
        phase_first(std::integral_constant<int, 2>{});
        phase_first(std::integral_constant<int, 3>{});
    };
    // second K step of a tile: its bias is fetched here (used by the tile's epilogue at least two K steps later; every
    // phase-end wait in between leaves at most 63 younger instructions in flight), and the stores of the first step are
    // still younger than the halves waited for
    auto kstep_second = [&](int n0) {
        using S1 = std::integral_constant<int, 1>;
        load_bias(n0);
        phase(std::integral_constant<int, 0>{}, S1{}, std::integral_constant<int, q_allow(4 * E_PHASE)>{});
        phase(std::integral_constant<int, 1>{}, S1{}, std::integral_constant<int, q_allow(3 * E_PHASE)>{});
        phase(std::integral_constant<int, 2>{}, S1{}, std::integral_constant<int, q_allow(2 * E_PHASE)>{});
        phase(std::integral_constant<int, 3>{}, S1{}, std::integral_constant<int, q_allow(E_PHASE)>{});
    };

    for (int ts = 0; ts < my_tiles; ++ts) {
        int m0, n0;
        coord(ts, m0, n0);
        kstep_first();
        kstep_second(n0);
        for (int kt = 2; kt < KT; kt += 2) {
            kstep_plain(std::integral_constant<int, 0>{});
            kstep_plain(std::integral_constant<int, 1>{});
        }
        set_ptile(m0, n0);
        if (ts == 0) rsrcC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)c_bytes, 0x00020000);   // a finished tile exists
    }
    if (my_tiles > 0) epilogue();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may land after the block has left the CU
#undef Q_DMA
}

template <typename T>
int launch_h16p_t(const GemmArgs& a, hipStream_t s) {
    int dev = 0;
    static bool attr_set[64] = {};
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_h16p_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, Q_LDS);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(gemm_h16p)");
        attr_set[dev] = true;
    }
    const int tiles = ((a.M + QT - 1) / QT) * (a.N / QT);
    const int ncu = device_num_cus();
    hipLaunchKernelGGL((gemm_h16p_kernel<T>), dim3(tiles < ncu ? tiles : ncu), dim3(256), Q_LDS, s, a);
    VITSEG_LAUNCH_CHECK("gemm_h16p");
    return VITSEG_OK;
}

}  // namespace

// true when this kernel takes the GEMM: the conditions of gemm_p8_applies (plain A, whole 256-column tiles, an even number
// of 64-deep K steps, at least 4 of them), the bias epilogue with 16-bit output, and a reduction short enough for the
// epilogue to matter (K <= 1024: measured 1.10-1.13 x gemm_p8.hip at K = 768, 0.94-0.97 x at K >= 2304)
bool gemm_h16p_applies(const GemmArgs& a, int epi) {
    if (opt(OPT_NO_H16P)) return false;   // A/B against gemm_p8.hip
    const size_t a_bytes = ((size_t)a.M + QT) * a.lda * 2, w_bytes = (size_t)a.N * (a.ldw ? a.ldw : a.K) * 2;
    return epi == EPI_BIAS && a.M >= 2048 && a.N % QT == 0 && a.K % 128 == 0 && a.K >= 256 && a.K <= 1024 && a.lda % 8 == 0 &&
           a.ldc % 8 == 0 && (a.ldw == 0 || a.ldw % 8 == 0) && a_bytes < 0x7fffffffull && w_bytes < 0x7fffffffull &&
           a.splitk <= 1 && (long long)a.M * a.ldc * 2 < 0x7fffffffll;   // (C through one buffer descriptor)
}

int launch_gemm_h16p(const GemmArgs& a_in, hipStream_t s, bool f16) {
    GemmArgs a = a_in;
    if (a.ldw == 0) a.ldw = a.K;
    return f16 ? launch_h16p_t<f16_t>(a, s) : launch_h16p_t<bf16_t>(a, s);
}

}  // namespace vitseg
