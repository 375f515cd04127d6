#!/bin/bash
# builds tools/probes/p8_where.hip against the product kernel and against patched copies (gpurun_out/p8_where/) and prints
# one timing block per variant.  Run on the GPU box:  bash tools/probes/p8_where.sh     (P8_WHERE_VARIANTS="base noepi")
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=${P8_WHERE_OUT:-gpurun_out/p8_where}   # P8_WHERE_OUT=tools/probes/_build/p8 P8_WHERE_BUILD_ONLY=1 here, then P8_WHERE_RUN_ONLY=1 on the box
mkdir -p $out
[ -z "$P8_WHERE_RUN_ONLY" ] && python3 - visiontransformer_amd/csrc/gemm_p8.hip "$out" <<'PY'
import os, sys
src, out = sys.argv[1], sys.argv[2]
s = open(src).read().replace('#include "kernels.hpp"', '#include "%s/visiontransformer_amd/csrc/kernels.hpp"' % os.getcwd())
def w(tag, t):
    open("%s/%s.hip" % (out, tag), "w").write(t)
def rep(t, a, b):
    assert a in t, a
    return t.replace(a, b)
w("base", s)
w("h16p", open("visiontransformer_amd/csrc/gemm_h16p.hip").read().replace('#include "kernels.hpp"', '#include "%s/visiontransformer_amd/csrc/kernels.hpp"' % os.getcwd()))
if os.path.exists("tools/probes/tmp_old_p8.hip"):   # an untracked copy of an earlier version, for A/B on the same box
    w("old", open("tools/probes/tmp_old_p8.hip").read().replace('#include "kernels.hpp"', '#include "%s/visiontransformer_amd/csrc/kernels.hpp"' % os.getcwd()))
# the accumulators must stay live (a kernel whose MFMA results are unused loses its MFMAs): sum them and compare
noepi = rep(s, "            epilogue(tc_cur);\n", """            { float t = 0.f;
              for (int mt = 0; mt < 8; ++mt) for (int nt = 0; nt < 4; ++nt) for (int r = 0; r < 4; ++r) t += acc[mt][nt][r];
              if (t == 1.2345f) ((float*)p.C)[0] = t; }
""")
# epilogue without the LDS round trip (values land in the wrong places; the store pattern is the product's)
nolds = rep(s, """            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private: no barrier needed
""", "")
nolds = rep(nolds, """                    const f32x4 t = *(const f32x4*)(stage + row * 64 + ((((rcol >> 2) + c4) ^ row) << 2));
""", """                    const f32x4 t = acc[mt][(i * (CPL / 4) + c4) & 3];
""")
nolds = rep(nolds, """                *(f32x4*)(stage + l15 * 64 + (((nt * 4 + lq) ^ l15) << 2)) = acc[mt][nt];
""", """                ;
""")
w("epi_nolds", nolds)
w("noepi", noepi)
nostore = rep(s, "                    if (grow < p.M) {\n                        if (EPI == EPI_GELU && p.aux) *(uint4*)((T*)p.aux + o) = ha;\n                        *(uint4*)(Cbase + o) = h;\n                    }",
              "                    if (grow < 0) {\n                        if (EPI == EPI_GELU && p.aux) *(uint4*)((T*)p.aux + o) = ha;\n                        *(uint4*)(Cbase + o) = h;\n                    }")
nostore = rep(nostore, "if (grow < p.M) *(f32x4*)((float*)Cbase + o)", "if (grow < 0) *(f32x4*)((float*)Cbase + o)")
w("nostore", nostore)
# tile-transition ablations on top of noepi: (a) the epilogue bracket barriers that re-align the two wave rows,
# (b) the tile coordinates without integer divisions (valid for 256 row tiles: M = 65536)
notilesync = rep(noepi, "            if (wr == 0) P8_BAR();\n", "")
notilesync = rep(notilesync, "            if (wr == 1) P8_BAR();\n            zero_acc();", "            zero_acc();")
w("noepi_notilesync", notilesync)
nocoord = rep(noepi, "    const int t = logical_item(round, tiles_m * tiles_n);\n    const int gsz = tiles_m * gn, ngroups = (tiles_n + gn - 1) / gn;",
              "    const int t = round * (int)gridDim.x + (int)blockIdx.x;\n    if (tiles_m == 256) { TileCoord q; q.m0 = (t & 255) * PT; q.n0 = (t >> 8) * PT; q.split = 0; return q; }\n    const int gsz = tiles_m * gn, ngroups = (tiles_n + gn - 1) / gn;")
w("noepi_nocoord", nocoord)
# main loop without any fragment read (the registers keep whatever they hold): MFMAs + DMA + barriers only
noread = rep(noepi, "        for (int mt = 0; mt < 4; ++mt)\n#pragma unroll\n            for (int ks = 0; ks < 2; ++ks) {\n                if (TT)\n                    xa[mt][ks]",
             "        for (int mt = 0; mt < (p.M < 0 ? 4 : 0); ++mt)\n#pragma unroll\n            for (int ks = 0; ks < 2; ++ks) {\n                if (TT)\n                    xa[mt][ks]")
noread = rep(noread, "        for (int nt = 0; nt < 2; ++nt)\n#pragma unroll\n            for (int ks = 0; ks < 2; ++ks) {\n                if (TT)\n                    w[nt][ks]",
             "        for (int nt = 0; nt < (p.M < 0 ? 2 : 0); ++nt)\n#pragma unroll\n            for (int ks = 0; ks < 2; ++ks) {\n                if (TT)\n                    w[nt][ks]")
w("noepi_noread", noread)
# in-kernel clock (MI355X_MICROARCH.md, DVFS give-back item 6): shader-clock ticks (s_memtime) over 100 MHz ticks
# (s_memrealtime) around the whole kernel of wave 0 of every block, written to GemmArgs::thin_scratch (unused by this kernel)
def clocked(t):
    t = rep(t, "    const int tid = threadIdx.x, lane = tid & 63;\n    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);\n    const int wr = wave >> 2, wc = wave & 3;",
            "    const int tid = threadIdx.x, lane = tid & 63;\n    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);\n    const int wr = wave >> 2, wc = wave & 3;\n    const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();")
    t = rep(t, "    if (wr == 0) P8_BAR();\n    P8_VMCNT(0);\n#undef P8_KSTEP",
            "    if (wr == 0) P8_BAR();\n    P8_VMCNT(0);\n    if (tid == 0 && p.thin_scratch) {\n        ((unsigned long long*)p.thin_scratch)[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk_t0;\n        ((unsigned long long*)p.thin_scratch)[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;\n    }\n#undef P8_KSTEP")
    return t
# de-phasing experiment: half of the CUs (odd XCD-local index) start ~DELAY us late, so that their epilogues (store bursts,
# VALU) fall into the other half's main loops
for delay_us in (4, 8, 16):
    w("stagger%d" % delay_us, rep(s, "    // ---- prologue: fill the stream (halves 0 .. 5 of the block's sequence), first B0 fragments ----",
      "    if (((blockIdx.x >> 3) & 1) && !TT) { const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); while (__builtin_amdgcn_s_memrealtime() - t0 < %dull) __builtin_amdgcn_s_sleep(8); }\n    // ---- prologue: fill the stream (halves 0 .. 5 of the block's sequence), first B0 fragments ----" % (delay_us * 100)))
# epilogue without its two explicit LDS waits per m tile (LDS operations of one wave execute in order)
nowaits = rep(s, """            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private: no barrier needed
""", "")
nowaits = rep(nowaits, """            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slab is free for the next m tile
""", "")
w("nowaits", nowaits)
w("clock", clocked(s))
w("noepi_clock", clocked(noepi))
nobar = rep(noepi, "        __builtin_amdgcn_s_barrier();         \\\n", "        ;         \\\n")
w("noepi_nobarrier", nobar)
def strip_dma(t):   # the DMA statement's instruction text (one or two string literals) becomes empty
    import re
    t2, n = re.subn(r'asm volatile\("s_mov_b32 m0, %0[^;]*?"\s*\n?\s*::', 'asm volatile("" ::', t, flags=re.S)
    assert n == 1, n
    return t2
nodma = strip_dma(noepi)
w("noepi_nodma", nodma)
w("noepi_nodma_clock", clocked(nodma))
w("noepi_noread_clock", clocked(noread))
w("noepi_noread_nodma_clock", clocked(strip_dma(noread)))
w("noepi_nobarrier_clock", clocked(nobar))
# round 5, VERDICT item 2 (iii): ONE barrier per phase instead of two (the barrier behind each MFMA quadrant is gone; the wave
# rows then run a whole phase apart instead of half a phase; the vmcnt margins are one phase short, so the tiles race: timing
# only).  Between the product (8 barriers per K step) and noepi_nobarrier (none).
import re
onebar, n_onebar = re.subn(r"(mfma_quad\(\d, \d, w\d\);\s*\\\n\s*)P8_BAR\(\);", r"\1;        ", noepi)
assert n_onebar == 4, n_onebar
w("noepi_onebar_clock", clocked(onebar))
# the cursors never move (every K step streams the same rows): the scalar work of advance() is gone
noadv = rep(noepi, "        advance(cur0);                                                                               \\\n        advance(cur1);                                                                               \\\n", "")
w("noepi_noadvance_clock", clocked(noadv))
# what a 256 x 128 tile would cost per K step: the B1 half is neither streamed nor read nor multiplied (phases (a0, b1) and
# (a1, b1) are gone, A0 / A1 / B0 still stream; the waits are approximate and the tiles race: timing only)
halfn = noepi
i0 = halfn.index("        /* phase 0: quadrant (a0, b0) */")
i1 = halfn.index("    } while (0)\n\n    for (int G = 0; G < total_k; G += 2) {")
halfn = halfn[:i0] + """        /* phase 0: (a0, b0) */                                                                      \\
        read_a(4 * (s) + 0);                                                                         \\
        issue_half(cur0, 3, 4 * ((s) ^ 1) + 3);                                                         \\
        P8_VMCNT(6);                                                                                 \\
        P8_BAR();                                                                                    \\
        mfma_quad(0, 0, w0);                                                                         \\
        P8_BAR();                                                                                    \\
        /* phase 3: (a1, b0) */                                                                      \\
        read_a(4 * (s) + 3);                                                                         \\
        read_b(w1, 4 * ((s) ^ 1) + 1);                                                               \\
        issue_half(cur1, 0, 4 * (s) + 0);                                                               \\
        issue_half(cur1, 1, 4 * (s) + 1);                                                               \\
        advance(cur0);                                                                               \\
        advance(cur1);                                                                               \\
        P8_VMCNT(6);                                                                                 \\
        P8_BAR();                                                                                    \\
        mfma_quad(1, 0, w0);                                                                         \\
        P8_BAR();                                                                                    \\
""" + halfn[i1:]
w("noepi_halfn_clock", clocked(halfn))
# wave priorities: (a) none at all, (b) the fragment reads + DMA issue of a phase above the other wave row's MFMAs
noprio = rep(rep(s, "        __builtin_amdgcn_s_setprio(1);\n", ""), "        __builtin_amdgcn_s_setprio(0);\n", "")
w("noprio_clock", clocked(noprio))
readprio = rep(rep(s, "        __builtin_amdgcn_s_setprio(1);\n", "        __builtin_amdgcn_s_setprio(0);\n"), "        __builtin_amdgcn_s_setprio(0);\n    };", "        __builtin_amdgcn_s_setprio(1);\n    };")
w("readprio_clock", clocked(readprio))
PY
for v in ${P8_WHERE_VARIANTS:-base noepi nostore epi_nolds noepi_nobarrier noepi_nodma}; do
  p8src=$v; hsrc=h16p
  [ -z "$P8_WHERE_RUN_ONLY" ] && { hipcc --offload-arch=gfx950 -O3 -std=c++17 -DP8_SRC="\"$PWD/$out/$p8src.hip\"" -DH16P_SRC="\"$PWD/$out/$hsrc.hip\"" tools/probes/p8_where.hip -o $out/$v 2> $out/$v.err || { cat $out/$v.err; exit 1; }; }
  echo "== $v"
  if [ -z "$P8_WHERE_BUILD_ONLY" ]; then $out/$v $P8_WHERE_M || exit 1; fi
done
