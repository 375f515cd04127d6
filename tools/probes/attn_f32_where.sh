#!/bin/bash
# builds tools/probes/attn_f32_where.hip against the product kernel and against patched copies (gpurun_out/attn_f32_where/)
# and prints one timing per variant.  Run on the GPU box:  bash tools/probes/attn_f32_where.sh
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=${ATTN_WHERE_OUT:-gpurun_out/attn_f32_where}   # ATTN_WHERE_OUT=tools/probes/_build/af32 ATTN_WHERE_BUILD_ONLY=1 here, ATTN_WHERE_RUN_ONLY=1 on the box
mkdir -p $out
[ -z "$ATTN_WHERE_RUN_ONLY" ] && python3 - visiontransformer_amd/csrc/attention_f32.hip "$out" <<'PY'
import os, sys
src, out = sys.argv[1], sys.argv[2]
s = open(src).read().replace('#include "kernels.hpp"', '#include "%s/visiontransformer_amd/csrc/kernels.hpp"' % os.getcwd())
def w(tag, t):
    open("%s/%s.hip" % (out, tag), "w").write(t)
def rep(t, a, b, cnt=1):
    assert a in t, a
    return t.replace(a, b, cnt)
w("base", s)
# the per-tile barrier of the patch-query kernel (first __syncthreads in the key loop)
TILE_END = "        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");   // this wave's pieces of tile kt + 1 have landed\n        __syncthreads();\n    }\n\n    // ---- normalise and store"
w("nobarrier", rep(s, TILE_END, "    }\n\n    // ---- normalise and store"))
# no global loads / no LDS staging writes in the loop (tiles stay what the prologue staged)
ng = rep(s, "        stage(min(kt + 1, nkt - 1), buf ^ 1);  // the last tile re-stages itself: keeps the body branch-free\n", "")
w("nostaging", ng)
# no CLS-query work in the block of row tile 0
w("nocls", rep(s, "        if (cls_blk) cls_tile(kt, Ks, Vs);\n", ""))
# V operand from a register instead of LDS
w("nov_lds", rep(s, "                    const float vf = Vs[key * HD + dt * 32 + li];", "                    const float vf = qreg[(key + dt) & 31];"))
# K fragments from registers instead of LDS
w("nok_lds", rep(s, "                    const f32x4 kf = *(const f32x4*)&Ks[key * HD + (((2 * c + lh) ^ (key & 15)) << 2)];", "                    const f32x4 kf = {qreg[(c + kb) & 31], qreg[(c + 8 + kb) & 31], qreg[(c + 16 + kb) & 31], qreg[(c + 24 + kb) & 31]};   // (kb: the two key blocks must stay different products)"))
# no exponentials / row sums (the scores go straight into PV)
w("noexp", rep(s, "                const float pv = __builtin_amdgcn_exp2f(st[kb][r]);\n                st[kb][r] = pv;\n                psum += pv;", "                psum += st[kb][r];"))
PY
for v in ${ATTN_WHERE_VARIANTS:-base nobarrier nostaging nov_lds nok_lds noexp nocls}; do
  [ -z "$ATTN_WHERE_RUN_ONLY" ] && { hipcc --offload-arch=gfx950 -O3 -std=c++17 -DATTN_SRC="\"$PWD/$out/$v.hip\"" tools/probes/attn_f32_where.hip -o $out/$v 2> $out/$v.err || { cat $out/$v.err; exit 1; }; }
  echo "== $v"
  if [ -z "$ATTN_WHERE_BUILD_ONLY" ]; then $out/$v || exit 1; fi
done
