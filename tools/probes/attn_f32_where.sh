#!/bin/bash
# builds tools/probes/attn_f32_where.hip against the product kernel and against patched copies (gpurun_out/attn_f32_where/)
# and prints one timing per variant.  Run on the GPU box:  bash tools/probes/attn_f32_where.sh
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/attn_f32_where
mkdir -p $out
python3 - visiontransformer_amd/csrc/attention_f32.hip "$out" <<'PY'
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
w("nobarrier", rep(s, "        swrite(buf ^ 1);\n        __syncthreads();\n    }\n\n    // ---- normalise and store", "        swrite(buf ^ 1);\n    }\n\n    // ---- normalise and store"))
# no global loads / no LDS staging writes in the loop (tiles stay what the prologue staged)
ng = rep(s, "        gload(min(kt + 1, nkt - 1));  // the last tile re-stages itself: keeps the body branch-free\n", "")
ng = rep(ng, "        swrite(buf ^ 1);\n        __syncthreads();\n    }\n\n    // ---- normalise and store", "        __syncthreads();\n    }\n\n    // ---- normalise and store")
w("nostaging", ng)
# V operand from a register instead of LDS
w("nov_lds", rep(s, "                    const float vf = Vs[key * HD + dt * 32 + li];", "                    const float vf = qreg[(key + dt) & 31];"))
# K fragments from registers instead of LDS
w("nok_lds", rep(s, "                    const f32x4 kf = *(const f32x4*)&Ks[key * HD + (((2 * c + lh) ^ (key & 15)) << 2)];", "                    const f32x4 kf = {qreg[c], qreg[c + 8], qreg[c + 16], qreg[c + 24]};"))
# no exponentials / row sums (the scores go straight into PV)
w("noexp", rep(s, "                const float pv = __builtin_amdgcn_exp2f(st[kb][r]);\n                st[kb][r] = pv;\n                psum += pv;", "                psum += st[kb][r];"))
PY
for v in ${ATTN_WHERE_VARIANTS:-base nobarrier nostaging nov_lds nok_lds noexp}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DATTN_SRC="\"$PWD/$out/$v.hip\"" tools/probes/attn_f32_where.hip -o $out/$v 2> $out/$v.err || { cat $out/$v.err; exit 1; }
  echo "== $v"
  if [ -z "$ATTN_WHERE_BUILD_ONLY" ]; then $out/$v || exit 1; fi
done
