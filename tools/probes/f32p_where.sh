#!/bin/bash
# builds tools/probes/f32p_where.hip against the product kernel and against patched copies (gpurun_out/f32p_where/) and
# prints one timing block per variant.  Run on the GPU box:  bash tools/probes/f32p_where.sh
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/f32p_where
mkdir -p $out
src=visiontransformer_amd/csrc/gemm_f32p.hip
python3 - "$src" "$out" <<'PY'
import re, sys
src, out = sys.argv[1], sys.argv[2]
s = open(src).read().replace('#include "kernels.hpp"', '#include "%s/visiontransformer_amd/csrc/kernels.hpp"' % __import__("os").getcwd())
def w(tag, t): open("%s/%s.hip" % (out, tag), "w").write(t)
w("base", s)
import os
if os.path.exists("tools/probes/tmp_old_f32p.hip"):   # an untracked copy of an earlier version, for A/B on the same box
    w("old", open("tools/probes/tmp_old_f32p.hip").read().replace('#include "kernels.hpp"', '#include "%s/visiontransformer_amd/csrc/kernels.hpp"' % os.getcwd()))
nobar = s.replace("__builtin_amdgcn_s_barrier();", ";")
w("nobarrier", nobar)
nodma = s.replace("buffer_load_dwordx4 %1, %2, %3 offen lds", "s_nop 0")
w("nodma_load", nodma)
nodma2 = s.replace('"s_mov_b32 m0, %0\\n\\ts_nop 0\\n\\tbuffer_load_dwordx4 %1, %2, %3 offen lds"', '""')
assert nodma2 != s
w("nodma_at_all", nodma2)
nofr = s.replace("read_frags(roff, jn, (slot) ^ 1);", ";").replace("        read_frags(nroff, 0, 0);\n        F32P_SB();\n        mfma8(1, 0);", "        F32P_SB();\n        mfma8(1, 0);")
assert nofr != s
w("nofrags", nofr)
allx = nofr.replace("__builtin_amdgcn_s_barrier();", ";").replace('"s_mov_b32 m0, %0\\n\\ts_nop 0\\n\\tbuffer_load_dwordx4 %1, %2, %3 offen lds"', '""')
w("mfma_only", allx)
PY
for v in ${F32P_WHERE_VARIANTS:-base nobarrier nodma_load nodma_at_all nofrags mfma_only}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DF32P_SRC="\"$PWD/$out/$v.hip\"" tools/probes/f32p_where.hip -o $out/$v 2> $out/$v.err || { cat $out/$v.err; exit 1; }
  echo "== $v"
  if [ -z "$F32P_WHERE_BUILD_ONLY" ]; then $out/$v $F32P_WHERE_WSCALE || exit 1; fi
done
