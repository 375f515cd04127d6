#!/bin/bash
# builds tools/probes/attn_bwd_trace.hip against a copy of csrc/attention_bwd_bf16.hip with s_memtime stamps in the dK/dV
# kernel's tile loop (gpurun_out/attn_bwd_trace/) and runs it.   bash tools/probes/attn_bwd_trace.sh
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/attn_bwd_trace
mkdir -p $out
python3 - visiontransformer_amd/csrc/attention_bwd_bf16.hip "$out" <<'PY'
import os, sys
src, out = sys.argv[1], sys.argv[2]
s = open(src).read().replace('#include "kernels.hpp"', '#include "%s/visiontransformer_amd/csrc/kernels.hpp"' % os.getcwd())
def rep(t, a, b, cnt=1):
    assert t.count(a) >= 1, a
    return t.replace(a, b, cnt)
# work only inside the dK/dV kernel: split the file at its definition
i = s.index("void attn_bwd_dkv_bf16_kernel(")
head, k = s[:i], s[i:]
STAMP = "    if (blockIdx.x == 200 && lane == 0 && qt < 20) g_trace[(wave * 20 + qt) * 16 + %d] = __builtin_readcyclecounter();\n"
k = rep(k, "        const int buf = qt & 1;\n", "        const int buf = qt & 1;\n    " + STAMP % 0)
# q block loop body is unrolled over qb by the compiler; stamps index 1..3 for qb = 0 and 4..6 for qb = 1
k = rep(k, "            unsigned pp[8], pd[8];  // P~ and dS fragments (B operands), query = register index\n",
        "        " + (STAMP % 99).replace("+ 99]", "+ 1 + 3 * qb]") + "            unsigned pp[8], pd[8];  // P~ and dS fragments (B operands), query = register index\n")
k = rep(k, "            // (reading these fragments ahead of their MFMAs", "        " + (STAMP % 99).replace("+ 99]", "+ 2 + 3 * qb]") + "            // (reading these fragments ahead of their MFMAs")
k = rep(k, "        swrite(buf ^ 1);\n        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");   // this wave's pieces of tile qt + 1 have landed\n",
        "    " + STAMP % 6 + "        swrite(buf ^ 1);\n        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");   // this wave's pieces of tile qt + 1 have landed\n    " + STAMP % 7)
# stamp 3 + 3 qb: after the dV/dK MFMAs of a q block = end of the qb loop body: insert before the closing of the qb loop,
# i.e. right before "        swrite(buf ^ 1);" is too late for qb = 0, so stamp at the top of the qb loop for qb = 1 instead
k = rep(k, "            f32x16 st, dp;\n", "        " + (STAMP % 99).replace("+ 99]", "+ 3 * qb]").replace("qt < 20)", "qt < 20 && qb == 1)") + "            f32x16 st, dp;\n")
open(out + "/traced.hip", "w").write(head + k)
PY
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DATTN_SRC="\"$PWD/$out/traced.hip\"" tools/probes/attn_bwd_trace.hip -o $out/traced 2> $out/traced.err || { cat $out/traced.err; exit 1; }
if [ -z "$TRACE_BUILD_ONLY" ]; then $out/traced; fi
