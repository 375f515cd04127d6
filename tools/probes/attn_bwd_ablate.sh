#!/bin/bash
# Where do the bf16 attention backward kernels spend their time?  Builds tools/probes/attn_bwd_ablate.hip against patched
# copies of csrc/attention_bwd_bf16.hip (one per variant) and times the dQ and dK/dV kernels at the training geometry.
#   bash tools/probes/attn_bwd_ablate.sh            ABL_VARIANTS="base noexp ..." to choose
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=${ABL_OUT:-gpurun_out/attn_bwd_ablate}   # ABL_OUT=tools/probes/_build/abl + ABL_BUILD_ONLY=1 here, ABL_RUN_ONLY=1 on the GPU box
mkdir -p $out
variants=${ABL_VARIANTS:-"base noexp novalu notr noqo nomfma2 nosync occ2 base"}
for v in $variants; do
[ -n "$ABL_RUN_ONLY" ] && continue
python3 - visiontransformer_amd/csrc/attention_bwd_bf16.hip "$out" "$v" <<'PY' || exit 1
import os, sys
src, out, v = sys.argv[1], sys.argv[2], sys.argv[3]
if v.startswith("old"):    # the committed kernels (git HEAD), for a same-box A/B against the working tree
    import subprocess
    text = subprocess.check_output(["git", "show", "HEAD:" + src]).decode()
    v_apply = "base"
else:
    text = open(src).read()
    v_apply = v
s = text.replace('#include "kernels.hpp"', '#include "%s/visiontransformer_amd/csrc/kernels.hpp"' % os.getcwd())
def rep(t, a, b, cnt=-1):
    assert t.count(a) >= 1, a
    return t.replace(a, b) if cnt < 0 else t.replace(a, b, cnt)
# the harness times one kernel at a time
s = rep(s, "        hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<DROP, RG, MWORDS>)", "        if (!g_skip_dq) hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<DROP, RG, MWORDS>)")
s = rep(s, "        hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<DROP, RG, MWORDS>)", "        if (!g_skip_dkv) hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<DROP, RG, MWORDS>)")
i = s.index("// ---------------------------------------------------------------------------------- dQ (patch queries)")
j = s.index("// ---------------------------------------------------------------------------------- dK, dV (patch keys)")
k = s.index("// ---------------------------------------------------------------------------------- the CLS token")
head, dq, dkv, tail = s[:i], s[i:j], s[j:k], s[k:]
if v_apply == "base":
    pass
elif v_apply.startswith("stagger"):   # co-resident blocks of a CU start a third (two thirds) of a tile apart: N x 64 cycles per slot
    n = int(v_apply[len("stagger"):])
    delay = "    { const int slot = ((int)blockIdx.x >> 8) %% 3; for (int i = 0; i < slot; ++i) __builtin_amdgcn_s_sleep(%d); }\n" % n
    dq = rep(dq, "    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n", "    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n" + delay, 1)
    dkv = rep(dkv, "    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n", "    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n" + delay, 1)
elif v_apply == "notail":      # no CLS-token work at the end of a block (vector update, partial records): the stores stay
    dq = rep(dq, "        cls_stage(qf, q_valid ? cls_ds : 0.f, red, tid);\n        cls_stage(dof, q_valid ? cls_pm : 0.f, red + 2048, tid);\n", "")
    dq = rep(dq, "        cls_finish(red, rec + 64, 2, tid);\n", "")
    dq = rep(dq, "        const float ds = p * fmaf(dpc, m, ndelta);\n#pragma unroll\n        for (int dt = 0; dt < 2; ++dt)\n#pragma unroll\n            for (int g4 = 0; g4 < 4; ++g4) {   // accumulator register", "        const float ds = p * fmaf(dpc, m, ndelta);\n#pragma unroll\n        for (int dt = 0; dt < 0; ++dt)\n#pragma unroll\n            for (int g4 = 0; g4 < 4; ++g4) {   // accumulator register")
    dkv = rep(dkv, "        cls_stage(kf, k_valid ? cls_ds : 0.f, red, tid);\n", "")
    dkv = rep(dkv, "        cls_finish(red, clsp + rec, 1, tid);\n", "")
    dkv = rep(dkv, "        const float pm = p * m, ds = p * fmaf(dpc, m, -delta[stat0 + Np]);\n#pragma unroll\n        for (int dt = 0; dt < 2; ++dt)", "        const float pm = p * m, ds = p * fmaf(dpc, m, -delta[stat0 + Np]);\n#pragma unroll\n        for (int dt = 0; dt < 0; ++dt)")
elif v_apply == "nocls_stage":   # dQ kernel: no partial records of the CLS key's gradients (the vector update stays)
    dq = rep(dq, "        cls_stage(qf, q_valid ? cls_ds : 0.f, red, tid);\n        cls_stage(dof, q_valid ? cls_pm : 0.f, red + 2048, tid);\n", "        if (cls_ds + cls_pm == 1.2345f) red[tid] = cls_ds;\n")
    dq = rep(dq, "        cls_finish(red, rec + 64, 2, tid);\n", "")
elif v_apply == "nocls_update":  # dQ kernel: no rank-1 update of dQ by the CLS key (the partial records stay)
    dq = rep(dq, "        const float ds = p * fmaf(dpc, m, ndelta);\n#pragma unroll\n        for (int dt = 0; dt < 2; ++dt)\n#pragma unroll\n            for (int g4 = 0; g4 < 4; ++g4) {   // accumulator register", "        const float ds = p * fmaf(dpc, m, ndelta);\n#pragma unroll\n        for (int dt = 0; dt < 0; ++dt)\n#pragma unroll\n            for (int g4 = 0; g4 < 4; ++g4) {   // accumulator register")
elif v_apply == "noexp":       # the transcendental replaced by a multiply
    dq = rep(dq, "__builtin_amdgcn_exp2f(st[r] * c)", "(st[r] * c)"); dq = rep(dq, "__builtin_amdgcn_exp2f(st[r + 1] * c)", "(st[r + 1] * c)")
    dkv = rep(dkv, "__builtin_amdgcn_exp2f(st[r] * c)", "(st[r] * c)"); dkv = rep(dkv, "__builtin_amdgcn_exp2f(st[r + 1] * c)", "(st[r + 1] * c)")
elif v_apply == "novalu":      # no element work between the two MFMA groups: the accumulators are packed as they are
    dq = rep(dq, "                float d0, d1;\n                if (DROP) {", "                float d0, d1;\n                if (true) { d0 = st[r] + dp[r]; d1 = st[r + 1] + dp[r + 1]; } else if (DROP) {")
    dkv = rep(dkv, "                    if (DROP) {\n                        float m0, m1;", "                    if (true) { pp[r >> 1] = pack2_bf16(st[r], st[r + 1]); pd[r >> 1] = pack2_bf16(dp[r], dp[r + 1]); } else if (DROP) {\n                        float m0, m1;")
elif v_apply == "notr":        # the transposed fragments of the second MFMA group from registers instead of the LDS
    dq = rep(dq, "auto ldf = [&](int i) { fr[i] = tr_frag(Ks, kb * 32 + 16 * (i >> 1), i & 1, lane); };", "auto ldf = [&](int i) { fr[i] = __builtin_bit_cast(bf16x8, qf[i]); };")
    dkv = rep(dkv, "tr_frag(Os, qb * 32 + 16 * s, dt, lane)", "__builtin_bit_cast(bf16x8, vf[2 * s + dt])")
    dkv = rep(dkv, "tr_frag(Qs, qb * 32 + 16 * s, dt, lane)", "__builtin_bit_cast(bf16x8, kf[2 * s + dt])")
elif v_apply == "noqo":        # the A operands of the first MFMA group from registers instead of the LDS
    dq = rep(dq, "const f32x4 kf = *(const f32x4*)&Ks[tile_off(key, 2 * s + lh)];", "const f32x4 kf = qf[(s + 1) & 3];")
    dq = rep(dq, "const f32x4 vf = *(const f32x4*)&Vs[tile_off(key, 2 * s + lh)];", "const f32x4 vf = dof[(s + 1) & 3];")
    dkv = rep(dkv, "const f32x4 qa = *(const f32x4*)&Qs[tile_off(q, 2 * s + lh)];", "const f32x4 qa = kf[(s + 1) & 3];")
    dkv = rep(dkv, "const f32x4 oa = *(const f32x4*)&Os[tile_off(q, 2 * s + lh)];", "const f32x4 oa = vf[(s + 1) & 3];")
elif v_apply == "nomfma2":     # the second MFMA group (dQ; dV, dK) left out, its operands kept alive
    dq = rep(dq, "dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i], dsf, dq[dt], 0, 0, 0);", 'asm volatile("" :: "v"(fr[i]), "v"(dsf));')
    dkv = rep(dkv, "dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Os, qb * 32 + 16 * s, dt, lane), pf, dv[dt],\n                                                                     0, 0, 0);", 'asm volatile("" :: "v"(tr_frag(Os, qb * 32 + 16 * s, dt, lane)), "v"(pf));')
    dkv = rep(dkv, "dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Qs, qb * 32 + 16 * s, dt, lane), df, dk[dt],\n                                                                     0, 0, 0);", 'asm volatile("" :: "v"(tr_frag(Qs, qb * 32 + 16 * s, dt, lane)), "v"(df));')
elif v_apply == "nosync":      # no staging wait and no barrier at the end of a tile (the tiles race: timing only)
    dq = rep(dq, "        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");   // this wave's pieces of tile kt + 1 have landed\n        __syncthreads();\n", "")
    dkv = rep(dkv, "        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");   // this wave's pieces of tile qt + 1 have landed\n        __syncthreads();\n", "")
elif v_apply == "occ2":        # two waves per SIMD (256 registers each)
    dq = rep(dq, "__launch_bounds__(256, 3) void attn_bwd_dq_bf16_kernel", "__launch_bounds__(256, 2) void attn_bwd_dq_bf16_kernel")
    dkv = rep(dkv, "__launch_bounds__(256, 3) void attn_bwd_dkv_bf16_kernel", "__launch_bounds__(256, 2) void attn_bwd_dkv_bf16_kernel")
else:
    raise SystemExit("unknown variant " + v)
open("%s/%s.hip" % (out, v), "w").write(head + dq + dkv + tail)
PY
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DATTN_SRC="\"$PWD/$out/$v.hip\"" tools/probes/attn_bwd_ablate.hip -o $out/$v 2> $out/$v.err || { cat $out/$v.err; exit 1; }
done
if [ -z "$ABL_BUILD_ONLY" ]; then
  for v in $variants; do timeout -k 10 120 $out/$v $v || exit 1; done
fi
