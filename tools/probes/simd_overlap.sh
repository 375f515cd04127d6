#!/bin/bash
# builds and runs tools/probes/simd_overlap.hip on the GPU box: wall times, then one rocprofv3 --pmc pass for the counter view
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for vop in ${SIMD_OVERLAP_VARIANTS:-0 1 f32}; do
  extra=""; v=$vop; if [ $vop = f32 ]; then extra="-DF32MFMA"; v=0; echo "== fp32 MFMA (v_mfma_f32_32x32x2_f32), 48 v_fma_f32"; fi
  hipcc --offload-arch=gfx950 -O3 -DVOP=$v $extra tools/probes/simd_overlap.hip -o gpurun_out/simd_overlap$vop 2>/dev/null || exit 1
  gpurun_out/simd_overlap$vop
  rm -rf gpurun_out/simd_overlap_pmc$vop
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv \
     -d gpurun_out/simd_overlap_pmc$vop -o p -- gpurun_out/simd_overlap$vop > gpurun_out/simd_overlap_pmc$vop.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/simd_overlap_pmc$vop/**/p_counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    g = {n: sum(v) / len(v) for n, v in c.items()}
    cyc = g.get("GRBM_GUI_ACTIVE", 0) / 8
    print(k[:40], {n: round(v) for n, v in g.items()})
    if cyc:
        print("   per SIMD: MFMA busy %.3f   VALU active (quad-cycles x4 / 1024 SIMDs) %.3f of the kernel's %d cycles" %
              (g.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc, g.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / cyc, cyc))
PY
done
