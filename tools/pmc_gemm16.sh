#!/bin/bash
# What bounds the 16-bit GEMM main loops?  Three rocprofv3 --pmc passes (SQ has 8 slots; no trace domain beside --pmc) over
# each of three probes -- the QKV projection (gemm_h16p_kernel), an NT gemm_p8 with the plain epilogue (dgrad through fc1:
# K = 3072) and the TT weight-gradient form -- reduced to per-kernel ratios.   bash tools/pmc_gemm16.sh [tag]
#   -> gpurun_out/pmc_gemm16_<tag>.json   (copied to profiles/r04_pmc_gemm16.json)
set -e
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_gemm16_$TAG
rm -rf $OUT; mkdir -p $OUT
run_passes() {   # <name> <probe args...>
  local name=$1; shift
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/$name/a -o p -- python3 tools/op_probe.py "$@" > $OUT/$name.a.log 2>&1
  rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/$name/b -o p -- python3 tools/op_probe.py "$@" > $OUT/$name.b.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/$name/c -o p -- python3 tools/op_probe.py "$@" > $OUT/$name.c.log 2>&1 || echo "pass c failed for $name (counter set); continuing"
  echo "passes done: $name"
}
run_passes qkv_h16p   linear_ex --M 65600 --N 2304 --K 768  --epi 0 --iters 10
run_passes p8_nt_k3072 linear_ex --M 65600 --N 768  --K 3072 --epi 0 --iters 10
run_passes p8_nt_gelu linear_ex --M 65600 --N 3072 --K 768  --epi 1 --aux --iters 10
run_passes p8_tt_wgrad wgrad    --M 768   --N 3072 --K 65600 --iters 10
python3 - "$OUT" <<'PY' > gpurun_out/pmc_gemm16_$TAG.json
import csv, glob, collections, sys, json, os
out = sys.argv[1]
res = {}
for probe in sorted(d for d in os.listdir(out) if os.path.isdir(os.path.join(out, d))):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out, probe, "**", "p_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_p8" in r["Kernel_Name"] or "gemm_h16p" in r["Kernel_Name"]:
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        g = {n: sum(v) / len(v) for n, v in c.items()}
        cyc = g["GRBM_GUI_ACTIVE"] / 8            # kernel cycles (the counter sums over the 8 XCDs)
        wc = g["SQ_WAVE_CYCLES"]                  # quad-cycles summed over waves
        e = {"kernel": k[:120], "kernel_cycles": round(cyc),
             "mfma_busy": round(g["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc, 3),          # per SIMD (1024 of them)
             "lds_unit_busy": round(g.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / cyc, 3),        # per CU: LDS-array cycles / kernel cycles
             "lds_bank_conflict_share": round(g.get("SQ_LDS_BANK_CONFLICT", 0) / max(g.get("SQ_LDS_IDX_ACTIVE", 1), 1), 4),
             "waves_per_simd": round(wc * 4 / 1024 / cyc, 2),
             "wait_any": round(g["SQ_WAIT_ANY"] / wc, 3), "wait_inst_any": round(g["SQ_WAIT_INST_ANY"] / wc, 3),
             "active_inst_any": round(g["SQ_ACTIVE_INST_ANY"] / wc, 3),
             "wait_inst_lds": round(g.get("SQ_WAIT_INST_LDS", 0) / wc, 4),
             "valu_active_per_simd": round(g["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, 3),
             "lds_inst_active_per_simd": round(g["SQ_ACTIVE_INST_LDS"] * 4 / 1024 / cyc, 3),
             "vmem_inst_cycles_per_simd": round(g.get("SQ_INST_CYCLES_VMEM", 0) * 4 / 1024 / cyc, 4),
             "vmem_latency_cycles": round(4 * g.get("SQ_INST_LEVEL_VMEM", 0) / max(g.get("SQ_INSTS_VMEM", 1), 1)),
             "mfma_valu_coexec_share_of_mfma_busy": round(g.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0) / max(g["SQ_VALU_MFMA_BUSY_CYCLES"], 1), 3),
             "insts": {n[9:]: round(g[n]) for n in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_VMEM", "SQ_INSTS_SALU") if n in g},
             "raw": {n: round(v) for n, v in g.items() if n in ("SQ_LDS_DATA_FIFO_FULL", "SQ_LDS_ADDR_CONFLICT")}}
        res[probe] = e
print(json.dumps(res, indent=1))
PY
cat gpurun_out/pmc_gemm16_$TAG.json
