#!/bin/bash
# Where do a kernel's wave-cycles go?  Two rocprofv3 --pmc passes (SQ has 8 slots) over one probe command, reduced to
# per-kernel ratios: MFMA busy, VALU / LDS active, parked (s_waitcnt / barrier), issue stalls, LDS issue stalls, average
# VMEM latency (SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM).   bash tools/pmc_waits.sh <tag> <filter> python3 tools/op_probe.py ...
set -e
TAG=$1; FILTER=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcw_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o p -- "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o p -- "$@" > $OUT/b.log 2>&1
python3 - "$OUT" "$FILTER" <<'PY'
import csv, glob, collections, sys, json
out, flt = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, c in acc.items():
    g = {n: sum(v) / len(v) for n, v in c.items()}
    cyc = g["GRBM_GUI_ACTIVE"] / 8
    wc = g["SQ_WAVE_CYCLES"]
    e = {"kernel_cycles": round(cyc), "mfma_busy": round(g["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc, 3),
         "valu_active_per_simd": round(g["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, 3),
         "lds_active_per_simd": round(g["SQ_ACTIVE_INST_LDS"] * 4 / 1024 / cyc, 3),
         "waves_per_simd": round(wc * 4 / 1024 / cyc, 2),
         "wait_any": round(g["SQ_WAIT_ANY"] / wc, 3), "wait_inst": round(g["SQ_WAIT_INST_ANY"] / wc, 3),
         "active": round(g["SQ_ACTIVE_INST_ANY"] / wc, 3), "wait_inst_lds": round(g.get("SQ_WAIT_INST_LDS", 0) / wc, 3),
         "vmem_latency_cycles": round(4 * g.get("SQ_INST_LEVEL_VMEM", 0) / max(g.get("SQ_INSTS_VMEM", 1), 1)),
         "lds_unit_busy": round(g.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / cyc, 3),
         "insts_per_wave_cycle": {n[9:]: round(g[n]) for n in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_VMEM") if n in g}}
    res[k[:100]] = e
print(json.dumps(res, indent=1))
PY
