#!/usr/bin/env python3
"""Per-kernel HBM traffic per launch from the FETCH_SIZE / WRITE_SIZE passes of
`tools/collect_pmc.sh bench_<prec> python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision <prec>`.
gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE is in KiB-like units of 1024 B and reports HALF the bytes
of wide coalesced reads -> bytes_read = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is exact for 16-B-per-lane stores."""
import collections
import csv
import glob
import json
import sys

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
rnd = sys.argv[2] if len(sys.argv) > 2 else "r03"
base = f"gpurun_out/pmc_bench_{prec}"
acc = collections.defaultdict(lambda: {"fetch": [], "write": []})
for kind in ("fetch", "write"):
    rows = [r for f in glob.glob(f"{base}/{kind}/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
    for r in rows:
        if r["Counter_Name"] != {"fetch": "FETCH_SIZE", "write": "WRITE_SIZE"}[kind]:
            continue
        name = r["Kernel_Name"]
        if "vitseg" not in name:
            continue
        short = name.replace("(anonymous namespace)::", "").replace("vitseg::", "")
        if short.endswith(")"):                      # drop the argument list, keep template arguments
            depth = 0
            for i in range(len(short) - 1, -1, -1):
                depth += short[i] == ")"
                depth -= short[i] == "("
                if depth == 0:
                    short = short[:i]
                    break
        short = short.replace("void ", "")
        acc[short][kind].append(float(r["Counter_Value"]))
out = {}
for k, v in acc.items():
    if not v["fetch"] or not v["write"]:
        continue
    rd = 2.0 * 1024.0 * sum(v["fetch"]) / len(v["fetch"])
    wr = 1024.0 * sum(v["write"]) / len(v["write"])
    out[k] = {"launches_sampled": len(v["fetch"]), "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
              "hbm_bytes_per_launch": rd + wr}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 2; "
                     "read = 2 * FETCH_SIZE * 1024 (gfx950 wide-read correction), write = WRITE_SIZE * 1024",
           "precision": prec, "kernels": out}, open(f"profiles/{rnd}_traffic_{prec}.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
    print(f"{k[:70]:70s} read {v['read_bytes_per_launch']/1e6:9.1f} MB  write {v['write_bytes_per_launch']/1e6:9.1f} MB  x{v['launches_sampled']}")
