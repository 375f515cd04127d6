#!/usr/bin/env python3
"""Top kernels of a rocprofv3 results database (rocpd sqlite: the default output of rocprofv3 --kernel-trace --stats):
python tools/db_top.py <results.db> [launches-per-unit] -> name, calls, us per call, ms per unit, share."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
per = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(db.execute("select name, total_calls, total_duration from top_kernels"))
tot = sum(r[2] for r in rows)
for name, calls, dur in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{name[:104]:104s} {calls:6d} {dur / calls * 1e3:9.2f} us {dur / per:8.3f} ms {dur / tot * 100:5.1f} %")   # the view holds milliseconds
print(f"total {tot / per:.3f} ms per unit")
