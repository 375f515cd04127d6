#!/usr/bin/env python3
"""Prints the top rows of a rocprofv3 kernel_stats.csv (average duration per kernel).  tools/kstats.py <csv> [n]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for r in rows[:n]:
    name = r["Name"].replace("vitseg::(anonymous namespace)::", "").replace("vitseg::", "")[:70]
    print(f"{float(r['AverageNs']) / 1e3:9.1f} us  x{r['Calls']:>5}  {float(r['Percentage']):5.1f}%  {name}")
