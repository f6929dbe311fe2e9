#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel calls / average / share."""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import kernel_key

d = sys.argv[1]
f = glob.glob(d + "/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {f}\n# total kernel time {tot/1e6:.3f} ms")
for r in rows:
    n = kernel_key(r["Name"])[:44]
    print("%-44s calls=%5s avg_us=%9.1f total_ms=%8.3f pct=%6.2f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                               float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
