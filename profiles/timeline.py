#!/usr/bin/env python3
"""One frame of a rocprofv3 --kernel-trace CSV as a timeline (start/end in us relative to the frame's cull kernel)."""
import csv, sys, glob, re
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_cull_compact' in r['Kernel_Name']]
k, k2 = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = int(rows[k]['Start_Timestamp'])
for r in rows[k - 3:k2 + 1]:
    n = re.sub(r"\(anonymous namespace\)::", "", r['Kernel_Name'])
    n = re.sub(r"^void ", "", n).split("(")[0].split("<")[0].split("::")[-1][:28]
    s = (int(r['Start_Timestamp']) - t0) / 1000; e = (int(r['End_Timestamp']) - t0) / 1000
    print(f"{n:30s} start {s:8.1f} end {e:8.1f} dur {e - s:6.1f} queue={r.get('Queue_Id')}")
