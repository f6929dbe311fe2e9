#!/usr/bin/env python3
"""A window of a rocprofv3 --kernel-trace CSV as a timeline across queues: timeline_window.py <dir> [frames=2]
Prints the kernels between the (k)th and (k + frames)th k_cull_compact in the middle of the trace, one column per queue,
and the fraction of the window during which a renderer / a sort-chain kernel was running."""
import csv, sys, glob, re
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_cull_compact' in r['Kernel_Name']]
k, k2 = idx[len(idx) * 2 // 3], idx[len(idx) * 2 // 3 + frames]
t0, t1 = int(rows[k]['Start_Timestamp']), int(rows[k2]['Start_Timestamp'])
queues = sorted({r.get('Queue_Id') for r in rows[k:k2]})
print(f"# window: {frames} frame starts, {(t1 - t0) / 1000:.1f} us = {(t1 - t0) / 1000 / frames:.1f} us per frame; queues {queues}")
busy = {}
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if e < t0 or s > t1: continue
    n = re.sub(r"\(anonymous namespace\)::", "", r['Kernel_Name'])
    n = re.sub(r"^void ", "", n).split("(")[0].split("<")[0].split("::")[-1][:26]
    q = queues.index(r.get('Queue_Id')) if r.get('Queue_Id') in queues else -1
    print(f"{(s - t0) / 1000:8.1f} {(e - t0) / 1000:8.1f} {(e - s) / 1000:7.1f}  " + "    " * max(q, 0) + f"q{q} {n}")
    kind = "render" if "render" in n else ("records" if "build_records" in n else "chain")
    busy.setdefault(kind, []).append((max(s, t0), min(e, t1)))
for kind, iv in busy.items():
    iv.sort(); tot = 0; cur_s, cur_e = iv[0]
    for s, e in iv[1:]:
        if s > cur_e: tot += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    tot += cur_e - cur_s
    print(f"# {kind}: some kernel of this kind running during {100 * tot / (t1 - t0):.0f} % of the window")
