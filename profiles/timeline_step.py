import csv, sys, glob, re
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_render_backward' in r['Kernel_Name']]
k = idx[len(idx) // 2]
# go back to the cull of this step
c = max(i for i in range(k) if 'k_cull_compact' in rows[i]['Kernel_Name'])
c2 = min(i for i in range(k, len(rows)) if 'k_cull_compact' in rows[i]['Kernel_Name'])
t0 = int(rows[c]['Start_Timestamp'])
for r in rows[c - 2:c2 + 1]:
    n = re.sub(r"\(anonymous namespace\)::", "", r['Kernel_Name'])
    n = re.sub(r"^void ", "", n).split("(")[0].split("<")[0].split("::")[-1][:28]
    if n in ("k_hist", "k_rowscan", "k_scatter"): continue
    s = (int(r['Start_Timestamp']) - t0) / 1000; e = (int(r['End_Timestamp']) - t0) / 1000
    print(f"{n:30s} start {s:8.1f} end {e:8.1f} dur {e - s:6.1f} queue={r.get('Queue_Id')}")
