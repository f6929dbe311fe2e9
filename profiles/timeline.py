import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find frames: k_cull_compact starts
idx = [i for i, r in enumerate(rows) if 'k_cull_compact' in r['Kernel_Name']]
k = idx[len(idx) // 2]
k2 = idx[len(idx) // 2 + 1]
t0 = int(rows[k]['Start_Timestamp'])
for r in rows[k - 2:k2 + 1]:
    n = r['Kernel_Name'].split('(')[0].split('::')[-1][:40]
    s = (int(r['Start_Timestamp']) - t0) / 1000; e = (int(r['End_Timestamp']) - t0) / 1000
    print(f"{n:42s} start {s:9.1f} end {e:9.1f} dur {e - s:7.1f} q={r.get('Queue_Id')} grid={r.get('Grid_Size_X', r.get('Grid_Size'))}")
