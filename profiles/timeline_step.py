"""One plain forward+backward step (the bench line's fwd_bwd leg: lcgs_render_forward(keep_state) + lcgs_render_backward, dense
rows) of a rocprofv3 --kernel-trace CSV as a timeline: every kernel from one step's cull pass to the next step's."""
import csv, sys, glob, re
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", r['Kernel_Name'])).split("(")[0].split("<")[0].split("::")[-1][:28]
idx = [i for i, r in enumerate(rows) if 'k_render_backward' in r['Kernel_Name']]
# the fwd_bwd leg runs first; its steps have no loss kernel between the renderer and the render-backward
for k in idx[len(idx) // 8:]:
    c = max(i for i in range(k) if 'k_cull_compact' in rows[i]['Kernel_Name'])
    c2 = min((i for i in range(k, len(rows)) if 'k_cull_compact' in rows[i]['Kernel_Name']), default=len(rows) - 1)
    names = [name(r) for r in rows[c:c2]]
    if 'k_l2_loss_backward' not in names and 'k_preprocess_backward_jac' in names and names.count('k_cull_compact') == 1:
        break
t0 = int(rows[c]['Start_Timestamp'])
print(f"# one forward+backward step, dense rows: {(int(rows[c2]['Start_Timestamp']) - t0) / 1000:.1f} us from cull to cull (rocprofv3 stretches kernels by ~12 %)")
for r in rows[c - 1:c2 + 1]:
    s = (int(r['Start_Timestamp']) - t0) / 1000; e = (int(r['End_Timestamp']) - t0) / 1000
    print(f"{name(r):30s} start {s:8.1f} end {e:8.1f} dur {e - s:6.1f} queue={r.get('Queue_Id')}")
