# One frame of the three stage-level operators (exact mode) as a kernel timeline: gpurun -- bash tools/gpu/timeline_stage.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/tls
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tls/raw -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-train-step --no-backward --no-batch --no-spatial --no-moving-camera > gpurun_out/tls/bench.json 2> gpurun_out/tls/err.log
python3 - <<'P' > gpurun_out/tls/timeline.txt 2>&1
import csv, glob, re
f = glob.glob('gpurun_out/tls/raw/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_sh_process' in r['Kernel_Name']]
k, k2 = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = int(rows[k]['Start_Timestamp'])
print(f"# one frame through SHProcessor::process / GSProjector::forward / GSTileSplatter::forward (exact mode): {(int(rows[k2]['Start_Timestamp']) - t0) / 1000:.1f} us from k_sh_process to k_sh_process")
for r in rows[k:k2 + 1]:
    n = re.sub(r"\(anonymous namespace\)::", "", r['Kernel_Name'])
    n = re.sub(r"^void ", "", n).split("(")[0].split("<")[0].split("::")[-1][:28]
    s = (int(r['Start_Timestamp']) - t0) / 1000; e = (int(r['End_Timestamp']) - t0) / 1000
    print(f"{n:30s} start {s:8.1f} end {e:8.1f} dur {e - s:6.1f} queue={r.get('Queue_Id')}")
P
rm -rf gpurun_out/tls/raw
cat gpurun_out/tls/timeline.txt
