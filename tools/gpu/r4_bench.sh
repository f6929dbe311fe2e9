cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err; echo rc=$?
python - <<'P'
import json
d=json.loads(open('gpurun_out/final/bench_default.json').readline())
print(json.dumps({k:d[k] for k in ('value','ms_per_step','per_frame_events','roofline','camera_batch','parity')},indent=0)[:3500])
print(json.dumps(d['fwd_bwd'].get('roofline'),indent=0)[:2500])
print(d['frame_roofline'].get('target_60pct_hbm'), d['frame_roofline'].get('pmc'))
print({k:(v.get('value') if isinstance(v,dict) else v) for k,v in d['fwd_bwd'].items() if k!='roofline'})
print(d.get('train_step'), d.get('stage_path'), d.get('moving_camera',{}).get('value'), d.get('cpu_baseline'))
P
