cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_owner.py tests/test_gpu_comm.py -m gpu -q -x 2>&1 | tail -4 || exit 1
LCGS_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_force_dist.json 2> gpurun_out/bench_force_dist.err; echo "force_dist rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/bench_force_dist.json').readline())
print(json.dumps(d.get('train_step'),indent=0)[:1500]); print(d.get('leg_errors'))
P
