cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/final
LCGS_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/final/bench_force_dist.json 2> gpurun_out/final/bench_force_dist.err; echo "force_dist rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/final/bench_force_dist.json').readline())
print(d.get('leg_errors'), json.dumps(d['train_step'],indent=0))
P
LCGS_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 2 --splats 1000000 --steps 6 --warmup 2 --no-cpu-baseline --no-stage-path --no-spatial --collective torch > gpurun_out/final/bench_two_ranks.json 2> gpurun_out/final/bench_two_ranks.err; echo "two ranks rc=$?"
python - <<'P'
import json
d=[json.loads(l) for l in open('gpurun_out/final/bench_two_ranks.json') if l.startswith('{')][-1]
print(d.get('leg_errors'), json.dumps(d['train_step'],indent=0))
P
