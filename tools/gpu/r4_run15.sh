cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/final
LCGS_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 4 --splats 1000000 --steps 4 --warmup 1 --no-cpu-baseline --no-stage-path --no-spatial --collective torch > gpurun_out/final/bench_four_ranks.json 2> gpurun_out/final/bench_four_ranks.err; echo "four ranks rc=$?"
python - <<'P'
import json
d=[json.loads(l) for l in open('gpurun_out/final/bench_four_ranks.json') if l.startswith('{')][-1]
print(d.get('leg_errors'), d['n_gpus'], d['value'], json.dumps({k:{kk:v[kk] for kk in ('value','ms_per_step','xgmi_bytes_sent_per_gpu') if kk in v} for k,v in d['train_step'].items()}))
P
tail -3 gpurun_out/final/bench_four_ranks.err
