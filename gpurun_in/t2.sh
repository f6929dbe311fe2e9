cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t2
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t2/pytest.log 2>&1
rc=$?
tail -25 gpurun_out/t2/pytest.log
[ $rc -eq 0 ] || exit $rc
B="python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-stage-path --no-batch"
for rep in 1 2; do
timeout 100 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], 'fwdbwd', d['fwd_bwd']['value'], d['train_step'])"
done
