cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t4
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t4/pytest.log 2>&1
rc=$?
tail -25 gpurun_out/t4/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout 200 python bench.py 2>gpurun_out/t4/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['fwd_bwd']['value'], d['fwd_bwd']['compact_rows'], d['spatially_ordered'], d['parity'])"
