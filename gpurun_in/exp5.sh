cd $GRAFT_REPO_ROOT
B="python bench.py --steps 50 --warmup 10 --no-stage-path"
for rep in 1 2; do for m in "" "--exp-morton"; do
timeout 200 $B $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$m', d['value'], 'batch', d['camera_batch']['value'], 'fwdbwd', d['fwd_bwd']['value'], d['fwd_bwd']['compact_rows']['value'], d['fwd_bwd']['backward_stages_ms'], {k:v['value'] for k,v in d['train_step'].items()}, d['stages_ms'], d['parity'])"
done; done
