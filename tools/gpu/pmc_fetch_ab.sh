# FETCH_SIZE of the forward renderer under two settings of an environment hook:
#   gpurun -- bash tools/gpu/pmc_fetch_ab.sh LCGS_TILE_ORDER_XCD 0 1
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
V=$1; shift
CMD="bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-batch --no-train-step --no-stage-path --no-spatial --no-moving-camera --no-backward"
for val in "$@"; do
  export $V=$val
  rm -rf gpurun_out/pmcab_$val
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcab_$val -- python3 $CMD > gpurun_out/pmcab_$val.log 2>&1 || { tail -3 gpurun_out/pmcab_$val.log; exit 1; }
  python3 - <<PY
import csv, glob
f=glob.glob('gpurun_out/pmcab_$val/**/*counter_collection.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'k_render_forward_b' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE']
v=[float(r['Counter_Value']) for r in rows]
print('$V=$val k_render_forward_b launches', len(v), 'FETCH_SIZE mean (raw units as reported)', sum(v)/max(len(v),1), 'min', min(v), 'max', max(v))
PY
  rm -rf gpurun_out/pmcab_$val
done
