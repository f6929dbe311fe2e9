#!/bin/bash
# Regenerates the judged profile artefacts of a round on the GPU box (run through gpurun from the repo root):
#   gpurun_out/prof/bench_default.json  default `python bench.py` line
#   gpurun_out/prof/kernel_stats.{csv,txt}  rocprofv3 --kernel-trace --stats of the bench command
#   gpurun_out/prof/pmc_traffic.{json,txt}  FETCH_SIZE / WRITE_SIZE (separate passes), per launch
#   gpurun_out/prof/pmc_sq.txt              SQ instruction counters, per launch
# Copy them to profiles/rNN_* afterwards (profiles/ is tracked, gpurun_out/ is scratch).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/prof
mkdir -p $O
CMD="bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-batch --no-train-step --no-stage-path --no-spatial --no-moving-camera"
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
echo "bench done" && cut -c1-200 $O/bench_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-spatial > $O/stats.log 2>&1 || exit 1
f=$(find $O/stats -name '*_kernel_stats.csv' | sort | tail -1)
cp "$f" $O/kernel_stats.csv
python3 profiles/summarize.py $O/stats > $O/kernel_stats.txt
rm -rf $O/stats
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $CMD > $O/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $CMD > $O/write.log 2>&1 || exit 1
python3 profiles/make_pmc_traffic.py $O/fetch $O/write $O/pmc_traffic "python3 $CMD" > $O/pmc_traffic.log 2>&1
rm -rf $O/fetch $O/write
echo "traffic done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python3 $CMD > $O/sq.log 2>&1 || exit 1
python3 profiles/pmc_summary.py $O/sq > $O/pmc_sq.txt
rm -rf $O/sq
echo "sq done"
# issue-cycle counters (round 4): how busy the SIMDs' VALU issue really is -- measured, not priced from an instruction mix.
# Quad-cycle units (MI355X_MICROARCH.md); bench.py turns them into roofline.valu_issue / fwd_bwd.roofline.valu_issue.
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/sqi -- python3 $CMD > $O/sqi.log 2>&1 || { tail -5 $O/sqi.log; exit 1; }
python3 profiles/pmc_summary.py $O/sqi > $O/pmc_sq_issue.txt
rm -rf $O/sqi
echo "sq issue done"
# LDS-side counters of the renderers (round 5): instruction counts by kind, LDS issue stalls, bank conflicts, lane utilisation.  (No
# counter separates waves parked at s_waitcnt from waves parked at s_barrier; see tools/gpu/pmc_wait_split.sh.)
bash tools/gpu/pmc_wait_split.sh $O "$CMD" || exit 1
ls -la $O
