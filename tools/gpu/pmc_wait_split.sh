# The wait-split and LDS counter passes of the two renderers (separate rocprofv3 --pmc runs, no trace domains):
#   gpurun -- bash tools/gpu/pmc_wait_split.sh [outdir] ["bench.py args"]      -> <outdir>/pmc_wait_split.txt, pmc_lds.txt
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=${1:-gpurun_out/prof}
CMD=${2:-"bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-batch --no-train-step --no-stage-path --no-spatial --no-moving-camera"}
mkdir -p $O
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/ws -- python3 $CMD > $O/ws.log 2>&1 || { tail -5 $O/ws.log; exit 1; }
python3 profiles/pmc_summary.py $O/ws > $O/pmc_wait_split.txt
rm -rf $O/ws
# (the derived metrics LdsLatency / VmemLatency / SmemLatency -- accumulate(SQ_INST_LEVEL_*, HIGH_RES) / count, which would bound the
#  cycles a wave can have been parked at a waitcnt -- abort rocprofv3 on this pool with an incomplete dispatch after minutes of silence:
#  not collected.  The raw SQ_INST_LEVEL_* counters without the accumulate are event counts, not cycles.)
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/lds -- python3 $CMD > $O/lds.log 2>&1 || { tail -5 $O/lds.log; exit 1; }
python3 profiles/pmc_summary.py $O/lds > $O/pmc_lds.txt
rm -rf $O/lds
echo "wait split done"
