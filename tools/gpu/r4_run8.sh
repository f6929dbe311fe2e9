cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof
mkdir -p $O
CMD="bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-batch --no-train-step --no-stage-path --no-spatial --no-moving-camera"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/sqi -- python3 $CMD > $O/sqi.log 2>&1 || { tail -5 $O/sqi.log; exit 1; }
python3 profiles/pmc_summary.py $O/sqi > $O/pmc_sq_issue.txt
rm -rf $O/sqi
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sqj -- python3 $CMD > $O/sqj.log 2>&1 || { tail -5 $O/sqj.log; }
python3 profiles/pmc_summary.py $O/sqj > $O/pmc_sq_issue2.txt
rm -rf $O/sqj
grep "k_render\|k_cull\|k_build\|k_scatter \|k_preprocess" $O/pmc_sq_issue.txt $O/pmc_sq_issue2.txt
