cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/gpu/timeline_step.sh | tail -70
