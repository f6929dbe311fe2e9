# bench.py with TWO real ranks on the one GPU of a pool box (process group on gloo -- RCCL refuses two ranks on a device): the
# launch, per-rank views, barriers, max-over-ranks timing and the per-leg error handling as on a node.
#   --collective torch: the gradient legs run over the process group;  --collective rccl: communicator creation fails on every
#   rank ("duplicate GPU") and the line must still come out, with the failure recorded.
# gpurun --timeout 900 -- bash tools/gpu/two_ranks_one_gpu.sh
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/final
export LCGS_BENCH_BACKEND=gloo
for coll in torch rccl; do
    timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2953$([ $coll = torch ] && echo 3 || echo 4) \
        bench.py --gpus 2 --steps 20 --warmup 3 --collective $coll --no-cpu-baseline --splats 1000000 \
        > gpurun_out/final/bench_two_ranks_one_gpu_gloo_$coll.json 2> gpurun_out/final/bench_two_ranks_one_gpu_gloo_$coll.err
    echo "$coll rc=$?"
    cut -c1-200 gpurun_out/final/bench_two_ranks_one_gpu_gloo_$coll.json
done
