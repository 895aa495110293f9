# the multi-rank flow of bench.py on a ONE-GPU box: 2 ranks share device 0, the exchange is host-staged (--dist-backend gloo).
# Checks the sharding / recording / timing logic of --gpus N; the RCCL path itself needs one GPU per rank (driver's node).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --dist-backend gloo --no-cpu-baseline > gpurun_out/r02/bench_2rank_gloo.json 2> gpurun_out/r02/bench_2rank_gloo.err
echo rc=$?; tail -c 600 gpurun_out/r02/bench_2rank_gloo.json; tail -5 gpurun_out/r02/bench_2rank_gloo.err
