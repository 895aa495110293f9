# two ranks sharing the one GPU of the box (gloo for the collective): exercises the N>1 flow of bench.py end to end
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 20 --warmup 5 --dist-backend gloo > gpurun_out/two_rank.json 2> gpurun_out/two_rank.err
echo "rc=$?"; tail -2 gpurun_out/two_rank.err | cut -c1-300; cut -c1-900 gpurun_out/two_rank.json
