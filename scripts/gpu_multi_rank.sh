# N ranks sharing the one GPU of the box (gloo for the collective): exercises the N>1 flow of bench.py end to end
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
N=${1:-4}
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29541 \
  bench.py --gpus $N --steps 20 --warmup 5 --dist-backend gloo > gpurun_out/multi_rank_$N.json 2> gpurun_out/multi_rank_$N.err
echo "N=$N rc=$?"; grep -v "Gloo\|socket\|amdgpu.ids" gpurun_out/multi_rank_$N.err | tail -3 | cut -c1-300; cut -c1-700 gpurun_out/multi_rank_$N.json
