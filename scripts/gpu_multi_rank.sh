# N ranks of examples/sharded_loop on ONE GPU (flow check of the RCCL exchange: unique id through a file, hdsm_comm_create,
# publish + one all-gather per round). RCCL may refuse several ranks on one device; the result is recorded either way.
# usage (GPU box): bash scripts/gpu_multi_rank.sh [ranks = 2] [agents = 64] [rounds = 40]
R=${1:-2}; A=${2:-64}; K=${3:-40}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/multi_rank; rm -f /tmp/hdsm_uid
export HSA_ENABLE_IPC_MODE_LEGACY=0
for r in $(seq 0 $((R-1))); do
  timeout 240 ./examples/sharded_loop $r $R /tmp/hdsm_uid $A $K > gpurun_out/multi_rank/rank$r.log 2>&1 &
done
wait
tail -n 3 gpurun_out/multi_rank/rank*.log
