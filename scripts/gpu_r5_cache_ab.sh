cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -2
for v in 1 0; do
HDSM_POLY_CACHE=$v timeout 900 bash scripts/gpu_dloop_trace.sh r05c_dloop_forest --scenario forest --agents 256 --first-round 60 > gpurun_out/dloop_tmp.log 2>&1; python -c "
import json,sys
d=json.load(open('gpurun_out/r05c_dloop_forest/dloop_trace.json'))
print('cache $v forest', d['round_period_us_mean'], {k: round(v['dur_us'],1) for k,v in d['kernels_per_round_us'].items()})"
done
timeout 900 bash scripts/gpu_dloop_trace.sh r05_dloop_fwf --scenario fwf --agents 4096 --horizon 15 --first-round 8 --steps 6 --warmup 2 > gpurun_out/dloop_tmp.log 2>&1; python -c "
import json,sys
d=json.load(open('gpurun_out/r05_dloop_fwf/dloop_trace.json'))
print('fwf', d['round_period_us_mean'], {k: round(v['dur_us'],1) for k,v in d['kernels_per_round_us'].items()})"
rm -rf gpurun_out/r05c_dloop_forest/trace gpurun_out/r05_dloop_fwf/trace
