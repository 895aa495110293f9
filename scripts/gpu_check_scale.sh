cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/scale_check
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log
echo "bench_md5=$(md5sum bench.py | cut -c1-12) so_md5=$(md5sum multi_agent_pkgs_amd/libhdsm.so | cut -c1-12)"
p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['agents'], 'H', d['config']['horizon'], 'replans/s', round(d['value']), 'kernel_ms', round(d['kernel_ms_mean'],4), 'ms_step', round(d['ms_per_step'],4), 'failed', d['failed_instances_recorded'], d['solver_stats_last_round'], 'frac', round(d['roofline']['frac'],4))"; }
run() { tag=$1; shift; s=$(date +%s.%N); timeout 1500 python bench.py --no-cpu-baseline "$@" > gpurun_out/scale_check/$tag.json 2> gpurun_out/scale_check/$tag.err; rc=$?
  e=$(date +%s.%N); echo "$tag rc=$rc wall_s=$(python -c "print(round($e-$s,1))")"; tail -1 gpurun_out/scale_check/$tag.json | p; }
run a64
run a128 --agents 128 --steps 20 --warmup 5
run a256 --agents 256 --steps 20 --warmup 5
run a1024early --agents 1024 --steps 10 --warmup 2 --first-round 30
run a1024late --agents 1024 --steps 10 --warmup 2 --first-round 150
run a4096h15 --agents 4096 --horizon 15 --steps 6 --warmup 2 --first-round 20
