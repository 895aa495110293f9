# attribution run: nonce + checksum + per-bench wall clock, stderr kept
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/scale_check
echo "nonce=$1 bench_md5=$(md5sum bench.py | cut -c1-12) host=$(hostname) t0=$(date +%s)"
run() { tag=$1; shift; s=$(date +%s.%N)
  timeout 2400 python bench.py --no-cpu-baseline "$@" > gpurun_out/scale_check/$tag.json 2> gpurun_out/scale_check/$tag.err; rc=$?
  e=$(date +%s.%N); echo "$tag rc=$rc wall_s=$(python -c "print(round($e-$s,1))") json_bytes=$(stat -c %s gpurun_out/scale_check/$tag.json) err_lines=$(wc -l < gpurun_out/scale_check/$tag.err)"
  tail -1 gpurun_out/scale_check/$tag.json | cut -c1-600; tail -3 gpurun_out/scale_check/$tag.err; }
run a128 --agents 128 --steps 20 --warmup 5
run a4096h15 --agents 4096 --horizon 15 --steps 6 --warmup 2 --first-round 20
