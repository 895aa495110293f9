# round 6, final binary: the -m gpu suite, the fuzz with 3000 configurations, then all four parts of the round's evidence (scripts/gpu_r6_evidence.sh)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4 | tee gpurun_out/r06_gpu_tests.log
HDSM_FUZZ_CASES=3000 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -s 2>&1 | grep -i "fuzz:\|passed\|failed" | tee gpurun_out/r06_fuzz.txt
bash scripts/gpu_r6_evidence.sh r06 "1 2 3 4"
