# round 6: where the solver / scanner wavefronts of the headline launch sit, and whether sharing a SIMD costs (scripts/timeline_simd.py)
cd $GRAFT_REPO_ROOT
bash scripts/gpu_timeline.sh --steps 20 --warmup 5 > gpurun_out/r6_simd_timeline.log 2>&1
tail -3 gpurun_out/r6_simd_timeline.log | cut -c1-300
python scripts/timeline_simd.py gpurun_out/timeline.bin 20 | tee gpurun_out/r6_simd.json
python scripts/timeline_fit.py gpurun_out/timeline.bin 20 simd > gpurun_out/r6_simd_fit.json
rm -f gpurun_out/timeline.bin
