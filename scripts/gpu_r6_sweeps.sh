# round 6: the instances of the bench launches by the number of sweeps they make (scripts/timeline_sweeps.py on a -DHDSM_TIMELINE build)
cd $GRAFT_REPO_ROOT
bash scripts/gpu_timeline.sh --steps 20 --warmup 5 > gpurun_out/r6_sweeps_timeline.log 2>&1
python scripts/timeline_sweeps.py gpurun_out/timeline.bin 20 > gpurun_out/r6_sweeps.json
rm -f gpurun_out/timeline.bin
python - <<'P'
import json
d=json.load(open('gpurun_out/r6_sweeps.json'))
for r in d['launches']:
    print(r['launch'], round(r['slowest_us'],1), {k:(v['n'],v['dur_mean'],v['dur_max'],v['runs_mean'],v['ops_mean'],v['seq_sweep_us_mean']) for k,v in r.items() if k.startswith('sweeps=')}, r['slowest'])
P
