bash scripts/gpu_check.sh 2>&1 | grep -v "^+" | tail -8
bash scripts/gpu_prof_phases.sh 2>&1 | grep -v "^+" | tail -14
