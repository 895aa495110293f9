# kernel development loop on the GPU box: timeline fit + bench A/B of one env knob + op-level phase profile
# usage: bash scripts/gpu_kernel_check.sh VAR "v1 v2" [profile: 0/1]
var=${1:-HDSM_SCANNER}; vals=${2:-"0 1"}; prof=${3:-1}
cd $GRAFT_REPO_ROOT
bash scripts/gpu_timeline_ab.sh $var "$vals" 2>&1 | grep label
bash scripts/gpu_ab_env.sh $var "$vals $vals" 2>&1 | grep "^$var"
if [ "$prof" = "1" ]; then EXTRA_DEFS="-DHDSM_PROF_OP" bash scripts/gpu_prof_bench.sh 2>&1 | grep HDSM_PROFILE; fi
