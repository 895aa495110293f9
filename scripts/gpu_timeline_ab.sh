# launch-timeline fit (scripts/timeline_fit.py) of the bench rounds for several settings of one environment knob, one -DHDSM_TIMELINE build
# usage: bash scripts/gpu_timeline_ab.sh VAR "v1 v2 ..." [bench args]
var=$1; vals=$2; shift 2
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp multi_agent_pkgs_amd/libhdsm.so /tmp/libhdsm_orig.so
make -C multi_agent_pkgs_amd/csrc -B CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-parameter -DHDSM_TIMELINE $EXTRA_DEFS" 2>&1 | grep -E "error"
for v in $vals; do
  rm -f /tmp/timeline.bin
  env $var=$v HDSM_TIMELINE_DUMP=/tmp/timeline.bin timeout 900 python bench.py --no-cpu-baseline --no-secondary "$@" > /tmp/timeline_bench.log 2>&1
  python scripts/timeline_fit.py /tmp/timeline.bin 20 "$var=$v" | tee -a gpurun_out/timeline_ab.jsonl
done
cp /tmp/libhdsm_orig.so multi_agent_pkgs_amd/libhdsm.so
