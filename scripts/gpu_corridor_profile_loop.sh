# where the corridor kernel's cycles go in the device-resident loop of the forest (a -DCD_PROFILE build, made here:
# make -C multi_agent_pkgs_amd/csrc OUT=../libhdsm_prof.so CXXFLAGS="-O3 -std=c++17 -fPIC -Wno-unused-parameter -DCD_PROFILE")
# usage: bash scripts/gpu_corridor_profile_loop.sh [tag]
TAG=${1:-r05c}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
[ -f multi_agent_pkgs_amd/libhdsm_prof.so ] || make -C multi_agent_pkgs_amd/csrc -s OUT=../libhdsm_prof.so CXXFLAGS="-O3 -std=c++17 -fPIC -Wno-unused-parameter -DCD_PROFILE" 2>&1 | grep error
HDSM_LIBRARY=$GRAFT_REPO_ROOT/multi_agent_pkgs_amd/libhdsm_prof.so timeout 600 python bench.py --no-cpu-baseline --no-secondary --scenario forest --agents 256 --first-round 60 --repeats 1 > gpurun_out/$TAG/loop_profile.json 2> gpurun_out/$TAG/loop_profile.err
python - <<PY
import json
d = json.loads(open("gpurun_out/$TAG/loop_profile.json").read().strip().splitlines()[-1])
print(json.dumps(d.get("device_resident_loop"), indent=1))
PY
