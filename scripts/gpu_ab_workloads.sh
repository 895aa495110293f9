# A/B of prebuilt libraries (multi_agent_pkgs_amd/libhdsm_<name>.so, scripts/build_variants.sh) on the secondary workloads:
# cfg 5 (4096 x H15, forest + wall + forest), cfg 3 (256 agents, pillar forest), 4096 x H15 circle.   usage: gpu_ab_workloads.sh "base new"
cd $GRAFT_REPO_ROOT
cp multi_agent_pkgs_amd/libhdsm.so /tmp/libhdsm_orig.so
run() { python bench.py --no-cpu-baseline --no-secondary --no-event-pass --repeats 2 "${@:3}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2], '%.4f ms' % d['ms_per_step'], 'limit', d['limit_instances_timed_rounds'], 'failed', d['failed_instances_timed_rounds'])" "$1" "$2"; }
for rep in 1 2; do
for n in $1; do
  cp multi_agent_pkgs_amd/libhdsm_$n.so multi_agent_pkgs_amd/libhdsm.so
  run $n cfg5 --scenario fwf --agents 4096 --horizon 15 --first-round 8 --steps 6 --warmup 2
  run $n cfg3 --scenario forest --agents 256 --first-round 60 --steps 8 --warmup 2
  run $n c4096h15 --agents 4096 --horizon 15 --first-round 20 --steps 8 --warmup 2
done
done
cp /tmp/libhdsm_orig.so multi_agent_pkgs_amd/libhdsm.so
