# instruction-cache counters of the bench launches (one --pmc pass per group, kernel trace only)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/icache; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  N=$(echo $C | tr ' ' '_')
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-event-pass > $OUT/$N.json 2> $OUT/$N.err)
  echo "== $C rc=$?"; tail -2 $OUT/$N.err | cut -c1-200
  python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc_$N/*counter_collection.csv")
if f:
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if "k_replan" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (s, n) in acc.items(): print(k, "mean per dispatch", s / max(n, 1), "dispatches", n)
PY
done
