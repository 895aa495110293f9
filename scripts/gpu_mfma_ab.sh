# A/B of the MFMA leaf test (opt-in build -DHDSM_LEAF_MFMA) against the default build on the bench line, with counters.
# usage (GPU box): bash scripts/gpu_mfma_ab.sh   -> gpurun_out/mfma_ab/
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/mfma_ab; mkdir -p $OUT
export TMPDIR=/tmp
cp multi_agent_pkgs_amd/libhdsm.so /tmp/libhdsm_default.so
one() { tag=$1
  for k in 1 2 3; do python bench.py --no-cpu-baseline > $OUT/${tag}_bench$k.json 2> $OUT/${tag}_bench$k.err; done
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/${tag}_pmc_mfma -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/${tag}_pmc_mfma.json 2> $OUT/${tag}_pmc_mfma.err)
  # FETCH_SIZE and WRITE_SIZE each in its OWN pass (together the tool aborts on this box and then hangs in finalisation)
  for C in WRITE_SIZE FETCH_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${tag}_pmc_$C -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/${tag}_pmc_$C.json 2> $OUT/${tag}_pmc_$C.err)
  done
}
one default
make -C multi_agent_pkgs_amd/csrc -B CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-parameter -DHDSM_LEAF_MFMA" 2>&1 | grep -E "error"
one mfma
cp /tmp/libhdsm_default.so multi_agent_pkgs_amd/libhdsm.so
python scripts/summarize_mfma_ab.py
