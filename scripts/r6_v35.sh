cd $GRAFT_REPO_ROOT
REPO=$PWD
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-60)
  rm -rf /tmp/pmc_sq; mkdir -p /tmp/pmc_sq
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_sq -o p -- python "$REPO/bench.py" --steps 2 --warmup 1 --repeats 1 --graphs 0 --no-cpu-baseline --no-pmc --no-configs ) > /tmp/pmc_sq.log 2>&1
  python scripts/pmc_sq_by_kernel.py "$(find /tmp/pmc_sq -name '*counter_collection.csv' | head -n 1)" 2>&1 | grep -E "k_gemm_nn2<13, false, false, 2|k_gemm_nn2<7, false, false, 2|k_gemm_tn_ws<7, 13, false, 2|k_gemm_tn_split<13, 7, false, (false|true), [23]|k_edge_aggregate|k_edge_bwd_cls" | cut -c1-330
  echo
done > gpurun_out/r6_v35_sq_counters.txt 2>&1
cat gpurun_out/r6_v35_sq_counters.txt
