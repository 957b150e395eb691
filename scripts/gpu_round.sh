#!/bin/bash
# One GPU-box visit: kernel tests, parity tests, smoke, bench, rocprof.  Small logs land in gpurun_out/ (<64 MiB!).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO="$PWD"
rm -rf gpurun_out; mkdir -p gpurun_out
export TMPDIR=/tmp
( rocm-smi --showproductname 2>/dev/null | head -8; nproc; free -g | head -2 ) > gpurun_out/gpu.txt
( time python __graft_entry__.py ) > gpurun_out/build.log 2>&1
if [[ "$*" == *alltests* ]]; then   # what the driver runs at round end
  timeout 1500 python -m pytest tests -m gpu -x -q --tb=short -rf --timeout 300 -p no:cacheprovider --durations=8 2>&1 | tail -n 60 > gpurun_out/test_all.log
  echo "alltests exit ${PIPESTATUS[0]}" >> gpurun_out/summary.txt
fi
if [[ "$*" == *poison* ]]; then   # deferred weight gradients start as NaN: any reader that runs before the join fails parity
  QAGNN_WGRAD_POISON=1 timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q --tb=short -rf --timeout 300 -p no:cacheprovider 2>&1 | tail -n 30 > gpurun_out/test_poison.log
  echo "poison exit ${PIPESTATUS[0]}" >> gpurun_out/summary.txt
fi
if [[ "$*" == *kernels* ]]; then
  timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q --tb=short -rf --timeout 180 -p no:cacheprovider 2>&1 | tail -n 300 > gpurun_out/test_kernels.log
  echo "kernels exit ${PIPESTATUS[0]}" >> gpurun_out/summary.txt
  # the non-default GEMM launch shapes go through the same tests
  for cfg in "QAGNN_NN_PERSIST=0 QAGNN_TN_STRIP=0 QAGNN_TN_SPLIT=0" "QAGNN_NN_PERSIST=2 QAGNN_TN_CHUNK=256 QAGNN_TN_SPLIT=0" "QAGNN_NN_PERSIST=1 QAGNN_NN_BLOCKS_PER_CU=1" "QAGNN_TN_SPLIT=0"; do
    env $cfg timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q --tb=short -rf --timeout 180 -p no:cacheprovider -k "gemm" 2>&1 | tail -n 40 >> gpurun_out/test_kernels_variants.log
    echo "kernels[$cfg] exit ${PIPESTATUS[0]}" >> gpurun_out/summary.txt
  done
fi
if [[ "$*" == *parity* ]]; then
  timeout 1200 python -X faulthandler -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -rf --timeout 300 -p no:cacheprovider > /tmp/parity_full.log 2>&1; ( head -c 40000 /tmp/parity_full.log; echo; echo "......"; tail -c 20000 /tmp/parity_full.log ) > gpurun_out/test_parity.log
  echo "parity exit $?" >> gpurun_out/summary.txt
fi
if [[ "$*" == *smoke* ]]; then
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -n 50 > gpurun_out/smoke.log
  echo "smoke exit ${PIPESTATUS[0]}" >> gpurun_out/summary.txt
fi
if [[ "$*" == *bench* ]]; then
  timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -n 50 > gpurun_out/bench.log
  echo "bench exit ${PIPESTATUS[0]}" >> gpurun_out/summary.txt
  timeout 300 python bench.py --steps 30 --warmup 5 --questions 2 --no-cpu-baseline 2>&1 | tail -n 1 > gpurun_out/bench_b10.log
fi
if [[ " $* " == *" prof "* ]]; then
  rm -rf /tmp/prof; mkdir -p /tmp/prof
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r2 -- python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-pmc ) 2>&1 | tail -n 30 > gpurun_out/prof.log
  echo "prof exit $?" >> gpurun_out/summary.txt
  mkdir -p gpurun_out/prof
  find /tmp/prof -name "*stats*.csv" -exec cp {} gpurun_out/prof/ \;
  python scripts/trace_by_shape.py /tmp/prof/r2_kernel_trace.csv > gpurun_out/prof/by_shape.txt 2>&1
  ls -la /tmp/prof/* | head -20 >> gpurun_out/prof.log
fi
if [[ " $* " == *" pmc "* ]]; then
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$ctr; mkdir -p /tmp/pmc_$ctr
    ( cd /tmp && timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$ctr -o p -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline ) 2>&1 | tail -n 5 > gpurun_out/pmc_$ctr.log
    python scripts/pmc_by_kernel.py /tmp/pmc_$ctr/p_counter_collection.csv > gpurun_out/pmc_$ctr.txt 2>&1
    ls /tmp/pmc_$ctr >> gpurun_out/pmc_$ctr.log
  done
  python scripts/pmc_edge_traffic.py /tmp/pmc_FETCH_SIZE/p_counter_collection.csv /tmp/pmc_WRITE_SIZE/p_counter_collection.csv 64000 208 > gpurun_out/pmc_edge_fwd.json 2> gpurun_out/pmc_edge_traffic.err
fi
if [[ "$*" == *sqpmc* ]]; then   # SQ counters of the GEMM micro-benchmark (own pass, kernel-trace only)
  rm -rf /tmp/sqpmc; mkdir -p /tmp/sqpmc
  ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d /tmp/sqpmc -o p -- "$REPO/tools/bin/gemm_ablate_BASE" ) 2>&1 | tail -n 12 > gpurun_out/sqpmc.log
  python scripts/pmc_table.py /tmp/sqpmc/p_counter_collection.csv > gpurun_out/sqpmc_gemm.txt 2>&1
  rm -rf /tmp/sqpmc2; mkdir -p /tmp/sqpmc2
  ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d /tmp/sqpmc2 -o p -- "$REPO/tools/bin/gemm_ablate_BASE" ) 2>&1 | tail -n 12 >> gpurun_out/sqpmc.log
  python scripts/pmc_table.py /tmp/sqpmc2/p_counter_collection.csv > gpurun_out/sqpmc_gemm2.txt 2>&1
fi
if [[ "$*" == *mfma* ]]; then
  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/mfma_peak.hip && /tmp/mfma_peak > gpurun_out/mfma_peak.txt 2>&1
fi
for arg in "$@"; do
  if [[ "$arg" == ab1:* ]]; then   # one interleaved pair (GPU minutes are short)
    var="${arg#ab1:}"
    for v in 0 1 0 1; do
      echo "$var=$v" >> gpurun_out/ab_$var.txt
      env $var=$v timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline 2>&1 | tail -n 1 > /tmp/ab_line.txt
      cut -c1-140 /tmp/ab_line.txt >> gpurun_out/ab_$var.txt
      grep -o '"breakdown_ms_per_step.*' /tmp/ab_line.txt | cut -c1-160 >> gpurun_out/ab_$var.txt
    done
  fi
  if [[ "$arg" == ab10:* ]]; then   # the same at the reference's own mini-batch (2 questions x 5 choices): host-bound
    var="${arg#ab10:}"
    for v in 0 1 0 1; do
      echo "$var=$v (B=10)" >> gpurun_out/ab10_$var.txt
      env $var=$v timeout 300 python bench.py --steps 60 --warmup 8 --questions 2 --no-cpu-baseline 2>&1 | tail -n 1 | cut -c1-140 >> gpurun_out/ab10_$var.txt
    done
  fi
  if [[ "$arg" == ab:* ]]; then
    var="${arg#ab:}"
    for rep in 1 2; do for v in 0 1; do
      echo "$var=$v" >> gpurun_out/ab_$var.txt
      env $var=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -n 1 | cut -c1-140 >> gpurun_out/ab_$var.txt
    done; done
  fi
done
if [[ "$*" == *dp2* ]]; then
  QAGNN_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 5 --warmup 2 2>&1 | tail -n 20 > gpurun_out/bench_dp2_shared_gpu.log
  echo "dp2 exit ${PIPESTATUS[0]}" >> gpurun_out/summary.txt
fi
if [[ "$*" == *census* ]]; then
  timeout 300 python tools/op_census.py 2>&1 | cut -c1-260 | grep -v "Warning\|warn" | head -n 220 > gpurun_out/op_census.txt
fi
if [[ "$*" == *hostprof* ]]; then
  timeout 300 python -m cProfile -s tottime bench.py --steps 40 --warmup 5 --questions 2 --no-cpu-baseline 2>&1 | head -n 70 > gpurun_out/hostprof_b10.txt
fi
if [[ "$*" == *unrollab* ]]; then   # edge kernels with 4 / 6 / 8 edges in flight per wave (library variants prebuilt by tools/build_micro.sh)
  for u in 6 8; do
    QAGNN_LIB=$REPO/tools/bin/libqagnn_hip_u$u.so timeout 300 python -m pytest tests/test_hip_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "edge or hop" 2>&1 | tail -n 3 >> gpurun_out/test_unroll_variants.log
  done
  for rep in 1; do for u in 4 6 8 4; do
    lib=$REPO/qagnn_amd/libqagnn_hip.so; [[ $u != 4 ]] && lib=$REPO/tools/bin/libqagnn_hip_u$u.so
    echo "EDGE_UNROLL=$u" >> gpurun_out/ab_unroll.txt
    QAGNN_WGRAD_OVERLAP=0 QAGNN_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline 2>&1 | tail -n 1 > /tmp/ab_line.txt
    cut -c1-140 /tmp/ab_line.txt >> gpurun_out/ab_unroll.txt
    grep -o '"breakdown_ms_per_step.*' /tmp/ab_line.txt | cut -c1-160 >> gpurun_out/ab_unroll.txt
  done; done
fi
if [[ "$*" == *gather* ]]; then   # row-gather rate of the memory system (tools/gather_micro.hip, prebuilt by tools/build_micro.sh)
  timeout 120 tools/bin/gather_micro > gpurun_out/gather_micro.txt 2>&1
fi
if [[ "$*" == *hostfused* ]]; then
  QAGNN_FUSED_HOP=1 timeout 300 python -m cProfile -s tottime bench.py --steps 60 --warmup 5 --questions 2 --no-cpu-baseline 2>&1 | head -n 90 > gpurun_out/hostprof_b10_fused.txt
fi
if [[ "$*" == *ablate* ]]; then   # prebuilt here by tools/build_micro.sh (compiling on the box would burn GPU minutes)
  run() { echo "== $*" >> gpurun_out/gemm_micro.txt; env "${@:2}" timeout 120 tools/bin/gemm_ablate_$1 >> gpurun_out/gemm_micro.txt 2>&1; }
  run BASE
  run BASE
  run BASE QAGNN_NN_PERSIST=0 QAGNN_TN_STRIP=0 QAGNN_TN_CHUNK=256
  for v in NOGLOAD NOMMA NOEPI; do run $v; done
fi
for f in gpurun_out/*.log; do echo "== $f"; tail -n 6 "$f"; done
cat gpurun_out/summary.txt
du -sh gpurun_out
