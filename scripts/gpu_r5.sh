#!/bin/bash
# One GPU-box visit of round 5: legs are named on the command line.  Small logs land in gpurun_out/ (< 64 MiB).
#   tests      the driver's `pytest tests -m gpu` (no -x: everything is reported), with the parity report file
#   smoke      __graft_entry__.smoke()
#   bench      the driver's `python bench.py` (defaults) -> gpurun_out/bench.json
#   benchq     bench.py without configs / cpu baseline / pmc (quick headline + breakdown)
#   prof       rocprofv3 --kernel-trace --stats of a short bench run -> gpurun_out/prof/
#   ab:VAR     interleaved A/B of an environment switch (0 1 0 1), quick bench
#   b10        quick bench at 2 questions (the reference's mini-batch)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO="$PWD"
mkdir -p gpurun_out
export TMPDIR=/tmp
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" >> gpurun_out/summary.txt; }
: > gpurun_out/summary.txt
( rocm-smi --showproductname 2>/dev/null | head -4; nproc; free -g | head -2 ) > gpurun_out/gpu.txt
( time python __graft_entry__.py ) > gpurun_out/build.log 2>&1
stamp "build check done: $(tail -n 4 gpurun_out/build.log | head -n 1)"
for arg in "$@"; do
  case "$arg" in
    tests)
      rm -f gpurun_out/parity_report.txt
      QAGNN_PARITY_REPORT=$REPO/gpurun_out/parity_report.txt timeout 2400 python -m pytest tests -m gpu -q --tb=short -rf --timeout 1500 -p no:cacheprovider --durations=12 > /tmp/test_all.log 2>&1
      echo "tests exit $?" >> gpurun_out/summary.txt
      ( head -c 30000 /tmp/test_all.log; echo; echo "......"; tail -c 12000 /tmp/test_all.log ) > gpurun_out/test_all.log
      stamp tests ;;
    tests:*)
      sel="${arg#tests:}"
      QAGNN_PARITY_REPORT=$REPO/gpurun_out/parity_report.txt timeout 1800 python -m pytest tests -m gpu -q --tb=short -rf --timeout 1500 -p no:cacheprovider -k "$sel" > /tmp/test_sel.log 2>&1
      echo "tests[$sel] exit $?" >> gpurun_out/summary.txt
      ( head -c 20000 /tmp/test_sel.log; echo; echo "......"; tail -c 8000 /tmp/test_sel.log ) > "gpurun_out/test_sel.log"
      stamp "tests:$sel" ;;
    smoke)
      timeout 300 python __graft_entry__.py smoke 2>&1 | tail -n 20 > gpurun_out/smoke.log
      echo "smoke exit ${PIPESTATUS[0]}" >> gpurun_out/summary.txt; stamp smoke ;;
    bench)
      timeout 1200 python bench.py > /tmp/bench.out 2> gpurun_out/bench.err
      echo "bench exit $?" >> gpurun_out/summary.txt
      tail -n 1 /tmp/bench.out > gpurun_out/bench.json; tail -n 5 gpurun_out/bench.err > gpurun_out/bench.log; stamp bench ;;
    benchq)
      timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-configs 2> gpurun_out/benchq.err | tail -n 1 > gpurun_out/benchq.json
      echo "benchq exit ${PIPESTATUS[0]}" >> gpurun_out/summary.txt; stamp benchq ;;
    b10)
      timeout 300 python bench.py --steps 40 --warmup 8 --questions 2 --no-cpu-baseline --no-pmc --no-configs 2>&1 | tail -n 1 > gpurun_out/bench_b10.json; stamp b10 ;;
    prof)
      rm -rf /tmp/prof; mkdir -p /tmp/prof gpurun_out/prof
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r3 -- python "$REPO/bench.py" --steps 5 --warmup 2 --repeats 1 --graphs 0 --no-cpu-baseline --no-pmc --no-configs ) 2>&1 | tail -n 8 > gpurun_out/prof.log
      echo "prof exit $?" >> gpurun_out/summary.txt
      find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} gpurun_out/prof/kernel_stats.csv \;
      python scripts/trace_by_shape.py "$(find /tmp/prof -name '*kernel_trace.csv' | head -n 1)" > gpurun_out/prof/by_shape.txt 2>&1
      stamp prof ;;
    profno)   # the same with every side stream off: pure per-kernel times (nothing co-runs)
      rm -rf /tmp/profno; mkdir -p /tmp/profno gpurun_out/prof
      ( cd /tmp && QAGNN_WGRAD_OVERLAP=0 QAGNN_PREP_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profno -o r3 -- python "$REPO/bench.py" --steps 5 --warmup 2 --repeats 1 --graphs 0 --no-cpu-baseline --no-pmc --no-configs ) > gpurun_out/profno.log 2>&1
      python scripts/trace_by_shape.py "$(find /tmp/profno -name '*kernel_trace.csv' | head -n 1)" > gpurun_out/prof/by_shape_no_overlap.txt 2>&1
      tail -n 4 gpurun_out/profno.log > /tmp/x && mv /tmp/x gpurun_out/profno.log
      stamp profno ;;
    prof10)
      rm -rf /tmp/prof10; mkdir -p /tmp/prof10 gpurun_out/prof
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof10 -o r3 -- python "$REPO/bench.py" --steps 10 --warmup 3 --repeats 1 --graphs 0 --questions 2 --no-cpu-baseline --no-pmc --no-configs ) 2>&1 | tail -n 8 > gpurun_out/prof10.log
      python scripts/trace_by_shape.py "$(find /tmp/prof10 -name '*kernel_trace.csv' | head -n 1)" > gpurun_out/prof/by_shape_b10.txt 2>&1
      stamp prof10 ;;
    ab:*)
      var="${arg#ab:}"
      for v in 0 1 0 1; do
        echo "$var=$v" >> gpurun_out/ab_$var.txt
        env $var=$v timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-pmc --no-configs 2>/dev/null | tail -n 1 > /tmp/ab_line.txt
        python - <<PY >> gpurun_out/ab_$var.txt
import json
d = json.load(open('/tmp/ab_line.txt'))
print(d['value'], d['ms_per_step'], d['repeat_ms_per_step'], d['breakdown_ms_per_step'])
PY
      done; stamp "ab:$var" ;;
    abv:*)   # abv:VAR:a:b -- interleaved A/B of two VALUES of an environment switch
      IFS=: read -r _ var va vb <<< "$arg"
      for v in $va $vb $va $vb; do
        echo "$var=$v" >> gpurun_out/ab_$var.txt
        env $var=$v timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-pmc --no-configs 2>/dev/null | tail -n 1 > /tmp/ab_line.txt
        python - <<PY >> gpurun_out/ab_$var.txt
import json
d = json.load(open('/tmp/ab_line.txt'))
print(d['value'], d['ms_per_step'], d['repeat_ms_per_step'], d['breakdown_ms_per_step'])
PY
      done; stamp "abv:$var" ;;
    ab10:*)
      var="${arg#ab10:}"
      for v in 0 1 0 1; do
        echo "$var=$v (2 questions)" >> gpurun_out/ab10_$var.txt
        env $var=$v timeout 300 python bench.py --steps 40 --warmup 8 --questions 2 --no-cpu-baseline --no-pmc --no-configs 2>/dev/null | tail -n 1 | cut -c1-200 >> gpurun_out/ab10_$var.txt
      done; stamp "ab10:$var" ;;
    ablib:*)   # ablib:<name>  -- the shipped library vs tools/bin/libqagnn_hip_<name>.so, whole step, interleaved
      name="${arg#ablib:}"
      for lib in "" "$name" "" "$name"; do
        echo "library: ${lib:-shipped}" >> gpurun_out/ablib_$name.txt
        QAGNN_LIB=${lib:+$REPO/tools/bin/libqagnn_hip_$lib.so} timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-pmc --no-configs 2>/dev/null | tail -n 1 > /tmp/ab_line.txt
        python - <<PY >> gpurun_out/ablib_$name.txt
import json
d = json.load(open('/tmp/ab_line.txt'))
print(d['value'], d['ms_per_step'], d['repeat_ms_per_step'], d['breakdown_ms_per_step'], d['hip_graph'][-70:])
PY
      done; stamp "ablib:$name" ;;
    abq:*)   # abq:<questions>:<VAR>  -- interleaved A/B (0 1 0 1) of a switch at a given number of questions
      spec="${arg#abq:}"; q="${spec%%:*}"; var="${spec#*:}"
      for v in 0 1 0 1; do
        echo "$var=$v ($q questions)" >> gpurun_out/abq${q}_$var.txt
        env $var=$v timeout 300 python bench.py --steps 40 --warmup 8 --questions $q --no-cpu-baseline --no-pmc --no-configs 2>/dev/null | tail -n 1 | cut -c1-200 >> gpurun_out/abq${q}_$var.txt
      done; stamp "abq:$q:$var" ;;
    abx10:*)   # abx10:VAR=a,b  -- interleaved A/B of two values of a switch at 2 questions (hipGraph replay as bench.py chooses)
      spec="${arg#abx10:}"; var="${spec%%=*}"; vals="${spec#*=}"; va="${vals%%,*}"; vb="${vals#*,}"
      for v in $va $vb $va $vb; do
        echo "$var=$v (2 questions)" >> gpurun_out/abx10_$var.txt
        env $var=$v timeout 300 python bench.py --steps 40 --warmup 8 --questions 2 --no-cpu-baseline --no-pmc --no-configs 2>/dev/null | tail -n 1 | cut -c1-200 >> gpurun_out/abx10_$var.txt
      done; stamp "abx10:$var" ;;
    dp2)   # two ranks time-slicing the one GPU over gloo: exercises the N > 1 code paths of bench.py (comm events, strong scaling)
      QAGNN_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 5 --warmup 2 --repeats 2 2>&1 | tail -n 3 > gpurun_out/bench_dp2_weak.log
      QAGNN_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29514 bench.py --gpus 2 --steps 5 --warmup 2 --repeats 2 --global-batch 32 2>&1 | tail -n 3 > gpurun_out/bench_dp2_strong.log
      QAGNN_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29515 bench.py --gpus 2 --steps 5 --warmup 2 --repeats 2 --graphs 0 --comm-overlap 2>&1 | tail -n 3 > gpurun_out/bench_dp2_weak_comm_overlap.log
      stamp dp2 ;;
    pmc)   # fabric traffic of the edge kernels (FETCH_SIZE / WRITE_SIZE in separate passes) -> gpurun_out/pmc_edge_fwd.json
      for ctr in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_$ctr; mkdir -p /tmp/pmc_$ctr
        ( cd /tmp && timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$ctr -o p -- python "$REPO/bench.py" --steps 2 --warmup 1 --repeats 1 --graphs 0 --no-cpu-baseline --no-pmc --no-configs ) 2>&1 | tail -n 3 > gpurun_out/pmc_$ctr.log
        python scripts/pmc_by_kernel.py "$(find /tmp/pmc_$ctr -name '*counter_collection.csv' | head -n 1)" > gpurun_out/pmc_$ctr.txt 2>&1
      done
      python scripts/pmc_edge_traffic.py "$(find /tmp/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -n 1)" "$(find /tmp/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -n 1)" 64000 208 > gpurun_out/pmc_edge_fwd.json 2> gpurun_out/pmc_edge_traffic.err
      stamp pmc ;;
    census)   # kernel launches of one step by forward region (backward attributed through autograd sequence numbers)
      timeout 300 python tools/op_census.py 2>&1 | cut -c1-230 | grep -v "Warning\|warn" | head -n 700 > gpurun_out/op_census.txt; stamp census ;;
    hostprof)
      timeout 300 python -m cProfile -s tottime bench.py --steps 40 --warmup 5 --repeats 1 --questions 2 --no-cpu-baseline --no-pmc --no-configs 2>&1 | head -n 70 > gpurun_out/hostprof_b10.txt; stamp hostprof ;;
    edgepmc)   # what bounds the edge kernels: TA / TD busy, L1 and L2 hit rates, issue stalls (three passes; --pmc with --kernel-trace only)
      i=0; files=""
      for ctrs in "GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max TD_TD_BUSY_sum TCC_HIT_sum TCC_MISS_sum" \
                  "GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum" \
                  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS"; do
        i=$((i+1)); rm -rf /tmp/ep$i; mkdir -p /tmp/ep$i
        ( cd /tmp && timeout 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/ep$i -o p -- python "$REPO/bench.py" --steps 2 --warmup 1 --repeats 1 --graphs 0 --no-cpu-baseline --no-pmc --no-configs ) > gpurun_out/edgepmc$i.log 2>&1
        tail -n 3 gpurun_out/edgepmc$i.log > /tmp/x && mv /tmp/x gpurun_out/edgepmc$i.log
        files="$files $(find /tmp/ep$i -name '*counter_collection.csv' | head -n 1)"
      done
      python scripts/pmc_edge_counters.py $files > gpurun_out/edge_counters.txt 2>&1
      stamp edgepmc ;;
    profreplay)   # kernel trace of the REPLAYED step (one hipGraph launch per step): busy time and gaps of the step as bench.py times it
      rm -rf /tmp/profrp; mkdir -p /tmp/profrp gpurun_out/prof
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profrp -o r5 -- python "$REPO/bench.py" --steps 10 --warmup 3 --repeats 1 --graphs 1 --no-cpu-baseline --no-pmc --no-configs ) > gpurun_out/profreplay.log 2>&1
      python scripts/trace_by_shape.py "$(find /tmp/profrp -name '*kernel_trace.csv' | head -n 1)" > gpurun_out/prof/by_shape_replay.txt 2>&1
      tail -n 4 gpurun_out/profreplay.log > /tmp/x && mv /tmp/x gpurun_out/profreplay.log
      stamp profreplay ;;
    cmd:*)
      c="${arg#cmd:}"
      bash -c "$c" > gpurun_out/cmd.log 2>&1; echo "cmd exit $?" >> gpurun_out/summary.txt; stamp "cmd" ;;
  esac
done
for f in gpurun_out/*.log; do echo "== $f"; tail -n 5 "$f"; done
cat gpurun_out/summary.txt
du -sh gpurun_out
