cd $GRAFT_REPO_ROOT
REPO=$PWD
mkdir -p gpurun_out; rm -rf /tmp/prof /tmp/lkt; mkdir -p /tmp/prof
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r6 -- python "$REPO/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-configs ) > gpurun_out/r6_v22_prof.log 2>&1
tail -2 gpurun_out/r6_v22_prof.log | cut -c1-400
find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} gpurun_out/r6_v22_kernel_stats.csv \;
head -25 gpurun_out/r6_v22_kernel_stats.csv | cut -c1-200
bash scripts/r6_prof.sh r6_v22_final 2 | tail -75
python scripts/sum_by_shape.py gpurun_out/r6_v22_final_by_shape.txt 15
