cd $GRAFT_REPO_ROOT
timeout 300 python tools/op_census.py > gpurun_out/r6_v31_census_b320.txt 2>&1
grep -E "k_gemm|Cijk" gpurun_out/r6_v31_census_b320.txt | grep -v ", 2, true, 4>\|, 2>(" | cut -c1-150
echo; head -16 gpurun_out/r6_v31_census_b320.txt | tail -14
