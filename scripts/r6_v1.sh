cd $GRAFT_REPO_ROOT
for b in nn2_ablate_0 nn2_x_abl128 tn_abl_0 tn_abl_32; do echo "== $b"; timeout 120 tools/bin/$b; done > gpurun_out/r6_v1_ablate.txt 2>&1
timeout 900 python bench.py > gpurun_out/r6_v1_bench.json 2> gpurun_out/r6_v1_bench.err
tail -c 1500 gpurun_out/r6_v1_ablate.txt
