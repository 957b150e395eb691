cd $GRAFT_REPO_ROOT
for v in 0 1 2 4 16 32 64 33 35 39 96 103; do echo "== abl $v"; timeout 120 tools/bin/nn2_r6_abl$v | grep -v "six MFMAs"; done > gpurun_out/r6_v17_nn2_ablation.txt 2>&1
cat gpurun_out/r6_v17_nn2_ablation.txt
