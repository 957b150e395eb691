cd $GRAFT_REPO_ROOT
for v in 0 8 24 9 11; do echo "== abl $v"; timeout 120 tools/bin/nn2_r6_abl$v | grep "three MFMAs"; done | tee gpurun_out/r6_v36_nn2_ablation_barrier.txt
