cd $GRAFT_REPO_ROOT
(echo "== 2 waves per SIMD"; timeout 120 tools/bin/nn2_r6_w2 | grep -v "six MFMAs"; echo "== narrow blocks at 3 waves per SIMD"; timeout 120 tools/bin/nn2_r6_w3 | grep -v "six MFMAs") > gpurun_out/r6_v19_nn2_w3.txt 2>&1
cat gpurun_out/r6_v19_nn2_w3.txt
