cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 2>&1 | tail -32 > gpurun_out/r6_v21_tests.txt
tail -30 gpurun_out/r6_v21_tests.txt | cut -c1-200
