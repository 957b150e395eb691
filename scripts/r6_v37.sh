cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "launch_timing or absmax" 2>&1 | tail -5
