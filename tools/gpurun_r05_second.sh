#!/bin/bash
# round 5, second GPU call: late-window tests of the metric instance, dense truncated projection, equilibration variants,
# flag-wait microbenchmark
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_state_seam.py -x -q -m gpu -s > gpurun_out/seam2.log 2>&1; echo "seam rc $?" >> gpurun_out/seam2.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "krylov_dimension_beyond or equilibration or certified" > gpurun_out/sel2.log 2>&1; echo "sel rc $?" >> gpurun_out/sel2.log
timeout 120 tools/micro/flagsync > gpurun_out/flagsync.log 2>&1; echo "flag rc $?" >> gpurun_out/flagsync.log
tail -15 gpurun_out/seam2.log; tail -8 gpurun_out/sel2.log; cat gpurun_out/flagsync.log
