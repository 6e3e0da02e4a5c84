#!/bin/bash
# round 5, first GPU call: state seam tests, n = 4000 state capture, then the whole GPU suite on the ABI-9 build
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_state_seam.py -x -q -m gpu > gpurun_out/seam.log 2>&1; echo "seam rc $?" >> gpurun_out/seam.log
timeout 900 python tools/gen/gpurun_capture_maxcut_n4000.py > gpurun_out/cap4000.log 2>&1; echo "cap rc $?" >> gpurun_out/cap4000.log
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/gputests.log 2>&1; echo "suite rc $?" >> gpurun_out/gputests.log
tail -5 gpurun_out/seam.log; tail -12 gpurun_out/cap4000.log; tail -5 gpurun_out/gputests.log
