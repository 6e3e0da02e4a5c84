#!/bin/bash
# round 5: kEnd state capture, the tests behind the one that stopped the last suite run, then the rocprofv3 evidence run
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python tools/gen/gpurun_capture_maxcut_n4000.py > gpurun_out/cap4000b.log 2>&1; echo "cap rc $?" >> gpurun_out/cap4000b.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "krylov_dimension or multi_block_families" > gpurun_out/sel4.log 2>&1; echo "sel rc $?" >> gpurun_out/sel4.log
timeout 1200 python -m pytest tests/test_host_abi.py tests/test_julia_convention.py tests/test_sharded_gpu.py tests/test_state_seam.py -q -m gpu -s > gpurun_out/rest4.log 2>&1; echo "rest rc $?" >> gpurun_out/rest4.log
tail -8 gpurun_out/cap4000b.log; tail -4 gpurun_out/sel4.log; grep -E "passed|failed|max relative|mat-vecs|worst" gpurun_out/rest4.log | tail -12
bash tools/gpurun_r05_profiles.sh > gpurun_out/prof5.log 2>&1; echo "prof rc $?" >> gpurun_out/prof5.log; tail -3 gpurun_out/prof5.log
