#!/bin/bash
# round 5: SDPLIB sweep over all 65 files of the reference's test/data + the tests touched since the last full run
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "krylov_dimension_beyond or equilibration" > gpurun_out/sel3.log 2>&1; echo "sel rc $?" >> gpurun_out/sel3.log
timeout 2400 python tools/sdplib_sweep.py --all --limit 20 --out gpurun_out/sdplib_sweep.md > gpurun_out/sweep.log 2>&1; echo "sweep rc $?" >> gpurun_out/sweep.log
tail -4 gpurun_out/sel3.log; tail -5 gpurun_out/sweep.log
