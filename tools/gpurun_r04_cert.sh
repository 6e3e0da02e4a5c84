#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04c; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "certified or verified or implicit_full_eig or sdplib_500 or served" > $O/pytest.log 2>&1; tail -15 $O/pytest.log




