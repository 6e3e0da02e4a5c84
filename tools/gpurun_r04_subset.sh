#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04f
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -k "implicit_full_eig or end_state or served or certified or verified or maxG51_default or beyond_side or n1000_objective" > gpurun_out/r04f/pytest.log 2>&1; tail -6 gpurun_out/r04f/pytest.log
