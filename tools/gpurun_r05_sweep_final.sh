#!/bin/bash
# round 5, third session: the SDPLIB sweep of tools/gpurun_r05_sweep.sh again on the final build (after the 48 x 48 product tiles)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python tools/sdplib_sweep.py --all --limit 20 --out gpurun_out/sdplib_sweep_final.md > gpurun_out/sweep_final.log 2>&1; echo "sweep rc $?" >> gpurun_out/sweep_final.log
tail -5 gpurun_out/sweep_final.log
