#!/bin/bash
# round-3 evidence run (GPU box): rocprofv3 --kernel-trace --stats of the bench windows (summaries are copied to profiles/)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03prof; rm -rf $O; mkdir -p $O
# headline: the driver's command, side legs that only repeat round-2 evidence switched off
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --steps 20 --warmup 5 --no-cpu --no-time-to-tol --hbm-n 0 > $O/bench_under_rocprof.json 2> $O/kt.err
python tools/prof_summary.py $O/kt $O/r03_kernel_stats_bench_n4000.md "Kernel stats, round 3: bench.py --steps 20 --warmup 5 (rank-63 headline with --settle 200, early iterations, packed operator)" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu --no-time-to-tol --hbm-n 0" > /dev/null
# BASELINE config 4 on one GPU: batched multi-block Lanczos
rocprofv3 --kernel-trace --stats -d $O/ktm -- python bench.py --workload mimo --no-cpu > $O/bench_mimo_under_rocprof.json 2> $O/ktm.err
python tools/prof_summary.py $O/ktm $O/r03_kernel_stats_bench_mimo.md "Kernel stats, round 3: bench.py --workload mimo (MIMO n=512 x 8 blocks, batched Lanczos steps: grid.z = block)" "rocprofv3 --kernel-trace --stats -- python bench.py --workload mimo --no-cpu" > /dev/null
rm -rf $O/kt $O/ktm
ls -la $O
