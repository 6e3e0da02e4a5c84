#!/bin/bash
# round 6 evidence on the final build: the reference's benchmark script, the one-workgroup kernel A/B, kernel stats of the headline
# command, bench lines of every workload
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06f; rm -rf $O; mkdir -p $O
python tools/gpurun_runbench.py $O/r06_runbench.md > $O/runbench.log 2>&1; tail -2 $O/runbench.log
PROXSDP_HIP_DEBUG_B1=1 python tools/r06/gpurun_block1.py > $O/block1.log 2>&1; cp gpurun_out/r06/block1.json $O/block1.json; grep -c bit_identical $O/block1.log
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --steps 20 --warmup 5 --no-cpu --no-time-to-tol --hbm-n 0 --no-config-legs > $O/bench_under_rocprof.json 2> $O/kt.err
python tools/prof_summary.py $O/kt $O/r06_kernel_stats_bench_n4000.md "Kernel stats, round 6: bench.py --steps 20 --warmup 5 (rank-63 headline with --settle 200, early iterations, packed operator)" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu --no-time-to-tol --hbm-n 0 --no-config-legs" > /dev/null
rm -rf $O/kt
python bench.py > $O/r06_bench_line_n4000.json 2> $O/bench_default.err
python bench.py --workload mimo > $O/r06_bench_line_mimo.json 2> $O/bench_mimo.err
python bench.py --workload sdplib > $O/r06_bench_line_sdplib.json 2> $O/bench_sdplib.err
python bench.py --workload randsdp > $O/r06_bench_line_randsdp.json 2> $O/bench_randsdp.err
ls -la $O
