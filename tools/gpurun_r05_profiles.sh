#!/bin/bash
# round-5 evidence run (GPU box): rocprofv3 --kernel-trace --stats of the bench windows and of the whole default-options solve,
# phase counters, bench lines (summaries are copied to profiles/)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05prof; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --steps 20 --warmup 5 --no-cpu --no-time-to-tol --hbm-n 0 --no-config-legs > $O/bench_under_rocprof.json 2> $O/kt.err
python tools/prof_summary.py $O/kt $O/r05_kernel_stats_bench_n4000.md "Kernel stats, round 5: bench.py --steps 20 --warmup 5 (rank-63 headline with --settle 200, early iterations, packed operator)" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu --no-time-to-tol --hbm-n 0 --no-config-legs" > /dev/null
rocprofv3 --kernel-trace --stats -d $O/ktm -- python bench.py --workload mimo --no-cpu > $O/bench_mimo_under_rocprof.json 2> $O/ktm.err
python tools/prof_summary.py $O/ktm $O/r05_kernel_stats_bench_mimo.md "Kernel stats, round 5: bench.py --workload mimo (MIMO n=512 x 8 blocks, batched Lanczos steps: grid.z = block)" "rocprofv3 --kernel-trace --stats -- python bench.py --workload mimo --no-cpu" > /dev/null
T2T_ONLY=1 timeout 900 rocprofv3 --kernel-trace --stats -d $O/ktt -- python tools/gpurun_t2t_phases.py $O/t2t_under_rocprof.json > $O/ktt.log 2> $O/ktt.err
python tools/prof_summary.py $O/ktt $O/r05_kernel_stats_time_to_tol_default.md "Kernel stats, round 5 (final build): whole default-options solve of Max-Cut n=4000 to tol 1e-4" "T2T_ONLY=1 rocprofv3 --kernel-trace --stats -- python tools/gpurun_t2t_phases.py" > /dev/null
rm -rf $O/kt $O/ktm $O/ktt
python tools/gpurun_t2t_phases.py $O/r05_time_to_tol_phases.json > $O/t2t_phases.log 2>&1
python bench.py > $O/r05_bench_line_default.json 2> $O/bench_default.err
python bench.py --workload mimo > $O/r05_bench_line_mimo.json 2> $O/bench_mimo.err
python bench.py --workload sdplib > $O/r05_bench_line_sdplib.json 2> $O/bench_sdplib.err
python bench.py --workload randsdp > $O/r05_bench_line_randsdp.json 2> $O/bench_randsdp.err
ls -la $O
