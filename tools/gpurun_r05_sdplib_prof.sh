#!/bin/bash
# round 5, third session: BASELINE config 5 after the 48 x 48 product tiles -- rocprofv3 kernel stats of `bench.py --workload sdplib`,
# the bench line itself (with the CPU leg), and the same line with the 48-tiles switched off for the record
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05sdplib; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --workload sdplib --no-cpu > $O/bench_under_rocprof.json 2> $O/kt.err
python tools/prof_summary.py $O/kt $O/r05_kernel_stats_bench_sdplib.md "Kernel stats, round 5 (third session): bench.py --workload sdplib (maxG51 n = 1000 on 48 x 48 product tiles, gpp500-1 n = 501 on 32 x 32)" "rocprofv3 --kernel-trace --stats -- python bench.py --workload sdplib --no-cpu" > /dev/null
rm -rf $O/kt
python bench.py --workload sdplib > $O/r05_bench_line_sdplib.json 2> $O/bench_sdplib.err
PROXSDP_HIP_SIGN_TILE48=0 python bench.py --workload sdplib --no-cpu > $O/r05_bench_line_sdplib_tiles32.json 2> $O/bench_sdplib32.err
ls -la $O; head -30 $O/r05_kernel_stats_bench_sdplib.md
