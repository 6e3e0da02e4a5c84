#!/bin/bash
# round-3 evidence for the sign-function projection with the shortened, tested schedule (options.sign_start_row):
# bench lines of BASELINE config 5, rocprofv3 kernel stats of the same command, per-size accuracy / time table
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03sign; rm -rf $O; mkdir -p $O
python bench.py --workload sdplib > $O/r03_bench_line_sdplib.json 2> $O/bench_sdplib.err
python bench.py --workload sdplib --no-cpu > $O/bench_sdplib_2.json 2>> $O/bench_sdplib.err
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --workload sdplib --no-cpu > $O/bench_sdplib_under_rocprof.json 2> $O/kt.err
python tools/prof_summary.py $O/kt $O/r03d_kernel_stats_bench_sdplib.md "Kernel stats, round 3: bench.py --workload sdplib (maxG51 n=1000 / gpp500-1 n=501, full_eig_decomp = true, sign-function projection with the shortened schedule)" "rocprofv3 --kernel-trace --stats -- python bench.py --workload sdplib --no-cpu" > /dev/null
rm -rf $O/kt
python tools/gpurun_sign.py 100 501 1000 2000 3000 4000 > $O/sign_sizes.log 2>&1
cp gpurun_out/sign.json $O/r03d_sign_projection_sizes.json
ls -la $O
