#!/bin/bash
# round 6, first evidence run: sensorloc kernel stats after the long-column fix, the reference's iteration counts, the GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -- python tools/r06/run_one.py sensorloc400 400 lanczos_cycle_kernel=0 > $O/sl400_after.log 2> $O/err.log
python tools/prof_summary.py $O/kt $O/r06_kernel_stats_sensorloc400.md "Kernel stats, round 6 (after the long-column fix): sensorloc n=400, 400 iterations, step kernels" "rocprofv3 --kernel-trace --stats -- python tools/r06/run_one.py sensorloc400 400 lanczos_cycle_kernel=0" > /dev/null
rm -rf $O/kt
cat $O/sl400_after.log; head -12 $O/r06_kernel_stats_sensorloc400.md
for n in 100 200 300 400; do python tools/r06/run_one.py sensorloc$n 20000; done 2>&1 | tee $O/sensorloc_after.log
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest_gpu.log
