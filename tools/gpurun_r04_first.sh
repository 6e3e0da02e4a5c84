#!/bin/bash
# round-4 first evidence run: step timeline, whole-solve phases, whole-solve kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04a; rm -rf $O; mkdir -p $O
python tools/timeline/run_timeline.py --md $O/r04_step_timeline.md > $O/timeline.log 2>&1
cp gpurun_out/timeline.json gpurun_out/timeline_raw.npy $O/ 2>/dev/null
python tools/gpurun_t2t_phases.py $O/t2t_phases.json > $O/t2t_phases.log 2>&1
T2T_ONLY=1 timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt -- python tools/gpurun_t2t_phases.py $O/t2t_under_rocprof.json > $O/kt.log 2> $O/kt.err
python tools/prof_summary.py $O/kt $O/r04_kernel_stats_time_to_tol_default.md "Kernel stats, round 4: whole default-options solve of Max-Cut n=4000 to tol 1e-4" "T2T_ONLY=1 rocprofv3 --kernel-trace --stats -- python tools/gpurun_t2t_phases.py" > /dev/null 2>> $O/kt.err
rm -rf $O/kt
ls -la $O
tail -5 $O/timeline.log
