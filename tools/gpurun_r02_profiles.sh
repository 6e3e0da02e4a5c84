#!/bin/bash
# round-2 evidence run (GPU box): kernel-trace stats of the bench windows + PMC passes on isolated kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02prof; rm -rf $O; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|FETCH_SIZE|WRITE_SIZE|GRBM_GUI_ACTIVE|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES" | head -40 > $O/counters_available.txt
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --no-cpu --no-time-to-tol --hbm-n 0 > $O/bench_under_rocprof.json 2> $O/kt.err
python tools/prof_summary.py $O/kt profiles_r02_kernel_stats.md "Kernel stats, round 2: bench.py windows (rank-63 headline, early iterations, packed operator)" "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu --no-time-to-tol --hbm-n 0" > /dev/null
mv profiles_r02_kernel_stats.md $O/
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -- python tools/gpurun_pmc5.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -- python tools/gpurun_pmc5.py > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_mfma -- python tools/gpurun_pmc5.py > $O/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA --kernel-trace -d $O/pmc_mfma2 -- python tools/gpurun_pmc5.py > $O/pmc_mfma2.log 2>&1
python tools/pmc_to_json.py $O/pmc_raw.json fetch=$O/pmc_fetch write=$O/pmc_write mfma=$O/pmc_mfma mfma2=$O/pmc_mfma2 > $O/pmc_to_json.log 2>&1
python tools/prof_summary.py $O/pmc_fetch $O/pmc_kernel_times.md "isolated kernels (PMC pass timing)" "tools/gpurun_pmc5.py" > /dev/null
rm -rf $O/kt $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_mfma2     # raw databases stay on the box
ls -la $O
