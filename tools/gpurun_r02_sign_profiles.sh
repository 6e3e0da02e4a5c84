#!/bin/bash
# round-2 evidence for the sign-function projection: kernel stats of the sdplib bench + PMC passes (MFMA utilisation)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02sign; rm -rf $O; mkdir -p $O
trap 'rm -rf $O/kt $O/pmc_mfma $O/pmc_mfma2 $O/pmc_fetch' EXIT
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --workload sdplib --steps 100 --warmup 5 --no-rocsolver-leg > $O/bench_sdplib_under_rocprof.json 2> $O/kt.err
python tools/prof_summary.py $O/kt $O/kernel_stats_sdplib.md "Kernel stats, round 2: bench.py --workload sdplib (maxG51, gpp500-1; sign-function projection)" "rocprofv3 --kernel-trace --stats -- python bench.py --workload sdplib --steps 100 --warmup 5 --no-rocsolver-leg" > /dev/null
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_mfma -- python tools/gpurun_sign_pmc.py 501 1000 2000 > $O/pmc_mfma.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA --kernel-trace -d $O/pmc_mfma2 -- python tools/gpurun_sign_pmc.py 501 1000 2000 > $O/pmc_mfma2.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -- python tools/gpurun_sign_pmc.py 501 1000 2000 > $O/pmc_fetch.log 2>&1
python tools/pmc_to_json.py $O/pmc_raw.json mfma=$O/pmc_mfma mfma2=$O/pmc_mfma2 fetch=$O/pmc_fetch > $O/pmc_to_json.log 2>&1
python tools/prof_summary.py $O/pmc_mfma $O/pmc_kernel_times.md "isolated sign-function projections n = 501, 1000, 2000 (PMC pass timing)" "tools/gpurun_sign_pmc.py" > /dev/null
rm -rf $O/kt $O/pmc_mfma $O/pmc_mfma2 $O/pmc_fetch
ls -la $O
