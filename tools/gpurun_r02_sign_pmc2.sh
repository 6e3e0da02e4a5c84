#!/bin/bash
# per-size kernel times and MFMA-busy counters of the sign-function projection (isolated projections)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02sign2; rm -rf $O; mkdir -p $O
trap 'rm -rf $O/kt_* $O/pmc_*_db' EXIT
for n in 501 1000 2000 4000; do
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/kt_$n -- python tools/gpurun_sign_pmc.py $n > $O/kt_$n.log 2>&1
  python tools/prof_summary.py $O/kt_$n $O/kernel_times_n$n.md "isolated sign-function projections (2 calls), n = $n" "rocprofv3 --kernel-trace --stats -- python tools/gpurun_sign_pmc.py $n" > /dev/null
done
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES; do
  timeout 150 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${c}_db -- python tools/gpurun_sign_pmc.py 1000 2000 > $O/pmc_$c.log 2>&1
  echo "$c rc=$?" >> $O/rc.txt
done
python tools/pmc_to_json.py $O/pmc_raw.json busy=$O/pmc_SQ_VALU_MFMA_BUSY_CYCLES_db sqbusy=$O/pmc_SQ_BUSY_CYCLES_db wave=$O/pmc_SQ_WAVE_CYCLES_db > $O/pmc_to_json.log 2>&1
ls -la $O; cat $O/rc.txt
