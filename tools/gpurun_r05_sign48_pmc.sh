#!/bin/bash
# round 5, third session: the 48 x 48 product kernel at n = 1000 -- kernel times of isolated projections and the MFMA-busy / CU-busy /
# GPU-active cycle counters (separate --pmc passes, --kernel-trace only), as tools/gpurun_r02_sign_pmc2.sh did for the 32-tiles
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05sign48; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_moi_conic.py -m gpu -q -k Rotated > $O/rsoc.log 2>&1; tail -2 $O/rsoc.log
for k in 1 0; do
  PROXSDP_HIP_SIGN_TILE48=$k timeout 120 rocprofv3 --kernel-trace --stats -d $O/kt_$k -- python tools/gpurun_sign_pmc.py 1000 > $O/kt_$k.log 2>&1
  python tools/prof_summary.py $O/kt_$k $O/kernel_times_n1000_tile48_$k.md "isolated sign-function projection, n = 1000, PROXSDP_HIP_SIGN_TILE48=$k" "rocprofv3 --kernel-trace --stats -- python tools/gpurun_sign_pmc.py 1000" > /dev/null
  for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE; do
    PROXSDP_HIP_SIGN_TILE48=$k timeout 150 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${k}_$c -- python tools/gpurun_sign_pmc.py 1000 > $O/pmc_${k}_$c.log 2>&1
    echo "tile48=$k $c rc=$?" >> $O/rc.txt
    python tools/pmc_query.py $O/pmc_${k}_$c 2>&1 | grep "k_sym_gemm" > $O/pmc_${k}_$c.txt
  done
done
rm -rf $O/kt_* $O/pmc_*_SQ_VALU_MFMA_BUSY_CYCLES $O/pmc_*_SQ_BUSY_CU_CYCLES $O/pmc_*_GRBM_GUI_ACTIVE
cat $O/rc.txt; grep "k_sym_gemm" $O/kernel_times_n1000_tile48_*.md; cat $O/pmc_*.txt
