#!/bin/bash
# round 5 (final build): HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, kernel-trace only) of the Lanczos step
# kernels at the metric's n = 4000, target rank 63 -- the same command as round 4's (tools/gpurun_r04_pmc.sh), to confirm that the
# figure bench.py quotes from profiles/pmc_traffic.json still describes the kernels that ship
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05pmc; rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --pmc $c --kernel-trace -d $O/n4000_$c -- python bench.py --steps 5 --warmup 2 --settle 20 --no-cpu --no-time-to-tol --no-packed-leg --no-early-leg --no-config-legs > $O/n4000_$c.log 2>&1
  echo "n4000 $c rc=$?" >> $O/rc.txt
  python tools/pmc_query.py $O/n4000_$c > $O/n4000_$c.txt 2>&1
done
rm -rf $O/*_FETCH_SIZE $O/*_WRITE_SIZE
cat $O/rc.txt; head -12 $O/n4000_FETCH_SIZE.txt; head -12 $O/n4000_WRITE_SIZE.txt
