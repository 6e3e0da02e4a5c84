#!/bin/bash
# round 3: HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, kernel-trace only) of the batched
# multi-block mat-vec launch (MIMO n=512 x 8) and of the operator-form Lanczos step kernels at n = 2000 (rank 45)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03pmc; rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/mimo_$c -- python bench.py --workload mimo --no-cpu > $O/mimo_$c.log 2>&1
  echo "mimo $c rc=$?" >> $O/rc.txt
  python tools/pmc_query.py $O/mimo_$c > $O/mimo_$c.txt 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/n2000_$c -- python bench.py --n 2000 --steps 20 --warmup 5 --settle 60 --no-cpu --no-time-to-tol --no-packed-leg --no-early-leg > $O/n2000_$c.log 2>&1
  echo "n2000 $c rc=$?" >> $O/rc.txt
  python tools/pmc_query.py $O/n2000_$c > $O/n2000_$c.txt 2>&1
done
rm -rf $O/mimo_FETCH_SIZE $O/mimo_WRITE_SIZE $O/n2000_FETCH_SIZE $O/n2000_WRITE_SIZE
cat $O/rc.txt; head -12 $O/mimo_FETCH_SIZE.txt; head -12 $O/mimo_WRITE_SIZE.txt; head -12 $O/n2000_FETCH_SIZE.txt; head -12 $O/n2000_WRITE_SIZE.txt
