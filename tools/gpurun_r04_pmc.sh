#!/bin/bash
# round 4: HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, kernel-trace only) of the Lanczos step kernels
# after the record / load changes: operator form at n = 2000 (rank 45) and -- one more attempt, time-boxed -- at the metric's n = 4000;
# batched multi-block launches of MIMO n = 512 x 8
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04pmc; rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d $O/mimo_$c -- python bench.py --workload mimo --no-cpu > $O/mimo_$c.log 2>&1
  echo "mimo $c rc=$?" >> $O/rc.txt
  python tools/pmc_query.py $O/mimo_$c > $O/mimo_$c.txt 2>&1
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d $O/n2000_$c -- python bench.py --n 2000 --steps 20 --warmup 5 --settle 60 --no-cpu --no-time-to-tol --no-packed-leg --no-early-leg --no-config-legs > $O/n2000_$c.log 2>&1
  echo "n2000 $c rc=$?" >> $O/rc.txt
  python tools/pmc_query.py $O/n2000_$c > $O/n2000_$c.txt 2>&1
  timeout -k 10 240 rocprofv3 --pmc $c --kernel-trace -d $O/n4000_$c -- python bench.py --steps 5 --warmup 2 --settle 20 --no-cpu --no-time-to-tol --no-packed-leg --no-early-leg --no-config-legs > $O/n4000_$c.log 2>&1
  echo "n4000 $c rc=$?" >> $O/rc.txt
  python tools/pmc_query.py $O/n4000_$c > $O/n4000_$c.txt 2>&1
done
rm -rf $O/*_FETCH_SIZE $O/*_WRITE_SIZE
cat $O/rc.txt; for f in mimo n2000 n4000; do head -8 $O/${f}_FETCH_SIZE.txt; head -8 $O/${f}_WRITE_SIZE.txt; done
