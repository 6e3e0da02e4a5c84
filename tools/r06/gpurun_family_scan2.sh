#!/bin/bash
# round 6: the multi-block SDPLIB families (block structure kept) under rocprofv3, top 4 kernels each
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06/scan2; rm -rf $O; mkdir -p $O
for inst in blocks:arch8 blocks:control8 blocks:truss8 blocks:qpG11 blocks:thetaG51 blocks:qap10 blocks:equalG11 blocks:maxG32; do
  f=$(echo $inst | tr ':' '_')
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -- python tools/r06/run_one.py $inst 300 > $O/$f.log 2> $O/$f.err
  python tools/prof_summary.py $O/kt $O/$f.md "scan $inst" "scan" > /dev/null 2>&1
  rm -rf $O/kt
  echo "== $inst: $(tail -1 $O/$f.log | cut -c1-200)"
  sed -n 8,11p $O/$f.md
done
