#!/bin/bash
# round 6: scan one instance per SDPLIB family (plus the benchmark families) for a kernel that dominates -- the way k_spmvT_S_batch
# dominated sensor localisation: rocprofv3 --kernel-trace --stats, 300 iterations each, top 4 kernels per instance
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06/scan; rm -rf $O; mkdir -p $O
for inst in thetaG11 theta6 qap10 control8 truss8 arch8 maxG11 qpG11 gpp500-1 mcp500-1 sensorloc300 mimo500; do
  rocprofv3 --kernel-trace --stats -d $O/kt -- python tools/r06/run_one.py $inst 300 > $O/$inst.log 2> $O/$inst.err
  python tools/prof_summary.py $O/kt $O/$inst.md "scan $inst" "scan" > /dev/null 2>&1
  rm -rf $O/kt
  echo "== $inst: $(tail -1 $O/$inst.log)"
  sed -n 8,11p $O/$inst.md
done
