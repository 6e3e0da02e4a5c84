#!/bin/bash
# round 6, second evidence run: full GPU suite on the current build, bench.py driver-style, sensorloc after the chain double-buffer
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee $O/pytest_gpu2.log
python bench.py --steps 20 --warmup 5 > $O/r06_bench_line_n4000_driver_style.json 2> $O/bench.err; tail -c 1500 $O/r06_bench_line_n4000_driver_style.json; echo
for n in 300 400; do python tools/r06/run_one.py sensorloc$n 600; done
python tools/r06/run_one.py theta6 300
