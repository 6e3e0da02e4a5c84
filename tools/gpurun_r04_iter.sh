#!/bin/bash
# round-4 iteration run: GPU suite (optional), step timeline, headline window
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04b; rm -rf $O; mkdir -p $O
if [ "$1" != "notests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
fi
python tools/timeline/run_timeline.py --md $O/r04_step_timeline.md > $O/timeline.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu --no-time-to-tol --no-packed-leg --no-early-leg --hbm-n 0 --no-config-legs > $O/bench_headline.json 2> $O/bench.err
tail -3 $O/pytest.log; cat $O/r04_step_timeline.md | head -40; python -c "
import json;d=json.load(open('$O/bench_headline.json'));print('headline', d['value'], d['ms_per_step'], d.get('cold_start_window'), d['roofline']['launch_arithmetic'])"
