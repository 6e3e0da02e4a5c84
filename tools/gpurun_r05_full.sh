#!/bin/bash
# round 5: the whole GPU suite on the current build, then the driver-style bench line (driver flags: --steps 20 --warmup 5)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/gputests_b.log 2>&1; echo "suite rc $?" >> gpurun_out/gputests_b.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r05_driver_style.json 2> gpurun_out/bench_r05_driver_style.err; echo "bench rc $?" >> gpurun_out/bench_r05_driver_style.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/smoke.log
tail -6 gpurun_out/gputests_b.log; tail -3 gpurun_out/bench_r05_driver_style.err; tail -2 gpurun_out/smoke.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r05_driver_style.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"])
print(json.dumps(d["config"], indent=0)[:2500])
print("cpu_baseline", json.dumps({k: v for k, v in d.get("cpu_baseline", {}).items() if k != "parity_on_the_sample"})[:1500])
print("roofline frac", d["roofline"]["frac"], d["roofline"]["launch_arithmetic"])
PY
