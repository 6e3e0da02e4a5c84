#!/bin/bash
# round 5: final validation -- whole GPU suite (no -x: every failure shows), smoke, bench with no flags (default K / W)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/gputests_final.log 2>&1; echo "suite rc $?" >> gpurun_out/gputests_final.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_final.log 2>&1; echo "smoke rc $?" >> gpurun_out/smoke_final.log
( time timeout 1200 python bench.py > gpurun_out/bench_r05_noflags.json 2> gpurun_out/bench_r05_noflags.err ) 2> gpurun_out/bench_r05_noflags.time; echo "bench rc $?" >> gpurun_out/bench_r05_noflags.err
tail -5 gpurun_out/gputests_final.log; tail -2 gpurun_out/smoke_final.log; tail -3 gpurun_out/bench_r05_noflags.err; cat gpurun_out/bench_r05_noflags.time
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r05_noflags.json").read().strip().splitlines()[-1])
print("value", d["value"], "steps", d["steps"], "warmup", d["warmup"], "ms/step", d["ms_per_step"], "spread", d["config"]["window_spread_it_per_s"], "windows", d["config"]["timed_windows"])
print("t2t", d["config"].get("time_to_tol_s"), d["config"].get("time_to_tol_iterations"), "cpu steady", d["config"].get("cpu_steady_it_per_s"), d["config"].get("gpu_steady_it_per_s_same_iterations"))
for k in ("config_sdplib", "config_mimo_x8", "config_randsdp"):
    v = d.get(k, {})
    print(k, v.get("value"), (v.get("roofline") or {}).get("frac"), v.get("skipped"))
print("legs error:", d.get("config_legs_error"))
PY
