#!/bin/bash
# 48 x 48 product tiles of the sign-function projection (k_sym_gemm48) against the 32 x 32 tiles: accuracy / time per projection at
# n = 1000 / 1001 / 993 (tools/gpurun_sign.py) and BASELINE config 5 (bench.py --workload sdplib), both settings in ONE session
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/tile48
for k in 0 1; do
  PROXSDP_HIP_SIGN_TILE48=$k python tools/gpurun_sign.py 1000 1001 993 > gpurun_out/tile48/sign_$k.log 2>&1
  cp gpurun_out/sign.json gpurun_out/tile48/sign_$k.json
  PROXSDP_HIP_SIGN_TILE48=$k python bench.py --workload sdplib --no-cpu > gpurun_out/tile48/bench_$k.json 2> gpurun_out/tile48/bench_$k.err
done
python bench.py --workload sdplib --no-cpu > gpurun_out/tile48/bench_auto.json 2> gpurun_out/tile48/bench_auto.err
python - <<'PY'
import json
for k in (0, 1):
    d = json.load(open("gpurun_out/tile48/sign_%d.json" % k))
    for key, v in d.items():
        print(k, key, "ms %.3f err %.2e rank %d/%d products %d" % (v["sign_ms"], v["err_sign"], v["rank_sign"], v["npos"], v["products"]))
for k in ("0", "1", "auto"):
    try:
        line = [l for l in open("gpurun_out/tile48/bench_%s.json" % k) if l.startswith("{")][-1]
        d = json.loads(line)
        print(k, d["value"], json.dumps(d.get("config", {}))[:600])
    except Exception as e:
        print(k, "bench failed", e, open("gpurun_out/tile48/bench_%s.err" % k).read()[-800:])
PY
