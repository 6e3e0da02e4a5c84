cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/tile48
for k in 0 1; do PROXSDP_HIP_SIGN_TILE48=$k python tools/gpurun_sign.py 700 800 1200 1500 2000 2500 > gpurun_out/tile48/sizes_$k.log 2>&1; cp gpurun_out/sign.json gpurun_out/tile48/sizes_$k.json; done
python - <<'PY'
import json
a=json.load(open("gpurun_out/tile48/sizes_0.json")); b=json.load(open("gpurun_out/tile48/sizes_1.json"))
for k in a:
    print(k, "t32 %.3f ms  t48 %.3f ms  products %d/%d err %.1e/%.1e" % (a[k]["sign_ms"], b[k]["sign_ms"], a[k]["products"], b[k]["products"], a[k]["err_sign"], b[k]["err_sign"]))
PY
