#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04d; rm -rf $O; mkdir -p $O
run() { tag=$1; shift; env "$@" T2T_ONLY=1 python tools/gpurun_t2t_phases.py $O/t2t_$tag.json > $O/t2t_$tag.log 2>&1; }
run base X=1
run k40 T2T_OPTS=full_eig_lanczos_kdim10=40
run k50 T2T_OPTS=full_eig_lanczos_kdim10=50
run keep25 PROXSDP_HIP_PP_KEEP=25
run keep40 PROXSDP_HIP_PP_KEEP=40
run k40keep25 PROXSDP_HIP_PP_KEEP=25 T2T_OPTS=full_eig_lanczos_kdim10=40
run k50keep25 PROXSDP_HIP_PP_KEEP=25 T2T_OPTS=full_eig_lanczos_kdim10=50
python - <<'PY'
import json
for t in ("base","k40","k50","keep25","keep40","k40keep25","k50keep25"):
    try:
        d=json.load(open(f"gpurun_out/r04d/t2t_{t}.json")); T=d["total"]
        print(t, d["status"], d["iterations"], round(d["time_s"],3), "obj %.9f"%d["objective"], "mv", int(T["lanczos_matvecs"]), "restarts", int(T["lanczos_restarts"]), "host_eig_s", round(T["host_eig_time"],2), "cert_failed", T.get("full_eigs_lanczos_cert_failed"))
    except Exception as e: print(t, "ERR", e)
PY
