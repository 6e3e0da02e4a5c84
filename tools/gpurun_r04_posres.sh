cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04e; rm -rf $O; mkdir -p $O
for p in 1e-7 1e-6 1e-5; do T2T_ONLY=1 T2T_OPTS=full_eig_lanczos_posres=$p python tools/gpurun_t2t_phases.py $O/t2t_posres_$p.json > $O/l_$p.log 2>&1; done
python - <<'PY'
import json
for p in ("1e-7","1e-6","1e-5"):
    d=json.load(open(f"gpurun_out/r04e/t2t_posres_{p}.json")); T=d["total"]
    print(p, d["status"], d["iterations"], round(d["time_s"],3), "obj %.12f"%d["objective"], "mv", int(T["lanczos_matvecs"]), "restarts", int(T["lanczos_restarts"]), "cert_failed", T.get("full_eigs_lanczos_cert_failed"), "certified", T.get("full_eigs_lanczos_certified"))
PY
