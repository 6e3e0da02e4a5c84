"""Whole default-options solve of the metric instance (VERDICT r3 task 5): per-phase counters.

Phase 1 = Krylov branch (target_rank <= max_target_rank_krylov_eigs = 16), phase 2 = implicit full_eig! regime
(every projection is full_eig!, served by the Lanczos engine in positive-part mode).  The split comes from the
trace (target_rank column); phase-1 counters from a second solve stopped at the boundary (same trajectory:
deterministic), phase 2 = total - phase 1.  T2T_ONLY=1 runs just the one full solve (what rocprofv3 wraps).

    python tools/gpurun_t2t_phases.py [out.json]
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer

KEYS = ("lanczos_matvecs", "lanczos_restarts", "lanczos_calls", "full_eigs", "full_eigs_lanczos", "full_eigs_lanczos_checks",
        "full_eigs_lanczos_certified", "full_eigs_lanczos_cert_failed", "cert_matvecs", "full_eigs_sign", "sign_products", "host_eigs", "host_eig_merges", "host_eig_time", "host_eig_overlap_time",
        "linesearch_trials", "loop_time", "t_primal", "t_psd", "t_linesearch", "t_residual", "warm_starts")
n = int(os.environ.get("T2T_N", "4000"))
extra = {}
for kv in filter(None, os.environ.get("T2T_OPTS", "").split(",")):
    k, v = kv.split("=")
    extra[k] = float(v)
pr = P.maxcut(n, seed=0)
o = Optimizer(time_limit=200.0, **extra)
s = o.optimize(pr, trace_capacity=20000)
tr = np.asarray(s.trace)
tot = {k: float(s.stats[k]) for k in KEYS}
out = {"n": n, "status": o.termination_status(), "iterations": int(s.iter), "time_s": s.time, "objective": o.objective_value(),
       "options": extra, "total": tot}
print("total", json.dumps(out), flush=True)
if os.environ.get("T2T_ONLY") != "1":
    big = np.nonzero(tr[:, 10] > 16)[0]
    it1 = int(big[0]) if len(big) else len(tr)          # iterations [0, it1) ran with target_rank <= 16
    out["phase1_iterations"] = it1
    out["phase1_wall_s"] = float(tr[it1 - 1, 12]) if it1 > 0 else 0.0
    out["phase2_wall_s"] = float(tr[-1, 12]) - out["phase1_wall_s"]
    out["phase1_matvecs_trace"] = float(tr[:it1, 13].sum())
    out["phase2_matvecs_trace"] = float(tr[it1:, 13].sum())
    # rank schedule: iterations and wall time spent at each target rank
    sched = []
    for r in sorted(set(int(v) for v in tr[:, 10])):
        m = tr[:, 10] == r
        idx = np.nonzero(m)[0]
        t_in = float(tr[idx[-1], 12] - (tr[idx[0] - 1, 12] if idx[0] > 0 else 0.0))
        sched.append({"target_rank": r, "iterations": int(m.sum()), "wall_s": t_in, "matvecs_per_iteration": float(tr[m, 13].mean()),
                      "ms_per_iteration": 1e3 * t_in / int(m.sum())})
    out["rank_schedule"] = sched
    if 0 < it1 < len(tr):
        o1 = Optimizer(max_iter=it1, **extra)
        s1 = o1.optimize(pr)
        p1 = {k: float(s1.stats[k]) for k in KEYS}
        out["phase1"] = p1
        out["phase2"] = {k: tot[k] - p1[k] for k in KEYS}
        for ph, its in (("phase1", it1), ("phase2", len(tr) - it1)):
            d = out[ph]
            d["iterations"] = its
            d["matvecs_per_iteration"] = d["lanczos_matvecs"] / max(its, 1)
            d["restarts_per_iteration"] = d["lanczos_restarts"] / max(its, 1)
            d["host_syncs_per_iteration_lanczos"] = (d["lanczos_restarts"] + d["lanczos_calls"]) / max(its, 1)
            d["ms_per_iteration"] = 1e3 * d["loop_time"] / max(its, 1)
            d["us_per_matvec_all_in"] = 1e6 * d["t_psd"] / max(d["lanczos_matvecs"], 1)
dst = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/t2t_phases.json"
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
