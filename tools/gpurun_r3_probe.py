"""Round-3 probes (one gpurun call): (1) sign-engine engagement on mcp250-1 / mcp500-1 after the linesearch fidelity
change moved their trajectories; (2) where the implicit full_eig! regime at n = 4000 starts being served by the
Lanczos engine when entered at iteration 1; (3) the rank-63 window with the K x K eigensolve by split + rank-one
merge (host_eig_merge) against implicit QL."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathlib import Path
import numpy as np
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
g = Path(__file__).resolve().parent.parent / "tests" / "golden" / "sdplib"
which = set((os.environ.get("PROBE") or "sign,implicit,merge").split(","))
out = {}
if "sign" in which:
    for name in ("mcp250-1", "mcp500-1"):
        pr = P.sdplib(g / f"{name}.dat-s")
        for eng in (0, 1):
            o = Optimizer(tol_gap=1e-4, tol_feasibility=1e-4, psd_sign_engine=eng, time_limit=120.0)
            s = o.optimize(pr)
            out[f"{name}_engine{eng}"] = dict(status=s.status, iter=int(s.iter), obj=s.objval, time=s.time,
                                              served=int(s.stats["sign_engine_projections"]), checks=int(s.stats["sign_engine_checks"]),
                                              mismatches=int(s.stats["sign_engine_mismatches"]), rejected=int(s.stats["sign_engine_rejected"]),
                                              matvecs=int(s.stats["lanczos_matvecs"]))
            print(name, eng, out[f"{name}_engine{eng}"], flush=True)
if "implicit" in which:
    pr = P.maxcut(4000, seed=0)
    for it in (100, 300):
        o = Optimizer(max_iter=it, initial_target_rank=17, full_eig_lanczos=-1)
        s = o.optimize(pr)
        out[f"implicit_{it}"] = dict(iter=int(s.iter), full_eigs=int(s.stats["full_eigs"]), by_lanczos=int(s.stats["full_eigs_lanczos"]),
                                     by_sign=int(s.stats["full_eigs_sign"]), time=s.time, final_rank=int(s.final_rank))
        print("implicit", it, out[f"implicit_{it}"], flush=True)
if "merge" in which:
    pr = P.maxcut(4000, seed=0)
    for hm in (0, -1):
        o = Optimizer(max_iter=260, initial_target_rank=63, max_target_rank_krylov_eigs=64, host_eig_merge=hm)
        s = o.optimize(pr, trace_capacity=260)
        tr = s.trace
        t = float(tr[259, 12] - tr[199, 12])
        out[f"merge_{hm}"] = dict(it_per_s=60 / t, ms=1e3 * t / 60, host_eig_ms_per_it=1e3 * s.stats["host_eig_time"] / s.iter,
                                  overlapped_ms_per_it=1e3 * s.stats["host_eig_overlap_time"] / s.iter, merges=int(s.stats["host_eig_merges"]),
                                  host_eigs=int(s.stats["host_eigs"]), matvecs=int(s.stats["lanczos_matvecs"]), obj=s.objval,
                                  prim_obj_last=float(tr[259, 1]))
        print("merge", hm, out[f"merge_{hm}"], flush=True)
    a, b = out["merge_0"], out["merge_-1"]
    print("same matvecs:", a["matvecs"] == b["matvecs"], "obj rel diff %.2e" % (abs(a["prim_obj_last"] - b["prim_obj_last"]) / abs(a["prim_obj_last"])))
    for tk in (16, 64):
        for hm in ((0, -1) if tk == 16 else (0, -1, 1)):
            o = Optimizer(time_limit=200.0, max_target_rank_krylov_eigs=tk, host_eig_merge=hm)
            s = o.optimize(pr)
            out[f"t2t_k{tk}_merge{hm}"] = dict(status=s.status, iter=int(s.iter), obj=s.objval, time=s.time, host_eig_s=s.stats["host_eig_time"],
                                                overlapped_s=s.stats["host_eig_overlap_time"], merges=int(s.stats["host_eig_merges"]))
            print("t2t", tk, hm, out[f"t2t_k{tk}_merge{hm}"], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r3_probe.json", "w"), indent=1)
