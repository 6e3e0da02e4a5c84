"""Late windows of the metric instance (Max-Cut n = 4000, seed 0, default options) by the CPU oracle, resumed from
states the LIBRARY wrote on the GPU box (tools/gen/gpurun_capture_maxcut_n4000.py -> state_maxcut_n4000_*.npz here):

  window A  iterations 1001 .. 1060         the steady Krylov-phase window of SURVEY section 8d
  window C  iterations E+1 .. the oracle's stop   E = 31 iterations before the LIBRARY's solve stops (8651): the oracle continues
                                            with reference defaults until ITS stop rule fires (pdhg.jl:248-253)
  window B  iterations U+1 .. U+12+W        12 iterations at target rank 16, the rank update 16 -> 17 that leaves
                                            KrylovKit's range (options.jl:76), then W >= 30 iterations of the implicit
                                            full_eig! regime with LAPACK dsyevr in the loop
                                            (/root/reference/src/prox_operators.jl:46-59,111-126, src/pdhg.jl:267-283)

    python tests/golden/make_golden_late_n4000.py [W]        (~10 min of CPU for W = 36)
Writes tests/golden/trace_maxcut_n4000_late.json: per iteration the oracle's trace row, current_rank, min_eig and the
Lanczos mat-vec count, plus the wall time per iteration (the steady-window CPU baseline of bench.py)."""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                                     # noqa: E402
import oracle                                                          # noqa: E402
from proxsdp_jl_amd import problems                                    # noqa: E402
from helpers import expand_state, load_compact_state                  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 36
n = 4000
pr = problems.maxcut(n, seed=0)
out = {"instance": "maxcut n=4000 seed=0, default options", "windows": {}}
outp = os.path.join(HERE, "trace_maxcut_n4000_late.json")
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None          # e.g. "kEnd": (re)generate these windows, keep the others
if only and os.path.exists(outp):
    out = json.load(open(outp))
for tag, extra in (("k1000", 60), ("kU", 12 + W), ("kEnd", 0)):
    if only and tag not in only:
        continue
    sp_ = os.path.join(HERE, f"state_maxcut_n{n}_{tag}.npz")
    if not os.path.exists(sp_):
        continue
    st = expand_state(load_compact_state(sp_))
    k0 = int(st["iteration"])
    o = oracle.Options()
    if extra:
        o.max_iter = k0 + extra          # (kEnd: reference defaults -- the oracle runs to ITS stop)
    mv = []
    stamps = []

    def cb(it, xin, xout, p, arc_list, mv=mv, stamps=stamps):
        mv.append(int(arc_list[0].matvecs))
        stamps.append(time.time())

    t0 = time.time()
    res = oracle.solve(pr, o, trace=True, resume=st, proj_callback=cb)
    wall = time.time() - t0
    rows = []
    prev = 0
    for t, m in zip(res.trace, mv):
        rows.append(dict(iter=t["iter"], prim_obj=t["prim_obj"], dual_obj=t["dual_obj"], gap=t["gap"], feas=t["feas"],
                         prim_res=t["prim_res"], dual_res=t["dual_res"], primal_step=t["primal_step"], beta=t["beta"],
                         theta=t["theta"], target_rank=t["target_rank"][0], current_rank=t["current_rank"][0],
                         min_eig=t["min_eig"][0], trials=t["trials"], matvecs=m - prev))
        prev = m
    out["windows"][tag] = dict(resumed_from=k0, rows=rows, wall_s=wall, loop_s=res.stats["loop_time"],
                               full_eigs=int(res.stats["full_eigs"]), cores=os.cpu_count(),
                               per_iteration_s=[float(b - a) for a, b in zip(stamps[:-1], stamps[1:])],
                               final=dict(status=int(res.status), iterations=int(res.iter), objval=float(res.objval),
                                          gap=float(res.gap), final_rank=int(res.final_rank),
                                          primal_feasible=bool(res.primal_feasible_user_tol)))
    print(tag, "resumed from", k0, ":", len(rows), "iterations in %.1f s" % wall, "full_eigs", res.stats["full_eigs"],
          "target ranks", sorted(set(r["target_rank"] for r in rows)), flush=True)
json.dump(out, open(outp, "w"))
