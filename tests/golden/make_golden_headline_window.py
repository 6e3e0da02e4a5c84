"""The headline window of bench.py against the oracle: Max-Cut n = 4000 (seed 0) pinned at target rank 63 (krylovdim 127), the CPU
oracle resumed from the LIBRARY's state after iteration 250 (tools/gen/gpurun_capture_headline_window.py ->
state_maxcut_n4000_rank63_k250.npz) for 20 iterations -> trace_maxcut_n4000_rank63_window.json (~2 min of CPU)."""
import json, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from proxsdp_jl_amd import problems
from helpers import expand_state, load_compact_state
st = expand_state(load_compact_state(os.path.join(HERE, "state_maxcut_n4000_rank63_k250.npz")))
k0 = int(st["iteration"])
o = oracle.Options()
o.initial_target_rank, o.max_target_rank_krylov_eigs, o.max_iter = 63, 64, k0 + 20
mv = []
t0 = time.time()
res = oracle.solve(problems.maxcut(4000, seed=0), o, trace=True, resume=st,
                   proj_callback=lambda it, xin, xout, p, arc: mv.append(int(arc[0].matvecs)))
rows, prev = [], 0
for t, m in zip(res.trace, mv):
    rows.append(dict(iter=t["iter"], prim_obj=t["prim_obj"], dual_obj=t["dual_obj"], gap=t["gap"], feas=t["feas"], prim_res=t["prim_res"],
                     dual_res=t["dual_res"], primal_step=t["primal_step"], beta=t["beta"], theta=t["theta"], target_rank=t["target_rank"][0],
                     current_rank=t["current_rank"][0], min_eig=t["min_eig"][0], trials=t["trials"], matvecs=m - prev))
    prev = m
json.dump(dict(resumed_from=k0, rows=rows, wall_s=time.time() - t0, options=dict(initial_target_rank=63, max_target_rank_krylov_eigs=64)),
          open(os.path.join(HERE, "trace_maxcut_n4000_rank63_window.json"), "w"))
print(len(rows), "iterations in %.1f s" % (time.time() - t0), "mat-vecs", [r["matvecs"] for r in rows])
