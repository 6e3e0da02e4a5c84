"""Where does the library's n = 2000 default-options solve part from the oracle's (objective 3.1e-7 apart after the same
7098 iterations)?  Compares the first 2200 iterations with an oracle trace (tools/_trace2000_oracle.json, generated in the
build container: oracle.solve(maxcut(2000, seed 0), max_iter = 2200, trace))."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_trace2000_oracle.json")))
G = np.array(g["rows"]); mv = np.array(g["matvecs"], float)
pr = P.maxcut(2000, seed=0)
s = Optimizer(max_iter=len(G)).optimize(pr, trace_capacity=len(G))
T = s.trace
sc = np.abs(G[:, 1:3]).max()
d = np.abs(T[:, 1:3] - G[:, 1:3]).max(axis=1) / sc          # (columns: iter, prim_obj, dual_obj, target_rank)
samemv = T[:, 13] == mv
print("iterations", len(G), "first mat-vec mismatch at", (int(np.argmin(samemv)) + 1) if not samemv.all() else None)
for thr in (1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-7):
    idx = np.nonzero(d > thr)[0]
    print("objective difference first exceeds %.0e at iteration" % thr, (int(idx[0]) + 1) if len(idx) else None)
k = int(np.argmax(d))
print("max diff %.3e at iteration %d; matvecs there oracle %d gpu %d; target rank %d" % (d.max(), k + 1, mv[k], T[k, 13], G[k, 3]))
for it in (100, 445, 446, 500, 718, 1000, 1305, 1500, 2018, 2200):
    if it <= len(G):
        print(it, "diff %.2e" % d[it - 1], "mv", int(mv[it - 1]), int(T[it - 1, 13]), "rank", int(G[it - 1, 3]))
