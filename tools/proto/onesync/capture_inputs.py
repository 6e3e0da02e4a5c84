"""Capture projection inputs (dense symmetric X) of the metric instance for the one-sync Lanczos prototype:
 - headline window: rank-63 solve resumed from state_maxcut_n4000_rank63_k250.npz (K = 127, one cycle per projection)
 - Krylov phase of the default solve: resumed from state_maxcut_n4000_k1000.npz (rank 5, K = 25, thick restarts)
Writes /tmp/os/X_<tag>_<iter>.npy (n x n) + meta.  Test infrastructure only (uses oracle/)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from proxsdp_jl_amd import problems
from helpers import expand_state, load_compact_state, smat
G = os.path.join(ROOT, "tests", "golden")
OUT = "/tmp/os"

def run(tag, statefile, iters, **optkw):
    st = expand_state(load_compact_state(os.path.join(G, statefile)))
    k0 = int(st["iteration"])
    o = oracle.Options()
    for k, v in optkw.items():
        setattr(o, k, v)
    o.max_iter = k0 + iters
    n = 4000; N = n * (n + 1) // 2
    meta = []
    prev = [None]
    def cb(it, xin, xout, p, arc):
        a = arc[0]
        mv = a.matvecs if prev[0] is None else a.matvecs - prev[0][0]
        rs = a.restarts if prev[0] is None else a.restarts - prev[0][1]
        prev[0] = (a.matvecs, a.restarts)
        np.save(os.path.join(OUT, f"X_{tag}_{it}.npy"), smat(xin[:N], n))
        meta.append(dict(iter=int(it), target_rank=int(p.target_rank[0]), matvecs=int(mv), restarts=int(rs),
                         converged=int(a.converged_eigs), vals=[float(v) for v in a.vals]))
        print(tag, it, mv, rs, a.converged_eigs, flush=True)
    oracle.solve(problems.maxcut(4000, seed=0), o, resume=st, proj_callback=cb)
    json.dump(meta, open(os.path.join(OUT, f"meta_{tag}.json"), "w"))

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "head"):
        run("head", "state_maxcut_n4000_rank63_k250.npz", 3, initial_target_rank=63, max_target_rank_krylov_eigs=64)
    if which in ("all", "kry"):
        run("kry", "state_maxcut_n4000_k1000.npz", 3)
