"""what does the unpinned eigen layer leave open?  The committed golden instances under the orthogonaliser
variants KrylovKit 0.5 - 0.9 shipped as defaults (oracle/eig.py ORTH: mgs2 = the restatement, cgs2, mgsir, cgsir): per instance
the iteration count, the per-iteration mat-vec counts, restart totals, and the first trace departure from the mgs2 run.
CPU only (test infrastructure).  python tools/r06/orth_variants.py [quick|full] -> profiles/r06_orthogonaliser_variants.md"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from oracle import eig as oeig
from proxsdp_jl_amd import problems as P
from helpers import expand_state, load_compact_state
G = os.path.join(ROOT, "tests", "golden")
mode = sys.argv[1] if len(sys.argv) > 1 else "quick"

def run(pr, variant, iters, resume=None, **optkw):
    oeig.ORTH = variant
    o = oracle.Options()
    for k, v in optkw.items():
        setattr(o, k, v)
    if resume is not None:
        o.max_iter = int(resume["iteration"]) + iters
    elif iters:
        o.max_iter = iters
    mv, rs = [], []
    prev = [0, 0]
    def cb(it, xin, xout, p, arc):
        a = arc[0]
        mv.append(int(a.matvecs) - prev[0]); rs.append(int(a.restarts) - prev[1])
        prev[0], prev[1] = int(a.matvecs), int(a.restarts)
    t0 = time.time()
    r = oracle.solve(pr, o, trace=True, proj_callback=cb, resume=resume)
    oeig.ORTH = "mgs2"
    rows = np.array([[t["prim_obj"], t["dual_obj"], t["gap"], t["feas"], t["primal_step"], t["trials"], t["target_rank"][0]] for t in r.trace])
    return dict(status=int(r.status), iter=int(r.iter), objval=float(r.objval), mv=mv, rs=rs, rows=rows, wall=time.time() - t0)

cases = [("maxcut_readme_n4 (200 it)", lambda: P.maxcut_readme(), 200, None, {}),
         ("sdplib mcp124-1 (120 it)", lambda: P.sdplib(os.path.join(G, "sdplib", "mcp124-1.dat-s")), 120, None, {}),
         ("maxcut n=200 seed 0 (120 it)", lambda: P.maxcut(200, seed=0), 120, None, {}),
         ("sdplib mcp124-1 (to tol)", lambda: P.sdplib(os.path.join(G, "sdplib", "mcp124-1.dat-s")), 0, None, {}),
         ("sensorloc n=50 (to tol)", lambda: P.sensorloc(50, seed=0), 0, None, {}),
         ("maxcut n=150 seed 2 (to tol)", lambda: P.maxcut(150, seed=2), 0, None, {})]
if mode == "full":
    cases += [("maxcut n=1000 seed 0 (600 it)", lambda: P.maxcut(1000, seed=0), 600, None, {}),
              ("maxcut n=4000 headline window (rank 63, K 127; 12 it from the committed state)", lambda: P.maxcut(4000, seed=0), 12,
               "state_maxcut_n4000_rank63_k250.npz", dict(initial_target_rank=63, max_target_rank_krylov_eigs=64)),
              ("maxcut n=4000 default options (12 it from the state at 1000)", lambda: P.maxcut(4000, seed=0), 12, "state_maxcut_n4000_k1000.npz", {})]
variants = ["mgs2", "cgs2", "mgsir", "cgsir"]
out = ["# The unpinned eigen layer: orthogonaliser variants of KrylovKit 0.5 - 0.9 on the committed golden instances (round 6)", "",
       "`oracle/eig.py` restates KrylovKit's Lanczos with a modified Gram-Schmidt recurrence plus a second full pass (`mgs2`).  KrylovKit is not on",
       "disk and `Project.toml` admits versions whose `KrylovDefaults.orth` differ; `tools/r06/orth_variants.py` (CPU, test infrastructure) runs the",
       "oracle under each variant (`oeig.ORTH`): classical Gram-Schmidt with a second pass (`cgs2`) and both with iterative refinement -- the second",
       "pass only when the first lost more than a factor sqrt(2) of the norm (`mgsir`, `cgsir`).  Columns: PDHG iterations, total Lanczos mat-vecs,",
       "total restarts, iterations whose mat-vec count differs from the `mgs2` run, first iteration whose trace row (objectives, gap, feasibility,",
       "step, linesearch trials, target rank) differs from `mgs2` by more than 1e-9 of the column's largest magnitude, largest such difference over the common iterations.", "",
       "| instance | variant | status | iterations | mat-vecs | restarts | iterations with another mat-vec count | first trace departure > 1e-9 | max rel. trace difference | objective |",
       "|---|---|---|---|---|---|---|---|---|---|"]
res = {}
for name, mk, iters, st, kw in cases:
    pr = mk()
    base = None
    for v in variants:
        resume = expand_state(load_compact_state(os.path.join(G, st))) if st else None
        r = run(pr, v, iters, resume=resume, **kw)
        if base is None:
            base = r
        m = min(len(r["mv"]), len(base["mv"]))
        dmv = int(np.sum(np.array(r["mv"][:m]) != np.array(base["mv"][:m])))
        a, b = r["rows"][:m], base["rows"][:m]
        rel = np.abs(a - b) / (1e-300 + np.abs(b).max(axis=0))          # per column, against the column's largest magnitude
        bad = np.where(rel.max(axis=1) > 1e-9)[0]
        first = int(bad[0]) + 1 if len(bad) else "-"
        out.append(f"| {name} | {v} | {r['status']} | {r['iter']} | {sum(r['mv'])} | {sum(r['rs'])} | {dmv} | {first} | {rel.max():.1e} | {r['objval']:.10g} |")
        print(out[-1], "(%.0f s)" % r["wall"], flush=True)
        res[f"{name}|{v}"] = dict(iter=r["iter"], mv=sum(r["mv"]), rs=sum(r["rs"]), dmv=dmv, first=first, maxrel=float(rel.max()))
open(os.path.join(ROOT, "profiles", f"r06_orthogonaliser_variants_{mode}.md"), "w").write("\n".join(out) + "\n")
json.dump(res, open(os.path.join(ROOT, "profiles", f"r06_orthogonaliser_variants_{mode}.json"), "w"), indent=1)
