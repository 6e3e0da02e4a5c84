"""SDPLIB through the library at tol 1e-4 against the literature optima.

    sdplib_sweep.py [--engine] [name ...]      the Max-Cut family of rounds 1-4 (tests/golden/sdplib, rank-64 knob)
    sdplib_sweep.py --all [--limit S] [--out F] every *.dat-s of the reference's test/data (65 files; copied as DATA into the
                                               git-ignored tests/golden/sdplib_all/ so that they travel to the GPU box), reference
                                               DEFAULT options, time limit S (default 30 s) each, markdown table to F

Two models per file (proxsdp.jl_amd/problems.py): "harness" = the reference's own reader (test/base_sdplib.jl:1-45: all blocks
merged into ONE PSD variable whose side is `length(c)` -- it only loads where the file has as many constraints as matrix rows:
the mcp / maxG / gpp / qpG families its benchmark runs, test/runbench.jl:120-155) and "blocks" = the file's block structure kept
(what a JuMP user writes: one PSD cone per block, diagonal blocks as nonnegative scalars).  --all runs "harness" where the
reference's reader loads the file and "blocks" for every file.
--engine: psd_sign_engine = 1 (sign-function projection where it is the cheaper way to the same projection)"""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathlib import Path
import numpy as np
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
# SDPLIB README optima (|value|; the sign convention of the harness differs per family)
LIT = {"arch0": 0.5665, "arch2": 0.6715, "arch4": 0.9726, "arch8": 7.057, "control1": 17.78463, "control2": 8.300, "control3": 13.63327,
       "control4": 19.79423, "control5": 16.88360, "control6": 37.30440, "control7": 20.62510, "control8": 20.28640,
       "gpp100": 44.9435, "gpp124-1": 7.3431, "gpp124-2": 46.8623, "gpp124-3": 153.014, "gpp124-4": 418.988, "gpp250-1": 15.4449,
       "gpp250-2": 81.869, "gpp250-3": 303.539, "gpp250-4": 747.328, "gpp500-1": 25.3205, "gpp500-2": 156.06, "gpp500-3": 513.02,
       "gpp500-4": 1567.02, "maxG11": 629.1648, "maxG32": 1567.640, "maxG51": 4003.809, "maxG60": 15222.27,
       "mcp100": 226.1574, "mcp124-1": 141.9905, "mcp124-2": 269.8802, "mcp124-3": 467.7501, "mcp124-4": 864.4119,
       "mcp250-1": 317.2643, "mcp250-2": 531.9301, "mcp250-3": 981.1726, "mcp250-4": 1681.960, "mcp500-1": 598.1485,
       "mcp500-2": 1070.057, "mcp500-3": 1847.970, "mcp500-4": 3566.738, "qap5": 436.0, "qap6": 381.44, "qap7": 424.82,
       "qap8": 756.96, "qap9": 1409.94, "qap10": 1092.6, "qpG11": 2448.659, "qpG51": 11818.00, "theta1": 23.0, "theta2": 32.87917,
       "theta3": 42.16698, "theta4": 50.32122, "theta5": 57.23231, "theta6": 63.47709, "thetaG11": 400.0, "thetaG51": 349.0,
       "truss1": 8.999996, "truss2": 123.3804, "truss3": 9.109996, "truss4": 9.009996, "truss5": 132.6357, "truss6": 901.4,
       "truss7": 900.0014, "truss8": 133.1146}
ROOT = Path(__file__).resolve().parent.parent
g = ROOT / "tests" / "golden" / "sdplib"
ENGINE = "--engine" in sys.argv
ALL = "--all" in sys.argv


def argval(flag, default):
    return sys.argv[sys.argv.index(flag) + 1] if flag in sys.argv else default


def header_of(path):
    with open(path) as f:
        lines = [ln.strip() for ln in f if ln.strip() and ln.strip()[0] not in '"*']
    m = int(lines[0].split()[0])
    s = lines[2]
    for ch in "{}(),":
        s = s.replace(ch, " ")
    blks = [int(float(v)) for v in s.split()]
    return m, blks


if ALL:
    limit = float(argval("--limit", "30"))
    outp = Path(argval("--out", str(ROOT / "gpurun_out" / "sdplib_sweep.md")))
    d = ROOT / "tests" / "golden" / "sdplib_all"
    files = sorted(d.glob("*.dat-s"), key=lambda p: p.stat().st_size)
    only = [a for a in sys.argv[1:] if not a.startswith("--") and a not in (argval("--limit", ""), argval("--out", ""))]
    rows = []
    t_all = time.time()
    for f in files:
        name = f.name[:-6]
        if only and name not in only:
            continue
        m, blks = header_of(f)
        side = sum(abs(b) for b in blks)
        models = (["harness"] if m == side else []) + (["blocks"] if (len(blks) > 1 or m != side) else [])
        for model in models:
            try:
                pr = P.sdplib(f) if model == "harness" else P.sdplib_blocks(f)
                o = Optimizer(time_limit=limit)
                t = time.time(); s = o.optimize(pr); dt = time.time() - t
                lit = LIT.get(name)
                rel = abs(abs(s.objval) - lit) / max(1.0, lit) if lit is not None else None
                st = s.stats
                rows.append((name, model, m, "%d blocks, side %d" % (len(blks), side), o.termination_status(), int(s.iter), dt, s.objval, lit, rel,
                             int(s.final_rank), int(st["lanczos_matvecs"]), int(st["full_eigs"]), int(st["batched_small_eigs"])))
            except Exception as e:                                   # a file the model cannot express must not stop the sweep
                rows.append((name, model, m, "%d blocks, side %d" % (len(blks), side), "ERROR %s: %s" % (type(e).__name__, str(e)[:60]),
                             0, 0.0, float("nan"), LIT.get(name), None, 0, 0, 0, 0))
            r = rows[-1]
            print("%-10s %-8s m=%5d %-22s %-16s it %7d %6.1f s obj %+.6g lit %s rel %s" % (
                r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], ("%.1e" % r[9]) if r[9] is not None else "-"), flush=True)
    with open(outp, "w") as fo:
        fo.write("# SDPLIB sweep: every `.dat-s` of the reference's `test/data/` through the library\n\n")
        fo.write("Reference DEFAULT options (tol 1e-4), time limit %.0f s per solve, one MI355X; `tools/sdplib_sweep.py --all`.  "
                 "`harness` = the reference's own reader (`test/base_sdplib.jl`: blocks merged, side = `length(c)`; loads only where "
                 "m = side), `blocks` = block structure kept.  `lit` = SDPLIB README optimum (absolute value), `rel` = "
                 "| |objective| - lit | / max(1, lit).  A first-order method at tol 1e-4 is expected within ~1e-3 of the optimum on "
                 "the well-conditioned families; TIME_LIMIT rows are reported as they ended.  Total %.0f s.\n\n" % (limit, time.time() - t_all))
        fo.write("| instance | model | m | structure | status | iterations | time s | objective | lit | rel | final rank | Lanczos mat-vecs | full_eig! | batched small |\n|---|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            fo.write("| %s | %s | %d | %s | %s | %d | %.2f | %.6g | %s | %s | %d | %d | %d | %d |\n" % (
                r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], ("%.6g" % r[8]) if r[8] is not None else "-",
                ("%.1e" % r[9]) if r[9] is not None else "-", r[10], r[11], r[12], r[13]))
        ok = [r for r in rows if r[4] == "OPTIMAL"]
        fo.write("\n%d solves, %d OPTIMAL within the limit; of those with a literature value, %d within 1e-3 and %d within 1e-2.\n" % (
            len(rows), len(ok), sum(1 for r in ok if r[9] is not None and r[9] <= 1e-3), sum(1 for r in ok if r[9] is not None and r[9] <= 1e-2)))
    print("wrote", outp)
    sys.exit(0)

names = [a for a in sys.argv[1:] if not a.startswith("--")]
FAMILY = ["mcp124-1", "mcp250-1", "mcp500-1", "maxG11", "maxG32", "maxG51", "maxG55"]
LIT["maxG55"] = 9999.21
for name in (names or FAMILY):
    pr = P.sdplib(g / f"{name}.dat-s")
    o = Optimizer(tol_gap=1e-4, tol_feasibility=1e-4, max_target_rank_krylov_eigs=64, time_limit=150.0,
                  psd_sign_engine=1 if ENGINE else 0)
    t = time.time(); s = o.optimize(pr); dt = time.time() - t
    side = pr.psd_sides()[0]
    print("%-9s n=%5d status %d iter %6d time %6.2f s obj %.4f lit %.4f rel %.2e rank %d matvecs %d full_eigs %d fop %d sign_engine %d" % (
        name, side, s.status, s.iter, dt, s.objval, LIT[name], abs(abs(s.objval) - LIT[name]) / LIT[name],
        s.final_rank, s.stats["lanczos_matvecs"], s.stats["full_eigs"], s.stats["fop_projections"], s.stats["sign_engine_projections"]), flush=True)
