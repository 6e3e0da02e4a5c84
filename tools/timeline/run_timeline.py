"""Intra-kernel timeline of the operator-form Lanczos step (task 1a).

Runs the headline regime (Max-Cut n = 4000, target rank 63, krylovdim 127) on a MEASUREMENT build of the library
(-DPX_TIMELINE, tools/timeline/build.sh): thread 0 of every workgroup of `k_fop_finish` and `k_lz_orth` stamps the
100 MHz device clock (s_memrealtime) at fixed points; the table is indexed by the Lanczos step k, so after the
solve it holds the stamps of the LAST restart cycle of the last projection (steps keep .. krylovdim-1, consecutive
launches).  Output: gpurun_out/timeline.json (raw medians) and a markdown table for profiles/.

    python tools/timeline/run_timeline.py [--n 4000] [--rank 63] [--iters 230] [--md profiles/r04_step_timeline.md]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from proxsdp_jl_amd import binding  # noqa: E402

binding.LIB_PATH = binding.pathlib.Path(ROOT) / "tools" / "timeline" / "libproxsdp_hip_tl.so"
from proxsdp_jl_amd import problems as P  # noqa: E402
from proxsdp_jl_amd.optimizer import Optimizer  # noqa: E402

TL_WG, TL_SLOTS, TL_K = 160, 8, 260


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4000)
    ap.add_argument("--rank", type=int, default=63)
    ap.add_argument("--iters", type=int, default=230)
    ap.add_argument("--md", default="gpurun_out/r04_step_timeline.md")
    ap.add_argument("--opts", default="", help="extra options k=v,k=v")
    args = ap.parse_args()
    L = binding.lib()
    L.proxsdp_hip_debug_timeline.argtypes = [C.POINTER(C.c_uint64), C.c_int64, C.c_int32]
    extra = {}
    for kv in filter(None, args.opts.split(",")):
        k, v = kv.split("=")
        extra[k] = float(v)
    pr = P.maxcut(args.n, seed=0)
    o = Optimizer(max_iter=args.iters, initial_target_rank=args.rank, max_target_rank_krylov_eigs=args.rank + 1, **extra)
    s = o.optimize(pr, trace_capacity=args.iters)
    tr = np.asarray(s.trace)
    it_s = 20.0 / float(tr[-1, 12] - tr[-21, 12])
    total = 2 * TL_K * TL_WG * TL_SLOTS
    buf = (C.c_uint64 * total)()
    rc = L.proxsdp_hip_debug_timeline(buf, total, 0)
    assert rc > 0, rc
    T = np.frombuffer(buf, dtype=np.uint64).reshape(2, TL_K, TL_WG, TL_SLOTS).astype(np.int64)
    n = args.n
    nt = (n + 63) // 64
    kd = max(2 * args.rank + 1, 25)
    # steps of the last cycle: the largest run of consecutive k < kd with increasing entry stamps
    ent = T[1, :kd, 0, 0]
    ks = [kd - 1]
    while ks[-1] - 1 >= 0 and 0 < ent[ks[-1] - 1] < ent[ks[-1]] and ent[ks[-1]] - ent[ks[-1] - 1] < 5000:
        ks.append(ks[-1] - 1)
    ks = ks[::-1][1:]                     # drop the first step of the cycle (k = keep: `first` form, other launch before it)
    us = 0.01                             # 100 MHz ticks -> microseconds
    rows = {}

    def stat(name, vals):
        v = np.asarray(vals, dtype=float) * us
        rows[name] = dict(median=float(np.median(v)), p10=float(np.percentile(v, 10)), p90=float(np.percentile(v, 90)))

    acc = {k: [] for k in (
        "A.span (first entry -> last exit of k_fop_finish)", "B.span (first entry -> last exit of k_lz_orth)",
        "gap A->B (last exit A -> first entry B)", "gap B->A' (last exit B -> first entry of the next A)",
        "step (first entry A -> first entry next A)",
        "A.entry skew (last workgroup's entry - first's)", "B.entry skew",
        "A.close: entry -> stop flag loaded (first load round trip)", "A.close: -> all loads landed", "A.close: -> fold + barrier 1",
        "A.close: -> beta, V h2 + barrier 2", "A.close: -> v_k stored (exit)",
        "A.fop: entry -> stop flag loaded", "A.fop: -> ELL gather done", "A.fop: -> Vp'w partials + barrier", "A.fop: -> exit",
        "B: entry -> stop flag loaded (first load round trip)", "B: -> all loads landed", "B: -> reductions + barrier 1",
        "B: -> coefficients + barrier 2", "B: -> w' assembled + barrier 3", "B: -> measured pass stored (exit)")}
    for k in ks[:-1]:
        A = T[0, k]
        B = T[1, k]
        A2 = T[0, k + 1]
        a_cl, a_fo = A[:nt], A[nt:2 * nt]
        if not (a_cl[:, 0].min() > 0 and a_fo[:, 0].min() > 0 and B[:nt, 0].min() > 0):
            continue
        a_entry = min(a_cl[:, 0].min(), a_fo[:, 0].min())
        a_exit = max(a_cl[:, 5].max(), a_fo[:, 4].max())
        b_entry = B[:nt, 0].min()
        b_exit = B[:nt, 6].max()
        a2_entry = min(A2[:nt, 0].min(), A2[nt:2 * nt, 0].min())
        acc["A.span (first entry -> last exit of k_fop_finish)"].append(a_exit - a_entry)
        acc["B.span (first entry -> last exit of k_lz_orth)"].append(b_exit - b_entry)
        acc["gap A->B (last exit A -> first entry B)"].append(b_entry - a_exit)
        acc["gap B->A' (last exit B -> first entry of the next A)"].append(a2_entry - b_exit)
        acc["step (first entry A -> first entry next A)"].append(a2_entry - a_entry)
        acc["A.entry skew (last workgroup's entry - first's)"].append(max(a_cl[:, 0].max(), a_fo[:, 0].max()) - a_entry)
        acc["B.entry skew"].append(B[:nt, 0].max() - b_entry)
        d = np.median(a_cl[:, 1:6] - a_cl[:, 0:5], axis=0)
        for name, v in zip(("A.close: entry -> stop flag loaded (first load round trip)", "A.close: -> all loads landed",
                            "A.close: -> fold + barrier 1", "A.close: -> beta, V h2 + barrier 2", "A.close: -> v_k stored (exit)"), d):
            acc[name].append(v)
        d = np.median(a_fo[:, 1:5] - a_fo[:, 0:4], axis=0)
        for name, v in zip(("A.fop: entry -> stop flag loaded", "A.fop: -> ELL gather done", "A.fop: -> Vp'w partials + barrier", "A.fop: -> exit"), d):
            acc[name].append(v)
        d = np.median(B[:nt, 1:7] - B[:nt, 0:6], axis=0)
        for name, v in zip(("B: entry -> stop flag loaded (first load round trip)", "B: -> all loads landed", "B: -> reductions + barrier 1",
                            "B: -> coefficients + barrier 2", "B: -> w' assembled + barrier 3", "B: -> measured pass stored (exit)"), d):
            acc[name].append(v)
    for k, v in acc.items():
        if v:
            stat(k, v)
    out = dict(n=n, rank=args.rank, krylovdim=kd, iters=int(s.iter), it_per_s_last20=it_s, steps_used=[int(ks[0]), int(ks[-1])] if ks else [],
               matvecs=int(s.stats["lanczos_matvecs"]), rows=rows, options=extra)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/timeline.json", "w"), indent=1)
    np.save("gpurun_out/timeline_raw.npy", T[:, :kd])
    with open(args.md, "w") as f:
        f.write("# Intra-kernel timeline of the operator-form Lanczos step (s_memrealtime, 100 MHz)\n\n")
        f.write(f"Max-Cut n = {n}, target rank {args.rank}, krylovdim {kd}; measurement build (-DPX_TIMELINE: thread 0 of every workgroup\n"
                f"stamps the device-wide clock; the stamps themselves cost a scalar memory round trip each, so the spans below are\n"
                f"upper bounds: the instrumented solve ran its last 20 iterations at {it_s:.0f} it/s).  Steps k = {out['steps_used']} of the last restart\n"
                f"cycle of iteration {int(s.iter)}; A = `k_fop_finish` (workgroups [0, nt) close step k-1, [nt, 2nt) form the operator rows of step k),\n"
                f"B = `k_lz_orth`; nt = {nt} workgroups of 64 rows.  Per-workgroup phases: median over workgroups, then over steps.  Resolution 0.01 us.\n\n")
        f.write("| segment | median us | p10 | p90 |\n|---|---|---|---|\n")
        for k, v in rows.items():
            f.write(f"| {k} | {v['median']:.2f} | {v['p10']:.2f} | {v['p90']:.2f} |\n")
    print(open(args.md).read())


if __name__ == "__main__":
    main()
