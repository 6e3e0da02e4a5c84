#!/usr/bin/env python
"""bench.py -- PDHG iterations/sec on the metric's instance (BASELINE.json):
Max-Cut SDP, Erdos-Renyi graph, n = 4000, one PSD cone, tol 1e-4 defaults.

    python bench.py --gpus N --steps K --warmup W

A "step" is one PDHG iteration (primal step with the PSD projection, linesearch
dual step, residuals/gap) of ONE solve; W untimed iterations precede the K timed
ones inside the same solve (max_iter = W+K), the split being taken from the
per-iteration host clock the library stamps after each iteration's final stream
synchronisation.  The whole call is bracketed by barrier + cuda.synchronize.
The problem is uploaded before the loop starts (inputs resident in HBM).

N > 1: the single-PSD-block path does not shard (SURVEY.md section 8e,
DESIGN.md section 7) -> N independent replicas (seed = rank), no data-path
collective, scaling "weak"; value = total iterations of all ranks / max time.

Extra objects on the JSON line: "roofline" for the dominant kernel
(k_symv_packed; HIP events recorded by the library on its own stream around
every 16th launch inside the timed solve) and "cpu_baseline" (the NumPy/SciPy
oracle -- a restatement, NOT the Julia reference, which cannot run here -- on a
bounded sample of the same instance, rank 0, N = 1 only), plus "packed_operator"
(same window with the reference's mat-vec operator), "rank_sqrt_n" (window
started at the metric's target rank), "time_to_tol" and "config_maxcut_n1000"
(BASELINE config 2 on both sides).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--n", type=int, default=4000, help="PSD side (metric: 4000)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU-baseline sample budget")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-packed-leg", action="store_true",
                    help="skip the extra window with lanczos_operator=0 (packed-triangle mat-vec)")
    ap.add_argument("--lanczos-operator", type=int, default=-1, help="-1 auto (operator form), 0 packed triangle")
    ap.add_argument("--no-time-to-tol", action="store_true",
                    help="skip the full solve to tol 1e-4 (time_to_tol object)")
    ap.add_argument("--krylov-rank", type=int, default=64,
                    help="max_target_rank_krylov_eigs for the time-to-tol leg (metric: rank ~ sqrt(n))")
    ap.add_argument("--profile-every", type=int, default=16)
    ap.add_argument("--support-path", type=int, default=-1, help="-1 auto, 0 dense vector passes, 1 support-aware")
    ap.add_argument("--workload", choices=["maxcut", "mimo", "randsdp"], default="maxcut",
                    help="maxcut: the metric's instance, replicas for N>1; mimo: BASELINE config 4, a block-diagonal "
                         "model of --blocks MIMO n=512 instances, PSD blocks sharded over the ranks; randsdp: BASELINE "
                         "config 3, dense equality rows generated in HBM (--rand-n 2000 --rand-m 4000 = 64 GB)")
    ap.add_argument("--rand-rank", type=int, default=50,
                    help="randsdp: initial_target_rank (BASELINE config 3 names target rank 50; the reference starts at 2)")
    ap.add_argument("--no-rank-leg", action="store_true", help="skip the window at target rank ~ sqrt(n)")
    ap.add_argument("--rand-n", type=int, default=2000)
    ap.add_argument("--rand-m", type=int, default=4000)
    ap.add_argument("--blocks", type=int, default=8)
    ap.add_argument("--mimo-n", type=int, default=512)
    args = ap.parse_args()

    import torch
    from proxsdp_jl_amd import binding, problems, replicas
    from proxsdp_jl_amd.optimizer import Optimizer
    rank, local_rank, world = replicas.rank_info()
    dist = None
    backend = os.environ.get("PROXSDP_BENCH_BACKEND", "nccl")       # "gloo" only for 1-GPU validation runs
    ndev = max(1, torch.cuda.device_count())
    dev_id = local_rank % ndev
    # PROXSDP_BENCH_FORCE_DIST=1: build the process group even for one rank (validates the
    # RCCL barrier / reductions next to the library on a 1-GPU box)
    if world > 1 or os.environ.get("PROXSDP_BENCH_FORCE_DIST") == "1":
        torch.cuda.set_device(dev_id)
        dist = replicas.init(backend, rank, world, device=torch.device("cuda", dev_id) if backend == "nccl" else None)

    if binding.device_count() <= 0:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if args.workload == "mimo":
        return bench_mimo(args, torch, dist, rank, world, dev_id, backend)
    if args.workload == "randsdp":
        return bench_randsdp(args, torch, dist, rank, world, dev_id, backend)
    n = args.n
    K, W = args.steps, args.warmup
    pr = problems.maxcut(n, seed=replicas.replica_seed(args.seed, rank))
    N = n * (n + 1) // 2

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    opt = Optimizer(max_iter=W + K, device_id=dev_id, profile_symv_every=args.profile_every,
                    support_path=args.support_path, lanczos_operator=args.lanczos_operator)
    sync()
    t0 = time.time()
    sol = opt.optimize(pr, trace_capacity=W + K)
    sync()
    wall = time.time() - t0
    tr = sol.trace
    if len(tr) < W + K:
        raise SystemExit(f"solve stopped after {len(tr)} iterations (< warmup+steps): status {sol.status}")
    t_start = tr[W - 1, 12] if W > 0 else 0.0
    t_steps = float(tr[W + K - 1, 12] - t_start)
    total_steps, t_steps = replicas.aggregate(dist, K, t_steps, device="cuda" if dist is not None else "cpu")
    value = total_steps / t_steps

    st = sol.stats
    symv_bytes = 8.0 * N + 16.0 * n                       # algorithmic bytes of one mat-vec

    def matvec_roofline(stats, packed):
        """roofline object of the Lanczos mat-vec launches of one solve.  `achieved` is the
        ALGORITHMIC figure of SURVEY section 8d (the 8N+16n bytes the reference's dsymv('U')
        reads per mat-vec) over the mean launch duration (kernel-only HIP events on the solve
        stream, every --profile-every-th launch).  HBM traffic from PMC counters cannot be
        taken by bench.py itself (and `rocprofv3 --pmc` segfaults on n = 4000 solves in this
        image): the figures are from separate --pmc FETCH_SIZE / WRITE_SIZE passes, see
        traffic_source."""
        ms = stats["symv_profiled_ms"] / max(1, stats["symv_profiled"])
        ach = symv_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
             "bytes_per_launch": symv_bytes, "avg_launch_ms": ms, "launches_profiled": int(stats["symv_profiled"]),
             "launches": int(stats["symv_launches"])}
        if packed:
            r["kernel"] = "k_symv_finish / k_symv_packed (tiles of the packed triangle)"
            r["traffic"] = (2 * 31492.7 + 1984.5) * 1024 if n == 4000 else None
            r["traffic_source"] = "profiles/r01_pmc_fetch_write_symv.md (2*FETCH_SIZE + WRITE_SIZE, isolated kernel)"
        else:
            r["kernel"] = "k_fop_finish / k_fop (operator-form mat-vec: previous factors + sparse update)"
            r["traffic"] = None
            r["traffic_source"] = ("operator form reads ~16 n r + O(|S|) bytes instead of the triangle, so the algorithmic "
                                   "figure can exceed the HBM peak; PMC at n = 2000: 2*FETCH_SIZE = 1.5 MB per launch vs "
                                   "16.0 MB algorithmic (profiles/r01_pmc_operator_form.md)")
        return r

    roof = matvec_roofline(st, packed=st["fop_projections"] == 0)
    roof["loop_algorithmic_GBs"] = st["algorithmic_bytes"] / max(st["loop_time"], 1e-9) / 1e9
    mv_timed = float(tr[W:W + K, 13].sum())
    trials_timed = float(tr[W:W + K, 11].sum())
    out = {
        "metric": "PDHG iterations/sec, Max-Cut SDP n=%d (tol_gap=tol_feasibility=1e-4 defaults)" % n,
        "value": value, "unit": "iterations/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * t_steps / K, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "Max-Cut SDP, Erdos-Renyi G(n,12/(n-1)) unit weights, n=%d, one PSD cone, "
                               "Nx=%d, p=%d equality rows; reference default options" % (n, N, n),
                   "parallelism": "replicas x%d (single PSD block does not shard)" % world,
                   "timed_iterations": [W + 1, W + K],
                   "lanczos_matvecs_per_step": mv_timed / K, "linesearch_trials_per_step": trials_timed / K,
                   "target_rank": int(tr[W + K - 1, 10])},
        "roofline": roof,
        "solve_wall_s": wall, "init_s": st["init_time"], "exit_s": st["exit_time"],
    }

    if rank == 0 and world == 1 and st["fop_projections"] > 0 and not args.no_packed_leg:
        # the same window with the reference's operator: every mat-vec streams the packed triangle
        # (HBM-bound tile kernel); kept beside the headline so the kernel-level roofline stays visible
        o1 = Optimizer(max_iter=W + K, device_id=dev_id, profile_symv_every=args.profile_every,
                       support_path=args.support_path, lanczos_operator=0)
        s1 = o1.optimize(pr, trace_capacity=W + K)
        t1 = float(s1.trace[W + K - 1, 12] - (s1.trace[W - 1, 12] if W > 0 else 0.0))
        out["packed_operator"] = {"value": K / t1, "unit": "iterations/s", "ms_per_step": 1e3 * t1 / K,
                                  "options": {"lanczos_operator": 0},
                                  "roofline": matvec_roofline(s1.stats, packed=True)}

    if rank == 0 and world == 1 and not args.no_rank_leg:
        # the metric's "rank ~ sqrt(n)" regime directly: a window started at target rank sqrt(n)
        # (library-only knob initial_target_rank; the reference hard-codes 2 and needs ~12 000
        # iterations of rank updates to get there), Lanczos path kept by max_target_rank_krylov_eigs
        r0 = max(2, int(round(n ** 0.5)))
        Kr, Wr = min(K, 200), min(W, 20)
        o3 = Optimizer(max_iter=Wr + Kr, device_id=dev_id, initial_target_rank=r0,
                       max_target_rank_krylov_eigs=max(args.krylov_rank, r0), support_path=args.support_path,
                       lanczos_operator=args.lanczos_operator)
        s3 = o3.optimize(pr, trace_capacity=Wr + Kr)
        t3 = float(s3.trace[Wr + Kr - 1, 12] - (s3.trace[Wr - 1, 12] if Wr > 0 else 0.0))
        out["rank_sqrt_n"] = {"value": Kr / t3, "unit": "iterations/s", "ms_per_step": 1e3 * t3 / Kr,
                              "timed_iterations": [Wr + 1, Wr + Kr], "target_rank": int(s3.trace[Wr + Kr - 1, 10]),
                              "lanczos_matvecs_per_step": float(s3.trace[Wr:Wr + Kr, 13].sum()) / Kr,
                              "full_eigs": int(s3.stats["full_eigs"]),
                              "host_eigensolve_ms_per_step": 1e3 * s3.stats["t_primal"] / max(1, int(s3.iter)),
                              "lanczos_restarts_per_step": s3.stats["lanczos_restarts"] / max(1, int(s3.iter)),
                              "options": {"initial_target_rank": r0, "max_target_rank_krylov_eigs": max(args.krylov_rank, r0)}}

    if rank == 0 and world == 1 and not args.no_time_to_tol:
        # second half of the metric: wall time to status OPTIMAL at tol_gap = tol_feasibility = 1e-4.
        # With the reference default max_target_rank_krylov_eigs = 16 the solve falls into a full
        # eigendecomposition per iteration once target_rank reaches 17 (prox_operators.jl:46-49);
        # the metric's "rank ~ sqrt(n)" regime keeps the Lanczos path, so the knob is raised here
        # (and must be raised identically for any CPU comparison).
        o2 = Optimizer(device_id=dev_id, time_limit=300.0, max_target_rank_krylov_eigs=args.krylov_rank)
        s2 = o2.optimize(pr)
        out["time_to_tol"] = {"status": o2.termination_status(), "time_s": s2.time, "iterations": int(s2.iter),
                              "objective": o2.objective_value(), "gap": s2.gap,
                              "whole_solve_it_per_s": s2.iter / max(s2.stats["loop_time"], 1e-9),
                              "final_rank": int(s2.final_rank),
                              "lanczos_matvecs": int(s2.stats["lanczos_matvecs"]),
                              "lanczos_restarts": int(s2.stats["lanczos_restarts"]),
                              "full_eigs": int(s2.stats["full_eigs"]),
                              "options": {"max_target_rank_krylov_eigs": args.krylov_rank}}

    if rank == 0 and world == 1 and not args.no_cpu:
        import oracle                                           # baseline leg only
        ncores = os.cpu_count() or 1
        o = oracle.Options()
        o.time_limit = args.cpu_seconds
        tc = time.time()
        ref = oracle.solve(pr, o)
        cpu_it = max(int(ref.iter), 1)
        cpu_loop = ref.stats["loop_time"]
        gpu_same = float(tr[min(cpu_it, len(tr)) - 1, 12])
        out["cpu_baseline"] = {"value": cpu_it / cpu_loop, "unit": "iterations/s", "cores": ncores,
                               "kind": "port",
                               "sample": "NumPy/SciPy oracle restatement (not Julia), iterations 1-%d of the same "
                                         "instance, %.1f s of CPU work, OpenBLAS threads=%d" % (cpu_it, cpu_loop, ncores),
                               "gpu_it_per_s_same_iterations": min(cpu_it, len(tr)) / max(gpu_same, 1e-9),
                               "wall_s": time.time() - tc}
        # BASELINE config 2 (Max-Cut n=1000, the size the CPU path handles comfortably): the same
        # 200 iterations on both sides ("iterations 1-200 at n=1000 fully")
        pr2 = problems.maxcut(1000, seed=args.seed)
        o.time_limit = 3600.0
        o.max_iter = 200
        tc2 = time.time()
        ref2 = oracle.solve(pr2, o)
        g2 = Optimizer(max_iter=200, device_id=dev_id).optimize(pr2, trace_capacity=200)
        out["config_maxcut_n1000"] = {
            "iterations": [1, int(ref2.iter)],
            "cpu_it_per_s": ref2.iter / max(ref2.stats["loop_time"], 1e-9), "cpu_kind": "port", "cpu_cores": ncores,
            "gpu_it_per_s": g2.iter / max(g2.stats["loop_time"], 1e-9),
            "same_iteration_count": bool(int(g2.iter) == int(ref2.iter)), "wall_s": time.time() - tc2}
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def bench_randsdp(args, torch, dist, rank, world, dev_id, backend):
    """BASELINE config 3: randSDP (test/base_randsdp.jl:4-23 + the bounds of test/moi_randsdp.jl),
    every A_k dense.  At n=2000, m=4000 the coefficient matrix is 4000 x 2 001 000 doubles = 64 GB:
    generated in HBM and handed to the library as a borrowed device pointer (M_dense).  Each PDHG
    iteration streams it twice (A x, and A'[y1 y2 y3] for the batched linesearch candidates); a
    single PSD block does not shard, so N > 1 runs replicas."""
    from proxsdp_jl_amd import problems, replicas
    from proxsdp_jl_amd.optimizer import Optimizer
    K, W = args.steps, args.warmup
    n, m = args.rand_n, args.rand_m
    torch.cuda.set_device(dev_id)
    t_gen = time.time()
    pr = problems.randsdp_device(n, m, seed=replicas.replica_seed(args.seed, rank), device="cuda:%d" % dev_id)
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    opt = Optimizer(max_iter=W + K, device_id=dev_id, initial_target_rank=args.rand_rank,
                    max_target_rank_krylov_eigs=max(16, args.rand_rank))
    sync()
    t0 = time.time()
    sol = opt.optimize(pr, trace_capacity=W + K)
    sync()
    wall = time.time() - t0
    tr = sol.trace
    if len(tr) < W + K:
        raise SystemExit(f"solve stopped after {len(tr)} iterations (< warmup+steps): status {sol.status}")
    t_steps = float(tr[W + K - 1, 12] - (tr[W - 1, 12] if W > 0 else 0.0))
    total_steps, t_steps = replicas.aggregate(dist, K, t_steps, device="cuda" if dist is not None else "cpu")
    if rank == 0:
        st = sol.stats
        N = n * (n + 1) // 2
        bytes_pass = 8.0 * m * N
        pass_ms = st["dense_ms"] / max(1, st["dense_passes"])
        print(json.dumps({
            "metric": "PDHG iterations/sec, randSDP n=%d m=%d (dense A, %.1f GB)" % (n, m, bytes_pass / 1e9),
            "value": total_steps / t_steps, "unit": "iterations/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * t_steps / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "randSDP n=%d, m=%d dense equality rows + %d bound rows, Nx=%d; A generated in HBM "
                                   "(torch), borrowed by the library" % (n, m, 2 * n, N),
                       "parallelism": "replicas x%d (single PSD block does not shard)" % world,
                       "dense_passes_per_step": st["dense_passes"] / max(1, int(sol.iter)),
                       "linesearch_trials_per_step": st["linesearch_trials"] / max(1, int(sol.iter)),
                       "lanczos_matvecs_per_step": st["lanczos_matvecs"] / max(1, int(sol.iter)),
                       "full_eigs": int(st["full_eigs"]), "target_rank": int(tr[W + K - 1, 10]),
                       "options": {"initial_target_rank": args.rand_rank,
                                   "max_target_rank_krylov_eigs": max(16, args.rand_rank)}},
            "roofline": {"bound": "hbm", "kernel": "k_dense_mtv / k_dense_mv (one pass over A)",
                         "achieved": bytes_pass / (pass_ms * 1e-3) / 1e9 if pass_ms > 0 else None, "peak": 8000.0,
                         "unit": "GB/s", "frac": (bytes_pass / (pass_ms * 1e-3) / 1e9 / 8000.0) if pass_ms > 0 else None,
                         "traffic": None,
                         "traffic_note": "rocprofv3 --pmc FETCH_SIZE segfaults at this size; on a 962 MB instance of the same "
                                         "kernels traffic/algorithmic = 1.005 (A'y) and 1.018 (A x): profiles/r01_pmc_dense.md",
                         "bytes_per_launch": bytes_pass, "avg_launch_ms": pass_ms,
                         "launches": int(st["dense_passes"])},
            "generate_s": t_gen, "solve_wall_s": wall, "init_s": st["init_time"], "exit_s": st["exit_time"]}))
    if dist is not None:
        dist.destroy_process_group()


def bench_mimo(args, torch, dist, rank, world, dev_id, backend):
    """BASELINE config 4: `--blocks` independent MIMO detection SDPs (test/base_mimo.jl, n=512 ->
    PSD side 513, 263 682 box rows each) as ONE block-diagonal model; its PSD blocks are sharded
    over the ranks (one block per GPU at N = blocks), scalars all-reduced twice per iteration
    (RCCL).  A step is one PDHG iteration of the coupled model; strong scaling in N."""
    from proxsdp_jl_amd import problems, replicas, sharded
    from proxsdp_jl_amd.optimizer import Optimizer
    K, W = args.steps, args.warmup
    model = problems.block_diag_problems([problems.mimo(args.mimo_n, seed=s) for s in range(args.blocks)],
                                         name="mimo-x%d" % args.blocks)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    sync()
    t0 = time.time()
    if dist is None:
        opt = Optimizer(max_iter=W + K, device_id=dev_id, support_path=args.support_path)
        sol = opt.optimize(model, trace_capacity=W + K)
    else:
        cdev = torch.device("cuda", dev_id) if backend == "nccl" else None
        opt, sol, _ = sharded.solve_sharded(model, dist, rank, world, device_id=dev_id,
                                            collective_device=cdev, max_iter=W + K)
    sync()
    wall = time.time() - t0
    tr = sol.trace
    if len(tr) < W + K:                      # MIMO models converge in O(100) iterations: time what there is
        W = min(W, max(len(tr) // 5, 0))
        K = len(tr) - W
    t_steps = float(tr[W + K - 1, 12] - (tr[W - 1, 12] if W > 0 else 0.0))
    _, t_steps = replicas.aggregate(dist, K, t_steps, device="cuda" if (dist is not None and backend == "nccl") else "cpu")
    if rank == 0:
        side = args.mimo_n + 1
        print(json.dumps({
            "metric": "PDHG iterations/sec, MIMO detection SDP n=%d x %d blocks (one block-diagonal model)" % (args.mimo_n, args.blocks),
            "value": K / t_steps, "unit": "iterations/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * t_steps / K, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "MIMO x%d: PSD side %d per block, Nx=%d, Q=%d; blocks sharded over %d rank(s)"
                                   % (args.blocks, side, model.n, model.A.shape[0] + model.G.shape[0], world),
                       "parallelism": "block-sharded, scalar all-reduce x2 per iteration" if world > 1 else "single GPU, blocks in sequence",
                       "status_after_window": int(sol.status), "objective": float(sol.objval)},
            "solve_wall_s": wall}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
