"""Round-3 probe: the solve to tolerance of the metric instance (rank-64 knob) with the K x K eigensolve split from
krylovdim 64 (default) and from 24 (host_eig_merge = 1).  Used for two experiments: (1) read-backs of the Lanczos
record and of the iteration's scalars as kernel stores into pinned host memory instead of copy commands -- measured
neutral (A/B on one box: 7.27 / 7.10 s against 7.16 / 7.13 s), not kept; (2) the restart rotation's upload from
PINNED staging buffers instead of a pageable vector -- kept (7.34 / 7.47 -> 7.16 / 7.13 s)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
pr = P.maxcut(4000, seed=0)
out = {}
for rep in (0, 1):
    for hm in (-1, 1):
        o = Optimizer(time_limit=200.0, max_target_rank_krylov_eigs=64, host_eig_merge=hm)
        s = o.optimize(pr)
        out[f"t2t_k64_merge{hm}_rep{rep}"] = dict(status=s.status, iter=int(s.iter), obj=s.objval, time=s.time,
                                                  host_eig_s=s.stats["host_eig_time"], overlapped_s=s.stats["host_eig_overlap_time"],
                                                  merges=int(s.stats["host_eig_merges"]), matvecs=int(s.stats["lanczos_matvecs"]))
        print("t2t", hm, rep, out[f"t2t_k64_merge{hm}_rep{rep}"], flush=True)
for hm in (-1, 1):
    o = Optimizer(time_limit=200.0, max_target_rank_krylov_eigs=64, host_eig_merge=hm, lanczos_warm_start=1)
    s = o.optimize(pr)
    out[f"t2t_k64_warm_merge{hm}"] = dict(status=s.status, iter=int(s.iter), obj=s.objval, time=s.time)
    print("warm", hm, out[f"t2t_k64_warm_merge{hm}"], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r3_zero_copy.json", "w"), indent=1)
