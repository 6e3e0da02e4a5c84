import sys; sys.path.insert(0,'.')
from proxsdp_jl_amd import problems
from proxsdp_jl_amd.optimizer import Optimizer
model = problems.block_diag_problems([problems.mimo(512, seed=s) for s in range(8)], name="mimo-x8")
for kw in (dict(), dict(lanczos_cycle_kernel=1), dict(support_path=1, lanczos_cycle_kernel=1)):
    try:
        s = Optimizer(max_iter=60, **kw).optimize(model, trace_capacity=60)
        st = s.stats
        print(kw, s.status, s.iter, {k: st[k] for k in ("lanczos_matvecs","fop_projections","cycle_launches","cycle_steps","lanczos_restarts","lanczos_calls")}, "loop_s", st["loop_time"])
    except Exception as e:
        print(kw, "ERR", e)
