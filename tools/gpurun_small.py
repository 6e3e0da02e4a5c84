"""multi-block SDPLIB instances: batched small-block projection vs one dense eigensolver call per block.  gpurun helper."""
import sys, json, time
sys.path.insert(0, ".")
from proxsdp_jl_amd import problems
from proxsdp_jl_amd.optimizer import Optimizer
for f, it in (("truss1", 0), ("control1", 3000), ("theta1", 3000)):
    pr = problems.sdplib_blocks(f"tests/golden/sdplib/{f}.dat-s")
    for sbb in (0, -1, 1):
        kw = dict(max_iter=it) if it else {}
        o = Optimizer(small_block_batch=sbb, **kw)
        s = o.optimize(pr)
        print(json.dumps(dict(inst=f, blocks=pr.psd_sides(), small_block_batch=sbb, iters=int(s.iter), loop_s=s.stats["loop_time"],
                              it_per_s=s.iter / s.stats["loop_time"], batched=int(s.stats["batched_small_eigs"]))))
