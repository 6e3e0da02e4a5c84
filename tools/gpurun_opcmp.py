"""operator form vs packed triangle at small n (where the packed mat-vec is only a few MB)."""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from pathlib import Path
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
g = Path(__file__).resolve().parent.parent / "tests" / "golden" / "sdplib"
for name, pr in (("er1000", P.maxcut(1000, seed=0)), ("maxG51", P.sdplib(g / "maxG51.dat-s")), ("er2000", P.maxcut(2000, seed=0)),
                 ("maxG32", P.sdplib(g / "maxG32.dat-s"))):
    for op in (0, 1):
        Optimizer(max_iter=20, lanczos_operator=op).optimize(pr)
        s = Optimizer(max_iter=300, lanczos_operator=op).optimize(pr)
        st = s.stats
        print("%-7s op %d: fop %3d matvecs %6d loop %.3f s  -> %.1f us per mat-vec (all-in)" % (
            name, op, st["fop_projections"], st["lanczos_matvecs"], st["loop_time"], 1e6 * st["loop_time"] / st["lanczos_matvecs"]), flush=True)
