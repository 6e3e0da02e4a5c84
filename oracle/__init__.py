"""TEST INFRASTRUCTURE ONLY.

CPU oracle: a NumPy/SciPy restatement of the reference's PDHG hot path
(/root/reference/src/pdhg.jl, prox_operators.jl, eigsolver.jl, residuals.jl,
scaling.jl).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this package -- never the product path (proxsdp.jl_amd/).

Parity status: the reference is pure Julia and cannot be run or compiled in
this image (no julia binary, no package depot), so there is no oracle/_ref.
The restatement is pinned by the reference's own known-answer tests rebuilt in
standard form (tests/test_oracle_kat.py).  The third-party eigen-solver layer
(KrylovKit.jl) is "parity unpinned": the reference's tests hold no vectors for
it; see oracle/eig.py.  Likewise "parity unpinned": equilibrate! and the
approx_norm=false (Arpack.svds) step size -- off by default and never switched
on by the reference's tests; restated from src/equilibration.jl and
src/pdhg.jl:64-119,751-755 and pinned only by the invariance of the known
answers under them.
"""
from .api import solve, to_standard_form  # noqa: F401
from .options import Options  # noqa: F401
