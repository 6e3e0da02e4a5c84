"""PDHG iterations/s on SMALL models (launch-latency territory): the reference's 3 x 3 / mixed-cone known answers, sensor
localisation (22 x 22), Max-Cut 150, MIMO 32 -- with the small-block batch kernel off / auto / forced.
gpurun -- python tools/gpurun_small_rate.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from proxsdp_jl_amd import problems as P, moi
from proxsdp_jl_amd.optimizer import Optimizer
import kat_problems as K

def sensorloc(n):
    import test_moi_mirror as T
    mm, x_true, a, d, d_bar = T.sensorloc_data(0, n)
    rng = np.random.default_rng(0)
    picks = [(i, j) for i in range(n) for j in range(i) if rng.random() > 0.9]
    m = moi.Model()
    X = m.add_variables(moi.sympackedlen(n + 2)); Xsq = moi.ivech(X)
    m.add_constraint(T.VOV(X), moi.PositiveSemidefiniteConeTriangle(n + 2))
    def eq(terms, rhs): m.add_constraint(T.VAF([T.VAT(1, T.SAT(c, int(v))) for c, v in terms], [-rhs]), moi.Zeros(1))
    for j in range(n):
        for k in range(mm):
            eq([(a[k][0]*a[k][0], Xsq[0,0]), (a[k][1]*a[k][1], Xsq[1,1]), (-2*a[k][0], Xsq[0,j+2]), (-2*a[k][1], Xsq[1,j+2]), (1.0, Xsq[j+2,j+2])], d_bar[k,j]**2)
    for (i, j) in picks: eq([(1.0, Xsq[i+2,i+2]), (1.0, Xsq[j+2,j+2]), (-2.0, Xsq[i+2,j+2])], d[i,j]**2)
    for (i, j, v) in ((0,0,1.0), (0,1,0.0), (1,0,0.0), (1,1,1.0)): m.add_constraint(T.vaf1(1.0, int(Xsq[i,j]), -v), moi.Zeros(1))
    m.set_objective_function(T.SAF([T.SAT(0.0, int(Xsq[0,0]))], 0.0)); m.set_objective_sense(moi.MIN_SENSE)
    return m.problem("sensorloc%d" % n)

tight = dict(tol_gap=1e-8, tol_feasibility=1e-8, min_iter=10**9)   # (1e-8: the loosest setting the sign engines accept; never stops)
cases = [("sdp_wiki 3x3", K.sdp_wiki(False), dict(max_iter=3000, **tight)),
         ("sensorloc 22x22", sensorloc(20), dict(max_iter=20000, **tight)),
         ("mixed_cones (1,3,104,1,5)", K.mixed_cones(), dict(max_iter=3000, **tight)),
         ("maxcut60", P.maxcut(60, seed=0), dict(max_iter=3000, **tight)),
         ("maxcut150 (Lanczos)", P.maxcut(150, seed=0), dict(max_iter=3000, **tight)),
         ("mimo32", P.mimo(32, seed=0), dict(max_iter=3000, **tight))]
for name, pr, kw in cases:
    for sbb in (0, -1, 1, 2):
        s = Optimizer(small_block_batch=sbb, **kw).optimize(pr)
        st = s.stats
        print(f"{name}: small_block_batch {sbb:2d}: iter {s.iter} status {s.status} {s.iter/st['loop_time']:.0f} it/s, {1e6*st['loop_time']/s.iter:.1f} us/iter "
              f"(psd {1e6*st['t_psd']/s.iter:.0f} linesearch {1e6*st['t_linesearch']/s.iter:.0f}) small_eigs {st['batched_small_eigs']} full_eigs {st['full_eigs']} obj {s.objval:.9f}", flush=True)
