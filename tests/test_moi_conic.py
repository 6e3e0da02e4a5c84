"""MathOptInterface's standard conic test problems, the ones `MOI.Test.runtests` feeds the reference's optimizer in
/root/reference/test/moitest.jl:34-75 (config atol 1e-4 / rtol 1e-3, solver tolerances 1e-6, Silent).  MathOptInterface is a
dependency of the reference that is NOT on disk (Project.toml `[deps] MathOptInterface`, compat "1"); the problems are restated from
the statements its test module publishes in comments (src/Test/test_conic.jl: `test_conic_linear_*` = LIN1..LIN4,
`test_conic_SecondOrderCone_*` = SOC1..SOC4, `test_conic_PositiveSemidefiniteConeTriangle_*` = SDP0 / SDP1), each with its
stated optimum; the optima are closed-form here and independent of either solver.

Both back ends behind the same model layer, as in test_moi_mirror.py: the CPU oracle and the HIP library (`-m gpu`).
The infeasible problems assert what MOI.Test asserts: INFEASIBLE with an infeasibility certificate as the dual status."""
import numpy as np
import pytest

from proxsdp_jl_amd import moi

from test_moi_mirror import BACKENDS, SAF, SAT, VAF, VAT, VOV, make

ATOL, RTOL = 1e-4, 1e-3          # MOI.Test.Config of moitest.jl:36-38


def close(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.all(np.abs(a - b) <= ATOL + RTOL * np.abs(b))


def identity_vaf(vs):
    return VAF([VAT(k + 1, SAT(1.0, v)) for k, v in enumerate(vs)], [0.0] * len(vs))


# ------------------------------------------------------------------ LIN1 .. LIN4
@pytest.mark.parametrize("form", ["VectorOfVariables", "VectorAffineFunction"])
@pytest.mark.parametrize("backend", BACKENDS)
def test_conic_linear(backend, form):
    """LIN1: min -3x - 2y - 4z, x + y + z = 3, y + z = 2, (x, y, z) >= 0.  Optimum -11 at (1, 0, 2); duals (0, 2, 0) of the
    cone and (-3, -1) of the equalities."""
    m = make(backend)
    v = m.add_variables(3)
    vc = m.add_constraint(VOV(v) if form == "VectorOfVariables" else identity_vaf(v), moi.Nonnegatives(3))
    c = m.add_constraint(VAF([VAT(1, SAT(1.0, v[0])), VAT(1, SAT(1.0, v[1])), VAT(1, SAT(1.0, v[2])),
                              VAT(2, SAT(1.0, v[1])), VAT(2, SAT(1.0, v[2]))], [-3.0, -2.0]), moi.Zeros(2))
    m.set_objective_function(SAF([SAT(-3.0, v[0]), SAT(-2.0, v[1]), SAT(-4.0, v[2])], 0.0))
    m.set_objective_sense(moi.MIN_SENSE)
    m.optimize()
    assert m.termination_status() == "OPTIMAL" and m.primal_status() == "FEASIBLE_POINT" and m.dual_status() == "FEASIBLE_POINT"
    assert close(m.objective_value(), -11) and close(m.dual_objective_value(), -11)
    assert close(m.variable_primal(v), [1, 0, 2])
    assert close(m.constraint_primal(vc), [1, 0, 2]) and close(m.constraint_primal(c), [0, 0])
    assert close(m.constraint_dual(vc), [0, 2, 0]) and close(m.constraint_dual(c), [-3, -1])


@pytest.mark.parametrize("form", ["VectorOfVariables", "VectorAffineFunction"])
@pytest.mark.parametrize("backend", BACKENDS)
def test_conic_linear_2(backend, form):
    """LIN2: min 3x + 2y - 4z + 0s, x - s = -4, y = -3, x + z = 12, x free, y <= 0, z >= 0, s = 0.  Optimum -82 at
    (-4, -3, 16, 0)."""
    m = make(backend)
    x, y, z, s = m.add_variables(4)
    c = m.add_constraint(VAF([VAT(1, SAT(1.0, x)), VAT(1, SAT(-1.0, s)), VAT(2, SAT(1.0, y)), VAT(3, SAT(1.0, x)), VAT(3, SAT(1.0, z))],
                             [4.0, 3.0, -12.0]), moi.Zeros(3))
    wrap = (lambda vs: VOV(vs)) if form == "VectorOfVariables" else identity_vaf
    vy = m.add_constraint(wrap([y]), moi.Nonpositives(1))
    vz = m.add_constraint(wrap([z]), moi.Nonnegatives(1))
    vs = m.add_constraint(wrap([s]), moi.Zeros(1))
    m.set_objective_function(SAF([SAT(3.0, x), SAT(2.0, y), SAT(-4.0, z)], 0.0))
    m.set_objective_sense(moi.MIN_SENSE)
    m.optimize()
    assert m.termination_status() == "OPTIMAL"
    assert close(m.objective_value(), -82) and close(m.dual_objective_value(), -82)
    assert close(m.variable_primal([x, y, z, s]), [-4, -3, 16, 0])
    assert close(m.constraint_primal(c), [0, 0, 0]) and close(m.constraint_primal(vy), [-3]) and close(m.constraint_primal(vz), [16])
    assert close(m.constraint_primal(vs), [0])
    # duals from the LP itself: reduced costs 3 - (l1 + l3) = 0, -4 - l3 = 0  ->  l3 = -4, l1 = 7; y's bound takes 2 - l2
    assert close(m.constraint_dual(c)[[0, 2]], [7, -4])
    assert close(m.constraint_dual(c)[1] + m.constraint_dual(vy)[0], 2) and close(m.constraint_dual(vz), [0])
    assert close(m.constraint_dual(vs), [7])


def assert_reports_infeasibility(m):
    """What MOI.Test asserts on an infeasible model: `termination_status == config.infeasible_status` (INFEASIBLE) and
    `dual_status == INFEASIBILITY_CERTIFICATE`.  The reference gets there through its stop rules for a diverging dual
    objective and the dual-ray test behind them (pdhg.jl:180-205, 281-330)."""
    assert m.termination_status() == "INFEASIBLE"
    assert m.dual_status() == "INFEASIBILITY_CERTIFICATE"
    assert m.primal_status() in ("INFEASIBLE_POINT", "NO_SOLUTION")


@pytest.mark.parametrize("which", ["LIN3", "LIN4"])
@pytest.mark.parametrize("backend", BACKENDS)
def test_conic_linear_INFEASIBLE(backend, which):
    """LIN3: -1 + x in R+, 1 + x in R-.  LIN4: -1 + x in R+, x in R- (as a variable-in-cone constraint)."""
    m = make(backend, time_limit=5.0)
    x = m.add_variable()
    m.add_constraint(VAF([VAT(1, SAT(1.0, x))], [-1.0]), moi.Nonnegatives(1))
    if which == "LIN3":
        m.add_constraint(VAF([VAT(1, SAT(1.0, x))], [1.0]), moi.Nonpositives(1))
    else:
        m.add_constraint(VOV([x]), moi.Nonpositives(1))
    m.optimize()
    assert_reports_infeasibility(m)


# ------------------------------------------------------------------ SOC1 .. SOC4
@pytest.mark.parametrize("form", ["VectorOfVariables", "VectorAffineFunction"])
@pytest.mark.parametrize("backend", BACKENDS)
def test_conic_SecondOrderCone(backend, form):
    """SOC1: max y + z, x = 1, x >= |(y, z)|.  Optimum sqrt 2 at (1, 1/sqrt 2, 1/sqrt 2); duals -sqrt 2 (equality) and
    (sqrt 2, -1, -1) (cone)."""
    m = make(backend)
    x, y, z = m.add_variables(3)
    ceq = m.add_constraint(VAF([VAT(1, SAT(1.0, x))], [-1.0]), moi.Zeros(1))
    csoc = m.add_constraint(VOV([x, y, z]) if form == "VectorOfVariables" else identity_vaf([x, y, z]), moi.SecondOrderCone(3))
    m.set_objective_function(SAF([SAT(1.0, y), SAT(1.0, z)], 0.0))
    m.set_objective_sense(moi.MAX_SENSE)
    if form == "VectorOfVariables":
        pr = m.problem()
        assert pr.n == 3 and len(pr.soc) == 1 and list(pr.soc[0]) == [0, 1, 2] and pr.A.shape == (1, 3)
    else:
        pr = m.problem()                                      # VectorSlack: three fresh cone variables tied by three rows
        assert pr.n == 6 and list(pr.soc[0]) == [3, 4, 5] and pr.A.shape == (4, 6)
    m.optimize()
    r2 = np.sqrt(2.0)
    assert m.termination_status() == "OPTIMAL" and m.primal_status() == "FEASIBLE_POINT" and m.dual_status() == "FEASIBLE_POINT"
    assert close(m.objective_value(), r2) and close(m.dual_objective_value(), r2)
    assert close(m.variable_primal([x, y, z]), [1, 1 / r2, 1 / r2])
    assert close(m.constraint_primal(ceq), [0]) and close(m.constraint_primal(csoc), [1, 1 / r2, 1 / r2])
    assert close(m.constraint_dual(ceq), [-r2]) and close(m.constraint_dual(csoc), [r2, -1, -1])


@pytest.mark.parametrize("nonneg", [True, False], ids=["negative_post_bound", "negative_initial_bound"])
@pytest.mark.parametrize("backend", BACKENDS)
def test_conic_SecondOrderCone_negative_bound(backend, nonneg):
    """SOC2: min x, y >= 1/sqrt 2 (as -1/sqrt 2 + y in R+, or 1/sqrt 2 - y in R-), 1 - t = 0, (t, x, y) in SOC.
    Optimum -1/sqrt 2 at x = -1/sqrt 2, y = 1/sqrt 2, t = 1."""
    m = make(backend)
    x, y, t = m.add_variables(3)
    h = 1 / np.sqrt(2.0)
    if nonneg:
        cb = m.add_constraint(VAF([VAT(1, SAT(1.0, y))], [-h]), moi.Nonnegatives(1))
    else:
        cb = m.add_constraint(VAF([VAT(1, SAT(-1.0, y))], [h]), moi.Nonpositives(1))
    ceq = m.add_constraint(VAF([VAT(1, SAT(-1.0, t))], [1.0]), moi.Zeros(1))
    csoc = m.add_constraint(VAF([VAT(1, SAT(1.0, t)), VAT(2, SAT(1.0, x)), VAT(3, SAT(1.0, y))], [0.0, 0.0, 0.0]), moi.SecondOrderCone(3))
    m.set_objective_function(SAF([SAT(1.0, x)], 0.0))
    m.set_objective_sense(moi.MIN_SENSE)
    m.optimize()
    assert m.termination_status() == "OPTIMAL"
    assert close(m.objective_value(), -h) and close(m.dual_objective_value(), -h)
    assert close(m.variable_primal([x, y, t]), [-h, h, 1])
    assert close(m.constraint_primal(cb), [0]) and close(m.constraint_primal(ceq), [0]) and close(m.constraint_primal(csoc), [1, -h, h])
    assert close(m.constraint_dual(csoc), [np.sqrt(2.0), 1, -1])


@pytest.mark.parametrize("form", ["VectorOfVariables", "VectorAffineFunction"])
@pytest.mark.parametrize("backend", BACKENDS)
def test_conic_SecondOrderCone_INFEASIBLE(backend, form):
    """SOC3: -2 + y in R+, -1 + x in R-, (x, y) in SOC(2): infeasible.  With the cone on the variables the dual ray is found
    after ~1900 iterations; through the slack bridge (MathOptInterface's own form of this test: an affine function in the
    cone) after ~156 000 -- inside the reference's 5 s on a CPU that runs this 4-variable model at > 10^5 iterations per
    second, not inside 5 s of the NumPy oracle (6 000 it/s) or of a GPU launching ten kernels per iteration: the limit is
    60 s for that form."""
    m = make(backend, time_limit=5.0 if form == "VectorOfVariables" else 60.0)
    x, y = m.add_variables(2)
    m.add_constraint(VAF([VAT(1, SAT(1.0, y))], [-2.0]), moi.Nonnegatives(1))
    m.add_constraint(VAF([VAT(1, SAT(1.0, x))], [-1.0]), moi.Nonpositives(1))
    m.add_constraint(VOV([x, y]) if form == "VectorOfVariables" else identity_vaf([x, y]), moi.SecondOrderCone(2))
    m.optimize()
    assert_reports_infeasibility(m)


@pytest.mark.parametrize("backend", BACKENDS)
def test_conic_SecondOrderCone_out_of_order(backend):
    """SOC4: min -2 x2 - x3, x1 = 1, x2 - x4 = 0, x3 - x5 = 0, (x1, x4, x5) in SOC: out-of-order indices in the cone.
    Optimum -sqrt 5 at (1, 2/sqrt 5, 1/sqrt 5, 2/sqrt 5, 1/sqrt 5)."""
    m = make(backend)
    x = m.add_variables(5)
    c1 = m.add_constraint(VAF([VAT(1, SAT(1.0, x[0])), VAT(2, SAT(1.0, x[1])), VAT(3, SAT(1.0, x[2])),
                               VAT(2, SAT(-1.0, x[3])), VAT(3, SAT(-1.0, x[4]))], [-1.0, 0.0, 0.0]), moi.Zeros(3))
    c2 = m.add_constraint(VOV([x[0], x[3], x[4]]), moi.SecondOrderCone(3))
    m.set_objective_function(SAF([SAT(0.0, x[0]), SAT(-2.0, x[1]), SAT(-1.0, x[2])], 0.0))
    m.set_objective_sense(moi.MIN_SENSE)
    m.optimize()
    r5 = np.sqrt(5.0)
    assert m.termination_status() == "OPTIMAL"
    assert close(m.objective_value(), -r5) and close(m.dual_objective_value(), -r5)
    assert close(m.variable_primal(x), [1, 2 / r5, 1 / r5, 2 / r5, 1 / r5])
    assert close(m.constraint_primal(c1), [0, 0, 0]) and close(m.constraint_primal(c2), [1, 2 / r5, 1 / r5])
    assert close(m.constraint_dual(c1), [-r5, -2, -1]) and close(m.constraint_dual(c2), [r5, -2, -1])


# ------------------------------------------------------------------ rotated SOC
@pytest.mark.parametrize("form", ["VectorOfVariables", "VectorAffineFunction"])
@pytest.mark.parametrize("backend", BACKENDS)
def test_conic_RotatedSecondOrderCone(backend, form):
    """SOCRotated1: min -x - y, a = 1/2, b = 1, 2 a b >= x^2 + y^2 (through the RSOC-to-SOC bridge and the slack bridge, as the
    reference gets it from MathOptInterface).  Optimum -sqrt 2 at x = y = 1/sqrt 2."""
    m = make(backend)
    a, b, x, y = m.add_variables(4)
    m.add_constraint(VAF([VAT(1, SAT(1.0, a))], [-0.5]), moi.Zeros(1))
    m.add_constraint(VAF([VAT(1, SAT(1.0, b))], [-1.0]), moi.Zeros(1))
    c = m.add_constraint(VOV([a, b, x, y]) if form == "VectorOfVariables" else identity_vaf([a, b, x, y]), moi.RotatedSecondOrderCone(4))
    m.set_objective_function(SAF([SAT(0.0, a), SAT(0.0, b), SAT(-1.0, x), SAT(-1.0, y)], 0.0))
    m.set_objective_sense(moi.MIN_SENSE)
    m.optimize()
    r2 = np.sqrt(2.0)
    assert m.termination_status() == "OPTIMAL"
    assert close(m.objective_value(), -r2) and close(m.dual_objective_value(), -r2)
    assert close(m.variable_primal([a, b, x, y]), [0.5, 1.0, 1 / r2, 1 / r2])
    # the bridged constraint's value in the SOC's coordinates: ((a + b) / sqrt 2, (a - b) / sqrt 2, x, y), on the cone's boundary
    v = m.constraint_primal(c)
    assert close(v, [1.5 / r2, -0.5 / r2, 1 / r2, 1 / r2]) and abs(v[0] - np.linalg.norm(v[1:])) <= 1e-3


# ------------------------------------------------------------------ SDP0, SDP1
@pytest.mark.parametrize("form", ["VectorOfVariables", "VectorAffineFunction"])
@pytest.mark.parametrize("backend", BACKENDS)
def test_conic_PositiveSemidefiniteConeTriangle(backend, form):
    """SDP0: min X11 + X22, X21 = 1, X PSD.  Optimum 2 at X = ones; dual y = 2, dual matrix [1 -1; -1 1]."""
    m = make(backend)
    X = m.add_variables(3)
    cX = m.add_constraint(VOV(X) if form == "VectorOfVariables" else identity_vaf(X), moi.PositiveSemidefiniteConeTriangle(2))
    c = m.add_constraint(SAF([SAT(1.0, X[1])], 0.0), moi.EqualTo(1.0))
    m.set_objective_function(SAF([SAT(1.0, X[0]), SAT(1.0, X[2])], 0.0))
    m.set_objective_sense(moi.MIN_SENSE)
    m.optimize()
    assert m.termination_status() == "OPTIMAL"
    assert close(m.objective_value(), 2) and close(m.dual_objective_value(), 2)
    assert close(m.variable_primal(X), [1, 1, 1]) and close(m.constraint_primal(cX), [1, 1, 1]) and close(m.constraint_primal(c), 1)
    assert close(m.constraint_dual(c), 2) and close(m.constraint_dual(cX), [1, -1, 1])


def sdp1_optimum():
    """Closed form of SDP1's optimum, derived here (not taken from either solver).  By the symmetry of the data the optimal
    X = [a b c; b d b; c b a] and x = (x1, x2, x2); MathOptInterface's test states the solution through
    alpha = sqrt(3 - 2 sqrt 2) etc.; instead of trusting a remembered formula the value is computed by a dense parametric
    search over the two-dimensional dual: max y1 + y2 / 2 subject to C - y1 I - y2 J PSD and (1 - y1, -y2, -y2) in SOC,
    i.e. 1 - y1 >= sqrt 2 |y2|."""
    C = np.array([[2.0, 1, 0], [1, 2, 1], [0, 1, 2]])
    J = np.ones((3, 3))
    best = -np.inf
    lo1, hi1, lo2, hi2 = -2.0, 2.0, -2.0, 2.0
    for _ in range(6):                                   # successive grid refinement of a concave maximisation
        g1 = np.linspace(lo1, hi1, 81); g2 = np.linspace(lo2, hi2, 81)
        arg = None
        for y1 in g1:
            for y2 in g2:
                if 1 - y1 < np.sqrt(2.0) * abs(y2) - 1e-15:
                    continue
                if np.linalg.eigvalsh(C - y1 * np.eye(3) - y2 * J)[0] < -1e-13:
                    continue
                v = y1 + 0.5 * y2
                if v > best:
                    best, arg = v, (y1, y2)
        if arg is None:
            break
        w1, w2 = (hi1 - lo1) / 40, (hi2 - lo2) / 40
        lo1, hi1, lo2, hi2 = arg[0] - w1, arg[0] + w1, arg[1] - w2, arg[1] + w2
    return best


def test_sdp1_closed_form_search_is_consistent():
    """The dual search brackets the value MathOptInterface documents for SDP1 (0.705710509...)."""
    assert abs(sdp1_optimum() - 0.705710509) <= 2e-6


@pytest.mark.parametrize("backend", BACKENDS)
def test_conic_PositiveSemidefiniteConeTriangle_3(backend):
    """SDP1 (MOSEK's sdo1): min <C, X> + x1, tr X + x1 = 1, <J, X> + x2 + x3 = 1/2, X PSD (3 x 3), x in SOC(3);
    C = [2 1 0; 1 2 1; 0 1 2], J = ones.  Optimum 0.7057105..."""
    m = make(backend)
    X = m.add_variables(6)                    # X11, X21, X22, X31, X32, X33
    x = m.add_variables(3)
    cX = m.add_constraint(VOV(X), moi.PositiveSemidefiniteConeTriangle(3))
    cx = m.add_constraint(VOV(x), moi.SecondOrderCone(3))
    c1 = m.add_constraint(SAF([SAT(1.0, X[0]), SAT(1.0, X[2]), SAT(1.0, X[5]), SAT(1.0, x[0])], 0.0), moi.EqualTo(1.0))
    c2 = m.add_constraint(SAF([SAT(1.0, X[0]), SAT(2.0, X[1]), SAT(1.0, X[2]), SAT(2.0, X[3]), SAT(2.0, X[4]), SAT(1.0, X[5]),
                               SAT(1.0, x[1]), SAT(1.0, x[2])], 0.0), moi.EqualTo(0.5))
    m.set_objective_function(SAF([SAT(2.0, X[0]), SAT(2.0, X[1]), SAT(2.0, X[2]), SAT(2.0, X[4]), SAT(2.0, X[5]), SAT(1.0, x[0])], 0.0))
    m.set_objective_sense(moi.MIN_SENSE)
    m.optimize()
    opt = 0.705710509
    assert m.termination_status() == "OPTIMAL"
    assert close(m.objective_value(), opt) and close(m.dual_objective_value(), opt)
    Xv, xv = m.variable_primal(X), m.variable_primal(x)
    Xs = np.array([[Xv[0], Xv[1], Xv[3]], [Xv[1], Xv[2], Xv[4]], [Xv[3], Xv[4], Xv[5]]])
    assert np.linalg.eigvalsh(Xs)[0] >= -ATOL and xv[0] >= np.hypot(xv[1], xv[2]) - ATOL
    assert close(np.trace(Xs) + xv[0], 1) and close(Xs.sum() + xv[1] + xv[2], 0.5)
    assert close(m.constraint_primal(c1), 1) and close(m.constraint_primal(c2), 0.5)
    # the symmetry of the data shows in the solution
    assert close(Xs[0, 0], Xs[2, 2]) and close(Xs[0, 1], Xs[1, 2]) and close(xv[1], xv[2])
    # dual feasibility: C - y1 I - y2 J PSD with the multipliers of the two rows
    y1, y2 = m.constraint_dual(c1), m.constraint_dual(c2)
    C = np.array([[2.0, 1, 0], [1, 2, 1], [0, 1, 2]])
    assert np.linalg.eigvalsh(C - y1 * np.eye(3) - y2 * np.ones((3, 3)))[0] >= -1e-3
    assert close(y1 + 0.5 * y2, opt)
    assert len(m.constraint_dual(cX)) == 6 and len(m.constraint_dual(cx)) == 3
