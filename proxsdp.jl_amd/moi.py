"""A small model layer with MathOptInterface's vocabulary, so that the reference's own MOI tests
(/root/reference/test/moi_proxsdp_unit.jl, moi_sensorloc.jl, moi_mimo.jl, test_terminationstatus.jl)
can be restated call by call, and so that the assembly `_optimize!` performs
(/root/reference/src/MOI_wrapper.jl:220-293) and the getters (:362-530) exist on this side of the
C ABI as well.  MathOptInterface itself is Julia and is NOT re-implemented: only what those tests and
the `Optimizer` surface use.

What the reference's optimizer supports natively (MOI_wrapper.jl:143-176) and what MOI's bridges turn
the rest into (with_bridge_type = Float64, as the tests instantiate it):

    VectorAffineFunction-in-Zeros          native: rows of A, b = -constants            (:233-240)
    VectorAffineFunction-in-Nonpositives   native: rows of G, h = -constants            (:241-247)
    VectorOfVariables-in-PositiveSemidefiniteConeTriangle / SecondOrderCone   native    (:275-288)
    ScalarAffineFunction-in-EqualTo(v)     VectorizeBridge:  f - v in Zeros(1)
    ScalarAffineFunction-in-LessThan(u)    VectorizeBridge:  f - u in Nonpositives(1)
    ScalarAffineFunction-in-GreaterThan(l) Vectorize + NonnegToNonpos:  -f + l in Nonpositives(1)
    VectorAffineFunction-in-Nonnegatives   NonnegToNonpos:  -f in Nonpositives
    VectorOfVariables-in-{Zeros, Nonpositives, Nonnegatives}   VectorFunctionize: the identity as an affine function
    VectorAffineFunction-in-{SecondOrderCone, PositiveSemidefiniteConeTriangle}   VectorSlack: y in the cone, f(x) - y in Zeros
    {VectorOfVariables, VectorAffineFunction}-in-RotatedSecondOrderCone   RSOCtoSOC: ((t + u) / sqrt 2, (t - u) / sqrt 2, x) in SecondOrderCone
    VariableIndex-in-{EqualTo, LessThan, GreaterThan}   as the scalar affine function 1.0 x

Variable indices are 1-based integers (`VariableIndex.value`).  Rows keep the order in which the
constraints were added, per native set, as `MOI.Utilities.MatrixOfConstraints` does.  A variable that
appears twice in one cone (moi_proxsdp_unit.jl `double_sdp_with_duplicates`) gets what MOI's bridges give it:
fresh slack variables in the cone and equality rows tying them to the user's variable.
"""
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp

from .optimizer import Optimizer
from .problems import Problem

MIN_SENSE, MAX_SENSE, FEASIBILITY_SENSE = "MIN_SENSE", "MAX_SENSE", "FEASIBILITY_SENSE"


# ---------------------------------------------------------------- functions
@dataclass
class ScalarAffineTerm:
    coefficient: float
    variable: int


@dataclass
class ScalarAffineFunction:
    terms: list
    constant: float = 0.0


@dataclass
class VectorAffineTerm:
    output_index: int                      # 1-based row of the vector function
    scalar_term: ScalarAffineTerm


@dataclass
class VectorAffineFunction:
    terms: list
    constants: list


@dataclass
class VectorOfVariables:
    variables: list


# ---------------------------------------------------------------- sets
@dataclass
class EqualTo:
    value: float


@dataclass
class LessThan:
    upper: float


@dataclass
class GreaterThan:
    lower: float


@dataclass
class Zeros:
    dimension: int


@dataclass
class Nonpositives:
    dimension: int


@dataclass
class Nonnegatives:
    dimension: int


@dataclass
class PositiveSemidefiniteConeTriangle:
    side_dimension: int


@dataclass
class SecondOrderCone:
    dimension: int


@dataclass
class RotatedSecondOrderCone:
    dimension: int                 # (t, u, x): 2 t u >= |x|^2, t, u >= 0


@dataclass(frozen=True)
class ConstraintIndex:
    kind: str        # "zeros" | "nonpos" | "psd" | "soc"  (the native set the rows live in)
    value: int       # position among the constraints of that kind (1-based, as ci.value)
    rows: tuple = ()         # 0-based rows in A / G
    flip: float = 1.0        # -1 for a bridged GreaterThan / Nonnegatives constraint
    shift: float = 0.0       # set constant a scalar bridge moved into the function
    scalar: bool = False


def sympackedlen(n):
    """src/MOI_wrapper.jl:218."""
    return n * (n + 1) // 2


def ivech(X):
    """Variables of a packed upper triangle (column by column, MOI's triangle order) as a full
    symmetric matrix of variable indices -- what the reference's tests build with `ProxSDP.ivech!`
    followed by `Symmetric(Xsq, :U)` (src/util.jl, test/moi_sensorloc.jl:14-17)."""
    L = len(X)
    n = int((np.sqrt(8 * L + 1) - 1) // 2)
    assert sympackedlen(n) == L
    M = np.zeros((n, n), dtype=np.int64)
    k = 0
    for j in range(n):
        for i in range(j + 1):
            M[i, j] = M[j, i] = X[k]
            k += 1
    return M


class Model:
    """`MOI.instantiate(() -> ProxSDP.Optimizer(...), with_bridge_type = Float64)`: a cache of the
    model plus the optimizer it is copied to by `optimize()`."""

    def __init__(self, optimizer=None, **options):
        self.optimizer = optimizer if optimizer is not None else Optimizer(**options)
        self.empty()

    # -- model
    def empty(self):
        self.nvar = 0
        self._zeros = []           # rows: (dict var -> coef, constant)
        self._nonpos = []
        self._psd = []
        self._soc = []
        self.sense = FEASIBILITY_SENSE
        self.objective = ScalarAffineFunction([], 0.0)
        self.optimizer.empty()

    def is_empty(self):
        return self.nvar == 0 and not (self._zeros or self._nonpos or self._psd or self._soc) and self.optimizer.is_empty()

    def add_variable(self):
        self.nvar += 1
        return self.nvar

    def add_variables(self, n):
        return [self.add_variable() for _ in range(n)]

    def _row(self, terms, sign=1.0):
        r = {}
        for t in terms:
            if not (1 <= t.variable <= self.nvar):
                raise ValueError(f"invalid variable index {t.variable}")
            r[t.variable] = r.get(t.variable, 0.0) + sign * float(t.coefficient)
        return r

    def add_constraint(self, f, s):
        if isinstance(f, (int, np.integer)):                       # VariableIndex
            f = ScalarAffineFunction([ScalarAffineTerm(1.0, int(f))], 0.0)
        if isinstance(s, RotatedSecondOrderCone):
            # RSOCtoSOCBridge: (t, u, x) in RSOC  <=>  ((t + u) / sqrt 2, (t - u) / sqrt 2, x) in SOC (the map is its own inverse);
            # the constraint index returned is the SOC constraint's: primal / dual values come back in ITS coordinates
            if isinstance(f, VectorOfVariables):
                f = VectorAffineFunction([VectorAffineTerm(k + 1, ScalarAffineTerm(1.0, int(x))) for k, x in enumerate(f.variables)],
                                         [0.0] * len(f.variables))
            d = len(f.constants)
            if d != s.dimension or d < 2:
                raise ValueError("dimension mismatch")
            r = 1.0 / np.sqrt(2.0)
            terms = []
            for t in f.terms:
                c, v = float(t.scalar_term.coefficient), t.scalar_term.variable
                if t.output_index == 1:
                    terms += [VectorAffineTerm(1, ScalarAffineTerm(r * c, v)), VectorAffineTerm(2, ScalarAffineTerm(r * c, v))]
                elif t.output_index == 2:
                    terms += [VectorAffineTerm(1, ScalarAffineTerm(r * c, v)), VectorAffineTerm(2, ScalarAffineTerm(-r * c, v))]
                else:
                    terms.append(t)
            k = [float(x) for x in f.constants]
            consts = [r * (k[0] + k[1]), r * (k[0] - k[1])] + k[2:]
            return self.add_constraint(VectorAffineFunction(terms, consts), SecondOrderCone(d))
        if isinstance(f, ScalarAffineFunction):
            if isinstance(s, EqualTo):
                self._zeros.append((self._row(f.terms), f.constant - s.value))
                return ConstraintIndex("zeros", len(self._zeros), (len(self._zeros) - 1,), 1.0, s.value, True)
            if isinstance(s, LessThan):
                self._nonpos.append((self._row(f.terms), f.constant - s.upper))
                return ConstraintIndex("nonpos", len(self._nonpos), (len(self._nonpos) - 1,), 1.0, s.upper, True)
            if isinstance(s, GreaterThan):
                self._nonpos.append((self._row(f.terms, -1.0), -(f.constant - s.lower)))
                return ConstraintIndex("nonpos", len(self._nonpos), (len(self._nonpos) - 1,), -1.0, s.lower, True)
            raise TypeError(f"unsupported scalar set {type(s).__name__}")
        if isinstance(f, VectorAffineFunction):
            d = len(f.constants)
            if isinstance(s, (Zeros, Nonpositives, Nonnegatives)) and s.dimension != d:
                raise ValueError("dimension mismatch")
            rows = [[] for _ in range(d)]
            for t in f.terms:
                rows[t.output_index - 1].append(t.scalar_term)
            if isinstance(s, Zeros):
                first = len(self._zeros)
                for k in range(d):
                    self._zeros.append((self._row(rows[k]), float(f.constants[k])))
                return ConstraintIndex("zeros", first + 1, tuple(range(first, first + d)))
            if isinstance(s, (Nonpositives, Nonnegatives)):
                sign = 1.0 if isinstance(s, Nonpositives) else -1.0
                first = len(self._nonpos)
                for k in range(d):
                    self._nonpos.append((self._row(rows[k], sign), sign * float(f.constants[k])))
                return ConstraintIndex("nonpos", first + 1, tuple(range(first, first + d)), sign)
            if isinstance(s, (PositiveSemidefiniteConeTriangle, SecondOrderCone)):
                # VectorSlackBridge: fresh variables y in the cone and the rows f(x) - y = 0 in Zeros
                y = self.add_variables(d)
                first = len(self._zeros)
                for k in range(d):
                    r = self._row(rows[k])
                    r[y[k]] = r.get(y[k], 0.0) - 1.0
                    self._zeros.append((r, float(f.constants[k])))
                self._slack_rows = getattr(self, "_slack_rows", []) + list(range(first, first + d))
                return self.add_constraint(VectorOfVariables(y), s)
            raise TypeError(f"unsupported vector set {type(s).__name__}")
        if isinstance(f, VectorOfVariables):
            v = [int(x) for x in f.variables]
            if isinstance(s, (Zeros, Nonpositives, Nonnegatives)):
                # VectorFunctionizeBridge: the identity as a VectorAffineFunction
                return self.add_constraint(VectorAffineFunction([VectorAffineTerm(k + 1, ScalarAffineTerm(1.0, x)) for k, x in enumerate(v)],
                                                                [0.0] * len(v)), s)
            if len(set(v)) != len(v):
                # a variable more than once in one cone (moi_proxsdp_unit.jl double_sdp_with_duplicates): what MOI's bridges do --
                # VectorFunctionize + VectorSlack: fresh variables y in the cone and the rows f(x) - y = 0 in Zeros
                y = self.add_variables(len(v))
                first = len(self._zeros)
                for xi, yi in zip(v, y):
                    self._zeros.append(({xi: 1.0, yi: -1.0}, 0.0))
                self._slack_rows = getattr(self, "_slack_rows", []) + list(range(first, first + len(v)))
                v = y
            if isinstance(s, PositiveSemidefiniteConeTriangle):
                if len(v) != sympackedlen(s.side_dimension):
                    raise ValueError("dimension mismatch")
                self._psd.append(v)
                return ConstraintIndex("psd", len(self._psd))
            if isinstance(s, SecondOrderCone):
                if len(v) != s.dimension:
                    raise ValueError("dimension mismatch")
                self._soc.append(v)
                return ConstraintIndex("soc", len(self._soc))
            raise TypeError(f"unsupported cone {type(s).__name__}")
        raise TypeError(f"unsupported function {type(f).__name__}")

    def set_objective_sense(self, sense):
        assert sense in (MIN_SENSE, MAX_SENSE, FEASIBILITY_SENSE)
        self.sense = sense

    def set_objective_function(self, f):
        assert isinstance(f, ScalarAffineFunction)
        self.objective = f

    # -- _optimize!  (src/MOI_wrapper.jl:220-293)
    def problem(self, name="moi"):
        n = self.nvar

        def mat(rows):
            r, c, v = [], [], []
            for i, (row, _) in enumerate(rows):
                for var, coef in sorted(row.items()):
                    r.append(i); c.append(var - 1); v.append(coef)
            return sp.csc_matrix((v, (r, c)), shape=(len(rows), n))

        A, G = mat(self._zeros), mat(self._nonpos)
        b = -np.array([k for _, k in self._zeros], dtype=float)                  # :239
        h = -np.array([k for _, k in self._nonpos], dtype=float)                 # :246
        max_sense = self.sense == MAX_SENSE
        c = np.zeros(n)
        if self.sense != FEASIBILITY_SENSE:
            for t in self.objective.terms:                                       # :254-256
                c[t.variable - 1] += (-1.0 if max_sense else 1.0) * float(t.coefficient)
        return Problem(n=n, A=A, b=b, G=G, h=h, c=c,
                       psd=[np.array(v, dtype=np.int64) - 1 for v in self._psd],
                       soc=[np.array(v, dtype=np.int64) - 1 for v in self._soc],
                       max_sense=max_sense,
                       objective_constant=float(self.objective.constant) if self.sense != FEASIBILITY_SENSE else 0.0,
                       name=name)

    def optimize(self, **kw):
        return self.optimizer.optimize(self.problem(), **kw)

    # -- attributes (src/MOI_wrapper.jl:84-139)
    def set(self, name, value):
        return self.optimizer.set_attribute(name, value)

    def get(self, name):
        return self.optimizer.get_attribute(name)

    # -- results (src/MOI_wrapper.jl:362-530 behind the bridges' own maps)
    def termination_status(self):
        return self.optimizer.termination_status()

    def primal_status(self, result_index=1):
        return self.optimizer.primal_status(result_index)

    def dual_status(self, result_index=1):
        return self.optimizer.dual_status(result_index)

    def objective_value(self):
        return self.optimizer.objective_value()

    def dual_objective_value(self):
        return self.optimizer.dual_objective_value()

    def solve_time_sec(self):
        return self.optimizer.solve_time_sec()

    def variable_primal(self, vi):
        x = self.optimizer.sol.primal
        if isinstance(vi, (int, np.integer)):
            return x[int(vi) - 1]
        return x[np.asarray(vi, dtype=np.int64) - 1]

    def constraint_primal(self, ci):
        sol = self.optimizer.sol
        if ci.kind == "psd":
            return sol.primal[np.array(self._psd[ci.value - 1]) - 1]
        if ci.kind == "soc":
            return sol.primal[np.array(self._soc[ci.value - 1]) - 1]
        slack = sol.slack_eq if ci.kind == "zeros" else sol.slack_in        # = native function value (pdhg.jl:757-758)
        v = ci.flip * slack[list(ci.rows)] + ci.shift
        return float(v[0]) if ci.scalar else v

    def constraint_dual(self, ci):
        sol = self.optimizer.sol
        if ci.kind == "psd":
            return sol.dual_cone[np.array(self._psd[ci.value - 1]) - 1]
        if ci.kind == "soc":
            return sol.dual_cone[np.array(self._soc[ci.value - 1]) - 1]
        d = -(sol.dual_eq if ci.kind == "zeros" else sol.dual_in)[list(ci.rows)]   # :495-513
        d = ci.flip * d
        return float(d[0]) if ci.scalar else d
