"""Host-side mirror of the reference's `ProxSDP.Optimizer` surface
(/root/reference/src/MOI_wrapper.jl) for the one path this build replaces.

MathOptInterface itself is Julia and is not re-implemented: a `Problem`
(problems.py) stands for the `OptimizerCache` that `MOI.copy_to` fills, and
`optimize()` is `_optimize!` (:220-342) with `chambolle_pock` (:310) replaced by
the C-ABI call.  Names, argument meaning and error behaviour follow the
reference so the parity tests read like its own tests:

    Optimizer(**kwargs)                  :69-81   unknown keyword -> error
    set_attribute / get_attribute        :84-103  RawOptimizerAttribute
    set_silent / silent                  :105-123 MOI.Silent
    set_time_limit_sec / time_limit_sec  :125-139 MOI.TimeLimitSec (None <-> 3600_00.0)
    termination_status ... dual_status   :377-441
    objective_value / dual_objective_value / solve_time_sec / pdhg_iterations
    variable_primal / constraint_* getters :451-530
"""
from . import binding

TERMINATION = {0: "OPTIMIZE_NOT_CALLED", 1: "OPTIMAL", 2: "TIME_LIMIT", 3: "ITERATION_LIMIT",
               4: "INFEASIBLE_OR_UNBOUNDED", 5: "DUAL_INFEASIBLE", 6: "INFEASIBLE"}


class Optimizer:
    SOLVER_NAME = "ProxSDP"            # MOI.SolverName, :77
    SOLVER_VERSION = "1.8.4"           # MOI.SolverVersion, :79 (the reference version mirrored)

    def __init__(self, **kwargs):
        self.options = binding.default_options()
        self.sol = None
        self.problem = None
        for k, v in kwargs.items():
            self.set_attribute(k, v)

    # -- RawOptimizerAttribute
    def set_attribute(self, name, value):
        binding.set_option(self.options, name, float(value))
        return value

    def get_attribute(self, name):
        return binding.get_option(self.options, name)

    # -- MOI.Silent
    def set_silent(self, value):
        if value:
            self.options.timer_verbose = 0
        self.options.log_verbose = 0 if value else 1

    def silent(self):
        return not (self.options.log_verbose or self.options.timer_verbose)

    # -- MOI.TimeLimitSec
    def set_time_limit_sec(self, value):
        self.options.time_limit = 360000.0 if value is None else float(value)

    def time_limit_sec(self):
        v = self.options.time_limit
        return None if v == 360000.0 else v

    def is_empty(self):
        return self.problem is None and self.sol is None

    def empty(self):
        self.problem, self.sol = None, None

    # -- _optimize!
    def optimize(self, problem, eig_resid=None, trace_capacity=0, reduce=None, coupling=None, index_base=0,
                 nccl_comm=None, resume=None, capture_iteration=None):
        self.empty()
        self.problem = problem
        sol = binding.solve(problem, self.options, eig_resid=eig_resid, trace_capacity=trace_capacity,
                            reduce=reduce, coupling=coupling, index_base=index_base, nccl_comm=nccl_comm,
                            resume=resume, capture_iteration=capture_iteration)
        sign = -1.0 if problem.max_sense else 1.0          # :336-337
        sol.objval = sign * sol.objval + problem.objective_constant
        sol.dual_objval = sign * sol.dual_objval + problem.objective_constant
        self.sol = sol
        return sol

    # -- attributes set by optimize
    def termination_status(self):
        return TERMINATION[0 if self.sol is None else self.sol.status]

    def raw_status_string(self):
        return "Problem not solved" if self.sol is None else self.sol.status_string

    def primal_status(self, result_index=1):
        s = 0 if self.sol is None else self.sol.status
        if result_index > 1 or s == 0:
            return "NO_SOLUTION"
        if s == 5 and self.sol.certificate_found:
            return "INFEASIBILITY_CERTIFICATE"
        return "FEASIBLE_POINT" if self.sol.primal_feasible_user_tol else "INFEASIBLE_POINT"

    def dual_status(self, result_index=1):
        s = 0 if self.sol is None else self.sol.status
        if result_index > 1 or s == 0:
            return "NO_SOLUTION"
        if s == 6 and self.sol.certificate_found:
            return "INFEASIBILITY_CERTIFICATE"
        return "FEASIBLE_POINT" if self.sol.dual_feasible_user_tol else "INFEASIBLE_POINT"

    def result_count(self):
        return 0 if self.sol is None else self.sol.result_count

    def objective_value(self):
        return self.sol.objval

    def dual_objective_value(self):
        return self.sol.dual_objval

    def solve_time_sec(self):
        return self.sol.time

    def pdhg_iterations(self):
        return int(self.sol.iter)

    def variable_primal(self, idx=None):
        return self.sol.primal if idx is None else self.sol.primal[idx]

    def constraint_primal_psd(self, k):
        return self.sol.primal[self.problem.psd[k]]

    def constraint_dual_psd(self, k):
        return self.sol.dual_cone[self.problem.psd[k]]

    def constraint_primal_zeros(self):
        return self.sol.slack_eq

    def constraint_primal_nonpositives(self):
        return self.sol.slack_in

    def constraint_dual_zeros(self):
        return -self.sol.dual_eq            # :495-503

    def constraint_dual_nonpositives(self):
        return -self.sol.dual_in            # :505-513
