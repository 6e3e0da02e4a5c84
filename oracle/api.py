"""TEST INFRASTRUCTURE ONLY -- glue between a standard-form problem (any object
with attributes n, A, b, G, h, c, psd, soc, max_sense, objective_constant) and
the oracle's chambolle_pock; applies the objective fix-up of
/root/reference/src/MOI_wrapper.jl:336-337."""
import numpy as np
import scipy.sparse as sp

from . import pdhg
from .options import Options


def to_standard_form(prob):
    """ConicSets/AffineSets as built at MOI_wrapper.jl:267-288."""
    aff = pdhg.AffineSets(prob.n, prob.A.shape[0], prob.G.shape[0],
                          sp.csc_matrix(prob.A), sp.csc_matrix(prob.G),
                          np.asarray(prob.b, float), np.asarray(prob.h, float),
                          np.asarray(prob.c, float))
    cones = pdhg.ConicSets()
    for idx in prob.soc:
        cones.socone.append(pdhg.SOCSet(np.asarray(idx, np.int64), len(idx)))
    for idx in prob.psd:
        L = len(idx)
        side = int((np.sqrt(8 * L + 1) - 1) // 2)
        cones.sdpcone.append(pdhg.SDPSet(np.asarray(idx, np.int64), L, side))
    return aff, cones


def solve(prob, opt=None, **kw):
    """optimize!: returns the Result with objval/dual_objval in user sense."""
    opt = opt or Options()
    aff, cones = to_standard_form(prob)
    res = pdhg.chambolle_pock(aff, cones, opt, **kw)
    sign = -1.0 if getattr(prob, "max_sense", False) else 1.0
    const = getattr(prob, "objective_constant", 0.0)
    res.objval = sign * res.objval + const
    res.dual_objval = sign * res.dual_objval + const
    return res
