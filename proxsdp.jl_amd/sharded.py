"""Block-sharded solve of a block-diagonal model: one process per GPU, GPU g owns the PSD
blocks assigned to it and the constraint rows that touch only them (DESIGN.md section 7,
SURVEY.md section 8e).  The reference processes the blocks of a model serially on one core
(/root/reference/src/prox_operators.jl:40); here every shard runs the full PDHG loop on
its own blocks and the shards exchange only scalars (linesearch norms, residual maxima,
objective sums, convergence flags, one clock) through two small all-reduces per
iteration -- `torch.distributed` with backend "nccl" (= RCCL over xGMI) on the GPU box,
"gloo" in the CPU-side tests.  All shards therefore take identical control-flow decisions
and the iterates are those of the single-process solve of the whole model.
"""
import numpy as np
import scipy.sparse as sp

from . import problems
from .optimizer import Optimizer


def split_block_diagonal(prob, owners, rank):
    """Sub-problem of `rank`: the PSD blocks with owners[k] == rank, and the rows of A and G
    whose entries all lie in those blocks' variables.  Raises if a row couples two shards or
    a variable belongs to no PSD block (free / SOC variables are not sharded)."""
    if prob.soc:
        raise ValueError("SOC cones are not supported by the block-sharded path")
    n = prob.n
    var_owner = np.full(n, -1, dtype=np.int64)
    for k, idx in enumerate(prob.psd):
        var_owner[idx] = owners[k]
    if np.any(var_owner < 0):
        raise ValueError("variables outside PSD blocks are not supported by the block-sharded path")
    mine = np.nonzero(var_owner == rank)[0]
    remap = np.full(n, -1, dtype=np.int64)
    remap[mine] = np.arange(len(mine))

    def rows_of(M):
        M = sp.csr_matrix(M)
        if M.shape[0] == 0:
            return np.zeros(0, dtype=np.int64), sp.csc_matrix((0, len(mine)))
        row_owner_min = np.full(M.shape[0], np.iinfo(np.int64).max)
        row_owner_max = np.full(M.shape[0], -1)
        coo = M.tocoo()
        np.minimum.at(row_owner_min, coo.row, var_owner[coo.col])
        np.maximum.at(row_owner_max, coo.row, var_owner[coo.col])
        if np.any((row_owner_max >= 0) & (row_owner_min != row_owner_max)):
            raise ValueError("a constraint row couples blocks of different shards")
        sel = np.nonzero(row_owner_max == rank)[0]
        return sel, sp.csc_matrix(M[sel][:, mine])

    ra, A = rows_of(prob.A)
    rg, G = rows_of(prob.G)
    psd = [remap[idx] for k, idx in enumerate(prob.psd) if owners[k] == rank]
    sub = problems.Problem(n=len(mine), A=A, b=np.asarray(prob.b)[ra], G=G, h=np.asarray(prob.h)[rg],
                           c=np.asarray(prob.c)[mine], psd=psd, max_sense=prob.max_sense,
                           objective_constant=prob.objective_constant, name=f"{prob.name}[shard {rank}]")
    return sub, dict(vars=mine, rows_eq=ra, rows_in=rg)


def make_reduce(dist, device=None):
    """reduce(sums, maxs) over the process group: two all-reduces (SUM, MAX)."""
    import torch

    def reduce(sums, maxs):
        if len(sums):
            t = torch.from_numpy(np.array(sums, dtype=np.float64))
            t = t.to(device) if device is not None else t
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            sums[:] = t.cpu().numpy()
        if len(maxs):
            t = torch.from_numpy(np.array(maxs, dtype=np.float64))
            t = t.to(device) if device is not None else t
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            maxs[:] = t.cpu().numpy()
    return reduce


def solve_sharded(prob, dist, rank, world, device_id=0, owners=None, collective_device=None, **options):
    """Every rank calls this with the SAME full model; returns (Optimizer, SolveResult of the
    local shard, index maps).  Objective / gap / status are global and identical on all ranks."""
    from . import replicas
    owners = owners if owners is not None else replicas.assign_blocks(len(prob.psd), world)
    sub, maps = split_block_diagonal(prob, owners, rank)
    opt = Optimizer(device_id=device_id, **options)
    sol = opt.optimize(sub, reduce=make_reduce(dist, collective_device),
                       trace_capacity=int(options.get("max_iter", 0)) if options.get("max_iter", 0) else 0)
    return opt, sol, maps
