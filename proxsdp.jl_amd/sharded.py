"""Block-sharded solve of a block-diagonal model: one process per GPU, GPU g owns the PSD
blocks assigned to it and the constraint rows that touch only them (DESIGN.md section 8,
SURVEY.md section 8e).  The reference processes the blocks of a model serially on one core
(/root/reference/src/prox_operators.jl:40); here every shard runs the full PDHG loop on
its own blocks and the shards exchange only scalars (linesearch norms, residual maxima,
objective sums, convergence flags, one clock) through one small collective per
iteration, plus -- when the model has rows that couple blocks of different shards -- one
all-reduce of the coupling rows of M x -- `torch.distributed` with backend "nccl" (= RCCL over xGMI) on the GPU box,
"gloo" in the CPU-side tests.  All shards therefore take identical control-flow decisions
and the iterates are those of the single-process solve of the whole model.
"""
import numpy as np
import scipy.sparse as sp

from . import problems
from .optimizer import Optimizer


def split_block_diagonal(prob, owners, rank, allow_coupling=True):
    """Sub-problem of `rank`: the PSD blocks with owners[k] == rank, the rows of A and G whose entries
    all lie in those blocks' variables (PRIVATE rows), and -- SURVEY.md section 8e -- every COUPLING
    row (entries in the variables of more than one shard): each shard carries all coupling rows,
    restricted to its own columns (possibly empty), with the same right-hand side; the partial
    products are summed over the shards once per iteration (proxsdp_problem.coupling_rows).
    Raises if a variable belongs to no PSD block (free / SOC variables are not sharded).
    Returns (sub-problem, maps); maps["coupling"] = dict(rows, owned) in the shard's row numbering
    (equalities first, then inequalities), or None."""
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
            z = np.zeros(0, dtype=np.int64)
            return z, sp.csc_matrix((0, len(mine))), z, np.zeros(0, dtype=np.int32)
        row_owner_min = np.full(M.shape[0], np.iinfo(np.int64).max)
        row_owner_max = np.full(M.shape[0], -1)
        coo = M.tocoo()
        np.minimum.at(row_owner_min, coo.row, var_owner[coo.col])
        np.maximum.at(row_owner_max, coo.row, var_owner[coo.col])
        coupled = (row_owner_max >= 0) & (row_owner_min != row_owner_max)
        if np.any(coupled) and not allow_coupling:
            raise ValueError("a constraint row couples blocks of different shards")
        sel = np.nonzero(((row_owner_max == rank) & ~coupled) | coupled)[0]       # original order kept
        local = np.nonzero(coupled[sel])[0]                                      # positions inside `sel`
        owned = (row_owner_min[sel][local] == rank).astype(np.int32)             # the lowest shard of a row owns it
        return sel, sp.csc_matrix(M[sel][:, mine]), local.astype(np.int64), owned

    ra, A, ca, oa = rows_of(prob.A)
    rg, G, cg, og = rows_of(prob.G)
    psd = [remap[idx] for k, idx in enumerate(prob.psd) if owners[k] == rank]
    sub = problems.Problem(n=len(mine), A=A, b=np.asarray(prob.b)[ra], G=G, h=np.asarray(prob.h)[rg],
                           c=np.asarray(prob.c)[mine], psd=psd, max_sense=prob.max_sense,
                           objective_constant=prob.objective_constant, name=f"{prob.name}[shard {rank}]")
    coupling = None
    if len(ca) + len(cg) > 0:
        coupling = dict(rows=np.concatenate([ca, len(ra) + cg]).astype(np.int64),
                        owned=np.concatenate([oa, og]).astype(np.int32))
    return sub, dict(vars=mine, rows_eq=ra, rows_in=rg, coupling=coupling)


class _DevPtr:
    """a raw device pointer as something torch.as_tensor understands (__cuda_array_interface__)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def make_reduce(dist, device=None, world=None):
    """reduce(sums, maxs) over the process group.  ONE collective per call: the packed record
    [sums | maxs] of every rank is all-gathered into a preallocated tensor (on `device` for RCCL, host
    memory for gloo) and combined on the host in rank order -- the same bits on every rank."""
    import torch
    world = world or dist.get_world_size()
    state = {}

    def reduce(sums, maxs):
        ns, nm = len(sums), len(maxs)
        if ns + nm == 0:
            return
        key = (ns, nm)
        if key not in state:
            state[key] = (torch.empty(ns + nm, dtype=torch.float64, device=device or "cpu"),
                          torch.empty(world * (ns + nm), dtype=torch.float64, device=device or "cpu"))
        mine, allr = state[key]
        mine.copy_(torch.from_numpy(np.concatenate([sums, maxs])))
        dist.all_gather_into_tensor(allr, mine)
        rec = allr.cpu().numpy().reshape(world, ns + nm)
        if ns:
            acc = rec[0, :ns].copy()
            for r in range(1, world):
                acc += rec[r, :ns]
            sums[:] = acc
        if nm:
            maxs[:] = rec[:, ns:].max(axis=0)
    return reduce


def make_reduce_vec(dist, device=None):
    """reduce_vec(ptr, length, on_device): element-wise SUM over the process group, in place.  On the
    GPU path `ptr` is a device pointer of the library (a different HIP runtime object than torch's,
    same process, same GPU VM): it is wrapped without a copy and handed to RCCL."""
    import ctypes
    import torch

    def reduce_vec(ptr, length, on_device):
        if on_device:
            t = torch.as_tensor(_DevPtr(ptr, length), device=device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize(device)
        else:
            buf = (ctypes.c_double * length).from_address(ptr)
            t = torch.from_numpy(np.ctypeslib.as_array(buf))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return reduce_vec


def make_native_comm(dist, rank, world, device_id=0):
    """RCCL communicator for the library's NATIVE collectives (proxsdp_problem.nccl_comm): rank 0 draws the
    unique id through the library's own librccl, the 128 bytes travel over the existing process group (any
    backend), every rank joins on the GPU it owns.  torch.distributed's communicator is not reachable from
    outside torch, and a communicator must belong to the librccl that issues the calls."""
    from . import binding
    box = [binding.rccl_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    return binding.rccl_comm_init(world, box[0], rank, device_id)


def solve_sharded(prob, dist, rank, world, device_id=0, owners=None, collective_device=None, native_comm=None,
                  **options):
    """Every rank calls this with the SAME full model; returns (Optimizer, SolveResult of the
    local shard, index maps).  Objective / gap / status are global and identical on all ranks.
    native_comm: handle from make_native_comm -> the library reduces scalars and coupling rows itself over RCCL
    on its own stream (no Python callback per iteration); otherwise the torch.distributed callbacks below
    (the path the gloo tests use).
    A failure on one rank (bad model, more ranks than blocks, an error inside the library) is
    all-reduced before anybody enters the solve loop's collectives, so all ranks raise together
    instead of leaving the others blocked in an all-reduce."""
    import torch
    from . import replicas
    err = None
    sub = maps = None
    try:
        if world > len(prob.psd):
            raise ValueError(f"{world} ranks for {len(prob.psd)} PSD blocks: every rank needs at least one block")
        owners = owners if owners is not None else replicas.assign_blocks(len(prob.psd), world)
        sub, maps = split_block_diagonal(prob, owners, rank)
    except Exception as e:                                   # noqa: BLE001 -- re-raised below on every rank
        err = e
    flag = torch.tensor([1.0 if err is not None else 0.0], dtype=torch.float64,
                        device=collective_device or "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if float(flag.item()) > 0:
        raise err if err is not None else RuntimeError("another rank failed to build its shard")
    opt = Optimizer(device_id=device_id, **options)
    coupling = None
    tcap = int(options.get("max_iter", 0)) if options.get("max_iter", 0) else 0
    if native_comm:
        if maps["coupling"] is not None:
            coupling = dict(maps["coupling"])
        sol = opt.optimize(sub, coupling=coupling, nccl_comm=native_comm, trace_capacity=tcap)
        return opt, sol, maps
    if maps["coupling"] is not None:
        coupling = dict(maps["coupling"], reduce_vec=make_reduce_vec(dist, collective_device),
                        on_device=collective_device is not None)
    sol = opt.optimize(sub, reduce=make_reduce(dist, collective_device, world), coupling=coupling, trace_capacity=tcap)
    return opt, sol, maps
