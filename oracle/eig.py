"""TEST INFRASTRUCTURE ONLY -- CPU oracle, never imported by the product path.

Spectral kernels of the reference, restated on NumPy/SciPy
(/root/reference/src/eigsolver.jl, /root/reference/src/prox_operators.jl:68-126).

Third-party arithmetic the reference delegates to and that is NOT under
/root/reference (Project.toml:6-29 gives version *ranges*, there is no
Manifest.toml):

  * KrylovKit.jl "0.5.2 - 0.9"  `eigsolve(A, x0, howmany, :LR, Lanczos(...))`
    call site eigsolver.jl:802-812.  Restated below (`krylovkit_eigsolve`) from
    its published algorithm: Lanczos with full re-orthogonalisation
    (KrylovDefaults.orth, a modified Gram-Schmidt with a second full pass),
    Krylov-Schur / thick restart keeping (3*krylovdim + 2*converged) div 5 Ritz
    vectors, absolute residual tolerance.  PARITY UNPINNED at the eigen-solver
    level: the reference's tests hold no eigenpair, iteration-count or mat-vec
    vectors for it; it is pinned by mathematics (both
    sides converge eigenpairs to <= 1e-12) and by the solve-level known answers
    in tests/test_oracle_kat.py.
  * Arpack.jl -> ARPACK-NG dsaupd/dseupd, call sites eigsolver.jl:671,723.
    SciPy wraps the same Fortran routines: `scipy.sparse.linalg.eigsh`.
  * LAPACK dsyevr through `LinearAlgebra.eigen!` (prox_operators.jl:113,
    pdhg.jl:685): `scipy.linalg.eigh(driver="evr")`.
  * BLAS dsymv('U') through `mul!(y, Symmetric, x)` (eigsolver.jl:678 and inside
    KrylovKit): `scipy.linalg.blas.dsymv(lower=0)`.

The Julia start vector `normalize!(randn(MersenneTwister(1234), n))`
(eigsolver.jl:392-411) cannot be reproduced outside Julia.  Oracle and HIP
library share the counter-based generator `start_vector` below instead
(bit-identical in C++: csrc/host/resid.hpp); tests may also pass an explicit
vector to both sides.
"""
import numpy as np
import scipy.linalg
import scipy.sparse.linalg
from scipy.linalg import blas

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _uniform53(seed, counter):
    """u in [0,1): top 53 bits of splitmix64(seed ^ (counter+1)*K)."""
    with np.errstate(over="ignore"):
        key = np.uint64(seed) ^ ((np.asarray(counter, dtype=np.uint64) + np.uint64(1))
                                 * np.uint64(0xD1342543DE82EF95))
    bits = _splitmix64(key) >> np.uint64(11)
    return bits.astype(np.float64) * (1.0 / 9007199254740992.0)


def start_vector(n, seed=1234, init=3):
    """Replacement for eigsolver_update_resid! (eigsolver.jl:397-411).
    init 3: normalised approx-normal (Irwin-Hall 12 minus 6, sequential adds);
    init 2: uniform [0,1); init 1: ones; else zeros."""
    idx = np.arange(n, dtype=np.uint64)
    if init == 3:
        z = np.zeros(n)
        for k in range(12):
            z = z + _uniform53(seed, idx * np.uint64(12) + np.uint64(k))
        z = z - 6.0
        nrm = np.sqrt(np.cumsum(z * z)[-1]) if n > 0 else 1.0   # sequential sum
        return z / nrm
    if init == 2:
        return _uniform53(seed, idx * np.uint64(12))
    if init == 1:
        return np.ones(n)
    return np.zeros(n)


class EigSolverAlloc:
    """EigSolverAlloc workspace (eigsolver.jl:336-390), fields actually used."""

    def __init__(self, n, opt, resid=None):
        self.n = n
        self.nev = 1
        self.converged = False
        self.converged_eigs = 0
        self.vals = np.zeros(0)
        self.vecs = np.zeros((n, 0))
        self.ncv = max(2 * self.nev + 1, opt.eigsolver_min_lanczos)
        init = opt.arpack_resid_init if opt.eigsolver == 1 else opt.krylovkit_resid_init
        # krylovkit_init! (eigsolver.jl:780-787): resid generated once at alloc
        self.resid = (np.array(resid, dtype=float) if resid is not None
                      else start_vector(n, opt.eigsolver_resid_seed, init))
        self.resid0 = self.resid.copy()
        # statistics (not in the reference)
        self.matvecs = 0
        self.restarts = 0


def symv_upper(Xdata, v):
    """mul!(y, Symmetric(X,:U), v) == dsymv('U'): reads only the upper triangle."""
    return blas.dsymv(1.0, Xdata, v, lower=0)


# Orthogonaliser of the Lanczos recurrence (TEST-ONLY switch).  KrylovKit's `KrylovDefaults.orth` is not the
# same object across the versions Project.toml admits (0.5.2 - 0.9): modified Gram-Schmidt with a second full pass ("mgs2", the
# restatement's default), its classical twin ("cgs2"), and the variants with ITERATIVE REFINEMENT that run the second pass
# only while the norm dropped by more than eta = 1/sqrt(2) in the first ("mgsir", "cgsir"; KrylovKit orthonormal.jl).
# tools/r06/orth_variants.py runs the committed golden instances under each and tables what moves (profiles/r06_orthogonaliser_variants.md).
ORTH = "mgs2"


def _second_passes(V, K, vnew, w, alpha, beta_in):
    """Re-orthogonalisation of w against V[:, :K] and vnew after the recurrence's own subtractions.  Returns (w, alpha)."""
    orth = ORTH
    if orth == "mgs2":
        s = 0.0
        for j in range(K):
            s = float(V[:, j] @ w)
            w = w - s * V[:, j]
        s = float(vnew @ w)
        w = w - s * vnew
        return w, alpha + s
    if orth == "cgs2":
        h = V[:, :K].T @ w
        s = float(vnew @ w)
        w = w - V[:, :K] @ h - s * vnew
        return w, alpha + s
    if orth in ("mgsir", "cgsir"):
        eta = 1.0 / np.sqrt(2.0)
        ab2 = alpha * alpha + beta_in * beta_in
        beta = float(np.linalg.norm(w))
        nold = np.sqrt(ab2 + beta * beta)
        while np.finfo(float).eps < beta < eta * nold:
            if orth == "mgsir":
                for j in range(K):
                    s = float(V[:, j] @ w)
                    w = w - s * V[:, j]
                s = float(vnew @ w)
                w = w - s * vnew
            else:
                h = V[:, :K].T @ w
                s = float(vnew @ w)
                w = w - V[:, :K] @ h - s * vnew
            alpha += s
            nold = beta
            beta = float(np.linalg.norm(w))
        return w, alpha
    raise ValueError(orth)


def krylovkit_eigsolve(matvec, x0, howmany, krylovdim, maxiter, tol, eager=False):
    """KrylovKit.eigsolve(A, x0, howmany, :LR, Lanczos(orth, krylovdim, maxiter,
    tol, eager)) -- restated from the published algorithm (see module header).

    Returns (values, vectors[n, len(values)], converged, numiter, numops).
    The projected matrix is kept as a dense K x K symmetric matrix (diagonal of
    kept Ritz values + the coupling row after a restart, tridiagonal after it);
    KrylovKit re-tridiagonalises with Householder reflections instead, which is
    the same matrix in another orthonormal basis of the kept subspace."""
    n = x0.shape[0]
    V = np.zeros((n, krylovdim + 1))
    T = np.zeros((krylovdim + 1, krylovdim + 1))
    # initialize (KrylovKit lanczos.jl `initialize`)
    beta0 = np.linalg.norm(x0)
    if beta0 == 0.0:
        raise ValueError("initial vector should not have norm zero")
    v = x0 / beta0
    r = matvec(v)
    numops = 1
    alpha = float(v @ r)
    r = r - alpha * v
    d = float(v @ r)                 # second pass of the orthogonaliser
    alpha += d
    r = r - d * v
    beta = float(np.linalg.norm(r))
    V[:, 0] = v
    T[0, 0] = alpha
    K = 1
    numiter = 1
    converged = 0
    coupling_set = False             # True right after a restart (row f already in T)
    D = U = f = None
    while True:
        if beta <= tol and K < howmany:
            howmany = K              # invariant subspace smaller than requested
        # (eager: right after a restart KrylovKit's test sees the kept pairs with their old residuals f[:keep] --
        # fewer than `howmany` converged, or the run would have ended before the restart -- and expands; the dense
        # form kept here has no residual vector for that state, so the test is skipped until the next expansion)
        if K == krylovdim or beta <= tol or (eager and K >= howmany and not coupling_set):
            if K == 1:
                D = np.array([T[0, 0]])
                U = np.ones((1, 1))
                f = np.array([beta])
                converged = int(beta <= tol)
            else:
                Dasc, Uasc = scipy.linalg.eigh(T[:K, :K])
                D = Dasc[::-1].copy()            # :LR -> descending
                U = Uasc[:, ::-1].copy()
                f = beta * U[K - 1, :]
                converged = 0
                while converged < K and abs(f[converged]) <= tol:
                    converged += 1
            if converged >= howmany:
                break
        if K < krylovdim:
            # expand!: v_{K+1} = r / beta, recurrence, full re-orthogonalisation
            vnew = r / beta
            V[:, K] = vnew
            if not coupling_set:
                T[K - 1, K] = T[K, K - 1] = beta
            coupling_set = False
            w = matvec(vnew)
            numops += 1
            tk = T[:K, K]
            w = w - V[:, :K] @ tk                # beta*v_K, or sum f_j v_j after restart
            alpha = float(vnew @ w)
            w = w - alpha * vnew
            if ORTH == "mgs2":
                s = 0.0
                for j in range(K + 1):           # second full modified G-S pass
                    s = float(V[:, j] @ w)
                    w = w - s * V[:, j]
                alpha += s                       # correction along v_{K+1}
            else:
                w, alpha = _second_passes(V, K, vnew, w, alpha, float(np.linalg.norm(tk)))
            T[K, K] = alpha
            beta = float(np.linalg.norm(w))
            r = w
            K += 1
        else:
            if numiter == maxiter:
                break
            keep = (3 * krylovdim + 2 * converged) // 5
            V[:, :keep] = V[:, :K] @ U[:, :keep]
            T[:, :] = 0.0
            T[np.arange(keep), np.arange(keep)] = D[:keep]
            T[keep, :keep] = f[:keep]
            T[:keep, keep] = f[:keep]
            coupling_set = True
            K = keep
            numiter += 1
    if converged > howmany:
        howmany = converged
    values = D[:howmany].copy()
    vectors = V[:, :K] @ U[:, :howmany]
    return values, vectors, converged, numiter, numops


def krylovkit_eig(arc, Xdata, nev, opt):
    """krylovkit_eig!(arc, A, nev, opt)  (eigsolver.jl:798-823)."""
    arc.converged = True
    arc.nev = nev                                        # krylovkit_update! :789-796
    if opt.krylovkit_reset_resid:
        arc.resid = arc.resid0.copy()
    arc.ncv = max(2 * arc.nev + 1, opt.eigsolver_min_lanczos)
    vals, vecs, conv, numiter, numops = krylovkit_eigsolve(
        lambda v: symv_upper(Xdata, v), arc.resid, arc.nev, arc.ncv,
        opt.krylovkit_max_iter, opt.krylovkit_tol, opt.krylovkit_eager)
    arc.vals, arc.vecs = vals, vecs
    arc.converged_eigs = conv
    arc.matvecs += numops
    arc.restarts += numiter - 1
    if conv == 0:
        arc.converged = False


def arpack_eig(arc, Xdata, nev, opt):
    """arpack_eig! (eigsolver.jl:748-770) through SciPy's dsaupd/dseupd.
    bmat="I", which="LA", mode 1, tol=arpack_tol, ncv=max(2nev+1,25), user resid.
    arc.maxiter is uninitialised memory in the reference;
    the oracle uses opt.arpack_max_iter, the value arpack_init! intended."""
    arc.nev = nev
    arc.ncv = max(2 * nev + 1, opt.eigsolver_min_lanczos)
    n = arc.n
    arc.converged = False
    arc.converged_eigs = 0
    if not (0 < nev < n) or arc.ncv > n:
        return                      # dsaupd info = -1/-3 -> arpackerror (eigsolver.jl:686-702)
    full = np.triu(Xdata) + np.triu(Xdata, 1).T
    count = [0]

    def mv(v):
        count[0] += 1
        return symv_upper(Xdata, v)

    op = scipy.sparse.linalg.LinearOperator((n, n), matvec=mv, dtype=float)
    try:
        d, v = scipy.sparse.linalg.eigsh(op, k=nev, which="LA", ncv=arc.ncv,
                                         tol=opt.arpack_tol, v0=arc.resid0.copy(),
                                         maxiter=opt.arpack_max_iter)
    except (scipy.sparse.linalg.ArpackNoConvergence, scipy.sparse.linalg.ArpackError):
        # info outside 0..1 -> arc.arpackerror -> not converged -> full_eig! (eigsolver.jl:697-702);
        # e.g. info = -9 "starting vector is zero" on the all-zero first iterate
        arc.matvecs += count[0]
        return
    del full
    arc.matvecs += count[0]
    arc.vals = d                   # ascending, as arc.d
    arc.vecs = v
    arc.converged_eigs = nev
    arc.converged = True


def full_eigh(Xdata):
    """LinearAlgebra.eigen!(Symmetric(X,:U)) == LAPACK dsyevr, ascending."""
    return scipy.linalg.eigh(Xdata, lower=False, driver="evr")
