"""Captured-PDHG-iterate projection fixtures ("projection pairs ... for
captured PDHG iterates").  From an oracle run of each instance, the FIRST THREE iterations whose
Lanczos needed a thick restart:

  <case>__it<k>__in     the packed vector handed to psd_projection! (prox_operators.jl:33-66)
  <case>__it<k>__vals   the eigenvalues the oracle's truncated projection used (positive ones among
                        the first min(target_rank, converged), prox_operators.jl:99-106)
  <case>__it<k>__vecs   their Ritz vectors (n x len(vals)): the oracle's output is
                        svec(vecs diag(vals) vecs'), stored in this factored form
  <case>__it<k>__meta   [n, iter, target_rank, current_rank, min_eig, matvecs, restarts, converged_eigs]
  <case>__it<k>__top    the target_rank + 3 largest eigenvalues of the input (LAPACK), for the
                        degeneracy criterion: the rank-`target_rank` truncation is only defined up to
                        the eigenvalue gap lambda_r - lambda_{r+1}

Run from the repo root:  python tests/golden/make_golden_captured.py"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import oracle  # noqa: E402
from oracle import Options  # noqa: E402
from proxsdp_jl_amd import problems as P  # noqa: E402
from helpers import smat, capture_restart_projections  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parent


def main():
    cases = [("sdplib_mcp124-1", P.sdplib(OUT / "sdplib" / "mcp124-1.dat-s"), 120),
             ("maxcut_er_n200_s0", P.maxcut(200, seed=0), 400)]
    data = {}
    for name, pr, iters in cases:
        caps = capture_restart_projections(pr, iters, 3)
        for c in caps:
            key = f"{name}__it{c['iter']}"
            data[key + "__in"] = c["x_in"]
            data[key + "__vals"] = c["vals"]
            data[key + "__vecs"] = c["vecs"]
            data[key + "__meta"] = np.array([c["n"], c["iter"], c["target_rank"], c["rank"], c["min_eig"],
                                             c["matvecs"], c["restarts"], c["converged_eigs"]], dtype=float)
            data[key + "__top"] = c["top"]
            print(key, "tr", c["target_rank"], "rank", c["rank"], "matvecs", c["matvecs"], "restarts", c["restarts"],
                  "top", c["top"])
    np.savez_compressed(OUT / "captured_projections.npz", **data)


if __name__ == "__main__":
    main()
