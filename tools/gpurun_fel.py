"""debug: full_eig!-by-Lanczos vs dense on gpp500-1 iterates captured from the oracle"""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle
from oracle import Options
from proxsdp_jl_amd import binding as B, problems as P
from helpers import smat, svec
import pathlib
pr = P.sdplib(pathlib.Path("tests/golden/sdplib/gpp500-1.dat-s"))
n = pr.psd_sides()[0]; N = n * (n + 1) // 2
caps = []
def cb(it, xin, xout, p, arc): caps.append((it, xin[:N].copy(), xout[:N].copy(), int(p.current_rank[0])))
o = Options(); o.max_iter = 12; o.full_eig_decomp = True
oracle.solve(pr, o, proj_callback=cb)
prev = 3
for it, xin, xout, rank in caps:
    w = np.linalg.eigvalsh(smat(xin, n))[::-1]
    npos = int((w > 0).sum())
    out, info = B.psd_project(xin, n, max(prev, 1), mode=2)
    nx = np.linalg.norm(xin)
    print(it, "npos", npos, "est", prev, "rank", rank, info, "err", np.linalg.norm(out - xout) / max(nx, 1e-300),
          "top", w[:npos + 2][-4:], "nx", nx)
    prev = npos
