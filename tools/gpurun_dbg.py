import numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))); sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'tests'))
from helpers import planted_packed, smat
from proxsdp_jl_amd import binding as B
for n, nev, top in [(101, 2, [40.0, 25.0, 9.0, 4.0]), (257, 4, [90.0, 60.0, 33.0, 12.0, 5.0]),
                    (1000, 3, [500.0, 20.0, 19.5, 19.0]), (2000, 24, list(np.linspace(60, 20, 24)))]:
    x = planted_packed(n, 11, top, bulk=(-5.0, 1.0))
    X = smat(x, n)
    vals, vecs, info = B.eigsolve(x, n, nev)
    ref = np.sort(np.linalg.eigvalsh(X))[::-1]
    print(n, nev, info, "orth", abs(vecs.T @ vecs - np.eye(vecs.shape[1])).max(),
          "val", abs(vals[:nev] - ref[:nev]).max(), "res", np.linalg.norm(X @ vecs - vecs * vals, axis=0).max())
