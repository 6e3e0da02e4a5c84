import sys, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))); sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'tests'))
import numpy as np, math
from proxsdp_jl_amd import binding as B
n, r = 4000, 24
rng = np.random.default_rng(0)
Z, _ = np.linalg.qr(rng.standard_normal((n, r)))
lam = np.linspace(400, 20, r)
ii = np.concatenate([np.arange(j + 1) for j in range(n)]); jj = np.repeat(np.arange(n), np.arange(1, n + 1))
x = np.einsum("ik,ik->i", (Z*lam)[ii], Z[jj])
# sparse symmetric perturbation (like -tau*(Mty+c)): diagonal + ~24k off-diagonals
x[jj*(jj+1)//2+ii == (jj*(jj+3)//2)] += rng.standard_normal(n)*0.3
sel = rng.choice(len(x), 24000, replace=False); x[sel] += rng.standard_normal(24000)*0.2
x = x*np.where(ii==jj,1.0,math.sqrt(2))
t=time.time(); vals, vecs, info = B.eigsolve(x, n, 26, cap=60); dt=time.time()-t
print("info", info, "wall %.3f s"%dt, "vals", vals[:3], vals[22:27])
t=time.time(); vals, vecs, info = B.eigsolve(x, n, 26, cap=60); dt=time.time()-t
print("second call wall %.3f s"%dt, "us/matvec %.1f"%(dt*1e6/info["nmatvec"]))
