import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from proxsdp_jl_amd import binding as B
n=4000; rng=np.random.default_rng(0)
x=rng.standard_normal(n*(n+1)//2); v=rng.standard_normal(n)
y,ms=B.symv_packed(x,n,v,repeat=5); print("symv ok", ms)
Z=rng.standard_normal((n,4)); out,ms=B.reconstruct(Z,np.ones(4),n,repeat=3); print("recon ok", ms)
