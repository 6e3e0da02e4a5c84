import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))); sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'tests'))
import numpy as np, time
from proxsdp_jl_amd import binding as B
for n in (1000, 2000, 4000, 8000):
    rng = np.random.default_rng(0)
    x = rng.standard_normal(n*(n+1)//2); v = rng.standard_normal(n)
    y, ms = B.symv_packed(x, n, v, repeat=200)
    byts = 8*n*(n+1)/2 + 16*n
    print(f"symv n={n}: {ms*1e3:.2f} us  {byts/ms/1e6:.0f} GB/s")
    Z = rng.standard_normal((n, 16)); lam = rng.uniform(1,2,16)
    out, ms = B.reconstruct(Z, lam, n, repeat=50)
    print(f"reconstruct n={n} r=16: {ms*1e3:.2f} us  {8*n*(n+1)/2/ms/1e6:.0f} GB/s")
