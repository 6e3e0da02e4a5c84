import json, sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))); sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'tests'))
from pathlib import Path
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
g = Path("tests/golden")
gold = json.loads((g / "traces.json").read_text())["sdplib_mcp124-1"]
pr = P.sdplib(g / "sdplib" / "mcp124-1.dat-s")
sol = Optimizer(max_iter=120).optimize(pr, trace_capacity=120)
G = np.array(gold["rows"]); T = sol.trace[:len(G)]
rel = np.abs(T[:, 1] - G[:, 1]) / (1e-300 + np.abs(G[:, 1]))
print("matvecs", T[:, 13].astype(int)[::8])
print("rel prim_obj", " ".join("%.1e" % r for r in rel[::4]))
for col, nm in ((1, "prim_obj"), (2, "dual_obj"), (7, "primal_step"), (8, "beta"), (9, "theta")):
    sc = np.abs(G[:, col]).max()
    d = np.abs(T[:, col] - G[:, col]) - 5e-2 * np.abs(G[:, col])
    print(nm, "scale %.3g max excess/scale %.3g at row %d" % (sc, d.max() / sc, d.argmax()), T[d.argmax(), col], G[d.argmax(), col])
