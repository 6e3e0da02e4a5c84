"""Time per sign-function projection with several builds of the library in ONE session:
gpurun -- python tools/gpurun_sign_ab.py <n> <lib.so | current | env:NAME=VALUE> ...   (each build in its own process, three rounds, interleaved)"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "child":
    sys.path.insert(0, root)
    import numpy as np
    from proxsdp_jl_amd import binding as B
    if sys.argv[3].startswith("env:"):                      # env:NAME=VALUE -- the current build with one environment knob
        k, v = sys.argv[3][4:].split("=")
        os.environ[k] = v
    elif sys.argv[3] != "current":
        B.LIB_PATH = B.pathlib.Path(os.path.abspath(sys.argv[3]))
    n = int(sys.argv[2])
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n)); A = (M + M.T) / 2
    iu = np.triu_indices(n)
    xp = np.zeros(n * (n + 1) // 2)
    for j in range(n):
        xp[j * (j + 1) // 2: j * (j + 1) // 2 + j + 1] = A[: j + 1, j] * np.sqrt(2.0)
        xp[j * (j + 1) // 2 + j] = A[j, j]
    ms = []
    for _ in range(4):
        o, t, r, p = B.full_eig_kernel(xp, n, sign=1, repeat=20)
        ms.append(t)
    print("%s n=%d products %d  ms per projection: %s" % (sys.argv[3], n, p, " ".join("%.4f" % v for v in ms)), flush=True)
else:
    n = sys.argv[1]
    for rnd in range(3):
        for lib in sys.argv[2:]:
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", n, lib])
