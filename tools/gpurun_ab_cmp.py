import numpy as np, json, sys, os
for name in ("maxcut2000", "mcp500-1", "maxcut4000r63"):
    a = np.load(f"gpurun_out/ab/old_{name}.npy"); b = np.load(f"gpurun_out/ab/new_{name}.npy")
    m = min(len(a), len(b))
    rel = np.abs(a[:m, 1] - b[:m, 1]) / (1 + np.abs(a[:m, 1]))
    mv = a[:m, 13] != b[:m, 13]; rk = a[:m, 10] != b[:m, 10]
    idx = [0, 10, 50, 100, 200, 400, 800, 1200, 1600, 2000, 2400]
    print(name, "rows", m, "first matvec diff at", int(np.argmax(mv)) if mv.any() else None, "first rank diff at", int(np.argmax(rk)) if rk.any() else None)
    print("   rel prim_obj diff at", [(i, float("%.2e" % rel[i])) for i in idx if i < m])
    if name == "mcp500-1" and os.path.exists("tests/golden/trace_sdplib500.json"):
        G = np.array(json.load(open("tests/golden/trace_sdplib500.json"))[name]["rows"]); gm = np.array(json.load(open("tests/golden/trace_sdplib500.json"))[name]["matvecs"])
        for tag, t in (("old", a), ("new", b)):
            k = min(len(G), len(t))
            r = np.abs(G[:k, 1] - t[:k, 1]) / (1 + np.abs(G[:k, 1]))
            mvd = gm[:k] != t[:k, 13]
            print("   vs oracle", tag, "rel diff at", [(i, float("%.2e" % r[i])) for i in (0, 10, 50, 100, 200, 300, 399) if i < k], "matvec counts differ in", int(mvd.sum()), "first", int(np.argmax(mvd)) if mvd.any() else None)
