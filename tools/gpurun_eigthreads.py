"""Host K x K eigensolve with spinning helper threads (host_util.hpp QlPool) on the GPU box's host: standalone timing at
K = 127 with 2.5 ms gaps (the rank-63 iteration period), then the rank-63 bench window under PROXSDP_HIP_EIG_THREADS."""
import os, sys, time, json, subprocess
sys.path.insert(0, ".")
if len(sys.argv) > 1 and sys.argv[1] == "standalone":
    import numpy as np
    from proxsdp_jl_amd import binding as B
    rng = np.random.default_rng(0)
    k = 127
    T = np.diag(rng.standard_normal(k)) + np.diag(rng.uniform(0.5, 1.5, k - 1), 1); T = np.triu(T) + np.triu(T, 1).T
    d0, U0 = B.host_symeig(T, threads=0)
    out = {}
    for thr in (0, 1, 2, 4, 8):
        ts = []
        for it in range(200):
            t1 = time.perf_counter()
            while time.perf_counter() - t1 < 2.5e-3: pass          # the GPU's Lanczos cycle
            t0 = time.perf_counter(); d1, U1 = B.host_symeig(T, threads=thr); ts.append(time.perf_counter() - t0)
            assert np.array_equal(d0, d1) and np.array_equal(U0, U1)
        out[thr] = dict(median_ms=1e3 * float(np.median(ts)), p90_ms=1e3 * float(np.quantile(ts, 0.9)), max_ms=1e3 * max(ts))
        print(thr, out[thr], flush=True)
    json.dump(out, open("gpurun_out/eigthreads_standalone.json", "w"), indent=1)
else:
    res = {}
    for thr in ("0", "2", "4", "8"):
        env = dict(os.environ, PROXSDP_HIP_EIG_THREADS=thr)
        r = subprocess.run([sys.executable, "bench.py", "--no-cpu", "--no-time-to-tol", "--hbm-n", "0"], env=env,
                           capture_output=True, text=True)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        res[thr] = dict(value=d["value"], ms=d["ms_per_step"], host_eig_ms=d["config"]["host_eigensolve_ms_per_step"])
        print(thr, res[thr], flush=True)
    json.dump(res, open("gpurun_out/eigthreads_bench.json", "w"), indent=1)
