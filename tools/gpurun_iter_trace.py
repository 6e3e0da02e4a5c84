"""Kernel sequence of ONE PDHG iteration from a rocprofv3 kernel trace (start, duration, gap to the previous kernel's end).
On the GPU box:
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/itrace -- python tools/gpurun_iter_trace.py run gpp500-1 40
  python tools/gpurun_iter_trace.py show gpurun_out/itrace [iteration-from-the-end]"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "run":
    from proxsdp_jl_amd import problems as P
    from proxsdp_jl_amd.optimizer import Optimizer
    name, iters = sys.argv[2], int(sys.argv[3])
    kw = {}
    if name.startswith("maxcut"):
        pr = P.maxcut(int(name[6:]), seed=0)
    elif name.startswith("mimo1x"):
        pr = P.mimo(int(name[6:]), seed=0)                      # ONE small block (side n + 1): the small-model regime
    elif name.startswith("mimo"):
        pr = P.block_diag_problems([P.mimo(512, seed=s) for s in range(8)], name="mimo8")
    else:
        pr = P.sdplib(os.path.join("tests", "golden", "sdplib", name + ".dat-s"))
        kw = dict(full_eig_decomp=1)
    for a in sys.argv[4:]:
        k, v = a.split("="); kw[k] = int(v)
    s = Optimizer(max_iter=iters, **kw).optimize(pr)
    print(name, s.iter, s.time)
else:
    files = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # iterations are delimited by the primal-update kernel
    marks = [i for i, r in enumerate(rows) if "k_primal_update" in r[2]]
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    a, b = marks[-back - 1], marks[-back]
    print("kernels in this iteration:", b - a, " wall %.1f us" % ((rows[b][0] - rows[a][0]) / 1e3))
    prev_end = rows[a - 1][1] if a > 0 else rows[a][0]
    agg = {}
    for s, e, nme in rows[a:b]:
        short = nme.split("(")[0].replace("void ", "").replace("proxsdp::dev::", "")[:60]
        d = agg.setdefault(short, [0, 0.0, 0.0])
        d[0] += 1; d[1] += (e - s) / 1e3; d[2] += max(0, s - prev_end) / 1e3
        prev_end = e
    for k, (c, dur, gap) in sorted(agg.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
        print("%-62s x%-4d busy %8.1f us  gaps before %8.1f us" % (k, c, dur, gap))
    # the large idle gaps of the iteration (> 15 us): which kernel the GPU waited for, and for how long
    prev_end = rows[a - 1][1] if a > 0 else rows[a][0]
    print("idle gaps > 15 us (kernel that ended the wait):")
    for s, e, nme in rows[a:b]:
        g = (s - prev_end) / 1e3
        if g > 15.0:
            print("   %8.1f us before %s" % (g, nme.split("(")[0].replace("void ", "").replace("proxsdp::dev::", "")[:70]))
        prev_end = e
