"""What a device-side restart could save at most: from a rocprofv3 kernel trace of a restart-heavy solve, the time
between the end of a cycle's closing launch (k_lz_finish) and the start of the restart rotation (k_lz_rotate) -- the host round trip
as the GPU sees it (read-back, K x K eigensolve, staging, upload, launch) -- and how much of it the speculated mat-vec covers.
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06/rg -- python tools/r06/run_one.py gpp500-1 400 lanczos_cycle_kernel=0
  python tools/r06/restart_gaps.py gpurun_out/r06/rg"""
import csv, glob, os, sys
import numpy as np
files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
gaps, busy, cyc = [], [], []
last_rot_end = None
for i, (s, e, n) in enumerate(rows):
    if "k_lz_rotate" in n or "k_lzb_rotate" in n:
        # walk back to the closing launch of the cycle
        j = i - 1
        inter = 0.0
        while j >= 0 and not ("k_lz_finish" in rows[j][2] or "k_lzb_mv" in rows[j][2]):
            inter += (rows[j][1] - rows[j][0]) / 1e3
            j -= 1
        if j >= 0 and i - j <= 6:
            gaps.append((s - rows[j][1]) / 1e3); busy.append(inter)
            if last_rot_end is not None: cyc.append((s - last_rot_end) / 1e3)
        last_rot_end = e
g, b = np.array(gaps), np.array(busy)
print("restart rotations with a closing launch in front: %d" % len(g))
print("closing launch end -> rotation start: median %.1f us (10%% %.1f, 90%% %.1f); kernels running in that window (speculated mat-vec, copies): median %.1f us"
      % (np.median(g), np.percentile(g, 10), np.percentile(g, 90), np.median(b)))
print("GPU idle per restart: median %.1f us;  rotation end -> next rotation start (a cycle between restarts): median %.1f us" % (np.median(g - b), np.median(cyc)))
