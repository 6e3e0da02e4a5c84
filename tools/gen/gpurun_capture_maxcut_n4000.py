"""GPU-box half of tests/golden/make_golden_late_n4000.py: run the LIBRARY on the metric instance (Max-Cut n = 4000,
seed 0, default options) and write its solver state (proxsdp_hip_solve_ex capture) at two iteration boundaries as
compact fixtures (tests/helpers.compact_state):

    state_maxcut_n4000_k1000.npz    the steady window of SURVEY section 8d ("1000-1200 from a saved state")
    state_maxcut_n4000_kU.npz       U = 12 iterations before the 16 -> 17 rank update, i.e. before the solve leaves
                                    KrylovKit's range (max_target_rank_krylov_eigs = 16) for the implicit full_eig! regime
    state_maxcut_n4000_kEnd.npz     31 iterations before the solve stops (OPTIMAL after 8651 iterations)

    gpurun -- python tools/gen/gpurun_capture_maxcut_n4000.py      -> gpurun_out/cap4000/
The CPU half resumes the oracle from these states.  Also stores the library's own trace of the whole solve."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                                     # noqa: E402
from proxsdp_jl_amd import problems                                    # noqa: E402
from proxsdp_jl_amd.optimizer import Optimizer                         # noqa: E402
from helpers import compact_state, expand_state, save_compact_state   # noqa: E402

out = os.path.join(ROOT, "gpurun_out", "cap4000")
os.makedirs(out, exist_ok=True)
n = int(os.environ.get("CAP_N", "4000"))
pr = problems.maxcut(n, seed=0)
t = time.time()
sol = Optimizer().optimize(pr, trace_capacity=20000, capture_iteration=1000)
print("run 1: %d iterations, status %d, obj %.9f, %.2f s (loop %.2f s)" % (sol.iter, sol.status, sol.objval, time.time() - t, sol.stats["loop_time"]), flush=True)
tr = sol.trace
np.save(os.path.join(out, "trace_default.npy"), tr)
tgt = tr[:, 10].astype(int)
first17 = int(tr[np.argmax(tgt > 16), 0]) if (tgt > 16).any() else None
print("first iteration run at target rank 17:", first17, "rank updates after:", [int(tr[i, 0]) for i in range(1, len(tr)) if tgt[i] != tgt[i - 1]])
info = dict(n=n, iterations=int(sol.iter), status=int(sol.status), objval=float(sol.objval), first17=first17,
            stats={k: v for k, v in sol.stats.items() if not isinstance(v, list)})


def store(state, tag):
    t0 = time.time()
    c = compact_state(state, pr.psd_sides())
    e = expand_state(c)
    err = float(np.abs(e["x"] - state["x"]).max() / np.abs(state["x"]).max())
    assert np.array_equal(e["Mty"], state["Mty"])
    save_compact_state(os.path.join(out, f"state_maxcut_n{n}_{tag}.npz"), c)
    print(f"state {tag}: iteration {state['iteration']}, target rank {state['target_rank']}, x rank {len(c['x_factors'][0][0])}, "
          f"round-trip error {err:.2e}, M'y nnz {len(c['Mty_idx'])}, {time.time() - t0:.1f} s", flush=True)
    info[tag] = dict(iteration=int(state["iteration"]), target_rank=int(state["target_rank"][0]), x_rank=len(c["x_factors"][0][0]),
                     round_trip=err)


store(sol.state, "k1000")
if first17 is not None:
    U = first17 - 1 - 12
    sol2 = Optimizer().optimize(pr, trace_capacity=20000, capture_iteration=U)
    same = np.array_equal(sol2.trace[:, :12], tr[:, :12])
    print("run 2: %d iterations, trace identical to run 1: %s" % (sol2.iter, same), flush=True)
    info["deterministic"] = bool(same)
    store(sol2.state, "kU")
    info["U"] = U
# third state: 31 iterations before the solve ends -- the oracle continues it to ITS stop (same iteration, same objective?)
kE = int(sol.iter) - 31
sol3 = Optimizer().optimize(pr, trace_capacity=20000, capture_iteration=kE)
info["deterministic_3"] = bool(np.array_equal(sol3.trace[:, :12], tr[:, :12]))
store(sol3.state, "kEnd")
info["kEnd"]["final"] = dict(iterations=int(sol3.iter), status=int(sol3.status), objval=float(sol3.objval), gap=float(sol3.gap),
                             final_rank=int(sol3.final_rank))
json.dump(info, open(os.path.join(out, "info.json"), "w"), indent=1)
