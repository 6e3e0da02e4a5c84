"""GPU-box half of the headline-window fixture: the LIBRARY's state inside bench.py's timed window (Max-Cut n = 4000, seed 0, window
pinned at target rank 63: initial_target_rank = 63, max_target_rank_krylov_eigs = 64 -- bench.py's headline options), captured
after iteration 250 (settle 200 + warm-up 5 + 45) -> gpurun_out/cap4000/state_maxcut_n4000_rank63_k250.npz.
CPU half: tests/golden/make_golden_headline_window.py."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from proxsdp_jl_amd import problems
from proxsdp_jl_amd.optimizer import Optimizer
from helpers import compact_state, expand_state, save_compact_state
out = os.path.join(ROOT, "gpurun_out", "cap4000"); os.makedirs(out, exist_ok=True)
pr = problems.maxcut(4000, seed=0)
sol = Optimizer(initial_target_rank=63, max_target_rank_krylov_eigs=64, max_iter=300).optimize(pr, trace_capacity=300, capture_iteration=250)
c = compact_state(sol.state, pr.psd_sides())
e = expand_state(c)
print("iteration", sol.state["iteration"], "target rank", sol.state["target_rank"], "x rank", len(c["x_factors"][0][0]),
      "round trip %.2e" % (np.abs(e["x"] - sol.state["x"]).max() / np.abs(sol.state["x"]).max()),
      "mat-vecs 251..270:", [int(v) for v in sol.trace[250:270, 13]])
save_compact_state(os.path.join(out, "state_maxcut_n4000_rank63_k250.npz"), c)
