"""Prototype support (CPU): resume the ORACLE from a committed state of the metric instance and dump the matrices handed
to psd_projection! for a few consecutive iterations of the implicit full_eig! regime (x_in / x_out, packed) to /tmp --
the inputs tools/proto/block_filter_proto.py experiments on.   python tools/proto/dump_phase2_inputs.py [first] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
from proxsdp_jl_amd import problems
from helpers import expand_state, load_compact_state

tag = os.environ.get("STATE", "kU")
st = expand_state(load_compact_state(os.path.join(ROOT, "tests", "golden", f"state_maxcut_n4000_{tag}.npz")))
k0 = int(st["iteration"])
first = int(sys.argv[1]) if len(sys.argv) > 1 else k0 + 20
count = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pr = problems.maxcut(4000, seed=0)
o = oracle.Options()
o.max_iter = first + count - 1


def cb(it, xin, xout, p, arc):
    if first <= it < first + count:
        np.save(f"/tmp/phase2_xin_{it}.npy", xin)
        np.save(f"/tmp/phase2_xout_{it}.npy", xout)
        print("dumped", it, "target rank", p.target_rank, "current rank", p.current_rank, flush=True)


oracle.solve(pr, o, resume=st, proj_callback=cb)
