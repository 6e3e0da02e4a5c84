import sys, json, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))); sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'tests'))
import numpy as np
from proxsdp_jl_amd import problems as P
from proxsdp_jl_amd.optimizer import Optimizer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
pr = P.maxcut(n, seed=0)
import os
o = Optimizer(time_limit=240.0, profile_symv_every=16, max_target_rank_krylov_eigs=int(os.environ.get('KR','64')))
t=time.time(); s = o.optimize(pr, trace_capacity=20000); wall=time.time()-t
tr = s.trace
print(json.dumps(dict(n=n, status=o.termination_status(), iters=int(s.iter), time=s.time, wall=wall, obj=o.objective_value(), gap=s.gap,
   rank=int(s.final_rank), stats=s.stats, target_rank_final=float(tr[-1,10]),
   mv_per_iter_quartiles=[float(np.percentile(tr[:,13],q)) for q in (0,25,50,75,100)],
   it_per_s=s.iter/s.stats["loop_time"])))
# per-500-iteration breakdown
for a in range(0, len(tr), 500):
    b=min(a+500,len(tr)); seg=tr[a:b]
    dt = seg[-1,12]-(tr[a-1,12] if a>0 else 0.0)
    print(a+1,b,"ms/iter %.3f"%(1e3*dt/(b-a)),"mv/iter %.1f"%seg[:,13].mean(),"trials %.2f"%seg[:,11].mean(),"tr",seg[-1,10],"gap %.2e feas %.2e"%(seg[-1,3],seg[-1,4]))
