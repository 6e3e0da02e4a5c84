import sys, os
mode = sys.argv[1]
if mode in ("torch", "torchinit", "torchthreads1"):
    import torch
    if mode == "torchthreads1":
        torch.set_num_threads(1)
    if mode in ("torchinit", "torchthreads1"):
        torch.cuda.init()
sys.argv = [sys.argv[0]]
exec(open("tools/gpurun_rank63.py").read())
