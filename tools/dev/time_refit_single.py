import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for dtype in (torch.float32, torch.float64):
    for Bt in (1, 16):
        for N in (128, 256, 512, 1024):
            p = make_instances(Bt, N, 2, 1, dtype=dtype, device="cuda", seed=1)
            t = timeit(lambda: ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"]))
            print("%s Bt=%2d N=%4d refit %8.1f us" % (str(dtype)[6:], Bt, N, t))
