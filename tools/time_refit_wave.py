import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _timing import timeit
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
os.environ["BCBF_REFIT_WAVE"] = "1"; os.environ["BCBF_REFIT_PAIR"] = "0"
r = []
for DT, Bt, N in ((torch.float64, 4096, 256), (torch.float64, 1024, 512), (torch.float64, 4096, 512), (torch.float64, 1024, 1024), (torch.float32, 4096, 256), (torch.float32, 1024, 512), (torch.float32, 4096, 512), (torch.float32, 1024, 1024), (torch.float32, 4096, 1024)):
    n, m = (2, 1) if N <= 256 else (3, 2)
    p = make_instances(Bt, N, n, m, dtype=DT, device="cuda", seed=5)
    r.append("%s %dx%d %.4f" % (str(DT)[11:], Bt, N, timeit(lambda: ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"]), reps=5)))
print(sys.argv[1] if len(sys.argv) > 1 else "", " | ".join(r))
