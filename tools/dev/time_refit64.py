import sys, os, torch, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _timing import timeit
os.environ["BCBF_REFIT_WAVE"] = "1"
out = {}
for Bt, N in ((1024, 512), (4096, 512), (1024, 1024)):
    p = make_instances(Bt, N, 3, 2, dtype=torch.float64, device="cuda", seed=5)
    out["%dx%d" % (Bt, N)] = round(timeit(lambda: ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"]), reps=5), 4)
    Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
    out["fails%dx%d" % (Bt, N)] = int((info != 0).sum())
print(os.environ.get("BCBF_RW64_SUPER_FORCE"), json.dumps(out))
