import sys, os, torch, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _timing import timeit
os.environ["BCBF_REFIT_WAVE"] = "1"
out = {}
for Bt, N in ((1024, 256), (4096, 256), (1024, 512), (4096, 512), (4096, 1024)):
    n, m = (2, 1) if N <= 256 else (3, 2)
    p = make_instances(Bt, N, n, m, dtype=torch.float32, device="cuda", seed=5)
    out["%dx%d" % (Bt, N)] = round(timeit(lambda: ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"]), reps=5), 4)
print(sys.argv[1] if len(sys.argv) > 1 else "", json.dumps(out))
