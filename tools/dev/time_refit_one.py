"""Latency of ONE model's refit (the facade's fit / clear_cache path), fp32 and fp64 (development)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _timing import timeit
for dtype in (torch.float32, torch.float64):
    out = []
    for N in (128, 256, 512, 1024):
        p = make_instances(1, N, 3, 2, dtype=dtype, device="cuda", seed=5)
        p["X"] = (2.0 * p["X"]).contiguous()
        t = timeit(lambda: ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"]), reps=20)
        r = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"], want_dense=True)
        out.append("%d: %.0f us%s" % (N, t * 1e3, "" if int(r[2][0]) == 0 else " FAIL"))
    print("pair=%s" % os.environ.get("BCBF_REFIT_PAIR", "-"), str(dtype)[6:], "Bt=1 refit", "  ".join(out))
