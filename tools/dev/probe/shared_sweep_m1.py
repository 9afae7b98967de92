import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _timing import timeit
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
for dtype in (torch.float64, torch.float32):
    p = make_instances(1, 512, 2, 1, dtype=dtype, device="cuda", seed=1)
    Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
    Vw, _ = ops.potrs(Lop, p["Xdot"], p["UH"], p["M0"])
    for b in (4096, 8192, 16384, 32768):
        xq = (p["X"][0, torch.randint(0, 512, (b,), device="cuda")] + 0.3 * torch.randn(b, 2, device="cuda", dtype=dtype)).contiguous()
        t = timeit(lambda: ops.posterior_shared(Lop, Vw, p["X"], UHB, p["ell"], p["s2"], p["Bm"], p["M0"], xq))
        print(json.dumps({"m": 1, "dtype": str(dtype)[6:], "queries": b, "ms": round(t, 4), "TFLOPs": round(b * 2 * 512 * 512 / (t * 1e-3) / 1e12, 2)}), flush=True)
