import os, torch, time, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import timeit
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
dev = "cuda:0"
for dtype in ((torch.float64,) if os.environ.get('BCBF_TS_F64') else (torch.float64, torch.float32)):
    for N in (128, 256, 512):
        p = make_instances(1, N, 3, 2, dtype=dtype, device=dev, seed=1)
        Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
        Vw, _ = ops.potrs(Lop, p["Xdot"], p["UH"], p["M0"])
        for b in (1024, 4096, 16384):
            xq = (p["X"][0, torch.randint(0, N, (b,), device=dev)] + 0.3 * torch.randn(b, 3, device=dev, dtype=dtype)).contiguous()
            t_m = timeit(lambda: ops.posterior_shared(Lop, Vw, p["X"], UHB, p["ell"], p["s2"], p["Bm"], p["M0"], xq))
            # streaming kernel on the shared model: small-b route is the same kernel; time it through posterior_query with b<16 chunks is unfair -> use env to disable? compare with replicated per-instance step instead
            fl = b * 3 * N * N / (t_m * 1e-3) / 1e12
            print(str(dtype)[6:], "N", N, "b", b, "mfma ms %.4f" % t_m, "TF %.2f" % fl, flush=True)
