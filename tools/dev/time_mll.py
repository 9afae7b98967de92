"""Development: bcbf_mll_grad alone at C3 scale (random symmetric K^-1 stand-in: the kernel's time does not depend on the values)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
for Bt, N, dt in ((4096, 512, torch.float64), (4096, 512, torch.float32), (4096, 256, torch.float64), (1024, 512, torch.float64)):
    p = make_instances(Bt, N, 3, 2, dtype=dt, device="cuda", seed=1)
    Lop, UHB, info = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])[:3]
    Kinv = torch.randn(Bt, N, N, dtype=dt, device="cuda") * 0.01
    R = p["Xdot"] - p["UH"] @ p["M0"] if "M0" in p else p["Xdot"]
    alpha = ops.kinv_apply(Kinv, R.contiguous())
    Ainv = torch.linalg.inv(p["A"]).contiguous()
    args = (Lop, alpha, Kinv, p["X"], p["UH"], R.contiguous(), Ainv, p["Bm"], p["ell"], p["s2"])
    for _ in range(3): ops.mll_grad(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.mll_grad(*args)
    e1.record(); torch.cuda.synchronize()
    print(Bt, N, dt, "mll_grad %.3f ms" % (e0.elapsed_time(e1) / 10), flush=True)
