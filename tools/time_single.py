import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
  for N in (64, 256, 512, 1024):
    p = make_instances(1, N, 2, 1, dtype=dtype, device="cuda", seed=1)
    f = lambda: ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
    Lop, UHB, info, _ = f()
    t_refit = timeit(f)
    g = lambda: ops.potrs(Lop, p["Xdot"], p["UH"], p["M0"])
    Vw, al = g(); t_potrs = timeit(g)
    xq = p["X"][0, :400 % N if N < 400 else 400].contiguous() if N >= 400 else p["X"][0].repeat(8, 1)[:400].contiguous()
    h = lambda: ops.posterior_query(Lop, Vw, p["X"], UHB, p["ell"], p["s2"], p["Bm"], p["M0"], xq, shared=True, want_W=True)
    t_q = timeit(h)
    print("%s N=%4d  refit %7.1f us  potrs %7.1f us  query(b=400,W) %7.1f us" % (str(dtype)[6:], N, t_refit, t_potrs, t_q))
