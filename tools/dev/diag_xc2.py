"""Development: the online loop's own ReservedGP after a run vs a freshly built one: plain reserved query and fused append + query."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.rollouts import learning_closed_loop
from _timing import timeit
out, final = learning_closed_loop(4096, 512, 40, 40, warmup=0, dtype=torch.float32, device="cuda", seed=1234)
rgp, p = final["rgp"], final["p"]
print("loop pass_ms", out["shares"]["pass_ms_per_step"], "N", rgp.N, flush=True)
N = rgp.N
xn, uhn, xdn, jn = (p[k][:, 500].contiguous() for k in ("X", "UH", "Xdot", "jitter"))
def fused(g):
    n0 = g.N
    g.append(xn, uhn, xdn, jn, query=p["xq"])
    g.N = n0
for tag, g in (("loop rgp", rgp),):
    print(tag, "plain reserved query", timeit(lambda: g.posterior(p["xq"])), "fused", timeit(lambda: fused(g)), flush=True)
# a fresh one from the same rows
lo = final["lo"]
sl = slice(lo, lo + N)
X, UH, Y, J = (p[k][:, sl].contiguous() for k in ("X", "UH", "Xdot", "jitter"))
Lop, UHB, info, _ = ops.refit(X, UH, p["Bm"], p["ell"], p["s2"], J)
Vw, _ = ops.potrs(Lop, Y, UH, p["M0"], want_alpha=False)
g2 = ops.ReservedGP(Lop, Vw, X, UHB, p["ell"], p["s2"], p["Bm"], p["M0"], 512)
print("fresh rgp", "plain reserved query", timeit(lambda: g2.posterior(p["xq"])), "fused", timeit(lambda: fused(g2)), flush=True)
g3 = ops.ReservedGP(Lop, Vw, X, UHB, p["ell"], p["s2"], p["Bm"], p["M0"], 512, window=472, drop=40, UH=UH, Xdot=Y, jitter=J)
print("fresh windowed rgp", "fused", timeit(lambda: fused(g3)), flush=True)
x = final["x"]
print("fresh rgp, query = loop state x", timeit(lambda: (g2.append(xn, uhn, xdn, jn, query=x), setattr(g2, "N", N))), flush=True)
