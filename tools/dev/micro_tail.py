"""Development: the tail step's kernels in isolation -- posterior only (do_append = 0) at t = 0 / 20 / 39, and appends -- for a kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
Bt, W, D, n, m = 4096, 472, 40, 3, 2
dev = torch.device("cuda")
p = make_instances(Bt, W + D + 1, n, m, dtype=torch.float32, device=dev, seed=3)
cut = lambda t, N: t[:, :N].contiguous()
jit0 = cut(p["jitter"], W) * 10
Lop, UHB, info, _ = ops.refit(cut(p["X"], W), cut(p["UH"], W), p["Bm"], p["ell"], p["s2"], jit0)
print("refit failures", int((info != 0).sum()))
Vw, _ = ops.potrs(Lop, cut(p["Xdot"], W), cut(p["UH"], W), p["M0"], want_alpha=False)
g = ops.ReservedGP(Lop, Vw, cut(p["X"], W), UHB, p["ell"], p["s2"], p["Bm"], p["M0"], W + D, window=W, drop=D,
                   UH=cut(p["UH"], W), Xdot=cut(p["Xdot"], W), jitter=jit0, tail=True)
xq = p["xq"]
for rep in range(3):
    g.posterior(xq)                      # t = 0, no append
torch.cuda.synchronize()
for k in range(D - 1):
    g.append(p["X"][:, W + k].contiguous(), p["UH"][:, W + k].contiguous(), p["Xdot"][:, W + k].contiguous(), p["jitter"][:, W + k].contiguous(), query=xq)
    if k in (19, 38):
        torch.cuda.synchronize()
        for rep in range(3):
            g.posterior(xq)
        torch.cuda.synchronize()
print("t", g.t, "fails", int((g.info != 0).sum()))
