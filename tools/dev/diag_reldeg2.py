"""Development: why tools/bench_reldeg2.py stopped coming back (round 5): each stage under a watchdog print."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
t0 = time.time()
def say(*a):
    print("%.1fs" % (time.time() - t0), *a, flush=True)
Bt, N, n, m, dtype = 4096, 512, 3, 2, torch.float32
p = make_instances(Bt, N, n, m, dtype=dtype, device="cuda", seed=3)
Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
torch.cuda.synchronize(); say("refit", int((info != 0).sum()), "failed")
Vw, _ = ops.potrs(Lop, p["Xdot"], p["UH"], p["M0"], want_alpha=False)
torch.cuda.synchronize(); say("potrs", bool(torch.isfinite(Vw).all()))
f = dict(dtype=dtype, device="cuda")
Mk, Bk, G, Mj = ops.posterior_jets(Lop, Vw, p["X"], UHB, p["ell"], p["s2"], p["Bm"], p["M0"], p["xq"])
torch.cuda.synchronize(); say("jets finite:", bool(torch.isfinite(G).all()), bool(torch.isfinite(Mj).all()), bool(torch.isfinite(Bk).all()))
hv, gh, Hh = torch.randn(Bt, **f), torch.randn(Bt, n, **f), torch.randn(Bt, n, n, **f)
Hh = (Hh + Hh.transpose(1, 2)).contiguous()
ka, u0 = torch.tensor([1.0, 2.0], **f), torch.rand(Bt, m, **f)
ok = torch.isfinite(G).all(dim=(1, 2)) & torch.isfinite(Mj).all(dim=(1, 2))
say("instances with finite jets:", int(ok.sum()), "of", Bt)
sel = ok.nonzero().flatten()[:64]
out = ops.cbc2_terms(Mk[sel], Bk[sel], G[sel], Mj[sel], p["A"][sel], p["Bm"][sel], p["ell"][sel], p["s2"][sel], hv[sel], gh[sel], Hh[sel], ka, u0[sel])
torch.cuda.synchronize(); say("terms on 64 finite instances: status", out[4].unique().tolist())
