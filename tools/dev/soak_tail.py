"""Development: soak of the row-major tail against the in-place form -- many windows, random failing pivots."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
dev = torch.device("cuda")
for dtype, tol in ((torch.float64, 1e-8), (torch.float32, 2e-3)):
    Bt, n, m, W, D, steps = 64, 3, 2, 96, 24, 600
    p = make_instances(Bt, W + steps + 1, n, m, dtype=dtype, device=dev, seed=11)
    p["X"] = (p["X"] * 2.0).contiguous()
    cut = lambda t, N: t[:, :N].contiguous()
    jit0 = cut(p["jitter"], W)
    Lop, UHB, info, _ = ops.refit(cut(p["X"], W), cut(p["UH"], W), p["Bm"], p["ell"], p["s2"], jit0)
    assert (info == 0).all()
    Vw, _ = ops.potrs(Lop, cut(p["Xdot"], W), cut(p["UH"], W), p["M0"], want_alpha=False)
    mk = lambda tail: ops.ReservedGP(Lop, Vw, cut(p["X"], W), UHB, p["ell"], p["s2"], p["Bm"], p["M0"], W + D, window=W, drop=D,
                                     UH=cut(p["UH"], W), Xdot=cut(p["Xdot"], W), jitter=jit0, tail=tail)
    gt, gi = mk(True), mk(False)
    g = torch.Generator(device="cpu").manual_seed(5)
    prior = float((p["s2"][:, None, None] * p["Bm"]).abs().max())
    worst, nfail = 0.0, 0
    for t in range(steps):
        N = W + t
        xq = (p["xq"] + 0.003 * t).contiguous()
        x_new, uh_new, xd_new, j_new = (p[k][:, N].clone().contiguous() for k in ("X", "UH", "Xdot", "jitter"))
        bad = (torch.rand(Bt, generator=g) < 0.1).to(dev)
        live_row = int(torch.randint(0, gt.N, (1,), generator=g))
        x_new[bad] = gt.X[bad, live_row]
        uh_new[bad] = gt._rUH[bad, live_row]
        j_new[bad] = -(j_new[bad].abs() + (1e-3 if dtype == torch.float32 else 0.0))     # (fp32: clear of the rounding of l'l, where the two forms may differ in sign)
        it, Mt, Bt_ = gt.append(x_new, uh_new, xd_new, j_new, query=xq)
        ii, Mi, Bi = gi.append(x_new, uh_new, xd_new, j_new, query=xq)
        assert torch.equal(it, ii), (t, it.cpu().tolist(), ii.cpu().tolist())
        nfail += int((it != 0).sum())
        e = max(float((Mt - Mi).abs().max() / max(1.0, float(Mi.abs().max()))), float((Bt_ - Bi).abs().max()) / prior)
        worst = max(worst, e)
        assert e < tol, (t, e)
    print(dtype, "steps", steps, "drops", gt.drops, "failed pivots", nfail, "drop failures", gt.drop_failures, gi.drop_failures, "worst deviation %.2e" % worst)
