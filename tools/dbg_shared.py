import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
DEV = "cuda"
for (N, n, m, b) in [(256, 2, 1, 64), (1024, 3, 3, 37), (64, 1, 1, 5)]:
    p = make_instances(1, N, n, m, dtype=torch.float32, device=DEV, seed=3 + N)
    p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in p.items()}
    Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
    Vw, _ = ops.potrs(Lop, p["Xdot"], p["UH"], p["M0"])
    Lop64, UHB64, info64, _ = ops.refit(p64["X"], p64["UH"], p64["Bm"], p64["ell"], p64["s2"], p64["jitter"])
    Vw64, _ = ops.potrs(Lop64, p64["Xdot"], p64["UH"], p64["M0"])
    g = torch.Generator(device="cpu").manual_seed(11)
    idx = torch.randint(0, N, (b,), generator=g)
    xq = (p["X"][0, idx.to(DEV)] + 0.3 * torch.randn(b, n, generator=g).to(DEV, torch.float32)).contiguous()
    Mk, Bk, W = ops.posterior_shared(Lop, Vw, p["X"], UHB, p["ell"], p["s2"], p["Bm"], p["M0"], xq, want_W=True)
    outs = []
    for i in range(0, b, 8):
        outs.append(ops.posterior_query(Lop, Vw, p["X"], UHB, p["ell"], p["s2"], p["Bm"], p["M0"], xq[i:i + 8].contiguous(), shared=True, want_W=True))
    Mks = torch.cat([o[0] for o in outs]); Bks = torch.cat([o[1] for o in outs]); Ws = torch.cat([o[2] for o in outs])
    o64 = []
    for i in range(0, b, 8):
        o64.append(ops.posterior_query(Lop64, Vw64, p64["X"], UHB64, p64["ell"], p64["s2"], p64["Bm"], p64["M0"], xq[i:i + 8].double().contiguous(), shared=True, want_W=True))
    Mk64 = torch.cat([o[0] for o in o64]); Bk64 = torch.cat([o[1] for o in o64]); W64 = torch.cat([o[2] for o in o64])
    print(N, n, m, b, "mfma-vs-f64 Mk %.2e Bk %.2e W %.2e | stream-vs-f64 Mk %.2e Bk %.2e W %.2e | Vw max %.2e" % (
        (Mk - Mk64).abs().max(), (Bk - Bk64).abs().max(), (W - W64).abs().max(),
        (Mks - Mk64).abs().max(), (Bks - Bk64).abs().max(), (Ws - W64).abs().max(), Vw.abs().max()))
