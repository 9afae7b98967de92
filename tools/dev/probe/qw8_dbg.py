import os, sys, torch, numpy as np, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
N, n, m = 256, 2, 1
p = make_instances(1, N, n, m, dtype=torch.float32, device="cuda", seed=41 + N)
cus = torch.cuda.get_device_properties(0).multi_processor_count
b = 16 * cus + 503
g = torch.Generator(device="cpu").manual_seed(7)
xq = (p["X"][0, torch.randint(0, N, (b,), generator=g).cuda()] + 0.3 * torch.randn(b, n, generator=g).cuda()).contiguous()
def run(dt):
    q = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in p.items()}
    if dt == torch.float64:
        Lop32, UHB32, _, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
    Lop, UHB, info, _ = ops.refit(q["X"], q["UH"], q["Bm"], q["ell"], q["s2"], q["jitter"])
    Vw, _ = ops.potrs(Lop, q["Xdot"], q["UH"], q["M0"])
    return ops.posterior_shared(Lop, Vw, q["X"], UHB, q["ell"], q["s2"], q["Bm"], q["M0"], xq.to(dt), want_W=True)
Mk, Bk, W = run(torch.float32)
Mk64, Bk64, W64 = run(torch.float64)
e = (Mk.double() - Mk64).abs().amax(dim=(1, 2))
print("QW env", os.environ.get("BCBF_PSR_QW"), "max err vs fp64", float(e.max()), "at query", int(e.argmax()), "median", float(e.median()), "count > 3e-4:", int((e > 3e-4).sum()))
np.save("/tmp/mk_%s.npy" % os.environ.get("BCBF_PSR_QW", "auto"), Mk.cpu().numpy())
