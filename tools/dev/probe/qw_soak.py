"""Development soak: regime-S queries over random shapes, the launcher's choice of queries per wave against four per wave
(BCBF_PSR_QW=4 in a second process), fp64 (differences must be at rounding level) and fp32."""
import os, sys, subprocess, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
if len(sys.argv) > 1:
    from bayesian_cbf_amd import ops
    from bayesian_cbf_amd.synthetic import make_instances
    rng = np.random.default_rng(123)
    out = {}
    for case in range(40):
        N = int(rng.integers(1, 513)); n = int(rng.integers(2, 5)); m = int(rng.integers(1, 3)); nq = int(rng.integers(4097, 12000))
        dt = torch.float64 if case % 2 == 0 else torch.float32
        p = make_instances(1, N, n, m, dtype=dt, device="cuda", seed=1000 + case)
        big = (p["jitter"] * 1e3).contiguous()            # well conditioned: the two forms must agree to rounding
        Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], big)
        if int(info[0]) != 0:
            continue
        Vw, _ = ops.potrs(Lop, p["Xdot"], p["UH"], p["M0"])
        g = torch.Generator(device="cpu").manual_seed(case)
        xq = (p["X"][0, torch.randint(0, N, (nq,), generator=g).cuda()] + 0.3 * torch.randn(nq, n, generator=g).cuda().to(dt)).contiguous()
        Mk, Bk, W = ops.posterior_shared(Lop, Vw, p["X"], UHB, p["ell"], p["s2"], p["Bm"], p["M0"], xq, want_W=True)
        out["Mk%d" % case], out["Bk%d" % case], out["W%d" % case] = Mk.cpu().numpy(), Bk.cpu().numpy(), W.cpu().numpy()
        out["meta%d" % case] = np.array([N, n, m, nq, case % 2])
    np.savez(sys.argv[1], **out)
else:
    subprocess.run([sys.executable, __file__, "/tmp/qw_auto.npz"], check=True)
    subprocess.run([sys.executable, __file__, "/tmp/qw_4.npz"], check=True, env=dict(os.environ, BCBF_PSR_QW="4"))
    a, b = np.load("/tmp/qw_auto.npz"), np.load("/tmp/qw_4.npz")
    worst = {0: 0.0, 1: 0.0}
    for k in a.files:
        if k.startswith("meta"):
            continue
        c = int(k.lstrip("MkBW"))
        f32 = int(a["meta%d" % c][4])
        err = np.abs(a[k] - b[k]).max() / max(1e-30, np.abs(b[k]).max())
        worst[f32] = max(worst[f32], err)
        if err > (1e-4 if f32 else 1e-11):
            print("MISMATCH", k, a["meta%d" % c], err)
    print("cases", sum(k.startswith("meta") for k in a.files), "worst relative difference fp64 %.2e fp32 %.2e" % (worst[0], worst[1]))
