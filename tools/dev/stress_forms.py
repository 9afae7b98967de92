"""Development: the LDS-DMA forms at full occupancy and an odd batch against the forms they replaced, element by element (each form in its own process:
the library reads its switches once).  python tools/dev/stress_forms.py [worker <tag>]"""
import os, sys, subprocess, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out", "stress")


def worker(tag):
    from bayesian_cbf_amd import ops
    from bayesian_cbf_amd._lib import lib
    from bayesian_cbf_amd.synthetic import make_instances
    res = {}
    for dt in (torch.float32, torch.float64):
        Bt, N = 4099, 512
        p = make_instances(Bt, N, 3, 2, dtype=dt, device="cuda", seed=11)
        Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"] * 100)
        Linv = torch.empty(Bt, N, N, dtype=dt, device="cuda"); Kinv = torch.empty_like(Linv)
        for rep in range(3):
            ops.check(getattr(lib, "bcbf_trtri" + ops._suf(Lop))(ops._p(Lop), ops._p(Linv), Bt, N, ops._stream(Lop)), "trtri")
            ops.check(getattr(lib, "bcbf_syrk_lt" + ops._suf(Lop))(ops._p(Linv), ops._p(Kinv), Bt, N, ops._stream(Lop)), "syrk")
        R = torch.randn(Bt, N, 3, dtype=dt, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
        al = ops.kinv_apply(Kinv, R)
        good = info == 0
        res[str(dt)] = dict(Linv=Linv[good][::97].double().cpu(), Kinv=Kinv[good][::97].double().cpu(), alpha=al[good][::97].double().cpu(), fails=int((~good).sum()))
        if dt == torch.float32:
            Vw, _ = ops.potrs(Lop, p["Xdot"], p["UH"], p["M0"], want_alpha=False)
            for rep in range(3):
                Mk, Bk, G, Mj = ops.posterior_jets(Lop, Vw, p["X"], UHB, p["ell"], p["s2"], p["Bm"], p["M0"], p["xq"])
            res["jets"] = dict(Mk=Mk[good][::97].double().cpu(), G=G[good][::97].double().cpu(), Mj=Mj[good][::97].double().cpu())
    torch.cuda.synchronize()
    os.makedirs(OUT, exist_ok=True)
    torch.save(res, os.path.join(OUT, tag + ".pt"))


if len(sys.argv) > 2 and sys.argv[1] == "worker":
    worker(sys.argv[2])
else:
    envs = {"new": {}, "old": {"BCBF_TRTRI_DMA": "0", "BCBF_SYRK_TILE": "0", "BCBF_JETS_MFMA": "0"}}
    for tag, e in envs.items():
        subprocess.run([sys.executable, os.path.abspath(__file__), "worker", tag], env=dict(os.environ, **e), check=True)
    a, b = torch.load(os.path.join(OUT, "new.pt")), torch.load(os.path.join(OUT, "old.pt"))
    ok = True
    for k in a:
        for name in a[k]:
            if name == "fails":
                print(k, "fails", a[k][name], b[k][name]); continue
            x, y = a[k][name], b[k][name]
            err = float((x - y).abs().max() / y.abs().max())
            tol = 5e-3 if ("float32" in k or k == "jets") else 1e-9
            print("%-16s %-6s max rel diff %.3e  finite %s" % (k, name, err, bool(torch.isfinite(x).all())))
            ok &= err < tol and bool(torch.isfinite(x).all())
    print("OK" if ok else "DIFFERENCES")
