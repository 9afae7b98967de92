#!/usr/bin/env python3
"""The four-waves-per-instance, LDS-staged form of the fp32 refit (BCBF_REFIT_SLAB=1, refit_slab.hip) against the form the launcher
picks without it: the packed operator element by element, both posteriors against the fp64 posterior of the same data, timings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances

def run(p, slab):
    if slab: os.environ["BCBF_REFIT_SLAB"] = "1"
    else: os.environ.pop("BCBF_REFIT_SLAB", None)
    out = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])[:3]
    torch.cuda.synchronize()
    return [o.clone() for o in out]

ok = True
for Bt, N, n, m in ((64, 512, 3, 2), (300, 500, 3, 2), (130, 448, 4, 2), (70, 384, 2, 1), (33, 490, 3, 3), (512, 256, 2, 1), (40, 128, 2, 1)):
    p = make_instances(Bt, N, n, m, dtype=torch.float32, device="cuda", seed=7)
    if Bt == 300: p["jitter"][::7] = -1.0
    a, s = run(p, False), run(p, True)
    good = (a[2] == 0) & (s[2] == 0)
    if not bool(good.any()):
        print(Bt, N, 'slab info', s[2][:8].tolist(), 'default info', a[2][:8].tolist()); ok = False; continue
    dL = ((s[0][good] - a[0][good]).abs().max() / a[0][good].abs().max()).item()
    dU = (s[1] - a[1]).abs().max().item()
    nanL = int(torch.isnan(s[0][good]).sum())
    print(Bt, N, n, m, "fails", int((a[2] != 0).sum()), int((s[2] != 0).sum()), "info equal", bool((a[2] == s[2]).all()), "max |dL| / max |L|", dL, "nan", nanL, "dUHB", dU)
    ok &= dU == 0.0 and nanL == 0
    q = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in p.items()}
    xq = p["X"][:, 0, :].contiguous() + 0.1
    res = {}
    for form, d, slab in (("ref64", q, False), ("default", p, False), ("slab", p, True)):
        Lop, UHB, info = run(d, slab)
        Vw = ops.potrs(Lop, d["Xdot"], d["UH"], d["M0"], want_alpha=False)
        Vw = Vw[0] if isinstance(Vw, tuple) else Vw
        Mk, Bk = ops.posterior_step(Lop, Vw, d["X"], UHB, d["ell"], d["s2"], d["Bm"], d["M0"], xq.to(d["X"].dtype))[:2]
        res[form] = (Mk.double(), Bk.double(), info)
    g = (res["default"][2] == 0) & (res["slab"][2] == 0) & (res["ref64"][2] == 0)
    for form in ("default", "slab"):
        eM = (res[form][0][g] - res["ref64"][0][g]).abs().max().item()
        eB = (res[form][1][g] - res["ref64"][1][g]).abs().max().item()
        print("   posterior vs fp64:", form, "max |dMk|", eM, "max |dBk|", eB)
    ok &= (res["slab"][0][g] - res["ref64"][0][g]).abs().max().item() <= 3 * (res["default"][0][g] - res["ref64"][0][g]).abs().max().item() + 1e-5
print("OK" if ok else "DIFFERENCES")
for Bt, N in ((4096, 512), (1024, 512), (2048, 512), (4096, 384), (8192, 512)):
    p = make_instances(Bt, N, 3, 2, dtype=torch.float32, device="cuda", seed=3)
    for slab in (False, True):
        run(p, slab); run(p, slab)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
        e1.record(); torch.cuda.synchronize()
        print("time", Bt, N, "slab" if slab else "default", "%.3f ms" % (e0.elapsed_time(e1) / 10))
