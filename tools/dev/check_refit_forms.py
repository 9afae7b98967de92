#!/usr/bin/env python3
"""The two-waves-per-instance form of the refit (BCBF_REFIT_PAIR=1) against the one-wave form, element by element (development):
the packed operator incl. the inverted diagonal blocks, relative to its largest element."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
ok = True
for DT in (torch.float64, torch.float32):
    for Bt, N, n, m in ((64, 32, 2, 1), (64, 64, 2, 1), (300, 100, 3, 2), (1024, 256, 2, 1), (130, 250, 4, 2), (256, 512, 3, 2), (70, 40, 6, 3)):
        p = make_instances(Bt, N, n, m, dtype=DT, device="cuda", seed=7)
        if Bt == 300:                                             # a failed pivot in some instances
            p["jitter"][::7] = -1.0
        out = {}
        for form in ("0", "1"):
            os.environ["BCBF_REFIT_WAVE"] = "1"; os.environ["BCBF_REFIT_PAIR"] = form
            Lop, UHB, info = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])[:3]
            torch.cuda.synchronize()
            out[form] = (Lop.clone(), UHB.clone(), info.clone())
        good = out["0"][2] == 0
        for form in ("1",):
            same_info = bool((out[form][2] == out["0"][2]).all())
            dL = ((out[form][0][good] - out["0"][0][good]).abs().max() / out["0"][0][good].abs().max()).item() if bool(good.any()) else 0.0
            dU = (out[form][1] - out["0"][1]).abs().max().item()
            print(str(DT)[6:], Bt, N, n, m, "form", form, "info equal", same_info, "fails", int((~good).sum()), "max |dL| / max |L|", dL, "max |dUHB|", dU)
            # (fp32 at this conditioning: a few pivots sit at the rounding level, which instances fail depends on the order of
            #  the additions -- reported, not an error; the posterior comparison below is the fp32 check)
            ok &= (same_info and dL <= 1e-8 and dU == 0.0) if DT == torch.float64 else dU == 0.0
# fp32: which form is closer to the fp64 posterior built from the same data?  (ill-conditioned K_b: element-wise differences
# between fp32 factors say little)
for Bt, N, n, m in ((256, 64, 2, 1), (256, 256, 2, 1), (128, 512, 3, 2)):
    p = make_instances(Bt, N, n, m, dtype=torch.float32, device="cuda", seed=13)
    xq = p["X"][:, 0, :].contiguous() + 0.1
    res = {}
    for form, q in (("ref64", {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in p.items()}), ("0", p), ("1", p)):
        os.environ["BCBF_REFIT_WAVE"] = "1"; os.environ["BCBF_REFIT_PAIR"] = "0" if form == "ref64" else form
        Lop, UHB, info = ops.refit(q["X"], q["UH"], q["Bm"], q["ell"], q["s2"], q["jitter"])[:3]
        Vw = ops.potrs(Lop, q["Xdot"], q["UH"], q["M0"], want_alpha=False)
        Vw = Vw[0] if isinstance(Vw, tuple) else Vw
        Mk, Bk = ops.posterior_step(Lop, Vw, q["X"], UHB, q["ell"], q["s2"], q["Bm"], q["M0"], xq.to(q["X"].dtype))[:2]
        res[form] = (Mk.double(), Bk.double(), info)
    good = (res["0"][2] == 0) & (res["1"][2] == 0) & (res["ref64"][2] == 0)
    for form in ("0", "1"):
        eM = (res[form][0][good] - res["ref64"][0][good]).abs().max().item()
        eB = (res[form][1][good] - res["ref64"][1][good]).abs().max().item()
        print("fp32 posterior vs fp64, N", N, "form", form, "fails", int((res[form][2] != 0).sum()), "max |dMk|", eM, "max |dBk|", eB)
print("ALL WITHIN TOLERANCE" if ok else "DIFFERENCES")
