#!/usr/bin/env python3
"""The team-of-waves form of the refit (BCBF_REFIT_TEAM=1) against the library's default choice without it, element by
element, and its time against the other forms for a few large systems (development)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from _timing import timeit
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
def run(p, env):
    for k in ("BCBF_REFIT_TEAM", "BCBF_REFIT_WAVE", "BCBF_REFIT_PAIR"):
        os.environ.pop(k, None)
    os.environ.update(env)
    return ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])[:3]
ok = True
for DT in (torch.float64, torch.float32):
    for Bt, N, n, m in ((1, 512, 3, 2), (3, 500, 2, 1), (5, 700, 6, 3), (2, 1024, 3, 2), (1, 2048, 3, 2), (4, 40, 2, 1), (2, 100, 3, 2)):
        p = make_instances(Bt, N, n, m, dtype=DT, device="cuda", seed=21)
        if Bt >= 3:
            p["jitter"][1, N // 2] = -1.0                         # a failed pivot in one instance
        a = run(p, {"BCBF_REFIT_TEAM": "1"}); b = run(p, {"BCBF_REFIT_TEAM": "0"})
        torch.cuda.synchronize()
        good = b[2] == 0
        same = bool((a[2] == b[2]).all())
        dL = ((a[0][good] - b[0][good]).abs().max() / b[0][good].abs().max()).item() if bool(good.any()) else 0.0
        print(str(DT)[6:], Bt, N, n, m, "info equal", same, "fails", int((~good).sum()), "max |dL| / max |L| %.2e" % dL, "UHB equal", bool(torch.equal(a[1], b[1])))
        ok &= same and bool(torch.equal(a[1], b[1])) and dL <= (1e-8 if DT == torch.float64 else 5e-3)
print("ALL WITHIN TOLERANCE" if ok else "DIFFERENCES")
for DT in (torch.float64, torch.float32):
    for N in (512, 1024, 2048):
        for Bt in (1, 8, 32, 64, 128):
            if Bt * N * N * 8 > 3e9:
                continue
            p = make_instances(Bt, N, 3, 2, dtype=DT, device="cuda", seed=5)
            t = {}
            for name, env in (("team", {"BCBF_REFIT_TEAM": "1"}), ("workgroup", {"BCBF_REFIT_WAVE": "0", "BCBF_REFIT_PAIR": "0"}), ("wave", {"BCBF_REFIT_WAVE": "1", "BCBF_REFIT_PAIR": "0"})):
                t[name] = timeit(lambda: run(p, env), reps=5)
            print(str(DT)[6:], "N", N, "batch", Bt, " ".join("%s %.3f" % kv for kv in t.items()))
