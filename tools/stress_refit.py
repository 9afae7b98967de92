#!/usr/bin/env python3
"""Soak test of the hand-off protocols of the two-wave / team refit kernels (spin waits on LDS counters): thousands of
launches over random shapes, forms and failure positions; every launch is synchronised, so a lost wake-up shows up as a
hang (run under `timeout`).  Development tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
t0 = time.time(); done = 0
cache = {}
while done < launches:
    DT = torch.float64 if rng.random() < 0.5 else torch.float32
    N = int(rng.choice([1, 31, 32, 33, 64, 100, 160, 256, 300, 512, 700, 1024]))
    Bt = int(rng.choice([1, 2, 7, 64, 200, 256, 300, 512, 700, 1024])) if N <= 512 else int(rng.choice([1, 2, 7, 64, 130]))
    n, m = [(2, 1), (3, 2), (6, 3)][int(rng.integers(3))]
    key = (DT, N, Bt, n, m)
    if key not in cache:
        if len(cache) > 6:
            cache.clear()
        cache[key] = make_instances(Bt, N, n, m, dtype=DT, device="cuda", seed=int(rng.integers(1000)))
    p = cache[key]
    jit = p["jitter"].clone()
    for _ in range(int(rng.integers(0, 4))):                      # failed pivots at random places in random instances
        jit[int(rng.integers(Bt)), int(rng.integers(N))] = -10.0
    form = rng.choice(["default", "pair", "team8", "team4", "wave"])
    for k in ("BCBF_REFIT_TEAM", "BCBF_REFIT_WAVE", "BCBF_REFIT_PAIR"):
        os.environ.pop(k, None)
    if form == "pair" and N <= 512:
        os.environ["BCBF_REFIT_WAVE"] = "1"; os.environ["BCBF_REFIT_PAIR"] = "1"
    elif form == "team8":
        os.environ["BCBF_REFIT_TEAM"] = "18"
    elif form == "team4":
        os.environ["BCBF_REFIT_TEAM"] = "14"
    elif form == "wave":
        os.environ["BCBF_REFIT_WAVE"] = "1"; os.environ["BCBF_REFIT_PAIR"] = "0"
    for rep in range(int(rng.integers(1, 6))):
        Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], jit, want_dense=bool(rng.random() < 0.15 and N <= 300))
        torch.cuda.synchronize()
        done += 1
print("ok: %d launches in %.1f s" % (done, time.time() - t0))
