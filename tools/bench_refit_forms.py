#!/usr/bin/env python3
"""refit: the workgroup-per-instance form against the one-wave-per-instance form (BCBF_REFIT_WAVE), same inputs.

    python tools/bench_refit_forms.py [f64|f32]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
from tools.bench_configs import timeit

DT = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == 'f32') else torch.float64
GRID = [(Bt, N) for N in (128, 256, 512, 1024) for Bt in (64, 128, 256, 512, 1024, 4096) if Bt * N * N * 8 * 0.6 < 24e9]
for Bt, N in GRID:
    n, m = (2, 1) if N <= 256 else (3, 2)
    p = make_instances(Bt, N, n, m, dtype=DT, device="cuda", seed=5)
    row = dict(batch=Bt, N=N, dtype=str(DT)[6:])
    for form in ("0", "1"):
        os.environ["BCBF_REFIT_WAVE"] = form
        t = timeit(lambda: ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"]), reps=5, warm=2)
        info = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])[2]
        row["ms_wave" if form == "1" else "ms_workgroup"] = t
        row["fail_" + form] = int((info != 0).sum())
        row["TFLOPs_wave" if form == "1" else "TFLOPs_workgroup"] = Bt * N ** 3 / 3.0 / (t * 1e-3) / 1e12
    print(json.dumps(row), flush=True)
