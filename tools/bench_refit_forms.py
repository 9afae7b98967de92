#!/usr/bin/env python3
"""refit: the workgroup-per-instance form against the one-wave-per-instance form (BCBF_REFIT_WAVE), the
two-waves-per-instance form (BCBF_REFIT_PAIR) and the team-of-eight-waves form (BCBF_REFIT_TEAM), same inputs.

    python tools/bench_refit_forms.py [f64|f32]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import timeit

DT = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == 'f32') else torch.float64
GRID = [(Bt, N) for N in (128, 256, 512, 1024) for Bt in (1, 64, 128, 256, 512, 1024, 4096) if Bt * N * N * 8 * 0.6 < 24e9]
for Bt, N in GRID:
    n, m = (2, 1) if N <= 256 else (3, 2)
    p = make_instances(Bt, N, n, m, dtype=DT, device="cuda", seed=5)
    row = dict(batch=Bt, N=N, dtype=str(DT)[6:])
    for form, name in (("0", "workgroup"), ("1", "wave"), ("pair", "pair"), ("team", "team")):
        for k in ("BCBF_REFIT_WAVE", "BCBF_REFIT_PAIR", "BCBF_REFIT_TEAM"):
            os.environ.pop(k, None)
        if form == "team":
            if Bt > 1024:
                continue
            os.environ["BCBF_REFIT_TEAM"] = "1"
        else:
            os.environ["BCBF_REFIT_WAVE"] = "1" if form == "pair" else form
            os.environ["BCBF_REFIT_PAIR"] = "1" if form == "pair" else "0"
        t = timeit(lambda: ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"]), reps=5)
        info = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])[2]
        row["ms_" + name] = t
        row["fail_" + name] = int((info != 0).sum())
        row["TFLOPs_" + name] = Bt * N ** 3 / 3.0 / (t * 1e-3) / 1e12
    for k in ("BCBF_REFIT_WAVE", "BCBF_REFIT_PAIR", "BCBF_REFIT_TEAM"):
        os.environ.pop(k, None)
    t = timeit(lambda: ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"]), reps=5)
    row["ms_default"] = t
    print(json.dumps(row), flush=True)
