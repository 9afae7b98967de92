#!/usr/bin/env python3
"""A few launches of the batched refit, nothing else (development: the program under rocprofv3 --pmc).

    python3 tools/refit_only.py [f32|f64] [batch] [N] [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
DT = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else torch.float64
Bt = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
N = int(sys.argv[3]) if len(sys.argv) > 3 else 256
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
n, m = (2, 1) if N <= 256 else (3, 2)
p = make_instances(Bt, N, n, m, dtype=DT, device="cuda", seed=5)
for _ in range(reps):
    ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
torch.cuda.synchronize()
