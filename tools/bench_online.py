#!/usr/bin/env python3
"""BASELINE configs[4] on one GPU: online GP growth N0 -> N1 with bcbf_gp_append (no refactorisation), one
posterior + SOCP control step per observation.  Prints per-N-range timings and the end-to-end check against a
from-scratch refit of the final data set."""
import os, sys, json, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesian_cbf_amd.rollouts import online_gp_growth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--n0", type=int, default=128)
ap.add_argument("--n1", type=int, default=2048)
ap.add_argument("--dtype", choices=["f32", "f64"], default="f64")
ap.add_argument("--no-control", action="store_true")
ap.add_argument("--packed", action="store_true", help="the packed layout of exactly N points (copy per append) instead of reserved storage")
ap.add_argument("--repeat", type=int, default=1, help="runs (reproducibility of the per-segment figures)")
a = ap.parse_args()
for _ in range(a.repeat):
    out = online_gp_growth(a.batch, a.n0, a.n1, dtype=torch.float64 if a.dtype == "f64" else torch.float32,
                           with_control=not a.no_control, reserved=not a.packed)
    print(json.dumps(out), flush=True)
