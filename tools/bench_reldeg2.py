#!/usr/bin/env python3
"""Rel-degree-2 path (cbc2_gp / GradientGP, cbc2.py:26-33, gp_algebra.py:319-402): posterior jets + closed-form
terms, batched over independent GPs.  Pendulum shape n=2, m=1 and the unicycle shape n=3, m=2."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import timeit          # warms the clocks up first (tools/_timing.py)

for (Bt, N, n, m, dtype) in ((4096, 512, 2, 1, torch.float32), (4096, 512, 3, 2, torch.float32), (1024, 256, 2, 1, torch.float64), (4096, 512, 2, 2, torch.float32), (4096, 512, 3, 1, torch.float32)):
    p = make_instances(Bt, N, n, m, dtype=dtype, device="cuda", seed=3)
    Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
    Vw, _ = ops.potrs(Lop, p["Xdot"], p["UH"], p["M0"], want_alpha=False)
    f = dict(dtype=dtype, device="cuda")
    hv, gh, Hh = torch.randn(Bt, **f), torch.randn(Bt, n, **f), torch.randn(Bt, n, n, **f)
    Hh = (Hh + Hh.transpose(1, 2)).contiguous()
    ka, u0 = torch.tensor([1.0, 2.0], **f), torch.rand(Bt, m, **f)
    jets = lambda: ops.posterior_jets(Lop, Vw, p["X"], UHB, p["ell"], p["s2"], p["Bm"], p["M0"], p["xq"])
    Mk, Bk, G, Mj = jets()
    t_j = timeit(jets)
    t_t = timeit(lambda: ops.cbc2_terms(Mk, Bk, G, Mj, p["A"], p["Bm"], p["ell"], p["s2"], hv, gh, Hh, ka, u0))
    t_v = timeit(lambda: ops.posterior_step(Lop, Vw, p["X"], UHB, p["ell"], p["s2"], p["Bm"], p["M0"], p["xq"]))
    isz = p["X"].element_size()
    by = Bt * isz * (N * (N + 1) // 2 + 2 * N * n + N * (1 + m))
    print(json.dumps(dict(batch=Bt, N=N, n=n, m=m, dtype=str(dtype), jets_ms=t_j, jets_GBs_algorithmic=by / (t_j * 1e-3) / 1e9,
                          values_only_ms=t_v, cbc2_terms_ms=t_t, rhs_columns=(1 + m) * (1 + n),
                          instance_steps_per_s=Bt / ((t_j + t_t) * 1e-3))))
