"""Micro-benchmark of regime S (b queries of one GP): MFMA kernel vs the streaming kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances

N, n, m = 512, 3, 2
dev = "cuda"
p = make_instances(1, N, n, m, dtype=torch.float32, device=dev, seed=1)
Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
Vw, _ = ops.potrs(Lop, p["Xdot"], p["UH"], p["M0"])
for b in (8, 512, 4096, 16384, 65536):
    xq = (p["X"][0, torch.randint(0, N, (b,), device=dev)] + 0.3 * torch.randn(b, n, device=dev)).contiguous()
    for name, fn in (("mfma", lambda: ops.posterior_shared(Lop, Vw, p["X"], UHB, p["ell"], p["s2"], p["Bm"], p["M0"], xq)),
                     ("stream", lambda: ops.lib.bcbf_posterior_step_f32 and ops.posterior_query(
                         Lop, Vw, p["X"], UHB, p["ell"], p["s2"], p["Bm"], p["M0"], xq[:min(b, 15)], shared=True))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        nb = b if name == "mfma" else min(b, 15)
        flops = nb * 4 * N * N  # 4 columns x N^2 (triangular solve, 2 flop per MAC, half the matrix)
        print("b=%6d %-6s %9.1f us  %8.2f Mquery/s  %6.2f TFLOP/s(useful cols incl. pad)" % (nb, name, us, nb / us, flops / us * 1e-6))
