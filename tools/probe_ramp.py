#!/usr/bin/env python3
"""How long after the device has been idle does the bench step reach its steady rate?  (development probe)
Per-step device time of the default two-stream schedule for 120 steps, started (a) after 0.5 s of idleness, (b) directly
behind a refit of the whole batch (5 ms of matrix-core work), (c) after 0.05 s of idleness.  Prints the mean step time of
consecutive groups of 10 steps."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances, make_unicycle_task

dev = "cuda"
Bt, N = 4096, 512
p = make_instances(Bt, N, 3, 2, dtype=torch.float32, device=dev, seed=1234)
task = make_unicycle_task(Bt, dtype=torch.float32, device=dev, seed=99)
jit = p["jitter"]
for _ in range(4):
    Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], jit)
    bad = info != 0
    if not bool(bad.any()):
        break
    jit = torch.where(bad[:, None], jit * 10, jit).contiguous()
Vw, _ = ops.potrs(Lop, p["Xdot"], p["UH"], p["M0"], want_alpha=False)
gp = dict(Lop=Lop, Vw=Vw, X=p["X"], UHB=UHB, ell=p["ell"], s2=p["s2"], Bm=p["Bm"], M0=p["M0"], A=p["A"])
x = task["x"].clone()
loop = ops.ConcurrentControlLoop(gp, task, x, parts=2, dt=1e-3, L_true=1.0, L_mean=4.0, clf_gamma=10.0, max_iters=20)
K = 120
ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
for e in ev:
    e.record(loop.streams[0])
torch.cuda.synchronize()


def series(tag, before):
    before()
    ev[0].record(loop.streams[0])
    for s in range(K):
        loop.step()
        loop.streams[0].wait_stream(loop.streams[1])
        ev[s + 1].record(loop.streams[0])
    torch.cuda.synchronize()
    t = np.array([ev[0].elapsed_time(e) for e in ev[1:]])
    per = np.diff(np.concatenate([[0.0], t]))
    print(tag, " ".join("%.3f" % per[i:i + 10].mean() for i in range(0, K, 10)))


series("idle 0.5 s      ", lambda: (torch.cuda.synchronize(), time.sleep(0.5)))
series("behind a refit  ", lambda: ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], jit))
series("idle 0.05 s     ", lambda: (torch.cuda.synchronize(), time.sleep(0.05)))
series("idle 0.005 s    ", lambda: (torch.cuda.synchronize(), time.sleep(0.005)))
series("back to back    ", lambda: None)
