"""Where one `custom_predict_fullmat(Xtest); clear_cache()` call of the published speed test goes (development)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bayesian_cbf_amd.control_affine_model import ControlAffineRegressorExact, ControlAffineRegressorVector
import math
from bayesian_cbf_amd.pendulum import PendulumDynamicsModel, ControlRandom, sampling_pendulum_data
from torch.profiler import profile, ProfilerActivity
torch.manual_seed(0)
dX, X, U = (a.numpy() for a in sampling_pendulum_data(PendulumDynamicsModel(m=1, n=2), D=2000, dt=0.01,
                                                       x0=torch.tensor([5 * math.pi / 6, -0.01]),
                                                       controller=ControlRandom(mass=1, gravity=10, length=1).control))
for cls in (ControlAffineRegressorExact, ControlAffineRegressorVector):
    for N in (256, 512):
        idx = np.random.default_rng(1).permutation(len(X) - 1)[:N]
        t = lambda a: torch.as_tensor(a[idx], dtype=torch.float32, device="cuda")
        th = np.linspace(X[idx, 0].min(), X[idx, 0].max(), 20); om = np.linspace(X[idx, 1].min(), X[idx, 1].max(), 20)
        Xtest = torch.as_tensor(np.stack(np.meshgrid(th, om), -1).reshape(-1, 2), dtype=torch.float32, device="cuda")
        dgp = cls(2, 1, device="cuda", dtype=torch.float32)
        dgp.fit(t(X), t(U), t(dX), training_iter=20)
        def call():
            dgp.custom_predict_fullmat(Xtest); dgp.clear_cache(); torch.cuda.synchronize()
        for _ in range(5): call()
        t0 = time.perf_counter()
        for _ in range(50): call()
        print(cls.__name__, "N", N, "ms per call %.3f" % ((time.perf_counter() - t0) / 50 * 1e3))
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            for _ in range(10): call()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=7, max_name_column_width=50))
