import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bayesian_cbf_amd.control_affine_model import ControlAffineRegressorExact
import math
from bayesian_cbf_amd.pendulum import PendulumDynamicsModel, ControlRandom, sampling_pendulum_data
torch.manual_seed(0)
dX, X, U = (a.numpy() for a in sampling_pendulum_data(PendulumDynamicsModel(m=1, n=2), D=2000, dt=0.01,
                                                       x0=torch.tensor([5 * math.pi / 6, -0.01]),
                                                       controller=ControlRandom(mass=1, gravity=10, length=1).control))
for N in (256, 512):
    idx = np.random.default_rng(1).permutation(len(X) - 1)[:N]
    t = lambda a: torch.as_tensor(a[idx], dtype=torch.float32, device="cuda")
    dgp = ControlAffineRegressorExact(2, 1, device="cuda", dtype=torch.float32)
    dgp.fit(t(X), t(U), t(dX), training_iter=3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dgp.fit(t(X), t(U), t(dX), training_iter=50)
    torch.cuda.synchronize(); print("N", N, "fit 50 iters: %.3f s" % (time.perf_counter() - t0))
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        dgp.fit(t(X), t(U), t(dX), training_iter=10)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=8, max_name_column_width=60))
