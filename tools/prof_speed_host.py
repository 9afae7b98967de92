"""Host-side profile (cProfile) of one `custom_predict_fullmat(Xtest); clear_cache()` call of the published speed test (development)."""
import sys, os, time, torch, math, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesian_cbf_amd.control_affine_model import ControlAffineRegressorExact
from bayesian_cbf_amd.pendulum import PendulumDynamicsModel, ControlRandom, sampling_pendulum_data
torch.manual_seed(0)
dX, X, U = (a.numpy() for a in sampling_pendulum_data(PendulumDynamicsModel(m=1, n=2), D=2000, dt=0.01,
                                                       x0=torch.tensor([5 * math.pi / 6, -0.01]),
                                                       controller=ControlRandom(mass=1, gravity=10, length=1).control))
N = 512
idx = np.random.default_rng(1).permutation(len(X) - 1)[:N]
t = lambda a: torch.as_tensor(a[idx], dtype=torch.float32, device="cuda")
th = np.linspace(X[idx, 0].min(), X[idx, 0].max(), 20); om = np.linspace(X[idx, 1].min(), X[idx, 1].max(), 20)
Xtest = torch.as_tensor(np.stack(np.meshgrid(th, om), -1).reshape(-1, 2), dtype=torch.float32, device="cuda")
dgp = ControlAffineRegressorExact(2, 1, device="cuda", dtype=torch.float32)
dgp.fit(t(X), t(U), t(dX), training_iter=20)
def call():
    dgp.custom_predict_fullmat(Xtest); dgp.clear_cache(); torch.cuda.synchronize()
for _ in range(20): call()
t0 = time.perf_counter()
for _ in range(200): call()
print("ms per call %.3f" % ((time.perf_counter() - t0) / 200 * 1e3))
def nosync():
    dgp.custom_predict_fullmat(Xtest); dgp.clear_cache()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): nosync()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host-only ms per call %.3f, with final sync %.3f" % ((t1 - t0) / 200 * 1e3, (t2 - t0) / 200 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): nosync()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
