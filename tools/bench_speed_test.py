#!/usr/bin/env python3
"""The reference's published speed test (pendulum.py:1305-1394, `speed_test_matrix_vector`), all four regressors
(MVGP full / diag, CoGP full / diag): pendulum n=2, m=1, N_train in {256, 320, 384, 512}, fit(training_iter=50), then
min(timeit.repeat('dgp.custom_predict_fullmat(Xtest); dgp.clear_cache()', repeat=5, number=50)) / 50 on a 20x20
(theta, omega) grid.  Every call is followed by a device synchronize (the reference's timing is host side).
Published values (unknown 2020 GPU) are in PUBLISHED below."""
import json, math, os, sys, time, timeit
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bayesian_cbf_amd.control_affine_model import (ControlAffineRegressorExact, ControlAffineRegMatrixDiag,
                                                   ControlAffineRegressorVector, ControlAffineRegVectorDiag)

PUBLISHED = {"matrix": {256: 0.0436, 320: 0.0453, 384: 0.0503, 512: 0.0775},
             "matrixdiag": {256: 0.0331, 320: 0.0363, 384: 0.0417, 512: 0.0511},
             "vector": {256: 0.0643, 320: 0.0865, 384: 0.1168, 512: 0.1915},
             "vectordiag": {256: 0.0590, 320: 0.0818, 384: 0.1123, 512: 0.1786}}


def pendulum_data(D=2000, tau=0.01, theta0=5 * math.pi / 6, omega0=-0.01, mass=1.0, gravity=10.0, length=1.0, seed=0):
    rng = np.random.default_rng(seed)
    x = np.array([theta0, omega0])
    X, U, dX = [], [], []
    for t in range(D):
        u = rng.uniform(-1, 1, 1) * mass * gravity * length          # random torque of the size of the gravity torque
        dx = np.array([x[1], -gravity / length * math.sin(x[0]) + u[0] / (mass * length ** 2)])
        X.append(x.copy()); U.append(u); dX.append(dx)
        x = x + tau * dx
    return np.array(X), np.array(U), np.array(dX)


def main():
    dtype = torch.float64 if "--f64" in sys.argv else torch.float32
    dev = "cuda"
    X, U, dX = pendulum_data()
    order = np.random.default_rng(1).permutation(len(X) - 1)
    out = []
    for cls_name, cls in (("matrix", ControlAffineRegressorExact), ("matrixdiag", ControlAffineRegMatrixDiag),
                          ("vector", ControlAffineRegressorVector), ("vectordiag", ControlAffineRegVectorDiag)):
        for N in (256, 320, 384, 512):
            idx = order[:N]
            t = lambda a: torch.as_tensor(a[idx], dtype=dtype, device=dev)
            Xtr, Utr, dXtr = t(X), t(U), t(dX)
            th = np.linspace(X[idx, 0].min(), X[idx, 0].max(), 20)
            om = np.linspace(X[idx, 1].min(), X[idx, 1].max(), 20)
            Xtest = torch.as_tensor(np.stack(np.meshgrid(th, om), -1).reshape(-1, 2), dtype=dtype, device=dev)
            torch.manual_seed(0)
            dgp = cls(2, 1, device=dev, dtype=dtype)
            t0 = time.perf_counter()
            dgp.fit(Xtr, Utr, dXtr, training_iter=50)
            torch.cuda.synchronize()
            fit_s = time.perf_counter() - t0

            def call():
                dgp.custom_predict_fullmat(Xtest)
                dgp.clear_cache()
                torch.cuda.synchronize()
            call()
            elapsed = min(timeit.repeat(call, repeat=5, number=50)) / 50
            hold = order[N:N + 300]                                   # held-out samples of the same trajectory
            th_ = lambda a: torch.as_tensor(a[hold], dtype=dtype, device=dev)
            pred, _ = dgp.custom_predict(th_(X), th_(U), compute_cov=False)
            err = float(np.sqrt(((pred.cpu().numpy() - dX[hold]) ** 2).mean()) / np.sqrt((dX[hold] ** 2).mean()))
            out.append(dict(regressor=cls_name, N=N, s_per_call=elapsed, published_s_per_call=PUBLISHED[cls_name][N],
                            speedup_vs_published=PUBLISHED[cls_name][N] / elapsed, fit_s=fit_s, heldout_rel_rms_err=err,
                            fit_loss_first_last=[dgp.fit_losses[0], dgp.fit_losses[-1]]))
            print(json.dumps(out[-1]))
    return out


if __name__ == "__main__":
    main()
