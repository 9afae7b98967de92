#!/usr/bin/env python3
"""The reference's published speed test through the façade recipe of the same name
(`bayesian_cbf_amd.pendulum.speed_test_matrix_vector_exp` = pendulum.py:1305-1394): all four regressors (MVGP full / diag,
CoGP full / diag), pendulum n=2, m=1, N_train in {256, 320, 384, 512}, fit(training_iter=50), then
min(timeit.repeat('dgp.custom_predict_fullmat(Xtest); dgp.clear_cache()', repeat=5, number=50)) / 50 on a 20x20
(theta, omega) grid, every call followed by a device synchronize.  Published values (unknown 2020 GPU) are
in PUBLISHED.    python tools/bench_speed_test.py [--f64] [--quick]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bayesian_cbf_amd.pendulum import speed_test_matrix_vector_exp

PUBLISHED = {"matrix": {256: 0.0436, 320: 0.0453, 384: 0.0503, 512: 0.0775},
             "matrixdiag": {256: 0.0331, 320: 0.0363, 384: 0.0417, 512: 0.0511},
             "vector": {256: 0.0643, 320: 0.0865, 384: 0.1168, 512: 0.1915},
             "vectordiag": {256: 0.0590, 320: 0.0818, 384: 0.1123, 512: 0.1786}}


def main():
    dtype = torch.float64 if "--f64" in sys.argv else torch.float32
    quick = "--quick" in sys.argv
    torch.manual_seed(0)
    np.random.seed(0)
    res = speed_test_matrix_vector_exp(errorbartries=2 if quick else 5, ntimes=10 if quick else 50, dtype=dtype)
    for name, per in res.items():
        for N, r in per.items():
            print(json.dumps(dict(regressor=name, N=N, s_per_call=r["elapsed"], published_s_per_call=PUBLISHED[name][N],
                                  speedup_vs_published=PUBLISHED[name][N] / r["elapsed"], fit_s=r["fit_s"],
                                  prior_model_error_mean=float(np.mean(r["errors"])), fit_loss_first_last=r["fit_loss_first_last"])))


if __name__ == "__main__":
    main()
