#!/usr/bin/env python3
"""The reference's unicycle speed test (`unicycle_speed_test_matrix_vector_exp`, unicycle_move_to_pose.py:2031-2152):
matrix-variate full / diag and vector-variate full / diag regressors (9 task outputs) inside
LearnedShiftInvariantDynamics, max_train in {64, 80, 96, 128}, fit 50 iterations, then
min(timeit.repeat('model.custom_predict_fullmat(Xtest); model.clear_cache()')) on 20 headings.  One JSON line per
(regressor, max_train).   python tools/bench_speed_test_unicycle.py [--quick]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bayesian_cbf_amd import unicycle_move_to_pose as ump

quick = "--quick" in sys.argv
np.random.seed(0)
torch.manual_seed(0)
out = ump.unicycle_speed_test_matrix_vector_exp(repeat=3 if quick else 10, ntimes=5 if quick else 10,
                                                errorbartries=2 if quick else 5,
                                                max_train_variations=(64, 128) if quick else (64, 80, 96, 128))
for name, rows in out.items():
    for N, d in rows.items():
        print(json.dumps(dict(regressor=name, max_train=N, ms_per_call=1e3 * d["elapsed"],
                              prior_error_mean=float(np.mean(d["errors"])), prior_error_std=float(np.std(d["errors"])))), flush=True)
