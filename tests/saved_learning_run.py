"""Shared set-up of the tests that pin the path against the committed learning run of the reference
(docs/saved-runs/..._learning_helps_avoid_getting_stuck_v1.6.3-1-g5fa08e8, written with the REAL gpytorch; fixture
tests/golden/saved_run_learning_v1p6p3.npz, extracted by tests/golden/extract_saved_runs.py).

What the log holds per control step t: state x_t, control u_t, the fitted kernel parameters as the reference's
accessors return them, `Fx_var` = custom_predict_fullmat([0, 0, theta_t])[1] = kron(B_k + jitter, A)  [9, 9]  and
`Fxu_var` = fu_func_gp(u_t).knl(x_t, x_t) = (uh' (B_k(x_t) + jitter) uh) A  [3, 3]  (unicycle_move_to_pose.py:970-982; the
first is queried at the shift-invariant state, the second at the raw state, :388-397, :413).  The model is refitted at
t = 40, 80, 120, 160 on the samples i < t - 1 (finite-difference targets, :340-349): X_i = [0, 0, theta_i], U_i = u_i.
The unknown 1e-5 rand jitters of make_psd (:907-910, on K_b and on B_k) bound what the log can pin after a refit."""
import os

import numpy as np

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "saved_run_learning_v1p6p3.npz"))
PRIOR_STEPS = (0, 1, 17, 39, 40)                  # before the first refit takes effect: the prior
POSTERIOR_STEPS = (41, 60, 80, 81, 119, 121, 150, 161, 199)


def hyper(t):
    return dict(A=G["knl_A"][t].astype(np.float64), B=G["knl_B"][t].astype(np.float64),
                ell=G["knl_lengthscale"][t].astype(np.float64), s2=float(G["knl_scalefactor"][t]))


def training_set(t):
    """(X[N,3], U[N,2]) the model holds while it answers step t (None before the first refit)."""
    every = int(G["train_every_n_steps"])
    last = (t - 1) // every * every                # the refit happens at the END of step `last`
    if last < every:
        return None
    N = last - 1
    X = np.zeros((N, 3))
    X[:, 2] = G["state"][:N, 2]
    return X, G["uopt"][:N].astype(np.float64)


def queries(t):
    x = G["state"][t].astype(np.float64)
    return np.array([0.0, 0.0, x[2]]), x, np.r_[1.0, G["uopt"][t].astype(np.float64)]
