"""Mirror of the hot-path pieces of bayes_cbf/pendulum.py: the ground-truth pendulum model (:82-130) and the
rel-degree-2 radial barrier (:643-696) whose condition `cbc(u)` goes through the jet kernel + closed-form terms
(`cbc2.RelDeg2Safety`).  Controllers / plotting of the reference module are out of scope."""
import math

import torch

from .cbc2 import RelDeg2Safety


class PendulumDynamicsModel:
    """theta'' = -(g/l) sin(theta) + u / (m l):  f(x) = [omega, -(g/l) sin theta],  g(x) = [0, 1/(m l)]'  (:106-130)."""
    ground_truth = True

    def __init__(self, m=1, n=2, mass=1, gravity=10, length=1, deterministic=True, model_noise=0,
                 dtype=torch.get_default_dtype()):
        self.m, self.n, self.mass, self.gravity, self.length, self.dtype = m, n, mass, gravity, length, dtype

    def to(self, dtype):
        self.dtype = dtype

    @property
    def ctrl_size(self):
        return self.m

    @property
    def state_size(self):
        return self.n

    def f_func(self, X):
        theta, omega = X[..., 0:1], X[..., 1:2]
        return torch.cat([omega, -(self.gravity / self.length) * torch.sin(theta)], dim=-1)

    def g_func(self, x):
        gx = torch.tensor([[0.0], [1.0 / (self.mass * self.length)]], dtype=x.dtype, device=x.device)
        return gx.expand(*x.shape[:-1], 2, 1).clone() if x.ndim >= 2 else gx

    def F_func(self, X):
        return torch.cat([self.f_func(X).unsqueeze(-1), self.g_func(X)], dim=-1)

    # the plant protocol of sampling.DynamicsModel (sampling.py:32-47); the reference's pendulum loop
    # (`sampling_pendulum`, :164-233) inlines the same Euler update and theta wrap
    def set_init_state(self, x0):
        self.current_state = x0.clone()

    def step(self, u, dt):
        x = self.current_state
        xdot = self.f_func(x) + (self.g_func(x) @ u.to(x).unsqueeze(-1)).squeeze(-1)
        xn = x + xdot * dt
        xn[..., 0] = ((xn[..., 0] + math.pi) % (2 * math.pi)) - math.pi
        self.current_state = xn
        return dict(xdot=xdot, x=xn)


class RadialCBFRelDegree2(RelDeg2Safety):
    """h(x) = cos(delta_col) - cos(theta - theta_c)  (:675-696): keep the pendulum out of a cone around theta_c."""

    def __init__(self, model, cbf_col_gamma=1, _k_alpha=(1.0, 3.0), cbf_col_delta=math.pi / 8,
                 cbf_col_theta=math.pi / 4, theta_c=math.pi / 4, gamma_col=1, max_unsafe_prob=0.01,
                 delta_col=math.pi / 8, name="cbf-r2", dtype=torch.get_default_dtype()):
        self._model, self._max_unsafe_prob, self._k_alpha = model, max_unsafe_prob, list(_k_alpha)
        self.cbf_col_delta, self.cbf_col_theta, self.name, self.dtype = cbf_col_delta, cbf_col_theta, name, dtype

    k_alpha = property(lambda self: self._k_alpha)
    model = property(lambda self: self._model)
    max_unsafe_prob = property(lambda self: self._max_unsafe_prob)

    def cbf(self, x):
        return math.cos(self.cbf_col_delta) - torch.cos(x[..., 0] - self.cbf_col_theta)

    value = cbf

    def grad_cbf(self, X_in):
        X = X_in.unsqueeze(0) if X_in.ndim == 1 else X_in
        g = torch.cat((torch.sin(X[:, 0:1] - self.cbf_col_theta), X.new_zeros(X.shape[0], 1)), dim=-1)
        return g.squeeze(0) if X_in.ndim == 1 else g

    def hess_cbf(self, x):
        """d grad_cbf / dx (what GradientGP differentiates through, gp_algebra.py:340-345)."""
        H = x.new_zeros(2, 2)
        H[0, 0] = torch.cos(x[0] - self.cbf_col_theta)
        return H


# ----------------------------------------------------------------------------------------------------------------
# The learning demo of BASELINE configs[0]: pendulum.learn_dynamics_matrix_vector (pendulum.py:1052-1271).  Data come
# from one simulated trajectory under a randomised controller; both regressors are fitted on a random subset, evaluated
# on a 20 x 20 (theta, omega) grid with `custom_predict_fullmat`, logged in the reference's tags, and compared with the
# variance-weighted error of `measure_batch_error`.  Plotting is out of scope.
class ControlTrivial:
    """u = m g sin(theta)  (pendulum.py:55-66)."""
    needs_ground_truth = True

    def __init__(self, m=1, mass=None, length=None, gravity=None, dt=None, true_model=None):
        self.m, self.mass, self.length, self.gravity = m, mass, length, gravity

    def control(self, xi, t=None):
        return (self.mass * self.gravity * torch.sin(xi[0])).reshape(1)


class ControlRandom:
    """ControlTrivial scaled by U[0.6, 1.4)  (pendulum.py:69-78)."""
    needs_ground_truth = True

    def __init__(self, **kwargs):
        self.control_trivial = ControlTrivial(**kwargs)

    def control(self, xi, t=None):
        return self.control_trivial.control(xi, t=t) * (torch.rand(1) * 0.8 + 0.60).to(xi)


def sampling_pendulum(dynamics_model, numSteps, controller=None, x0=None, dt=0.01, plot_every_n_steps=20, axs=None,
                      visualizer=None, visualizer_class=None, plotfile=None):
    """pendulum.py:164-233: numSteps controller calls along one explicit-Euler trajectory (theta wrapped to [-pi, pi) by
    the plant's `step`), through the rollout harness of `sampling.sample_generator_trajectory`.  Returns
    (damage %, time[numSteps], theta[numSteps], omega[numSteps], u[numSteps]); damage = share of steps with
    0 < theta < pi/4.  Plotting visualizers are out of scope: the default shows nothing."""
    from .sampling import sample_generator_trajectory, VisualizerZ
    assert controller is not None, "Surprise !! Changed interface to make controller a required argument"
    if visualizer is None:
        visualizer = visualizer_class(plotfile=plotfile, plot_every_n_steps=plot_every_n_steps) if visualizer_class else VisualizerZ()
    x0 = torch.as_tensor(x0, dtype=torch.float64)
    _, X, U = sample_generator_trajectory(dynamics_model, numSteps, dt=dt, x0=x0, controller=controller, visualizer=visualizer)
    theta_vec, omega_vec, u_vec = X[:numSteps, 0], X[:numSteps, 1], U[:, 0]
    assert torch.all((theta_vec <= math.pi) & (-math.pi <= theta_vec))
    damage = ((0 < theta_vec) & (theta_vec < math.pi / 4)).double().sum() * 100 / numSteps
    return damage, dt * torch.arange(numSteps, dtype=torch.float64), theta_vec, omega_vec, u_vec


def sampling_pendulum_data(dynamics_model, D=100, dt=0.01, **kwargs):
    """(dX[D,2], X[D+1,2], U[D+1,1]) with dX the finite differences of the (wrapped) states, pendulum.py:236-252."""
    _, _, theta_vec, omega_vec, u_vec = sampling_pendulum(dynamics_model, numSteps=D + 1, dt=dt, **kwargs)
    X = torch.stack((theta_vec, omega_vec), dim=1)
    U = u_vec.reshape(-1, 1)
    dX = (X[1:] - X[:-1]) / dt
    return dX, X, U


def get_grid_from_Xtrain(Xtrain):
    """20 x 20 grid over the training range (pendulum.py:421-428); [2, 20, 20] like np.mgrid."""
    import numpy as np
    th = slice(Xtrain[:, 0].min(), Xtrain[:, 0].max(), (Xtrain[:, 0].max() - Xtrain[:, 0].min()) / 20)
    om = slice(Xtrain[:, 1].min(), Xtrain[:, 1].max(), (Xtrain[:, 1].max() - Xtrain[:, 1].min()) / 20)
    return np.mgrid[th, om]


def measure_batch_error(FX_learned, var_FX, FX_true):
    """sqrt(mean_b (F - F^)' var_b^-1 (F - F^))  (pendulum.py:1091-1103)."""
    N, D = FX_learned.shape
    assert FX_true.shape == (N, D) and var_FX.shape == (N, D, D)
    diff = (FX_true - FX_learned).unsqueeze(-1).double().cpu()           # an evaluation metric: tiny systems, host side
    errors = diff.transpose(-2, -1) @ torch.linalg.solve(var_FX.double().cpu(), diff)
    assert bool((errors > 0).all())
    return float(torch.sqrt(errors.sum() / N))


def log_learned_model(Xtrain, model, true_f_func, key="Fx", logger=None):
    """Evaluate the model on the grid and log (Xtrain, grid, FX_learned, var_FX, FX_true) (pendulum.py:450-475);
    returns the logged arrays."""
    import numpy as np
    grid = get_grid_from_Xtrain(Xtrain)
    _, N_, M_ = grid.shape
    n, m = model.x_dim, model.u_dim
    Xtest = torch.as_tensor(grid.transpose(1, 2, 0).reshape(-1, 2), dtype=model.dtype, device=model.device)
    FX_learned, var_FX = model.custom_predict_fullmat(Xtest)
    assert FX_learned.shape == (N_ * M_ * (1 + m) * n,)
    assert not torch.isnan(FX_learned).any() and not torch.isnan(var_FX).any()
    FX_true = true_f_func(Xtest).transpose(-1, -2)                       # (b, 1+m, n)
    out = dict(Xtrain=np.asarray(Xtrain), theta_omega_grid=grid,
               FX_learned=FX_learned.reshape(N_, M_, 1 + m, n).cpu().numpy(),
               var_FX=var_FX.reshape(N_, M_, 1 + m, n, N_, M_, 1 + m, n).cpu().numpy(),
               FX_true=FX_true.reshape(N_, M_, 1 + m, n).cpu().numpy())
    if logger is not None:
        logger.add_tensors("/".join(("log_learned_model", key)), out, 0)
    return out


def learn_dynamics_from_data(dX, X, U, pend_env, regressor_class, logger, max_train, tags=(), training_iter=50,
                             device="cuda", dtype=torch.float32):
    """Random subset of the trajectory, fit (50 Adam steps), log the learned model (pendulum.py:345-371)."""
    numSteps = X.shape[0]
    N = min(numSteps - 1, max_train)
    idx = torch.randint(numSteps - 1, size=(N,))
    f = dict(dtype=dtype, device=device)
    Xtrain, Utrain, XdotTrain = X[idx].to(**f), U[idx].to(**f), dX[idx].to(**f)
    dgp = regressor_class(Xtrain.shape[-1], Utrain.shape[-1], device=device, dtype=dtype)
    dgp.fit(Xtrain, Utrain, XdotTrain, training_iter=training_iter)
    if logger is not None:
        logger.add_tensors("train", dict(Xtrain=Xtrain, Utrain=Utrain), 0)
    logged = log_learned_model(Xtrain.cpu().numpy(), dgp, pend_env.F_func, key="/".join(list(tags) + ["Fx"]), logger=logger)
    return dgp, logged


def learned_model_error(logged, n=2, m=1):
    """The number learn_dynamics_matrix_vector_vis writes to vector_matrix_learning_error.txt (pendulum.py:1121-1139):
    measure_batch_error with the per-point (1+m)n x (1+m)n covariance blocks."""
    FXl = torch.as_tensor(logged["FX_learned"]).double()
    b = int(FXl.shape[0] * FXl.shape[1])
    T_ = (1 + m) * n
    var = torch.as_tensor(logged["var_FX"]).double().reshape(b, T_, b, T_)
    blocks = torch.stack([var[i, :, i, :] for i in range(b)])
    return measure_batch_error(FXl.reshape(b, T_), blocks, torch.as_tensor(logged["FX_true"]).double().reshape(b, T_))


def learn_dynamics_matrix_vector_exp(exps=None, theta0=5 * math.pi / 6, omega0=-0.01, tau=0.01, mass=1, gravity=10,
                                     length=1, max_train=200, numSteps=1000, logger=None, device="cuda",
                                     dtype=torch.float32, training_iter=50):
    """pendulum.py:1052-1088: returns {name: (regressor, logged arrays, error)}."""
    from .control_affine_model import ControlAffineRegressorExact, ControlAffineRegressorVector
    exps = exps or dict(matrix=dict(regressor_class=ControlAffineRegressorExact),
                        vector=dict(regressor_class=ControlAffineRegressorVector))
    pend_env = PendulumDynamicsModel(m=1, n=2, mass=mass, gravity=gravity, length=length)
    dX, X, U = sampling_pendulum_data(pend_env, D=numSteps, x0=torch.tensor([theta0, omega0]), dt=tau,
                                      controller=ControlRandom(mass=mass, gravity=gravity, length=length).control)
    if logger is not None:
        for t, (dx, x, u) in enumerate(zip(dX, X, U)):
            logger.add_tensors("traj", dict(dx=dx, x=x, u=u), t)
    out = dict()
    for name, kw in exps.items():
        dgp, logged = learn_dynamics_from_data(dX, X, U, pend_env, kw["regressor_class"], logger, max_train=max_train,
                                               tags=[name], training_iter=training_iter, device=device, dtype=dtype)
        out[name] = (dgp, logged, learned_model_error(logged))
    return out


def compute_errors(regressor_class, sampling_callable, pend_env, ntries=5, max_train=200, test_on_grid=False, ntest=400,
                   device="cuda", dtype=torch.float32):
    """pendulum.py:1248-1303: `ntries` variance-weighted errors (`measure_batch_error`) of `regressor_class` on fresh
    trajectories.  As upstream, the regressor of a sample is constructed and queried WITHOUT being fitted (there is no
    fit call between `regressor_class(...)` and `custom_predict_fullmat`, :1279-1283): the figure is the error of the
    prior model."""
    import numpy as np
    errors = []
    for _ in range(ntries):
        dX, X, U = sampling_callable()
        order = np.arange(X.shape[0] - 1)
        np.random.shuffle(order)
        order_t = torch.from_numpy(order)
        Xtrain = X[order_t[:max_train]]
        if test_on_grid:
            grid = get_grid_from_Xtrain(Xtrain.cpu().numpy())
            Xtest = torch.from_numpy(grid.reshape(-1, Xtrain.shape[-1]))
        else:
            Xtest = X[order_t[-ntest:]]
        Xtest = Xtest.to(device=device, dtype=dtype)
        FX_true = pend_env.F_func(Xtest).transpose(-2, -1)                       # (b, 1+m, n)
        dgp = regressor_class(Xtrain.shape[-1], U.shape[-1], device=device, dtype=dtype)
        FX_learned, var_FX = dgp.custom_predict_fullmat(Xtest.reshape(-1, Xtest.shape[-1]))
        b, T_ = Xtest.shape[0], (1 + pend_env.ctrl_size) * pend_env.state_size
        idx = torch.arange(b, device=var_FX.device)
        blocks = var_FX.reshape(b, T_, b, T_)[idx, :, idx, :]
        errors.append(measure_batch_error(FX_learned.reshape(-1, T_), blocks, FX_true.reshape(-1, T_).to(FX_learned)))
    return errors


def speed_test_matrix_vector_exp(max_train_variations=(256, 256 + 64, 256 + 128, 256 + 256), ntimes=50, repeat=5,
                                 errorbartries=30, logger=None, exps=None, theta0=5 * math.pi / 6, omega0=-0.01, tau=0.01,
                                 mass=1, gravity=10, length=1, numSteps=2000, pendulum_dynamics_class=PendulumDynamicsModel,
                                 training_iter=50, device="cuda", dtype=torch.float32):
    """The reference's published speed test (pendulum.py:1305-1394, the only path of the repository with published
    numbers): one randomised pendulum trajectory; per training-set size a random subset, and per regressor
    (MVGP full / diag, CoGP full / diag) `fit(training_iter=50)`, then
    min(timeit.repeat('dgp.custom_predict_fullmat(Xtest); dgp.clear_cache()', repeat, number=ntimes)) on the 20 x 20
    (theta, omega) grid of the training range, and `errorbartries` prior-model errors (`compute_errors`).  Every timed call
    ends with a device synchronize (the reference's timing is host side).  Logged under the reference's tags when a
    logger is given (`<name>/elapsed` is seconds PER CALL, as upstream logs it); returns
    {name: {max_train: dict(elapsed=s per call, errors=[...], fit_s=...)}}."""
    import time
    import timeit
    from functools import partial
    import numpy as np
    from .control_affine_model import (ControlAffineRegressorExact, ControlAffineRegMatrixDiag,
                                       ControlAffineRegressorVector, ControlAffineRegVectorDiag)
    exps = exps or dict(matrix=dict(regressor_class=ControlAffineRegressorExact),
                        vector=dict(regressor_class=ControlAffineRegressorVector),
                        matrixdiag=dict(regressor_class=ControlAffineRegMatrixDiag),
                        vectordiag=dict(regressor_class=ControlAffineRegVectorDiag))
    pend_env = pendulum_dynamics_class(m=1, n=2, mass=mass, gravity=gravity, length=length)
    sample = partial(sampling_pendulum_data, dynamics_model=pend_env, D=numSteps, x0=torch.tensor([theta0, omega0]), dt=tau,
                     controller=ControlRandom(mass=mass, gravity=gravity, length=length).control)
    dX, X, U = sample()
    if logger is not None:
        for t, (dx, x, u) in enumerate(zip(dX, X, U)):
            logger.add_tensors("traj", dict(dx=dx, x=x, u=u), t)
    order = np.arange(X.shape[0] - 1)
    f = dict(device=device, dtype=dtype)
    out = {name: {} for name in exps}
    for max_train in max_train_variations:
        np.random.shuffle(order)
        idx = torch.from_numpy(order[:max_train].copy())
        Xtrain, Utrain, XdotTrain = X[idx].to(**f), U[idx].to(**f), dX[idx].to(**f)
        grid = get_grid_from_Xtrain(X[idx].numpy())
        Xtest = torch.from_numpy(grid.reshape(-1, 2)).to(**f)           # raw reshape of the [2, 20, 20] mgrid, as upstream (:1351-1355)
        for name, kw in exps.items():
            dgp = kw["regressor_class"](2, 1, device=device, dtype=dtype)
            t0 = time.perf_counter()
            dgp.fit(Xtrain, Utrain, XdotTrain, training_iter=training_iter)
            torch.cuda.synchronize()
            fit_s = time.perf_counter() - t0

            def call():
                dgp.custom_predict_fullmat(Xtest)
                dgp.clear_cache()
                torch.cuda.synchronize()
            call()
            elapsed = min(timeit.repeat(call, repeat=repeat, number=ntimes)) / ntimes
            errors = compute_errors(kw["regressor_class"], sample, pend_env, max_train=max_train, ntries=errorbartries,
                                    device=device, dtype=dtype)
            out[name][max_train] = dict(elapsed=elapsed, errors=errors, fit_s=fit_s,
                                        fit_loss_first_last=[dgp.fit_losses[0], dgp.fit_losses[-1]] if training_iter else None)
            if logger is not None:
                logger.add_scalars(name, dict(elapsed=elapsed), max_train)
                logger.add_tensors(name, dict(errors=np.asarray(errors)), max_train)
    return out


# ------------------------------------------------------------------------------------------------
# The run-script entry points (`run.sh:17-21` calls them by name): experiment half + the NUMBERS half of the reference's
# `_vis` functions (what they read back from the event file and write next to it).  The plotting half (matplotlib figures,
# LaTeX labels, xdg-open) is out of scope (SURVEY 2 / DESIGN 8).
def _newest_events_file(logs_dir):
    import glob
    import os
    return max(glob.glob(os.path.join(logs_dir, "*.tfevents*")), key=lambda f: os.stat(f).st_mtime)


def learn_dynamics_matrix_vector_vis(exps=("matrix", "vector"), events_file=None):
    """pendulum.py:1215-1242 without the figure: reads the logged grids back from the event file, computes the
    variance-weighted error of every experiment (`measure_batch_error` on the per-point covariance blocks, :1121-1139) and
    writes them to `vector_matrix_learning_error.txt` beside the event file in the reference's format (one row, %.03f,
    header = names).  Returns (error_file, {name: error})."""
    import os.path as osp
    import numpy as np
    from .tblog import load_tensorboard_scalars
    logdata = load_tensorboard_scalars(events_file)
    errors = dict()
    for exp in exps:
        key = "log_learned_model/" + exp + "/Fx/"
        logged = dict(FX_learned=logdata[key + "FX_learned"][0][1], var_FX=logdata[key + "var_FX"][0][1],
                      FX_true=logdata[key + "FX_true"][0][1])
        errors[exp] = float(learned_model_error(logged))
    error_file = osp.join(osp.dirname(events_file), "vector_matrix_learning_error.txt")
    np.savetxt(error_file, [[errors[e] for e in exps]], fmt="%.03f", header=" ".join(exps))
    return error_file, errors


def learn_dynamics_matrix_vector(logger_class=None, **kw):
    """pendulum.py:1244-1246: the experiment (`learn_dynamics_matrix_vector_exp`, logged under `data/runs/
    learn_matrix_vector_<version>` unless `logger_class` says otherwise), then the numbers of `_vis`.  Returns the event
    file, as the reference's `_exp` does."""
    from functools import partial
    from .tblog import TBLogger
    logger = (logger_class or partial(TBLogger, exp_tags=["learn_matrix_vector"], runs_dir="data/runs"))()
    res = learn_dynamics_matrix_vector_exp(logger=logger, **kw)
    events_file = _newest_events_file(logger.experiment_logs_dir)
    learn_dynamics_matrix_vector_vis(exps=tuple(res), events_file=events_file)
    return events_file


def speed_test_matrix_vector_vis(events_file, exp_conf=("vectordiag", "matrixdiag", "vector", "matrix")):
    """pendulum.py:1396-1430 without the figure: {name: dict(training_samples, elapsed [s per call], errors)} read back
    from the event file (tags `<name>/elapsed`, `<name>/errors`)."""
    from .tblog import load_tensorboard_scalars
    logdata = load_tensorboard_scalars(events_file)
    out = dict()
    for gp in exp_conf:
        if gp + "/elapsed" not in logdata:
            continue
        training_samples, elapsed = zip(*logdata[gp + "/elapsed"])
        _, errors = zip(*logdata[gp + "/errors"])
        out[gp] = dict(training_samples=list(training_samples), elapsed=list(elapsed), errors=list(errors))
    return out


def speed_test_matrix_vector(logger_class=None, **kw):
    """pendulum.py:1433-1435: the published speed test (`speed_test_matrix_vector_exp`, logged under `data/runs/
    speed_test_matrix_vector_<version>`), then the read-back of `_vis`.  Returns the event file."""
    from functools import partial
    from .tblog import TBLogger
    logger = (logger_class or partial(TBLogger, exp_tags=["speed_test_matrix_vector"], runs_dir="data/runs"))()
    speed_test_matrix_vector_exp(logger=logger, **kw)
    events_file = _newest_events_file(logger.experiment_logs_dir)
    speed_test_matrix_vector_vis(events_file)
    return events_file
