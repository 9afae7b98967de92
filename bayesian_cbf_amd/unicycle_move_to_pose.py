"""Mirror of the hot-path classes of bayes_cbf/unicycle_move_to_pose.py on libbcbf.

`ControllerCLFBayesian.control(x, t)` keeps the reference's signature (:926) and accepts either
one state [3] (reference behaviour) or a batch of states [Bt, 3] -- one independent control loop
per row (new capability).  The chance-constraint assembly and the SOCP run in one fused launch
sequence (`ops.unicycle_control_step`); infeasible programs raise `ValueError` for a single state,
exactly like the reference (:954-964), and are masked (status != 0, control = ctrl_ref) in a batch."""
import math

import numpy as np
import torch

from . import ops
from .cbc2 import cbc1_safety_factor
from .planner import PiecewiseLinearPlanner  # noqa: F401  (re-export, as in the reference module)


class AckermannDrive:
    """Plant / prior mean dynamics (unicycle_move_to_pose.py:200-292)."""
    state_size, ctrl_size = 3, 2

    def __init__(self, L=0.2, kernel_diag_A=(1.0, 1.0, 1.0)):
        self.L = L
        self.kernel_diag_A = torch.as_tensor(kernel_diag_A, dtype=torch.float64)
        self.current_state = None

    def set_init_state(self, x):
        self.current_state = x.clone()

    def f_func(self, x):
        return torch.zeros_like(x)

    def g_func(self, state_in):
        state = state_in.unsqueeze(0) if state_in.dim() <= 1 else state_in
        th = state[..., 2]
        z, o = torch.zeros_like(th), torch.ones_like(th)
        gX = torch.stack([torch.stack([th.cos(), z], -1), torch.stack([th.sin(), z], -1),
                          torch.stack([z, o / self.L], -1)], -2)
        return gX.squeeze(0) if state_in.dim() <= 1 else gX

    def F_func(self, X):
        return torch.cat([self.f_func(X).unsqueeze(-1), self.g_func(X)], dim=-1)

    def fixed_kernel(self):
        """(A, B) of the fixed-kernel GP ((u_hom' B u_hom) A, :261-275)."""
        return torch.diag(self.kernel_diag_A), torch.eye(self.ctrl_size + 1, dtype=torch.float64)

    def fu_func_gp(self, u):
        return _fixed_kernel_gp(self, u, "AckermannDrive")

    def step(self, u, dt):
        """Explicit Euler (:277-282): returns dict(xdot, x).  A device batch of states [Bt, 3] advances in ONE launch of
        the HIP plant kernel (`bcbf_unicycle_step`, in place on `current_state`)."""
        return _euler_step(self, u, dt, float(self.L))


def _euler_step(model, u, dt, L):
    x = model.current_state
    xdot = model.f_func(x) + (model.g_func(x) @ u.unsqueeze(-1)).squeeze(-1)
    if x.is_cuda and x.dim() == 2:
        ops.unicycle_step(x, u.to(x).contiguous(), float(dt), L)
    else:
        model.current_state = x + xdot * dt
    return dict(xdot=xdot, x=model.current_state)


def _fixed_kernel_gp(model, u, name):
    """GaussianProcess(mean = f + g u, knl = (u_hom' B u_hom) A): a leaf of the expression algebra whose conditions
    lower onto bcbf_cbc_terms with M_k = 0, B_k = B (gp_algebra.lower, cbc2.FixedKernelGP)."""
    from .gp_algebra import GaussianProcess
    A, B = model.fixed_kernel()

    def mean(x):
        return model.f_func(x) + model.g_func(x) @ u.to(x)

    def knl(x, xp):
        uh = torch.cat([torch.ones(1).to(x), u.to(x)])
        return (uh @ B.to(x) @ uh) * A.to(x)

    return GaussianProcess(mean=mean, knl=knl, shape=(model.state_size,), name=name, source=(model, "fu", u))


class PolarDynamics:
    """The unicycle in polar coordinates relative to the goal, x = (rho, alpha, beta), u = (v, omega)
    (unicycle_move_to_pose.py:143-167): f = 0, g = [[-cos a, 0], [-sin a / rho, 1], [-sin a / rho, 0]]."""
    state_size, ctrl_size = 3, 2

    def __init__(self):
        self.current_state = None

    def set_init_state(self, x0):
        self.current_state = x0

    def f_func(self, x):
        return torch.zeros_like(x)

    def g_func(self, x):
        rho, alpha = x[..., 0], x[..., 1]
        assert bool((rho > 1e-6).all())
        z, o = torch.zeros_like(rho), torch.ones_like(rho)
        s = -torch.sin(alpha) / rho
        return torch.stack([torch.stack([-torch.cos(alpha), z], -1), torch.stack([s, o], -1), torch.stack([s, z], -1)], -2)

    def step(self, u_torch, dt):
        x = self.current_state
        xdot = self.f_func(x) + (self.g_func(x) @ u_torch.unsqueeze(-1)).squeeze(-1)
        self.current_state = x + xdot * dt
        return dict(xdot=xdot, x=self.current_state)


class CartesianDynamics:
    """Unit-wheelbase unicycle with the kernel (u'u + 1) I (:168-197)."""
    state_size, ctrl_size = 3, 2

    def __init__(self):
        self.current_state = None

    def set_init_state(self, x0):
        self.current_state = x0.clone()

    def f_func(self, x):
        return torch.zeros_like(x)

    def g_func(self, state_in):
        state = state_in.unsqueeze(0) if state_in.dim() <= 1 else state_in
        th = state[..., 2]
        z, o = torch.zeros_like(th), torch.ones_like(th)
        gX = torch.stack([torch.stack([th.cos(), z], -1), torch.stack([th.sin(), z], -1), torch.stack([z, o], -1)], -2)
        return gX.squeeze(0) if state_in.dim() <= 1 else gX

    def step(self, u_torch, dt):
        return _euler_step(self, u_torch, dt, 1.0)

    def fixed_kernel(self):
        return torch.eye(self.state_size, dtype=torch.float64), torch.eye(self.ctrl_size + 1, dtype=torch.float64)

    def fu_func_gp(self, u):
        return _fixed_kernel_gp(self, u, "CartesianDynamics")


class ZeroDynamicsModel:
    """misc.py:194-213."""

    def __init__(self, m, n):
        self.m, self.n = m, n

    ctrl_size = property(lambda self: self.m)
    state_size = property(lambda self: self.n)

    def f_func(self, X):
        return torch.zeros_like(X)

    def g_func(self, X):
        return torch.zeros(*X.shape, self.m, dtype=X.dtype, device=X.device)


class ZeroDynamicsBayesian(ZeroDynamicsModel):
    """Zero mean, kernel (u'u + 1) I (:794-798)."""

    def fixed_kernel(self):
        return torch.eye(self.n, dtype=torch.float64), torch.eye(self.m + 1, dtype=torch.float64)

    def fu_func_gp(self, U):
        return _fixed_kernel_gp(self, U, "ZeroDynamicsBayesian")


class CLFCartesian:
    """Parameters of the reference's CLFCartesian (:522-615); evaluated inside bcbf_unicycle_constraints."""

    def __init__(self, Kp=(0.9, 1.5, 4.0)):
        self.Kp = torch.as_tensor(Kp, dtype=torch.float64)


class ObstacleCBF:
    """Parameters of the reference's ObstacleCBF (:618-696)."""

    def __init__(self, center, radius, term_weights=(0.5, 0.5)):
        self.center = torch.as_tensor(center, dtype=torch.float64)
        self.radius = torch.as_tensor(radius, dtype=torch.float64)
        self.term_weights = tuple(term_weights)


def obstacles_at_mid_from_start_and_goal(x, x_g, term_weights=(0.5, 0.5)):
    """unicycle_move_to_pose.py:1562-1570 (x, x_g: [3] or [Bt,3])."""
    R90 = torch.tensor([[0.0, -1.0], [1.0, 0.0]], dtype=x.dtype, device=x.device)
    d = x[..., :2] - x_g[..., :2]
    mid = (x[..., :2] + x_g[..., :2]) / 2
    off = d @ R90.T / 3
    rad = d.norm(dim=-1) / 4
    return [ObstacleCBF(mid + off, rad, term_weights), ObstacleCBF(mid - off, rad, term_weights)]


class LearnedShiftInvariantDynamics:
    """unicycle_move_to_pose.py:295-428: prior mean dynamics + a learned control-affine residual.  `train(x, u)` buffers
    the closed-loop samples and every `train_every_n_steps` refits the regressor on finite-difference targets minus the
    prior mean (random subsample to `max_train`), exactly the reference's schedule; inputs are made shift invariant
    ((x, y) zeroed) for f_func / g_func / fit / custom_predict_fullmat but -- as in the reference (:388-397) -- not for
    the GP the controller queries."""
    state_size, ctrl_size = 3, 2

    def __init__(self, dt=None, learned_dynamics=None, learned_dynamics_class=None, mean_dynamics=None, max_train=200,
                 training_iter=100, shift_invariant=True, train_every_n_steps=20, enable_learning=True, device="cuda",
                 dtype=torch.float64, hyper_refit_every=1, online_update=False, window=None):
        """hyper_refit_every / online_update: see `online.OnlineLearner` (defaults = the reference's schedule)."""
        from .control_affine_model import ControlAffineRegressorExactRankOne
        from .online import OnlineLearner
        self.max_train, self.training_iter, self.dt = max_train, training_iter, dt
        self.mean_dynamics = mean_dynamics or AckermannDrive()
        cls = learned_dynamics_class or ControlAffineRegressorExactRankOne
        self.learned_dynamics = learned_dynamics if learned_dynamics is not None else cls(
            self.state_size, self.ctrl_size, device=device, dtype=dtype)
        self.shift_invariant = shift_invariant
        self.train_every_n_steps, self.enable_learning = train_every_n_steps, enable_learning
        self.current_state = None

        def subsample(count, k):                      # :377-381: numpy shuffle of all indices, first max_train kept
            idx = np.arange(count)
            np.random.shuffle(idx)
            return torch.from_numpy(idx[:k])

        self._learner = OnlineLearner(self.learned_dynamics, self._residual_targets, dt, train_every_n_steps, max_train,
                                      training_iter, subsample, enable_learning=enable_learning,
                                      hyper_refit_every=hyper_refit_every, online_update=online_update,
                                      transform=self._trans_invariant_wrapper, window=window)

    # the controller's buffers (:340-354), owned by the learner
    Xtrain = property(lambda self: self._learner.Xtrain)
    Utrain = property(lambda self: self._learner.Utrain)

    def _trans_invariant_wrapper(self, X):
        if not self.shift_invariant:
            return X
        return torch.cat([torch.zeros_like(X[..., :self.state_size - 1]), X[..., self.state_size - 1:]], dim=-1)

    def _residual_targets(self, X, U, Xdot):
        """Xdot minus the prior-mean dynamics at the shift-invariant inputs (:361-375)."""
        x0 = self._trans_invariant_wrapper(X)
        md = self.mean_dynamics
        return Xdot - (md.f_func(x0) + (md.g_func(x0) @ U.unsqueeze(-1)).squeeze(-1))

    def f_func(self, X):
        x0 = self._trans_invariant_wrapper(X)
        return self.mean_dynamics.f_func(x0) + self.learned_dynamics.f_func(x0).to(x0)

    def g_func(self, X):
        x0 = self._trans_invariant_wrapper(X)
        return self.mean_dynamics.g_func(x0) + self.learned_dynamics.g_func(x0).to(x0)

    def train(self, xi, uopt):
        """Hand one visited (state, control) to the learner: refit / append on its schedule (:340-354)."""
        ln = self._learner                  # the schedule's knobs are plain attributes upstream: honour later changes
        ln.dt, ln.enable_learning, ln.training_iter = self.dt, self.enable_learning, self.training_iter
        ln.max_train, ln.train_every_n_steps = self.max_train, self.train_every_n_steps
        ln.observe(xi, uopt)

    def get_kernel_param(self, name):
        return self.learned_dynamics.get_kernel_param(name)

    def fit(self, Xtrain, Utrain, XdotTrain, training_iter=None):
        """Fit the residual model on given data (:356-386): prior mean removed, shift-invariant inputs, random subsample
        to max_train."""
        if not len(Xtrain):
            return
        XdotError = self._residual_targets(Xtrain, Utrain, XdotTrain)
        Xtrain = self._trans_invariant_wrapper(Xtrain)
        if XdotTrain.shape[0] > self.max_train:
            idx = self._learner.subsample(XdotTrain.shape[0], self.max_train).to(Xtrain.device)
            Xtrain, Utrain, XdotError = Xtrain[idx], Utrain[idx], XdotError[idx]
        self.learned_dynamics.fit(Xtrain, Utrain, XdotError,
                                  training_iter=self.training_iter if training_iter is None else training_iter)
        self._learner.has_been_trained_once, self._learner.n_in_model = True, None

    def fu_func_gp(self, U):
        if self.enable_learning:
            return self.learned_dynamics.fu_func_gp(U)          # + the deterministic prior mean, added by the controller
        raise NotImplementedError("fixed-kernel model: use ControllerCLFBayesian(dynamics=None, mean_dynamics=...)")

    def step(self, u_torch, dt):
        x = self.current_state
        xdot = self.f_func(x) + self.g_func(x) @ u_torch
        self.current_state = x + xdot * dt
        return dict(xdot=xdot, x=self.current_state)

    def custom_predict_fullmat(self, Xtest_in, **kw):
        Xtest = Xtest_in.unsqueeze(0) if Xtest_in.ndim == 1 else Xtest_in
        x0 = self._trans_invariant_wrapper(Xtest)
        diffFX, diffVarFX = self.learned_dynamics.custom_predict_fullmat(x0, **kw)
        md = self.mean_dynamics
        F = torch.cat([md.f_func(x0).unsqueeze(-1), md.g_func(x0)], dim=-1).to(diffFX)       # [b, n, 1+m]
        return F.transpose(-2, -1).reshape(-1) + diffFX, diffVarFX

    def clear_cache(self):
        self.learned_dynamics.clear_cache()

    def as_dict(self):
        """GP tensors for `ops.unicycle_control_step` (one shared learned model), or None before the first fit."""
        reg = self.learned_dynamics
        if not self.enable_learning or reg.Xtrain is None:
            return None
        reg._require_rbf("the fused control step")
        st = reg._state()
        return {k: st[k] for k in ("Lop", "Vw", "X", "UHB", "ell", "s2", "Bm", "M0", "A")}


class ControllerCLFBayesian:
    """unicycle_move_to_pose.py:801-998.  `dynamics` is a `BatchedControlAffineGP` (learned residual,
    regime I), or None for the fixed-kernel model of AckermannDrive.fu_func_gp (:262-275:
    M_k = 0, B_k = I, A = diag(kernel_diag_A))."""

    def __init__(self, planner, u_dim=2, coordinate_converter=None, dynamics=None, clf=None, clf_gamma=10.0,
                 cost_weights=(0.33, 0.33, 0.33), cbfs=(), cbf_gammas=(), ctrl_min=(-10.0, -np.pi * 5),
                 ctrl_max=(10.0, np.pi * 5), ctrl_ref=(0.0, 0.0), max_risk=1e-2, visualizer=None,
                 mean_dynamics=None, device="cuda", dtype=torch.float64):
        self.u_dim = 2
        self.planner, self.dynamics, self.clf = planner, dynamics, clf or CLFCartesian()
        self.clf_gamma, self.cost_weights = clf_gamma, cost_weights
        self.cbfs, self.cbf_gammas = list(cbfs), list(cbf_gammas)
        self.ctrl_min, self.ctrl_max, self.ctrl_ref = np.array(ctrl_min), np.array(ctrl_max), np.array(ctrl_ref)
        self.max_risk = max_risk
        self.mean_dynamics = mean_dynamics or AckermannDrive(L=1.0)
        self.device, self.dtype = torch.device(device), dtype
        self._ws = None

    def _factor(self):
        assert 0 <= self.max_risk <= 0.5
        return 0.0 if self.max_risk == 0.5 else cbc1_safety_factor(self.max_risk)

    def _task(self, Bt, t):
        f = dict(dtype=self.dtype, device=self.device)
        exp = lambda v: torch.as_tensor(v, **f).reshape(-1, v.shape[-1] if hasattr(v, "shape") and v.ndim else 1)
        plan = self.planner.plan(t).to(**f)
        dplan = self.planner.dot_plan(t).to(**f)
        plan = plan.expand(Bt, 3).contiguous() if plan.dim() == 1 else plan.contiguous()
        dplan = dplan.expand(Bt, 3).contiguous() if dplan.dim() == 1 else dplan.contiguous()
        Kob = len(self.cbfs)
        if Kob:
            def per_instance(v, width, what):          # one value for every loop, or one per loop: anything else is an error
                v = v.to(**f).reshape(-1, width)
                if v.shape[0] == 1:
                    return v.expand(Bt, width)
                if v.shape[0] != Bt:
                    raise ValueError("%s: %d values for a batch of %d control loops" % (what, v.shape[0], Bt))
                return v
            centers = torch.stack([per_instance(c.center, 2, "obstacle center") for c in self.cbfs], dim=1).contiguous()
            radii = torch.stack([per_instance(c.radius, 1, "obstacle radius")[:, 0] for c in self.cbfs], dim=1).contiguous()
        else:       # the reference's default cbfs=[] (:811): a pure CLF controller, the kernels take Kob = 0
            centers, radii = torch.empty(Bt, 0, 2, **f), torch.empty(Bt, 0, **f)
        tw = torch.tensor(self.cbfs[0].term_weights if Kob else (0.5, 0.5), **f)
        return dict(plan=plan, dot_plan=dplan, Kp=self.clf.Kp.to(**f), centers=centers, radii=radii, tw=tw,
                    gammas=torch.tensor(list(self.cbf_gammas), **f),
                    w=torch.tensor(list(self.cost_weights), **f).expand(Bt, 3).contiguous(),
                    r=torch.tensor(self.ctrl_ref, **f).expand(Bt, 2).contiguous(),
                    sign=torch.tensor([-1.0] + [1.0] * Kob, **f),
                    relax_mask=torch.tensor([1.0] + [0.0] * Kob, **f),
                    rho=torch.full((Bt,), self._factor(), **f))

    def _gp(self, Bt):
        f = dict(dtype=self.dtype, device=self.device)
        if self.dynamics is not None:
            return self.dynamics.as_dict()
        raise NotImplementedError

    def control(self, x_torch, t):
        single = x_torch.dim() == 1
        x = x_torch.reshape(-1, 3).to(device=self.device, dtype=self.dtype).contiguous()
        Bt = x.shape[0]
        task = self._task(Bt, t)
        Kob = len(self.cbfs)
        if self._ws is None or self._ws["y"].shape[0] != Bt:
            self._ws = ops.control_workspace(Bt, Kob, self.dtype, self.device)
        ws = self._ws
        L_mean = float(self.mean_dynamics.L)
        learned = isinstance(self.dynamics, LearnedShiftInvariantDynamics)
        if learned:
            L_mean = float(self.dynamics.mean_dynamics.L)
        gp = self.dynamics.as_dict() if self.dynamics is not None else None
        if gp is not None:
            ops.unicycle_control_step(gp, task, ws, x, dt=0.0, L_mean=L_mean, clf_gamma=float(self.clf_gamma))
        elif learned and self.dynamics.enable_learning:   # learning enabled, no data yet: the GP prior (M0, s2 B, A)
            m = self.dynamics.learned_dynamics.model
            f = dict(dtype=self.dtype, device=self.device)
            ops.unicycle_constraints(x, task["plan"], task["dot_plan"], task["Kp"], float(self.clf_gamma),
                                     task["centers"], task["radii"], task["tw"], task["gammas"], L_mean,
                                     out=(ws["grad"], ws["cst"], ws["fhat"], ws["ghat"]))
            with torch.no_grad():
                ws["Mk"].copy_(m.M0.t().to(**f).expand(Bt, 3, 3))
                ws["Bk"].copy_((m.outputscale * m.B).to(**f).expand(Bt, 3, 3))
                A = m.A.to(**f).expand(Bt, 3, 3).contiguous()
            y, status, iters, cones, cstatus, _ = ops.cbc_socp(ws["Mk"], ws["Bk"], A, ws["grad"], ws["cst"], task["sign"],
                                                              ws["fhat"], ws["ghat"], task["w"], task["r"],
                                                              task["relax_mask"], task["rho"])
            ws["y"].copy_(y)
            ws["status"].copy_(status)
        else:   # fixed-kernel model: no posterior kernel, M_k = 0, B_k = I
            ops.unicycle_constraints(x, task["plan"], task["dot_plan"], task["Kp"], float(self.clf_gamma),
                                     task["centers"], task["radii"], task["tw"], task["gammas"], L_mean,
                                     out=(ws["grad"], ws["cst"], ws["fhat"], ws["ghat"]))
            ws["Mk"].zero_()
            ws["Bk"].copy_(torch.eye(3, dtype=self.dtype, device=self.device).expand(Bt, 3, 3))
            A = torch.diag(self.mean_dynamics.kernel_diag_A.to(dtype=self.dtype, device=self.device)).expand(Bt, 3, 3).contiguous()
            y, status, iters, cones, cstatus, _ = ops.cbc_socp(ws["Mk"], ws["Bk"], A, ws["grad"], ws["cst"], task["sign"],
                                                              ws["fhat"], ws["ghat"], task["w"], task["r"],
                                                              task["relax_mask"], task["rho"])
            ws["y"].copy_(y)
            ws["status"].copy_(status)
        u = ws["y"][:, :2]
        if single:
            st = int(ws["status"][0])
            if st != 0:
                raise ValueError({1: "max_iterations", 2: "infeasible", 3: "bad_cone"}.get(st, "solver_error"))
            uopt = u[0].to(device=x_torch.device, dtype=x_torch.dtype)
            if hasattr(self.dynamics, "train"):            # :990-993: the controller feeds the learner
                self.dynamics.train(x_torch, uopt)
            return uopt
        bad = ws["status"] != 0
        if bool(bad.any()):
            u = torch.where(bad[:, None], task["r"], u)
        self.last_status = ws["status"]
        return u.to(dtype=x_torch.dtype)


class ControllerCLF(ControllerCLFBayesian):
    """unicycle_move_to_pose.py:703-791: the mean-only CLF-CBF quadratic program (also the data-generating
    controller of the reference's speed tests, :2075-2080)
        min |u|^2 + clf_relax_weight * relax
        s.t. ctrl_min <= u <= ctrl_max,   a_clc'u + b_clc - relax <= 0,   a_k'u + b_k >= 0  (one per obstacle)
    with a = grad' g(x), b = grad' f(x) + const from the same task-function kernel as the Bayesian controller.
    Batched over independent states; solved by the generic cone-QP kernel (`bcbf_coneqp_f64`).
    NB the reference's constructor ignores its clf_gamma / clf_relax_weight arguments (:720-721: 10 and 10)."""

    def __init__(self, planner, u_dim=2, coordinate_converter=None, dynamics=None, clf=None, clf_gamma=10.0,
                 clf_relax_weight=10.0, cbfs=(), cbf_gammas=(), visualizer=None, device="cuda", dtype=torch.float64,
                 **kw):
        super().__init__(planner, u_dim=u_dim, dynamics=None, clf=clf, clf_gamma=10.0, cbfs=cbfs,
                         cbf_gammas=cbf_gammas, mean_dynamics=dynamics if isinstance(dynamics, AckermannDrive) else None,
                         device=device, dtype=dtype, **kw)
        self.clf_relax_weight = 10.0

    def control(self, x_torch, t):
        single = x_torch.dim() == 1
        f64 = dict(dtype=torch.float64, device=self.device)
        x = x_torch.reshape(-1, 3).to(**f64).contiguous()
        Bt, Kob = x.shape[0], len(self.cbfs)
        dtype, self.dtype = self.dtype, torch.float64            # the generic solver is fp64
        task = self._task(Bt, t)
        self.dtype = dtype
        grad, cst, fhat, ghat = ops.unicycle_constraints(x, task["plan"], task["dot_plan"], task["Kp"],
                                                         float(self.clf_gamma), task["centers"], task["radii"],
                                                         task["tw"], task["gammas"], float(self.mean_dynamics.L))
        a = torch.einsum("bkn,bnm->bkm", grad, ghat)             # [Bt, 1+Kob, 2]
        b = torch.einsum("bkn,bn->bk", grad, fhat) + cst
        nv, K = 3, 4 + 1 + Kob                                   # y = [u0, u1, relax];  G y <= h
        P = torch.zeros(Bt, nv, nv, **f64)
        P[:, 0, 0] = P[:, 1, 1] = 2.0
        q = torch.zeros(Bt, nv, **f64)
        q[:, 2] = self.clf_relax_weight
        G = torch.zeros(Bt, K, nv, **f64)
        h = torch.zeros(Bt, K, **f64)
        lo, hi = torch.as_tensor(self.ctrl_min, **f64), torch.as_tensor(self.ctrl_max, **f64)
        G[:, 0, 0] = G[:, 1, 1] = -1.0; h[:, 0], h[:, 1] = -lo[0], -lo[1]
        G[:, 2, 0] = G[:, 3, 1] = 1.0; h[:, 2], h[:, 3] = hi[0], hi[1]
        G[:, 4, :2], G[:, 4, 2], h[:, 4] = a[:, 0], -1.0, -b[:, 0]
        G[:, 5:, :2], h[:, 5:] = -a[:, 1:], b[:, 1:]
        y, status, iters = ops.coneqp(P.contiguous(), q, G.contiguous(), h.contiguous(), K, [])
        self.last_status = status
        u = y[:, :2]
        if single:
            if int(status[0]) != 0:
                raise ValueError({1: "max_iterations", 2: "infeasible"}.get(int(status[0]), "solver_error"))
            return u[0].to(device=x_torch.device, dtype=x_torch.dtype)
        return u.to(dtype=x_torch.dtype)


# ----------------------------------------------------------------------------------------------------------------
# Entry points of the reference module (unicycle_move_to_pose.py:1689-2013): one closed loop per call, logged in the
# reference's event-file format and read back by `tblog.playback_logfile` (plots / animation are out of scope).
def track_trajectory_ackerman_clf_bayesian(x, x_g, dt=None, cbfs=None, cbf_gammas=None, numSteps=None,
                                           enable_learning=True, mean_dynamics_gen=lambda: AckermannDrive(L=10.0),
                                           true_dynamics_gen=lambda: AckermannDrive(L=1.0), visualizer_class=None,
                                           controller_class=None, train_every_n_steps=20, logger=None, device="cuda",
                                           dtype=torch.float64, learned_dynamics=None, training_iter=100, **kw):
    """:1689-1733.  Returns (X[numSteps+1,3], U[numSteps,2]); `logger` (a tblog.TBLogger) receives every step."""
    from . import tblog
    controller_class = controller_class or ControllerCLFBayesian
    f = dict(dtype=dtype, device=device)
    x, x_g = torch.as_tensor(x, **f), torch.as_tensor(x_g, **f)
    planner = PiecewiseLinearPlanner(x, x_g, numSteps, dt, frac_time_to_reach_goal=0.95)
    mean_dynamics, plant = mean_dynamics_gen(), true_dynamics_gen()
    dynamics = (LearnedShiftInvariantDynamics(dt=dt, mean_dynamics=mean_dynamics, enable_learning=True,
                                              train_every_n_steps=train_every_n_steps, learned_dynamics=learned_dynamics,
                                              training_iter=training_iter, device=device, dtype=dtype)
                if enable_learning else None)
    ctrl = controller_class(planner, coordinate_converter=lambda a, b: a, dynamics=dynamics, mean_dynamics=mean_dynamics,
                            clf=CLFCartesian(Kp=[0.9, 1.5, 0.0]), cbfs=cbfs(x, x_g), cbf_gammas=list(cbf_gammas),
                            device=device, dtype=dtype, **kw)
    rl = tblog.RolloutLogger(planner, dt, logger) if logger is not None else None
    X = torch.empty(numSteps + 1, 3, **f)
    U = torch.empty(numSteps, 2, **f)
    X[0] = x
    for t in range(numSteps):
        u = ctrl.control(X[t], t)
        if rl is not None:
            rl.setStateCtrl(X[t], u, t)
        X[t + 1] = X[t] + (plant.f_func(X[t]) + plant.g_func(X[t]) @ u) * dt          # AckermannDrive.step (:277-282)
        U[t] = u
    return X, U


def unicycle_demo(simulator, exp_tags=(), runs_dir="data/runs", state_start=(-3.0, -1.0, -math.pi / 4),
                  state_goal=(0.0, 0.0, math.pi / 4), config=None):
    """:1740-1778: run `simulator(x0, xg, logger=...)` into <runs_dir>/unicycle_move_to_pose_fixed_<tags>_<version>,
    with its config.json; returns the directory."""
    from . import tblog
    logger = tblog.TBLogger(["unicycle_move_to_pose_fixed"] + list(exp_tags), runs_dir=runs_dir)
    logger.write_config(dict(config or {}, state_start=list(state_start), state_goal=list(state_goal)))
    simulator(torch.tensor(state_start, dtype=torch.float64), torch.tensor(state_goal, dtype=torch.float64), logger=logger)
    logger.summary_writer.close()
    return logger.experiment_logs_dir


def _obstacle_recipe(max_risk, exp_tag, **over):
    cfg = dict(dt=0.001, numSteps=2000, cbf_gammas=[5.0, 5.0], term_weights=[0.7, 0.3], max_risk=max_risk, true_L=12.0,
               mean_L=1.0, kernel_diag_A=[1e-2, 1e-2, 1e-2], enable_learning=False, train_every_n_steps=20)
    cfg.update(over)

    def exp(runs_dir="data/runs", **kw):
        c = dict(cfg, **kw)
        sim = lambda x, xg, logger: track_trajectory_ackerman_clf_bayesian(
            x, xg, dt=c["dt"], numSteps=c["numSteps"],
            cbfs=lambda a, b: obstacles_at_mid_from_start_and_goal(a, b, term_weights=tuple(c["term_weights"])),
            cbf_gammas=c["cbf_gammas"], enable_learning=c["enable_learning"], train_every_n_steps=c["train_every_n_steps"],
            mean_dynamics_gen=lambda: AckermannDrive(L=c["mean_L"], kernel_diag_A=c["kernel_diag_A"]),
            true_dynamics_gen=lambda: AckermannDrive(L=c["true_L"]), max_risk=c["max_risk"], logger=logger,
            learned_dynamics=c.get("learned_dynamics"), training_iter=c.get("training_iter", 100))
        return unicycle_demo(sim, exp_tags=[exp_tag], runs_dir=runs_dir,
                             config={k: v for k, v in c.items() if k != "learned_dynamics"})
    return exp


# :1889-1945  fixed-kernel model, true L = 12 against a mean model with L = 1: the mean-only condition (max_risk 0.5)
# collides, the Bayesian one (max_risk 0.01) keeps clear
unicycle_mean_cbf_collides_obstacle_exp = _obstacle_recipe(0.5, "mean_cbf_collides")
unicycle_bayes_cbf_safe_obstacle_exp = _obstacle_recipe(0.01, "bayes_cbf_safe_obstacle")
# :1948-2013  learned residual on a deliberately wrong mean model (L = 12 vs 1): with periodic refits the loop passes
# the obstacles, without (one refit beyond the horizon) it gets stuck
unicycle_learning_helps_avoid_getting_stuck_exp = _obstacle_recipe(
    0.01, "learning_helps_avoid_getting_stuck", true_L=1.0, mean_L=12.0, kernel_diag_A=[1.0, 1.0, 1.0],
    enable_learning=True, train_every_n_steps=400)
unicycle_no_learning_gets_stuck_exp = _obstacle_recipe(
    0.01, "no_learning_gets_stuck", true_L=1.0, mean_L=12.0, kernel_diag_A=[1.0, 1.0, 1.0], enable_learning=True,
    train_every_n_steps=2000)


class NoPlanner:
    """unicycle_move_to_pose.py:1522-1530: the goal itself at every step."""

    def __init__(self, x_goal):
        self.x_goal = x_goal

    def plan(self, t):
        return self.x_goal

    def dot_plan(self, t):
        return torch.zeros_like(self.x_goal)


def unicycle_speed_test_matrix_vector_exp(max_train_variations=(64, 64 + 16, 64 + 32, 128), ntimes=10, repeat=50,
                                          errorbartries=20, state_start=(-3.0, -1.0, -math.pi / 4),
                                          state_goal=(0.0, 0.0, math.pi / 4), numSteps=512, dt=0.01,
                                          true_dynamics_gen=None, mean_dynamics_gen=None, logger=None, exps=None,
                                          training_iter=50, device="cuda", dtype=torch.float64):
    """unicycle_move_to_pose.py:2031-2152: inference time and learning error of the four regressors (matrix-variate
    full / diag, vector-variate full / diag; the vector ones carry (1+m) n = 9 task outputs) wrapped in
    LearnedShiftInvariantDynamics, on one closed-loop trajectory of the true unicycle under the mean CLF controller;
    `custom_predict_fullmat(Xtest); clear_cache()` on 20 headings spanning the training set, min over `repeat` of
    `ntimes` calls (device synchronised after every call: the reference's timing is host side).
    Returns {name: {max_train: dict(elapsed, errors)}}; logged under the reference's tags when a logger is given."""
    import timeit
    from functools import partial
    from .control_affine_model import (ControlAffineRegressorExact, ControlAffineRegMatrixDiag,
                                       ControlAffineRegressorVector, ControlAffineRegVectorDiag)
    from .pendulum import measure_batch_error
    from .sampling import sample_generator_trajectory
    f = dict(device=device, dtype=dtype)
    true_dynamics_gen = true_dynamics_gen or partial(AckermannDrive, L=1.0)
    mean_dynamics_gen = mean_dynamics_gen or partial(AckermannDrive, L=12.0)
    exps = exps or dict(matrix=ControlAffineRegressorExact, vector=ControlAffineRegressorVector,
                        vectordiag=ControlAffineRegVectorDiag, matrixdiag=ControlAffineRegMatrixDiag)
    true_model = true_dynamics_gen()
    goal = torch.tensor(state_goal, **f)

    def trajectory():
        ctrl = ControllerCLF(NoPlanner(goal), coordinate_converter=lambda x, x_g: x, dynamics=CartesianDynamics(),
                             clf=CLFCartesian(), device=device, dtype=dtype)
        Xdot, X, U = sample_generator_trajectory(true_model, numSteps, dt=dt, x0=torch.tensor([state_start], **f),
                                                 controller=ctrl.control)
        return Xdot[:, 0], X[:, 0], U[:, 0]

    def true_F(Xt):                                               # [b, 1+m, n]
        return torch.cat([true_model.f_func(Xt).unsqueeze(-1), true_model.g_func(Xt)], dim=-1).transpose(-2, -1)

    def heading_grid(Xtrain):                                     # (:2099-2110) mgrid with 1 x 1 x 20 cells
        lo, hi = Xtrain.min(dim=0).values, Xtrain.max(dim=0).values
        th = lo[2] + (hi[2] - lo[2]) / 20 * torch.arange(20, **f)
        return torch.stack([lo[0].expand(20), lo[1].expand(20), th], dim=-1).contiguous()

    def make(cls, max_train):
        return LearnedShiftInvariantDynamics(dt=dt, learned_dynamics_class=cls, mean_dynamics=mean_dynamics_gen(),
                                             max_train=max_train, device=device, dtype=dtype)

    Xdot, X, U = trajectory()
    if logger is not None:
        for t, (dx, x, u) in enumerate(zip(Xdot, X, U)):
            logger.add_tensors("traj", dict(dx=dx, x=x, u=u), t)
    order = np.arange(X.shape[0] - 1)
    out = {name: {} for name in exps}
    for max_train in max_train_variations:
        np.random.shuffle(order)
        idx = torch.from_numpy(order[:max_train].copy()).to(device)
        Xtrain, Utrain, XdotTrain = X[idx], U[idx], Xdot[idx]
        Xtest = heading_grid(Xtrain)
        for name, cls in exps.items():
            model = make(cls, max_train)
            model.fit(Xtrain, Utrain, XdotTrain, training_iter=training_iter)

            def call():
                model.custom_predict_fullmat(Xtest)
                model.clear_cache()
                torch.cuda.synchronize()
            call()
            elapsed = min(timeit.repeat(call, repeat=repeat, number=ntimes)) / ntimes
            errors = []
            for _ in range(errorbartries):
                # compute_errors (:2156-2221).  As upstream, the regressor of an error sample is constructed and queried
                # WITHOUT being fitted (no fit call there): the figure is the variance-weighted error of the prior model
                # (mean dynamics, prior covariance) on 400 states of a fresh trajectory
                dX2, X2, U2 = trajectory()
                np.random.shuffle(order)
                tdx = torch.from_numpy(order[-400:].copy()).to(device)
                mdl = make(cls, max_train)
                Xt = X2[tdx]
                mean, var = mdl.custom_predict_fullmat(Xt)
                b, D = Xt.shape[0], mean.numel() // Xt.shape[0]
                blocks = var.reshape(b, D, b, D)[torch.arange(b), :, torch.arange(b), :]
                errors.append(float(measure_batch_error(mean.reshape(b, D), blocks, true_F(Xt).reshape(b, D).to(mean))))
            out[name][max_train] = dict(elapsed=elapsed, errors=errors)
            if logger is not None:
                logger.add_scalars(name, dict(elapsed=elapsed), max_train)
                logger.add_tensors(name, dict(errors=np.asarray(errors)), max_train)
    return out


def _with_playback(exp):
    def run(**kw):
        from . import tblog
        return tblog.playback_logfile(exp(**kw))
    return run


unicycle_mean_cbf_collides_obstacle = _with_playback(unicycle_mean_cbf_collides_obstacle_exp)
unicycle_bayes_cbf_safe_obstacle = _with_playback(unicycle_bayes_cbf_safe_obstacle_exp)
unicycle_learning_helps_avoid_getting_stuck = _with_playback(unicycle_learning_helps_avoid_getting_stuck_exp)
unicycle_no_learning_gets_stuck = _with_playback(unicycle_no_learning_gets_stuck_exp)
