"""Mirror of the generic (pendulum-style) controllers of bayes_cbf/controllers.py on libbcbf.

Same classes, constructor arguments and method names as the reference; the arithmetic runs on the device:
constraint terms through `cbc2_quadratic_terms` (jet / posterior kernels + closed-form terms), the cone rows through
`bcbf_controller_cones` and the program through `bcbf_coneqp_f64` -- no cvxpy / cvxopt / GUROBI.  `control(x, t)`
takes one state like the reference, or a batch x[b, n] (one program per row, one launch per stage).

The small host-side nominal controllers (Zero / Greedy / epsilon-greedy, controllers.py:166-213, 269-285) and
`NamedAffineFunc` (:739-771) are here as upstream has them; `ControlCBFLearned()` is constructible with its defaults
(nominal controller = the greedy one-step tracker inside the epsilon-greedy explorer).  Not mirrored (SURVEY 2.1 #8:
outside the GP / conic-program path): LQR / ILQR (they need the external `lqr` / `mpc` packages) and the plotters."""
import math
import random
from abc import ABC, abstractmethod

import numpy as np
import torch

from . import ops
from .cbc2 import cbc2_quadratic_terms, pack_terms
from .gp_algebra import GaussianProcess
from .optimizers import InfeasibleProblemError


class NamedFunc:
    def __init__(self, func, name):
        self.__name__ = name
        self.func = func

    def __call__(self, *args, **kwargs):
        return self.func(*args, **kwargs)


def identity(x):
    return x


def to_numpy(x):
    return x.detach().double().cpu().numpy() if isinstance(x, torch.Tensor) else x


def add_diag_const(Q, const=1.0):
    """[[Q, 0], [0, const]]  (controllers.py:46-51)."""
    out = Q.new_zeros(Q.shape[0] + 1, Q.shape[1] + 1)
    out[:-1, :-1] = Q
    out[-1, -1] = const
    return out


class Controller(ABC):
    """Controller interface (controllers.py:54-61)."""
    needs_ground_truth = False

    @abstractmethod
    def control(self, xi, t=None):
        pass


def epsilon(i, interpolate={0: 1, 1000: 0.01}):
    """Log-linear interpolation between two (step, value) pairs (misc.py:261-265)."""
    (si, sv), (ei, ev) = list(interpolate.items())
    return math.exp((i - si) / (ei - si) * (math.log(ev) - math.log(sv)) + math.log(sv))


def clip(x, min_, max_):
    """misc.py:287-288."""
    return torch.max(torch.min(x, max_), min_)


class ZeroController(Controller):
    """u = 0 (controllers.py:166-171)."""

    def __init__(self, model, Q, R, x_goal, numSteps, dt, ctrl_range):
        self.u_dim = R.shape[-1]

    def control(self, x, t=None):
        return x.new_zeros(*x.shape[:-1], self.u_dim)


class GreedyController(Controller):
    """One-step greedy tracking of x_goal (controllers.py:174-213): with G = g(x) dt, lam = 1/2,
    u = (lam R dt + (1 - lam) G'P G)^-1 (1 - lam) G'P (x_g - x - f(x) dt).  One state or a batch of states."""

    def __init__(self, model, Q, R, x_goal, numSteps, dt, ctrl_range):
        self.x_goal, self.model, self.Q, self.R = x_goal, model, Q, R
        self.numSteps, self.dt, self.ctrl_range = numSteps, dt, ctrl_range

    def clf(self, x):
        return (x - self.x_goal) ** 2

    def grad_clf(self, x):
        return 2 * (x - self.x_goal)

    def control(self, x, t=None):
        with torch.no_grad():
            f = dict(dtype=x.dtype, device=x.device)
            x_g, P, R, lam = self.x_goal.to(**f), self.Q.to(**f), self.R.to(**f) * self.dt, 0.5
            xb = x.reshape(-1, x.shape[-1])
            fx = self.dt * torch.as_tensor(self.model.f_func(xb), **f)
            Gx = self.dt * torch.as_tensor(self.model.g_func(xb), **f)              # [b, n, m]
            Q = lam * R + (1 - lam) * Gx.transpose(-2, -1) @ P @ Gx
            c = (1 - lam) * Gx.transpose(-2, -1) @ (P @ (x_g - xb - fx).unsqueeze(-1))
            u = torch.linalg.solve(Q, c).squeeze(-1)
            assert u.shape[-1] == self.R.shape[-1]
            return u[0] if x.dim() == 1 else u


class EpsilonGreedyController(ABC):
    """A uniformly random action with probability eps(t), eps annealed log-linearly from egreedy_scheme[0] at t = 0 to
    egreedy_scheme[1] at t = numSteps; clipped to ctrl_range (controllers.py:269-285)."""

    def __init__(self, base_controller, u_dim, numSteps, egreedy_scheme, ctrl_range):
        self.base_controller, self.u_dim, self.numSteps = base_controller, u_dim, numSteps
        self.egreedy_scheme, self.ctrl_range = egreedy_scheme, ctrl_range

    def control(self, x, t=None):
        min_, max_ = self.ctrl_range
        eps = epsilon(t, interpolate={0: self.egreedy_scheme[0], self.numSteps: self.egreedy_scheme[1]})
        u0 = self.base_controller.control(x, t=t)
        randomact = (torch.rand(self.u_dim) * (max_ - min_) + min_).to(u0)
        uegreedy = randomact.expand_as(u0) if random.random() < eps else u0
        return clip(uegreedy, torch.as_tensor(min_).to(u0), torch.as_tensor(max_).to(u0))


class NamedAffineFunc(ABC):
    """A(x) u - b(x) with a name for plots (controllers.py:739-771); the pendulum / car barrier and Lyapunov classes of
    the reference derive from it."""

    @property
    def __name__(self):
        return self.name

    @abstractmethod
    def value(self, x):
        """Scalar value function."""

    @abstractmethod
    def b(self, x):
        """A(x) @ u - b(x)"""

    @abstractmethod
    def A(self, x):
        """A(x) @ u - b(x)"""

    def __call__(self, x, u):
        return self.A(x) @ u - self.b(x)


class _SummedGP(GaussianProcess):
    """f_func_gp / fu_func_gp of a sum of models (GaussianProcessAddExpr, gp_algebra.py:109-130): means add;
    deterministic summands carry no covariance, so the kernel is the learned model's.  A leaf of the expression
    algebra (`source` = the summed model): conditions built on it lower onto the kernels with the mean shifted."""

    def __init__(self, learned_gp, det_means, shape, source):
        self._gp, self._dets = learned_gp, det_means

        def mean(x):
            out = self._gp.mean(x) if self._gp is not None else 0
            for d in self._dets:
                out = out + torch.as_tensor(d(x)).to(x)
            return out

        def knl(x, xp):
            if self._gp is None:
                return x.new_zeros(max(shape), max(shape))
            return self._gp.knl(x, xp)

        super().__init__(mean, knl, shape, name="sum", source=source)

    def covar(self, Z, x, xp):
        if self._gp is None:
            return x.new_zeros(max(self.shape), max(Z.shape))
        return self._gp.covar(getattr(Z, "_gp", Z), x, xp)


class SumDynamicModels:
    """f, g and fu_func_gp of a sum of dynamics models (controllers.py:288-317)."""

    def __init__(self, *models):
        assert len(models) >= 2
        self.models = models

    @property
    def ctrl_size(self):
        return self.models[0].ctrl_size

    @property
    def state_size(self):
        return self.models[0].state_size

    def f_func(self, x):
        return sum(torch.as_tensor(m.f_func(x)).to(x) for m in self.models)

    def g_func(self, x):
        return sum(torch.as_tensor(m.g_func(x)).to(x) for m in self.models)

    def fu_func_gp(self, u):
        learned = [m for m in self.models if hasattr(m, "fu_func_gp")]
        dets = [m for m in self.models if not hasattr(m, "fu_func_gp")]
        assert len(learned) <= 1, "one learned summand (the cross-covariance of two learned models is not defined)"
        fu = lambda m: (lambda x: torch.as_tensor(m.f_func(x)).to(x) + torch.as_tensor(m.g_func(x)).to(x) @ u.to(x))
        return _SummedGP(learned[0].fu_func_gp(u) if learned else None, [fu(m) for m in dets], (self.state_size,),
                         source=(self, "fu", u))

    def f_func_gp(self):
        learned = [m for m in self.models if hasattr(m, "f_func_gp")]
        dets = [m for m in self.models if not hasattr(m, "f_func_gp")]
        f = lambda m: (lambda x: torch.as_tensor(m.f_func(x)).to(x))
        return _SummedGP(learned[0].f_func_gp() if learned else None, [f(m) for m in dets], (self.state_size,),
                         source=(self, "f", None))


class MeanAdjustedModel(SumDynamicModels):
    """A prior mean model + a learned residual, refit every `train_every_n_steps` control steps on finite-difference
    targets minus the prior (controllers.py:320-378; the reference reads `self.x_dim`, `self.u_dim`, `self.dt` in
    `_train` without ever setting them -- they are constructor arguments here).  The buffering / schedule / targets
    live in `online.OnlineLearner`, shared with `LearnedShiftInvariantDynamics`; `hyper_refit_every` and
    `online_update` switch on the incremental (`bcbf_gp_append`) updates described there."""

    def __init__(self, x_dim, u_dim, mean_dynamics_model_class, model, max_train=None, train_every_n_steps=None,
                 enable_learning=None, dt=None, training_iter=100, hyper_refit_every=1, online_update=False, window=None):
        from .online import OnlineLearner
        self.mean_dynamics_model = mean_dynamics_model_class()
        super().__init__(model, self.mean_dynamics_model)
        self.model, self.max_train, self.train_every_n_steps = model, max_train, train_every_n_steps
        self.enable_learning, self.x_dim, self.u_dim, self.dt, self.training_iter = enable_learning, x_dim, u_dim, dt, training_iter
        self._learner = OnlineLearner(
            model, self._residual_targets, dt, train_every_n_steps, max_train, training_iter,
            subsample=lambda count, k: torch.randint(count, (k,)),        # :360-362: WITH replacement, as upstream
            enable_learning=bool(enable_learning), hyper_refit_every=hyper_refit_every, online_update=online_update, window=window)

    Xtrain = property(lambda self: self._learner.Xtrain)
    Utrain = property(lambda self: self._learner.Utrain)
    _has_been_trained_once = property(lambda self: self._learner.has_been_trained_once)

    def _residual_targets(self, X, U, Xdot):
        md = self.mean_dynamics_model
        mean = torch.as_tensor(md.f_func(X)).to(X) + torch.as_tensor(md.g_func(X)).to(X).bmm(U.unsqueeze(-1)).squeeze(-1)
        return Xdot - mean

    def _train(self):
        """Refit on everything buffered so far (:336-369)."""
        self._learner.dt = self.dt
        self._learner._refit_from_scratch(self.training_iter)

    def train(self, xi, uopt):
        ln = self._learner
        ln.dt, ln.enable_learning, ln.training_iter = self.dt, bool(self.enable_learning), self.training_iter
        ln.max_train, ln.train_every_n_steps = self.max_train, self.train_every_n_steps
        ln.observe(xi, uopt)


def _rows_to_cone(G, h):
    """Gq = [-c'; -A], hq = [d; b]  ->  (A, b, c, d)   (inverse of optimizers.py:6-39)."""
    return -G[1:], h[1:], -G[0], h[0]


class SOCPController(Controller):
    """min y_1  s.t.  |[0, sqrt(lambda) rho; sqrt(Q)(u - u_ref)]| <= y_1,  safety and stability cones
    (controllers.py:382-591), over y = [y_1; rho; u]."""

    def __init__(self, x_dim, u_dim, ctrl_reg, clf_relax_weight, net_model, cbfs, clf, unsafe_controller,
                 summary_writer=None):
        self.x_dim, self.u_dim, self.ctrl_reg, self.clf_relax_weight = x_dim, u_dim, ctrl_reg, clf_relax_weight
        self.net_model, self.cbfs, self.clf = net_model, cbfs, clf
        self.unsafe_controller, self.summary_writer = unsafe_controller, summary_writer

    # ---- the individual builders of the reference, each a slice of the same device kernel
    def _socp_objective(self, i, x, u0, yidx=0, extravars=None):
        assert yidx == 0 and extravars == 2
        ub = u0.reshape(1, -1).contiguous()
        G, h, _, _, _ = ops.controller_cones(None, ub, [], ctrl_reg=self.ctrl_reg, relax_weight=self.clf_relax_weight,
                                             extravars=extravars, objective=True)
        return _rows_to_cone(G[0], h[0])

    @staticmethod
    def convert_cbc_terms_to_socp_terms(bfe, e, V, bfv, v, extravars, testing=False):
        """(A, bfb, bfc, d) with |A y + bfb| = sqrt(var(u)), bfc'y + d = mean(u) + rho  (controllers.py:423-482)."""
        terms = pack_terms(((bfe, torch.as_tensor(e).reshape(()).to(bfe)), (V, bfv, torch.as_tensor(v).reshape(()).to(bfe))))
        terms = terms.reshape(1, 1, -1).contiguous()
        G, h, _, _, cst = ops.controller_cones(terms, None, [0], extravars=extravars, objective=False)
        if int(cst[0, 0]) != 0:
            raise RuntimeError("cholesky: Asq + 1e-3 I is not positive definite")
        A, bfb, bfc, d = _rows_to_cone(G[0], h[0])
        if testing:
            m = bfe.shape[-1]
            u0 = torch.rand(m).to(bfe)
            y = torch.cat([torch.zeros(extravars).to(A), u0.to(A)])
            np.testing.assert_allclose(float((A @ y + bfb) @ (A @ y + bfb)),
                                       float(u0 @ V @ u0 + bfv @ u0 + v), rtol=1e-2, atol=1e-3)
        return A, bfb, bfc, d

    def _terms(self, cbc, x, u0):
        return pack_terms(cbc2_quadratic_terms(cbc, x, u0))

    def _socp_stability(self, clc, t, x, u0, extravars=None):
        terms = self._terms(lambda u: clc(t, u), x, u0).reshape(1, 1, -1).contiguous()
        G, h, _, _, cst = ops.controller_cones(terms, None, [0], extravars=extravars, objective=False)
        if int(cst[0, 0]) != 0:
            raise RuntimeError("cholesky: Asq + 1e-3 I is not positive definite")
        return _rows_to_cone(G[0], h[0])

    def _socp_safety(self, cbc2, x, u0, safety_factor=None, extravars=None):
        terms = self._terms(cbc2, x, u0).reshape(1, 1, -1).contiguous()
        G, h, _, _, _ = ops.controller_cones(terms, None, [1], [safety_factor], extravars=extravars, objective=False)
        return _rows_to_cone(G[0], h[0])

    def _named_socp_constraints(self, t, x, u_ref, convert_out=to_numpy, extravars=None):
        cons = [("Objective", list(map(to_numpy, self._socp_objective(t, x, u_ref, yidx=0, extravars=extravars))))]
        cons += [("Safety_%d gt 0" % i,
                  list(map(to_numpy, self._socp_safety(cbf.cbc, x, u_ref, safety_factor=cbf.safety_factor(),
                                                       extravars=extravars))))
                 for i, cbf in enumerate(self.cbfs)]
        if self.clf is not None:
            cons += [("Stability gt 0", list(map(convert_out, self._socp_stability(self.clf.clc, t, x, u_ref,
                                                                                   extravars=extravars))))]
        return cons

    # ---- the per-step program, batched
    def program(self, xi, t, u_ref, extravars=2):
        """Device rows (G[b,Kt,nv], h[b,Kt], qdims, cstatus) of the program at states xi[b,n], references u_ref[b,m]."""
        rows, kinds, factors = [], [], []
        for cbf in self.cbfs:
            rows.append(self._terms(cbf.cbc, xi, u_ref))
            kinds.append(1)
            factors.append(cbf.safety_factor())
        if self.clf is not None:
            rows.append(self._terms(lambda u: self.clf.clc(t, u), xi, u_ref))
            kinds.append(0)
            factors.append(1.0)
        terms = torch.stack(rows, dim=1).contiguous() if rows else None
        ub = u_ref.to(terms) if terms is not None else u_ref
        G, h, qdims, l, cst = ops.controller_cones(terms, ub.contiguous(), kinds, factors, ctrl_reg=self.ctrl_reg,
                                                   relax_weight=self.clf_relax_weight, extravars=extravars, objective=True)
        return G, h, qdims, cst

    def control(self, xi, t=None, extravars=2):
        assert extravars == 2, "I assumed extravars to be delta"
        single = xi.dim() == 1
        u_ref = self.unsafe_controller.control(xi, t=t)
        xb, ub = xi.reshape(-1, self.x_dim), u_ref.reshape(-1, self.u_dim)
        G, h, qdims, cst = self.program(xb, t, ub, extravars=extravars)
        b, nv = G.shape[0], G.shape[2]
        q = G.new_zeros(b, nv)
        q[:, 0] = 1.0                                                   # linear objective [1, 0, 0..] (:575)
        y, status, _ = ops.coneqp(G.new_zeros(b, nv, nv), q, G, h, 0, qdims)
        badcone = (cst != 0).any(dim=1)
        # solver status, with rows whose stability cone could not be factored reported as BADCONE (3)
        self.last_status = torch.where(badcone, torch.full_like(status, 3), status)
        bad = self.last_status != 0
        if single and bool(bad[0]):
            raise InfeasibleProblemError("Infeasible problem: solver status %d" % int(self.last_status[0]))
        # a batch cannot raise per row (the reference raises InfeasibleProblemError, optimizers.py:83-89): rows without
        # a solution return the nominal control and are flagged in last_status, as ControllerCLFBayesian does
        uopt = torch.where(bad[:, None], ub.to(y), y[:, extravars:]).to(dtype=xi.dtype, device=xi.device)
        return uopt[0] if single else uopt


class QPController(Controller):
    """min lambda rho^2 + Q |u|^2  s.t.  0 <= c'y + d (the mean of the relaxed Lyapunov condition), y = [rho; u]
    (controllers.py:594-662; as in the reference u_ref only seeds the solver, it is not in the objective)."""

    def __init__(self, x_dim, u_dim, ctrl_reg, clf_relax_weight, net_model, cbfs, clf, unsafe_controller,
                 summary_writer=None):
        self.x_dim, self.u_dim, self.ctrl_reg, self.clf_relax_weight = x_dim, u_dim, ctrl_reg, clf_relax_weight
        self.net_model, self.cbfs, self.clf = net_model, cbfs, clf
        self.unsafe_controller, self.summary_writer = unsafe_controller, summary_writer

    def _qp_stability(self, clc, t, x, u0, extravars=None):
        terms = pack_terms(cbc2_quadratic_terms(lambda u: clc(t, u), x, u0))
        single = terms.dim() == 1
        terms = terms.reshape(-1, 1, terms.shape[-1]).contiguous()
        G, h, _, _, _ = ops.controller_cones(terms, None, [2], extravars=extravars, objective=False)
        return (-G[0, 0], h[0, 0]) if single else (-G[:, 0], h[:, 0])

    def control(self, xi, t=None, extravars=1):
        assert extravars == 1, "I assumed extravars to be delta"
        single = xi.dim() == 1
        u_ref = self.unsafe_controller.control(xi, t=t)
        xb, ub = xi.reshape(-1, self.x_dim), u_ref.reshape(-1, self.u_dim)
        terms = pack_terms(cbc2_quadratic_terms(lambda u: self.clf.clc(t, u), xb, ub)).reshape(xb.shape[0], 1, -1)
        G, h, _, l, _ = ops.controller_cones(terms.contiguous(), None, [2], extravars=extravars, objective=False)
        b, nv = G.shape[0], G.shape[2]
        d = torch.cat([G.new_full((1,), float(self.clf_relax_weight)), G.new_full((self.u_dim,), float(self.ctrl_reg))])
        P = (2.0 * torch.diag(d)).expand(b, nv, nv).contiguous()        # 2 A'A, A = diag(sqrt(lambda), sqrt(Q)..)
        y, status, _ = ops.coneqp(P, G.new_zeros(b, nv), G, h, l, [])
        if single and int(status[0]) != 0:
            raise InfeasibleProblemError("Infeasible problem: solver status %d" % int(status[0]))
        self.last_status = status
        bad = status != 0                       # batch: unsolved rows fall back to the nominal control (flagged above)
        uopt = torch.where(bad[:, None], ub.to(y), y[:, extravars:]).to(dtype=xi.dtype, device=xi.device)
        return uopt[0] if single else uopt


class ControlCBFLearned(Controller):
    """Learning controller shell (controllers.py:665-736): wraps the regressor in a MeanAdjustedModel, an exploration
    controller around the nominal one, and a QP/SOCP safety filter; `control` also feeds the learner."""
    needs_ground_truth = False

    def __init__(self, x_dim=2, u_dim=1, model=None, train_every_n_steps=10, dt=0.001, constraint_plotter_class=None,
                 plots_dir='data/runs/', ctrl_range=(-5., 5.), x_goal=None, x_quad_goal_cost=None, u_quad_cost=None,
                 numSteps=1000, unsafe_controller_class=GreedyController, cbfs=(), ground_truth_cbfs=(), exp_tags=(),
                 exploration_controller_class=EpsilonGreedyController, clf_class=None, egreedy_scheme=(1, 0.1),
                 summary_writer=None, x0=None, ctrl_reg=1., clf_relax_weight=100., enable_learning=False,
                 mean_dynamics_model_class=None, max_train=None, controller_class=QPController, planner_class=None,
                 training_iter=100):
        """Defaults as upstream, except: the nominal controller defaults to `GreedyController` (upstream: `LQRController`,
        which needs the external `lqr` package), goal and costs default to the origin / identity when not given (upstream:
        `torch.tensor(None)` raises), `model=None` builds a `ControlAffineRegressorExact`, `mean_dynamics_model_class=None` a
        zero prior mean, and `exploration_controller_class=None` means "no exploration wrapper"."""
        self.x_dim, self.u_dim, self.model, self.dt, self.numSteps = x_dim, u_dim, model, dt, numSteps
        self.ctrl_reg, self.clf_relax_weight, self.summary_writer = ctrl_reg, clf_relax_weight, summary_writer
        if model is None:                         # upstream passes None on to the model sum, which fails at the first query
            from .control_affine_model import ControlAffineRegressorExact
            model = self.model = ControlAffineRegressorExact(x_dim, u_dim)
        if mean_dynamics_model_class is None:     # upstream: None() raises; a zero prior mean is the neutral choice
            from .unicycle_move_to_pose import ZeroDynamicsModel
            mean_dynamics_model_class = lambda: ZeroDynamicsModel(m=u_dim, n=x_dim)
        dev = getattr(model, "device", "cpu")
        self.ctrl_range = torch.tensor(ctrl_range)
        self.x_goal = torch.as_tensor([0.0] * x_dim if x_goal is None else x_goal, device=dev)
        self.x_quad_goal_cost = torch.as_tensor(np.eye(x_dim) if x_quad_goal_cost is None else x_quad_goal_cost,
                                                device=dev).to(self.x_goal.dtype)
        self.u_quad_cost = torch.as_tensor(np.eye(u_dim) if u_quad_cost is None else u_quad_cost,
                                           device=dev).to(self.x_goal.dtype)
        self.net_model = MeanAdjustedModel(x_dim, u_dim, mean_dynamics_model_class, model, max_train=max_train,
                                           train_every_n_steps=train_every_n_steps, enable_learning=enable_learning,
                                           dt=dt, training_iter=training_iter)
        self.unsafe_controller = unsafe_controller_class(self.net_model, self.x_quad_goal_cost, self.u_quad_cost,
                                                         self.x_goal, numSteps, dt, self.ctrl_range)
        if exploration_controller_class is not None:      # the epsilon-greedy wrapper (controllers.py:269-285)
            self.unsafe_controller = exploration_controller_class(self.unsafe_controller, u_dim, numSteps,
                                                                  egreedy_scheme, self.ctrl_range)
        self.cbfs, self.ground_truth_cbfs = list(cbfs), list(ground_truth_cbfs)
        if clf_class is None:          # the reference calls None(...) here (:722): unconstructible upstream; allow no CLF
            self.clf = None
        else:
            planner = planner_class(torch.tensor(x0), self.x_goal, numSteps, dt) if planner_class is not None else None
            self.clf = clf_class(self.net_model, planner=planner)
        self._controller = controller_class(self.x_dim, self.u_dim, self.ctrl_reg, self.clf_relax_weight, self.net_model,
                                            self.cbfs, self.clf, self.unsafe_controller, self.summary_writer)

    def control(self, xi, t=None):
        uopt = self._controller.control(xi, t=t)
        self.net_model.train(xi, uopt)
        return uopt
