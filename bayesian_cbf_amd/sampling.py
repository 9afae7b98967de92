"""bayes_cbf/sampling.py: the closed-loop rollout harness, with the reference's call surface.

`sample_generator_trajectory(dynamics_model, D, dt, x0, true_model, controller, controller_class, visualizer)`
(sampling.py:49-75) drives ANY plant through `dynamics_model.set_init_state(x0)` / `dynamics_model.step(u, dt) ->
{'xdot', 'x'}` and returns `(Xdot, X, U)`.  A 1-D `x0` is the reference's single trajectory ([D, n], [D+1, n], [D, m]);
a 2-D `x0` [Bt, n] runs Bt closed loops together (one Monte-Carlo rollout per row: [D, Bt, n], [D+1, Bt, n],
[D, Bt, m]) -- plants whose `step` takes a device batch (AckermannDrive: the HIP Euler kernel `bcbf_unicycle_step`)
advance all rows in one launch."""
from abc import ABC, abstractmethod

import torch


def controller_sine(xi, t=1):
    """sampling.py:7-9 (the default excitation of the reference: m = 1)."""
    m = 1
    return torch.sin(xi[..., 0:1]) * torch.abs(torch.rand(m)).to(xi) + 0.2 * torch.rand(1).to(xi)


class Visualizer(ABC):
    @abstractmethod
    def setStateCtrl(self, x, u, t=0, **kw):
        pass


class VisualizerZ(Visualizer):
    """The visualizer that shows nothing (sampling.py:16-18)."""

    def setStateCtrl(self, x, u, t=0, **kw):
        pass


def uncertainity_vis_kwargs(controller, x, u, dt):
    """One-step-ahead state distribution for visualizers (sampling.py:20-30): when `controller` is the bound `control`
    of an object whose `.model` is a Bayesian dynamics model, x_{t+1} ~ N(x + mean(x) dt, knl(x, x) dt^2)."""
    owner = getattr(controller, "__self__", None)
    model = getattr(owner, "model", None)
    if model is None or not hasattr(model, "fu_func_gp"):
        return dict()
    gp = model.fu_func_gp(u)
    return dict(xtp1=gp.mean(x) * dt + x, xtp1_var=gp.knl(x, x) * dt * dt)


class DynamicsModel(ABC):
    """What a plant offers the rollout (sampling.py:32-47)."""

    @property
    @abstractmethod
    def ctrl_size(self):
        pass

    @property
    @abstractmethod
    def state_size(self):
        pass

    @abstractmethod
    def step(self, u, dt):
        pass

    @abstractmethod
    def set_init_state(self, x0):
        pass


def sample_generator_trajectory(dynamics_model, D, dt=0.01, x0=None, true_model=None, controller=controller_sine,
                                controller_class=None, visualizer=None):
    """u_t = controller(x_t, t);  obs = dynamics_model.step(u_t, dt);  Xdot[t] = obs['xdot'], X[t+1] = obs['x'].

    controller_class: constructed as `controller_class(dt=dt, true_model=true_model)`, its `.control` is the controller
    (sampling.py:54-57).  visualizer: `setStateCtrl(x_t, u_t, t=t, **uncertainity_vis_kwargs(...))` every step; the
    default shows nothing (and then the one-step-ahead distribution, a GP query per step, is not evaluated)."""
    if controller_class is not None:
        controller = controller_class(dt=dt, true_model=true_model).control
    quiet = visualizer is None or type(visualizer) is VisualizerZ
    m, n = dynamics_model.ctrl_size, dynamics_model.state_size
    if x0 is None:
        x0 = torch.rand(n)
    elif not isinstance(x0, torch.Tensor):
        x0 = torch.tensor(x0)
    f = dict(dtype=x0.dtype, device=x0.device)
    lead = tuple(x0.shape[:-1])                      # () for the reference's single trajectory, (Bt,) for a batch
    U = torch.empty((D,) + lead + (m,), **f)
    X = torch.zeros((D + 1,) + lead + (n,), **f)
    Xdot = torch.zeros((D,) + lead + (n,), **f)
    X[0] = x0
    dynamics_model.set_init_state(X[0].clone() if lead else X[0])
    for t in range(D):
        x_t = X[t]
        U[t] = torch.as_tensor(controller(x_t, t=t)).to(**f).reshape(lead + (m,))
        if not quiet:
            visualizer.setStateCtrl(x_t, U[t], t=t, **uncertainity_vis_kwargs(controller, x_t, U[t], dt))
        obs = dynamics_model.step(U[t], dt)
        Xdot[t] = obs["xdot"]
        X[t + 1] = obs["x"]
    return Xdot, X, U


def sample_generator_independent(dynamics_model, D):
    """D independent uniform (x, u) samples and their state derivatives (sampling.py:76-88)."""
    m, n = dynamics_model.ctrl_size, dynamics_model.state_size
    U, X = torch.rand(D, m), torch.rand(D, n)
    Xdot = torch.stack([dynamics_model.f_func(X[i]) + dynamics_model.g_func(X[i]) @ U[i] for i in range(D)])
    return Xdot, X, U
