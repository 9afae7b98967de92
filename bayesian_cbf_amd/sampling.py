"""bayes_cbf/sampling.py:49-75 `sample_generator_trajectory`, batched: Bt closed control loops
advance together (one row per Monte-Carlo rollout), the plant step is the HIP Euler kernel."""
import torch

from . import ops


def sample_generator_trajectory(dynamics_model, D, dt=0.01, x0=None, controller=None, record=True):
    """Returns (Xdot[D,Bt,n] | None, X[D+1,Bt,n], U[D,Bt,m]).  `controller(x[Bt,n], t) -> u[Bt,m]`."""
    X0 = x0.clone()
    Bt, n = X0.shape
    m = dynamics_model.ctrl_size
    X = torch.empty(D + 1, Bt, n, dtype=X0.dtype, device=X0.device) if record else None
    U = torch.empty(D, Bt, m, dtype=X0.dtype, device=X0.device) if record else None
    x = X0
    if record:
        X[0] = x
    for t in range(D):
        u = controller(x, t=t).contiguous()
        ops.unicycle_step(x, u, float(dt), float(dynamics_model.L))      # x_{t+1} = x_t + (f + g u) dt
        if record:
            U[t] = u
            X[t + 1] = x
    return None, (X if record else x), U
