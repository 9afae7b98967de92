"""bayesian_cbf_amd.misc (the reference's bayes_cbf/misc.py helper names) against vectors recorded from the executed
reference (tests/golden/misc_surfaces.npz, generator: tests/golden/gen_golden.py misc)."""
import os

import numpy as np
import torch

from bayesian_cbf_amd import misc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
T = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64)


def close(a, b, tol=1e-12):
    np.testing.assert_allclose(a.detach().numpy() if torch.is_tensor(a) else a, b, rtol=tol, atol=tol)


def test_autograd_helpers():
    g = np.load(os.path.join(GOLDEN, "misc_surfaces.npz"))
    W, x, xp, P = T(g["W"]), T(g["x"]), T(g["xp"]), T(g["P"])
    fvec = lambda z: torch.tanh(W @ z) * (z @ z)
    with misc.variable_required_grad(x) as xg:
        assert xg is x and x.requires_grad
        close(misc.t_jac(fvec(xg), xg), g["jac_vec"])
        close(misc.t_jac(fvec(xg).sum(), xg), g["jac_scalar"])
    assert not x.requires_grad                                       # a leaf gets its flag back
    y = (x * 2.0).requires_grad_(False)
    with misc.variable_required_grad(x * 2.0) as yg:                 # a non-leaf: a detached copy
        assert yg.requires_grad and misc.isleaf(yg)
    f2 = lambda a, b: torch.sin(a @ P @ b) + (a * a) @ (b * b)
    close(misc.t_hessian(f2, x.clone(), xp.clone()), g["hess"])
    Qm, pv, r0 = T(g["Qm"]), T(g["pv"]), float(g["r0"])
    Q, p, r = misc.get_quadratic_terms(lambda z: z @ Qm @ z + pv @ z + r0, x.clone())
    close(Q, g["quad_Q"]); close(p, g["quad_p"]); close(r, g["quad_r"])
    close(Q, 0.5 * (g["Qm"] + g["Qm"].T))                            # (the Jacobian of the gradient / 2)
    a, b = misc.get_affine_terms(lambda z: pv @ z + r0, x.clone())
    close(a, g["aff_a"]); close(b, g["aff_b"])


def test_kron_schedule_and_small_helpers():
    g = np.load(os.path.join(GOLDEN, "misc_surfaces.npz"))
    close(misc.torch_kron(T(g["kron_A"]), T(g["kron_B"])), g["kron_AB"])
    close(misc.torch_kron(T(g["kron_A0"]), T(g["kron_B0"]), batch_dims=0), g["kron_AB0"])
    close(misc.torch_kron(T(g["kron_A0"]), T(g["kron_B0"]), batch_dims=0), np.kron(g["kron_A0"], g["kron_B0"]))
    eps = [misc.epsilon(i) for i in (0, 10, 500, 1000)] + [misc.epsilon(3, interpolate={0: 2.0, 10: 0.5})]
    close(np.array(eps), g["epsilon"])
    close(np.array([misc.normalize_radians(v) for v in (-7.0, -3.2, 0.0, 3.2, 9.5)]), g["normalize_radians"])
    assert misc.to_numpy(torch.ones(2, dtype=torch.float32)).dtype == np.float64 and misc.to_numpy("s") == "s"
    close(misc.t_hstack([torch.ones(2, 1), torch.zeros(2, 2)]), np.array([[1.0, 0, 0], [1.0, 0, 0]]))
    assert misc.t_vstack([torch.ones(1, 2), torch.zeros(2, 2)]).shape == (3, 2)
    close(misc.clip(torch.tensor([-3.0, 0.5, 9.0]), torch.tensor(-1.0), torch.tensor(2.0)), np.array([-1.0, 0.5, 2.0]))
    M = misc.random_psd(4)
    assert torch.linalg.eigvalsh(M).min() > -1e-12

    class Store:
        @misc.store_args
        def __init__(self, a, b=2, c="see"):
            self.ran = True
    st = Store(1, c="given")
    assert [str(getattr(st, k, "<unset>")) for k in ("a", "b", "c", "ran")] == list(g["store_args"])

    class Skip:
        @staticmethod
        def _d(m):
            return misc.store_args(m, skip=["b"])
    class S2:
        def __init__(self, a, b=2):
            pass
    S2.__init__ = misc.store_args(S2.__init__, skip=["b"])
    s2 = S2(5)
    assert s2.a == 5 and not hasattr(s2, "b")


def test_dynamics_model_base_classes():
    g = np.load(os.path.join(GOLDEN, "misc_surfaces.npz"))

    class Plant(misc.DynamicsModel):
        ctrl_size, state_size = 2, 3
        f_func = lambda self, X: torch.sin(X) * 0.5
        g_func = lambda self, X: torch.stack([torch.cos(X), X * 0.3], dim=-1)
    pl = Plant()
    Xb, Ub = T(g["dyn_X"]), T(g["dyn_U"])
    close(pl.forward(Xb, Ub), g["dyn_fwd_batch"])
    close(pl.forward(Xb[0], Ub[0, :, 0]), g["dyn_fwd_single"])
    close(pl.F_func(Xb), g["dyn_F"])
    pl.set_init_state(Xb[1])
    s1 = pl.step(Ub[1, :, 0], 0.05)
    s2 = pl.step(Ub[2, :, 0], 0.05)
    close(torch.stack([s1["x"], s2["x"]]), g["dyn_step_x"])
    close(torch.stack([s1["xdot"], s2["xdot"]]), g["dyn_step_xdot"])
    z = misc.ZeroDynamicsModel(2, 3)
    assert z.ctrl_size == 2 and z.state_size == 3
    close(z.f_func(Xb), g["zero_f"]); close(z.g_func(Xb), g["zero_g"])
    close(z.f_func(Xb[0]), g["zero_f1"]); close(z.g_func(Xb[0]), g["zero_g1"])
    try:
        misc.BayesianDynamicsModel()
    except TypeError:
        pass
    else:
        raise AssertionError("abstract classes must not instantiate")


def test_cbc2_quadratic_terms_falls_back_to_autograd_for_foreign_callables():
    """cbc2.py:7-23 on a callable u -> object with .mean(x) / .knl(x, x') that is NOT one of this package's expression trees
    (plain torch, differentiable in u): the reference's autograd extraction, against the reference's own output for the
    same functions (there built from its gp_algebra nodes: Det(tanh).t() @ GaussianProcess(...))."""
    from bayesian_cbf_amd.cbc2 import cbc2_quadratic_terms
    g = np.load(os.path.join(GOLDEN, "misc_surfaces.npz"))
    W, x, A1, G, u0 = T(g["W"]), T(g["x"]), T(g["cbc_A1"]), T(g["cbc_G"]), T(g["cbc_u0"])
    n = 3

    class Foreign:
        def __init__(self, u):
            self.u = u

        def mean(self, z):
            return torch.tanh(z) @ (torch.sin(W[:n] @ z) + G @ self.u)

        def knl(self, a, b):
            uh = torch.cat([torch.ones(1, dtype=a.dtype), self.u])
            return torch.tanh(a) @ (torch.exp(-0.5 * ((a - b) ** 2).sum()) * A1 * (uh @ uh)) @ torch.tanh(b)

    (mA, mb), (Q, p, r), mean, var = cbc2_quadratic_terms(Foreign, x.clone(), u0)
    for val, key in ((mA, "cbc_mean_A"), (mb, "cbc_mean_b"), (Q, "cbc_Q"), (p, "cbc_p"), (r, "cbc_r"), (mean, "cbc_mean"), (var, "cbc_var")):
        close(val.reshape(np.shape(g[key])), g[key], tol=1e-10)
