"""Host logic of the GP expression algebra (no GPU): the operators build the reference's tree shapes
(bayes_cbf/gp_algebra.py:30-49) and the lowering recognises exactly the closed-form shapes."""
import pytest
import torch


class _Model:
    state_size, ctrl_size = 2, 1


def _leaves(u):
    from bayesian_cbf_amd.gp_algebra import GaussianProcess
    mdl = _Model()
    z = lambda x, xp=None: None
    f_gp = GaussianProcess(z, z, (2,), name="f", source=(mdl, "f", None))
    fu_gp = GaussianProcess(z, z, (2,), name="fu", source=(mdl, "fu", u))
    return mdl, f_gp, fu_gp


def test_rel_degree_1_expression_flattens_to_one_condition():
    from bayesian_cbf_amd import gp_algebra as ga
    u = torch.zeros(1)
    mdl, f_gp, fu_gp = _leaves(u)
    g1 = ga.DeterministicGP(lambda x: x, shape=(2,))
    h1 = ga.DeterministicGP(lambda x: x.sum(), shape=(1,))
    expr = (g1.t() @ fu_gp + h1 * 2.0) * -1.0
    assert isinstance(expr, ga.GaussianProcessMulExpr) and expr.shape == (1,)
    terms, _ = ga._flatten(expr, 1.0)
    assert sorted((c, t[0]) for c, t in terms) == [(-2.0, "det"), (-1.0, "L1")]
    low = ga.lower(expr)
    assert low.rel_degree == 1 and low.model is mdl and low.u is u
    x = torch.tensor([0.5, 2.0])
    assert float(low.cst_fn(x)) == pytest.approx(-5.0)
    assert torch.allclose(low.grad_h(x), -x)
    assert sum([g1.t() @ fu_gp, h1]).shape == (1,)          # sum() starts from 0 (controllers.py:307-317)


def test_rel_degree_2_expression_as_cbc2_gp_writes_it():
    from bayesian_cbf_amd import gp_algebra as ga
    u = torch.ones(1)
    mdl, f_gp, fu_gp = _leaves(u)
    gh = ga.DeterministicGP(lambda x: x, shape=(2,), jac=lambda x: torch.eye(2))
    h = ga.DeterministicGP(lambda x: x[0], shape=(1,))
    L1h = gh.t() @ f_gp
    expr = ga.GradientGP(L1h, x_shape=(2,)).t() @ fu_gp + h * 0.5 + L1h * 3.0
    low = ga.lower(expr)
    assert low.rel_degree == 2 and low.k_alpha == [1.0, 3.0] and low.scale == 1.0 and low.u is u
    assert float(low.h(torch.tensor([4.0, 0.0]))) == pytest.approx(2.0)
    low2 = ga.lower(expr * 2.0)                               # an overall factor scales mean and variance terms
    assert low2.scale == 2.0 and low2.k_alpha == [1.0, 3.0]
    assert float(low2.h(torch.tensor([4.0, 0.0]))) == pytest.approx(2.0)


def test_unsupported_shapes_raise():
    from bayesian_cbf_amd import gp_algebra as ga
    mdl, f_gp, fu_gp = _leaves(torch.zeros(1))
    _, f2, fu2 = _leaves(torch.zeros(1))
    g = ga.DeterministicGP(lambda x: x, shape=(2,))
    with pytest.raises(NotImplementedError):
        ga.lower(f_gp.t() @ fu_gp)                            # product of two random vectors
    with pytest.raises(NotImplementedError):
        ga.lower(g.t() @ fu_gp + g.t() @ fu2)                 # two different models
    with pytest.raises(NotImplementedError):
        ga.lower(ga.DeterministicGP(lambda x: x[0], shape=(1,)) * 2.0)
    # gradient of F u, not of L_f h: no fused form -- it goes to the general evaluator (gp_eval, GPU tests)
    assert ga.GradientGP(g.t() @ fu_gp, x_shape=(2,))._lie1_or_none() is None
    assert ga.GradientGP(g.t() @ f_gp, x_shape=(2,))._lie1_or_none() is not None


def test_jet_product_rule_against_autograd():
    """gp_eval's jets (value, d/dx, d/dx', d2/dx dx') through sums, products, transposes and traces equal torch.autograd
    on the same scalar function of (x, x')."""
    from bayesian_cbf_amd import gp_eval as ge
    torch.manual_seed(0)
    n = 3
    A1, A2 = torch.randn(2, 2, n, dtype=torch.float64), torch.randn(2, 2, n, dtype=torch.float64)

    def mats(x, xp):
        Mx = torch.sin(A1 @ x) + torch.outer(x[:2], x[1:])            # [2,2] function of x
        Mp = torch.cos(A2 @ xp) * xp[0]                               # [2,2] function of x'
        Mxp = torch.exp(-0.5 * ((x - xp) ** 2).sum()) * (A1 @ x) @ (A2 @ xp).t()   # both
        return Mx, Mp, Mxp

    def scalar(x, xp):
        Mx, Mp, Mxp = mats(x, xp)
        return (Mx.t() @ Mxp @ Mp).trace() * 2.0 + (Mxp @ Mxp.t()).trace()

    def jet_of(fn, x, xp, r, c):
        v = fn(x, xp)
        Jx, Jp = torch.autograd.functional.jacobian(lambda a, b: fn(a, b).reshape(-1), (x, xp))
        H = torch.autograd.functional.jacobian(
            lambda b: torch.autograd.functional.jacobian(lambda a: fn(a, b).reshape(-1), x, create_graph=True), xp)
        return ge.Jet(v.reshape(r, c), Jx.reshape(r, c, n).permute(2, 0, 1), Jp.reshape(r, c, n).permute(2, 0, 1),
                      H.reshape(r, c, n, n).permute(2, 3, 0, 1))

    x, xp = torch.randn(n, dtype=torch.float64), torch.randn(n, dtype=torch.float64)
    jMx = jet_of(lambda a, b: mats(a, b)[0], x, xp, 2, 2)
    jMp = jet_of(lambda a, b: mats(a, b)[1], x, xp, 2, 2)
    jMxp = jet_of(lambda a, b: mats(a, b)[2], x, xp, 2, 2)
    out = ge.jadd(ge.jscale(ge.jtrace(ge.jmatmul(ge.jmatmul(ge.jT(jMx), jMxp), jMp)), 2.0),
                  ge.jtrace(ge.jmatmul(jMxp, ge.jT(jMxp))))
    ref = jet_of(scalar, x, xp, 1, 1)
    for got, want in ((out.v, ref.v), (out.dx, ref.dx), (out.dp, ref.dp), (out.dxp, ref.dxp)):
        assert torch.allclose(got, want, rtol=1e-10, atol=1e-12)


def test_general_evaluator_on_handmade_leaves_matches_reference_golden():
    """Every propagation rule of gp_algebra.py:109-255, 319-402 -- incl. the inner product of two RANDOM vectors and
    GradientGP's mean / derivative kernel / cross-covariance (total derivative when x' IS x) -- on leaves made of plain
    torch callables, against values recorded from the executed reference (tests/golden/gen_golden.py gp_algebra:
    `handmade_trees`, restated here on this package's algebra).  Runs on the CPU: such leaves have no device source."""
    import os
    import numpy as np
    from bayesian_cbf_amd import gp_algebra as ga
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gpalgebra_handmade.npz"))
    n = G["W1"].shape[0]
    W1, W2, A1, A2, C12 = (torch.as_tensor(G[k], dtype=torch.float64) for k in ("W1", "W2", "A1", "A2", "C12"))
    rbf = lambda x, xp, l: torch.exp(-0.5 * ((x - xp) ** 2).sum() / l ** 2)
    f = ga.GaussianProcess(lambda x: torch.sin(W1 @ x), lambda x, xp: rbf(x, xp, 0.9) * A1, (n,), name="f")
    g = ga.GaussianProcess(lambda x: torch.cos(W2 @ x) + x, lambda x, xp: rbf(x, xp, 1.3) * A2 * (1 + 0.1 * x @ xp),
                           (n,), name="g")
    f.register_covar(g, lambda x, xp: rbf(x, xp, 1.1) * C12)
    d = ga.DeterministicGP(lambda x: torch.tanh(x) + 0.5 * x.flip(0), shape=(n,), name="d")
    L1 = d.t() @ f
    gL1 = ga.GradientGP(L1, x_shape=(n,))
    L2 = gL1.t() @ g
    rr = f.t() @ g
    mix = (L1 * 0.7 + rr) * -1.3 + d.t() @ g
    trees = dict(L1=L1, gL1=gL1, L2=L2, rr=rr, mix=mix)
    for i in range(G["xs"].shape[0]):
        x, xp = torch.as_tensor(G["xs"][i]), torch.as_tensor(G["xps"][i])
        for name, e in trees.items():
            for key, val in (("mean", e.mean(x)), ("knl_xxp", e.knl(x, xp)), ("knl_xx", e.knl(x, x)),
                             ("covar_g_xxp", e.covar(g, x, xp)), ("covar_f_xx_same", e.covar(f, x, x))):
                want = G["t_%s_%s" % (name, key)][i]
                got = val.detach().numpy().reshape(want.shape)
                np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-11 * max(1.0, np.abs(want).max()),
                                           err_msg="%s.%s[%d]" % (name, key, i))
