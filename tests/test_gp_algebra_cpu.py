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
    with pytest.raises(NotImplementedError):
        ga.GradientGP(g.t() @ fu_gp, x_shape=(2,)).mean(torch.zeros(2))   # gradient of F u, not of L_f h
