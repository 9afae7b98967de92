"""The formula-level pins of the reference's own test suite, applied to the CPU oracle:
Kronecker structure of the matrix-variate kernel (tests/test_control_affine_kernel.py:37-50, 69-111), the cone
conversion identity (tests/test_controllers.py:14-32), analytic RBF gradient / Hessian (tests/test_gp_algebra.py:91-142)
and the regression fixture tests/data/Xtrain_Utrain_X_interpolate_lazy_tensor_error.npz (a data file of the
reference's tests, committed under tests/golden/)."""
import os

import numpy as np

from oracle import cbc as ocbc
from oracle import cbc2 as ocbc2
from oracle import gp_posterior as ogp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _toy_kernel(X1, X2):
    return np.exp(-((X1[:, None, :] - X2[None, :, :]) ** 2).sum(-1))      # the reference test's DataKernel (:31-34)


def test_train_block_kronecker_structure():
    """K_train = kron(H kron(K, B) H', A) with H = blockdiag(uh_i')  (test_control_affine_kernel.py:37-39): the oracle's
    K_b = K o (UH B UH') is exactly the first Kronecker factor, and its posterior on the training points themselves
    reproduces the vec-form solve with that full matrix."""
    rng = np.random.default_rng(0)
    D, n, m = 5, 1, 2
    X, U = rng.normal(size=(D, n)), rng.normal(size=(D, m))
    UH = ogp.homogeneous_controls(U)
    Wa, Wb = rng.normal(size=(n, n)), rng.normal(size=(1 + m, 1 + m))
    A, B = Wa @ Wa.T + np.eye(n), Wb @ Wb.T + np.eye(1 + m)
    K = _toy_kernel(X, X)
    H = np.zeros((D, D * (1 + m)))
    for i in range(D):
        H[i, i * (1 + m):(i + 1) * (1 + m)] = UH[i]
    first = H @ np.kron(K, B) @ H.T
    np.testing.assert_allclose(K * (UH @ B @ UH.T), first, rtol=1e-12, atol=1e-12)
    # RBF with unit lengthscale/outputscale differs from the toy kernel only by the 1/2 in the exponent
    np.testing.assert_allclose(ogp.kb_matrix(X * np.sqrt(2.0), UH, B, np.ones(n), 1.0), first, rtol=1e-12, atol=1e-12)
    # vec-form posterior mean at a test point with the full Kronecker matrix == matrix-variate formulas
    Y = rng.normal(size=(D, n))
    Kfull = np.kron(first + 1e-6 * np.eye(D), A)
    xs = rng.normal(size=(1, n))
    ks = _toy_kernel(X, xs)[:, 0]
    Phi = ks[:, None] * (UH @ B)                                            # [D, 1+m]
    Mk = np.linalg.solve(first + 1e-6 * np.eye(D), Y).T @ Phi               # [n, 1+m]
    cross = np.kron(H @ np.kron(ks[:, None], B), A)                         # cov(train vec, F(x*) vec): [(D n), (1+m) n]
    mean_vec = cross.T @ np.linalg.solve(Kfull, Y.reshape(-1))
    np.testing.assert_allclose(mean_vec.reshape(1 + m, n), Mk.T, rtol=1e-8, atol=1e-10)


def test_cone_conversion_identity():
    """|A y + b| = sqrt(u'Vu + bfv'u + v) and c'y + d = bfe'u + e  (test_controllers.py:14-32)."""
    rng = np.random.default_rng(1)
    for m, extravars in ((2, 2), (1, 1), (3, 0)):
        W = rng.normal(size=(m + 1, m + 1))
        Vh = W @ W.T + 1e-3 * np.eye(m + 1)
        V, bfv, v = Vh[1:, 1:], 2 * Vh[1:, 0], Vh[0, 0]
        bfe, e, u = rng.uniform(size=m), rng.uniform(), rng.uniform(size=m)
        A, b, c, d = ocbc.convert_cbc_terms_to_socp_terms(bfe, e, V, bfv, v, extravars)
        y = np.concatenate([np.zeros(extravars), u])
        np.testing.assert_allclose(np.linalg.norm(A @ y + b), np.sqrt(u @ V @ u + bfv @ u + v), rtol=1e-10)
        np.testing.assert_allclose(c @ y + d, bfe @ u + e, rtol=1e-12)


def test_rbf_kernel_derivatives_are_analytic():
    """d k / d x and d^2 k / dx dx' of the ARD RBF used by the jets (test_gp_algebra.py:112-127) against central
    differences of the oracle's kernel."""
    rng = np.random.default_rng(2)
    n = 3
    ell, s2 = rng.uniform(0.5, 1.5, n), 0.8
    x, xp = rng.normal(size=n), rng.normal(size=n)
    k = lambda a, b: ogp.rbf_ard_kernel(a[None], b[None], ell, s2)[0, 0]
    g = -(x - xp) / ell ** 2 * k(x, xp)
    Hxx = (np.diag(1 / ell ** 2) - np.outer((x - xp) / ell ** 2, (x - xp) / ell ** 2)) * k(x, xp)
    h = 1e-5
    for d in range(n):
        e = np.zeros(n); e[d] = h
        np.testing.assert_allclose((k(x + e, xp) - k(x - e, xp)) / (2 * h), g[d], rtol=1e-6, atol=1e-9)
        for d2 in range(n):
            e2 = np.zeros(n); e2[d2] = h
            fd = (k(x + e, xp + e2) - k(x + e, xp - e2) - k(x - e, xp + e2) + k(x - e, xp - e2)) / (4 * h * h)
            np.testing.assert_allclose(fd, Hxx[d, d2], rtol=1e-4, atol=1e-6)
    # the jets the oracle feeds to the rel-degree-2 terms are consistent with these derivatives at x = x'
    X = rng.normal(size=(6, n)); U = rng.normal(size=(6, 1)); Y = rng.normal(size=(6, n))
    st = ogp.refit_state(X, U, Y, np.eye(2), ell, s2, np.zeros((2, n)), np.full((1, 6), 0.5))
    assert "L" in st and hasattr(ocbc2, "posterior_jets")


def test_reference_regression_fixture_posterior_is_finite():
    """tests/test_control_affine_regression.py:237-247: fit + predict on the data set that once broke the reference must
    not raise; here: the oracle posterior on that data (finite-difference targets as in the reference test)."""
    d = np.load(os.path.join(GOLDEN, "reference_fixture_Xtrain_Utrain_X.npz"))
    Xtr, Utr, Xt = d["Xtrain"], d["Utrain"], d["X"]
    Xdot = Xtr[1:] - Xtr[:-1]
    n, m = Xtr.shape[1], Utr.shape[1]
    st = ogp.refit_state(Xtr[:-1], Utr, Xdot, np.eye(1 + m), np.full(n, 0.6931), 0.6931, np.zeros((1 + m, n)),
                         np.full((1, len(Utr)), 0.5))
    Mk, Bk = ogp.posterior_step(st["L"][None], st["alpha"][None], Xtr[:-1][None], st["UHB"][None], np.full((1, n), 0.6931),
                                np.array([0.6931]), np.eye(1 + m)[None], np.zeros((1, 1 + m, n)), Xt)
    assert np.isfinite(Mk).all() and np.isfinite(Bk).all()
    assert np.all(np.linalg.eigvalsh(Bk[0]) > -1e-9)


def test_matern52_option_formula_against_an_independent_implementation():
    """The OPT-IN Matern-5/2 data kernel (no reference counterpart: parity unpinned) at formula level:
    the oracle's restatement of gpytorch's ScaleKernel(MaternKernel(nu=2.5, ard)) equals scikit-learn's
    `Matern(length_scale, nu=2.5)` (an independent implementation of the same published kernel) on random ARD inputs, has
    k(x, x) = s2, is positive definite, and is below the RBF near the origin / above it in the tails (heavier tails)."""
    from sklearn.gaussian_process.kernels import Matern
    from oracle import gp_posterior as ogp
    rng = np.random.default_rng(0)
    for n in (1, 2, 3):
        X1, X2 = rng.normal(size=(17, n)) * 2, rng.normal(size=(9, n)) * 2
        ell, s2 = rng.uniform(0.3, 2.0, size=n), 0.7
        K = ogp.matern52_ard_kernel(X1, X2, ell, s2)
        np.testing.assert_allclose(K, s2 * Matern(length_scale=ell, nu=2.5)(X1, X2), rtol=1e-12, atol=1e-15)
        Kxx = ogp.matern52_ard_kernel(X1, X1, ell, s2)
        np.testing.assert_allclose(np.diag(Kxx), s2, rtol=1e-15)
        assert np.linalg.eigvalsh(Kxx).min() > 0
    r = np.array([[0.0], [0.3], [6.0]])
    km, kr = ogp.matern52_ard_kernel(r, r[:1], [1.0], 1.0)[:, 0], ogp.rbf_ard_kernel(r, r[:1], [1.0], 1.0)[:, 0]
    assert km[0] == kr[0] == 1.0 and km[1] < kr[1] and km[2] > kr[2]


def test_matern52_derivatives_of_the_option_against_central_differences():
    """The derivative formulas the opt-in Matern-5/2 path is built on (device jets, rel-degree-2 terms, expression trees):
    d k / d x and d2 k / dx dx' at x' = x ((5/3) s2 / ell_d^2 on the diagonal), against central differences of the kernel
    formula that is itself checked against scikit-learn above."""
    rng = np.random.default_rng(11)
    n = 3
    X = rng.normal(size=(7, n))
    x = rng.normal(size=n)
    ell, s2 = np.array([0.7, 1.3, 0.9]), 0.8
    g = ogp.matern52_ard_grad(X, x, ell, s2)
    h = 1e-6
    for d in range(n):
        e = np.zeros(n); e[d] = h
        fd = (ogp.matern52_ard_kernel(X, (x + e)[None], ell, s2)[:, 0] - ogp.matern52_ard_kernel(X, (x - e)[None], ell, s2)[:, 0]) / (2 * h)
        np.testing.assert_allclose(g[:, d], fd, rtol=1e-6, atol=1e-9)
    # mixed second derivative at x' = x by differences of the first: d/dx'_e [d k(x, x') / dx_d] = -d/dx_e(...) by symmetry
    hh = 1e-4
    for d in range(n):
        e = np.zeros(n); e[d] = hh
        # k(x + e, x - e) = k(2e): second difference of phi(t) = k(t e_d) at 0 gives -d2k/dx_d dx'_d
        k0 = ogp.matern52_ard_kernel(x[None], x[None], ell, s2)[0, 0]
        k2 = ogp.matern52_ard_kernel((x + e)[None], (x - e)[None], ell, s2)[0, 0]
        np.testing.assert_allclose(-(2 * k2 - 2 * k0) / (2 * hh) ** 2, (5.0 / 3.0) * s2 / ell[d] ** 2, rtol=2e-3)


def test_product_kernel_option_is_the_product_and_its_gradient_matches_central_differences():
    """The opt-in RBF x Matern-5/2 kernel (no reference counterpart): the product of the two factor kernels by definition;
    its gradient (product rule) against central differences; the curvature at x' = x is (1 + 5/3) s2 / ell_d^2."""
    rng = np.random.default_rng(5)
    X, x = rng.normal(size=(9, 3)), rng.normal(size=3)
    ell, s2 = np.array([0.7, 1.3, 0.9]), 1.7
    K = ogp.rbf_matern52_ard_kernel(X, x[None], ell, s2)[:, 0]
    np.testing.assert_allclose(K, ogp.rbf_ard_kernel(X, x[None], ell, s2)[:, 0] * ogp.matern52_ard_kernel(X, x[None], ell, s2)[:, 0] / s2, rtol=1e-14)
    g = ogp.rbf_matern52_ard_grad(X, x, ell, s2)
    h = 1e-6
    for d in range(3):
        e = np.zeros(3); e[d] = h
        fd = (ogp.rbf_matern52_ard_kernel(X, (x + e)[None], ell, s2)[:, 0] - ogp.rbf_matern52_ard_kernel(X, (x - e)[None], ell, s2)[:, 0]) / (2 * h)
        np.testing.assert_allclose(g[:, d], fd, rtol=1e-7, atol=1e-9)
        hh = 1e-4
        e2 = np.zeros(3); e2[d] = hh
        k0 = ogp.rbf_matern52_ard_kernel(x[None], x[None], ell, s2)[0, 0]
        k2 = ogp.rbf_matern52_ard_kernel((x + e2)[None], (x - e2)[None], ell, s2)[0, 0]
        # k(x + e, x - e) = k0 - (1/2) kxx s2 (2 hh)^2 / ell^2 + O(h^3): d2k/dx dx' at x' = x  =  (k0 - k2) / (2 hh^2)
        np.testing.assert_allclose((k0 - k2) / (2 * hh * hh), ogp.KERNEL_KXX["rbf_matern52"] * s2 / ell[d] ** 2, rtol=2e-3)
