"""Oracle: matrix-variate GP posterior over F(x) = [f(x) g(x)]  (test infrastructure).

numpy/scipy restatement of the reference's hand-written prediction path.  Notation
(SURVEY.md section 0): N train points, n state dim, m ctrl dim, UH[N,1+m] rows [1,u_i],
Bm[(1+m),(1+m)] covariance over "controls", A[n,n] covariance over state dims,
ell[n] ARD lengthscale, s2 outputscale, M0[(1+m),n] constant prior mean.

Randomness is never drawn here: every `torch.rand` of the reference (the `make_psd`
jitter) is an explicit argument.
"""
import numpy as np
import scipy.linalg as sla


# --------------------------------------------------------------------------- kernels
def softplus(x):
    """gpytorch's positive-constraint transform (third-party, published definition)."""
    x = np.asarray(x, dtype=np.float64)
    return np.log1p(np.exp(-np.abs(x))) + np.maximum(x, 0.0)


def rbf_ard_kernel(X1, X2, ell, s2):
    """k(x,x') = s2 * exp(-1/2 sum_d ((x_d - x'_d)/ell_d)^2).

    gpytorch ScaleKernel(RBFKernel(ard_num_dims=n)) as constructed at
    bayes_cbf/control_affine_model.py:164-171 (arithmetic lives in gpytorch; restated from
    its published definition; cross-checked by the analytic RBF of
    bayes_cbf/trigger_interval.py:32-43 and tests/test_gp_algebra.py:91-127).
    X1[a,n], X2[b,n] -> [a,b].
    """
    X1 = np.atleast_2d(X1)
    X2 = np.atleast_2d(X2)
    d = (X1[:, None, :] - X2[None, :, :]) / np.asarray(ell).reshape(1, 1, -1)
    return s2 * np.exp(-0.5 * np.sum(d * d, axis=-1))


def matern52_ard_kernel(X1, X2, ell, s2):
    """OPT-IN data kernel (no reference counterpart -- the reference has no Matern kernel; PARITY UNPINNED):
    gpytorch `ScaleKernel(MaternKernel(nu=2.5, ard_num_dims=n))`, restated from its published definition:
    k = s2 (1 + sqrt5 r + 5/3 r^2) exp(-sqrt5 r),  r = |(x - x') / ell|.  Checked against scikit-learn's
    `Matern(nu=2.5)` (an independent implementation) in tests/test_oracle_formulas.py."""
    X1, X2 = np.atleast_2d(X1), np.atleast_2d(X2)
    d = (X1[:, None, :] - X2[None, :, :]) / np.asarray(ell, dtype=np.float64).reshape(1, 1, -1)
    r = np.sqrt((d * d).sum(-1))
    a = np.sqrt(5.0) * r
    return s2 * (1.0 + a + 5.0 / 3.0 * r * r) * np.exp(-a)


def matern52_ard_grad(X, x, ell, s2):
    """d k(X_i, x) / d x_d [N, n] of the opt-in Matern-5/2 kernel: -(5/3) s2 (1 + a) exp(-a) (x_d - X_id) / ell_d^2, a = sqrt5 r
    (differentiating the definition above; checked against central differences in tests/test_oracle_formulas.py)."""
    ell = np.asarray(ell, dtype=np.float64)
    z = (np.asarray(x)[None, :] - np.atleast_2d(X)) / ell
    a = np.sqrt(5.0 * (z * z).sum(-1))
    return -(5.0 / 3.0) * s2 * ((1.0 + a) * np.exp(-a))[:, None] * z / ell


def rbf_matern52_ard_kernel(X1, X2, ell, s2):
    """OPT-IN data kernel, the PRODUCT RBF x Matern-5/2 on one set of ARD length scales (BASELINE.json north_star's
    "RBF x Matern"; no reference counterpart, PARITY UNPINNED): by definition the elementwise product of the two kernels
    above over s2 -- k = s2 exp(-r^2 / 2) (1 + sqrt5 r + 5/3 r^2) exp(-sqrt5 r)."""
    return rbf_ard_kernel(X1, X2, ell, 1.0) * matern52_ard_kernel(X1, X2, ell, s2)


def rbf_matern52_ard_grad(X, x, ell, s2):
    """d k(X_i, x) / d x_d [N, n] of the product kernel by the product rule from the two factors' own gradients."""
    ell = np.asarray(ell, dtype=np.float64)
    kr = rbf_ard_kernel(X, np.asarray(x)[None], ell, 1.0)[:, 0]
    km = matern52_ard_kernel(X, np.asarray(x)[None], ell, s2)[:, 0]
    z = (np.asarray(x)[None, :] - np.atleast_2d(X)) / ell
    dkr = -(z / ell) * kr[:, None]
    return dkr * km[:, None] + kr[:, None] * matern52_ard_grad(X, x, ell, s2)


DATA_KERNELS = {"rbf": rbf_ard_kernel, "matern52": matern52_ard_kernel, "rbf_matern52": rbf_matern52_ard_kernel}
KERNEL_KXX = {"rbf": 1.0, "matern52": 5.0 / 3.0, "rbf_matern52": 8.0 / 3.0}     # d2 k / dx_d dx'_d at x' = x, units of s2 / ell_d^2


def index_kernel_covar(covar_factor, raw_var):
    """gpytorch IndexKernel.covar_matrix = F F^T + diag(softplus(raw_var)); used for A and B
    (bayes_cbf/matrix_variate_multitask_kernel.py:37-41, control_affine_model.py:158-163)."""
    F = np.asarray(covar_factor, dtype=np.float64)
    return F @ F.T + np.diag(softplus(raw_var))


def homogeneous_controls(U, fill=1.0):
    """UH = [fill, u]  (control_affine_model.py:186-193 for training rows, :428-434 for queries)."""
    U = np.atleast_2d(U)
    return np.concatenate([np.full((U.shape[0], 1), fill, dtype=U.dtype), U], axis=1)


# --------------------------------------------------------------------------- refit state
def kb_matrix(X, UH, Bm, ell, s2, kernel="rbf"):
    """K_b = k(X,X) o (UH Bm UH^T)   (control_affine_model.py:370-372).  kernel="matern52" / "rbf_matern52": the opt-in data kernels."""
    k = DATA_KERNELS[kernel]
    return k(X, X, ell, s2) * (UH @ Bm @ UH.T)


def make_psd(Kb, rand_draws, cholesky_tries=10, perturb_init=1e-5, perturb_scale=10):
    """Jittered Cholesky with x10 retry  (control_affine_model.py:899-921).

    rand_draws[t] is the U[0,1)^N vector the reference draws with torch.rand on try t.
    Returns (Kb + diag(jitter), L, tries_used); raises LinAlgError after the last try.
    """
    rand_draws = np.atleast_2d(rand_draws)
    factor = perturb_init
    for ntry in range(cholesky_tries):
        r = rand_draws[min(ntry, rand_draws.shape[0] - 1)]
        Kbp = Kb + factor * np.diag(r)
        try:
            L = np.linalg.cholesky(Kbp)
            return Kbp, L, ntry + 1
        except np.linalg.LinAlgError:
            if ntry == cholesky_tries - 1:
                raise
            factor = factor * perturb_scale
    raise AssertionError("unreachable")


def perturbed_cholesky(X, UH, Bm, ell, s2, rand_draws):
    """L = chol(K_b + jitter)   (control_affine_model.py:366-377)."""
    _, L, _ = make_psd(kb_matrix(X, UH, Bm, ell, s2), rand_draws)
    return L


def residual_targets(Xdot, UH, M0):
    """Y = Xdot - M0^T uh_i per row   (control_affine_model.py:525-532, 1033-1042).

    The reference evaluates its mean module on raw X (no mask column) which, for a constant
    mean, gives vec(M0) on every row unless x[0]==1.0 exactly; M0 is
    therefore an explicit [(1+m),n] input here.
    """
    return Xdot - UH @ M0


def cholesky_solve(Y, L):
    """K^-1 Y given K = L L^T  (torch.cholesky_solve, control_affine_model.py:545,1053)."""
    return sla.solve_triangular(L, sla.solve_triangular(L, Y, lower=True), lower=True, trans='T')


# --------------------------------------------------------------------------- vector-variate view
def custom_predict(X, UH, Y, L, A, Bm, ell, s2, M0, Xtest, UHtest, Xtestp=None, UHtestp=None,
                   compute_cov=True):
    """ControlAffineRegressor.custom_predict  (control_affine_model.py:390-613), grad_gp=False.

    Returns (mean[b,n], scalar_var[b,b'], cov[1, b*n, b'*n] = kron(scalar_var, A)).
    """
    Xtest = np.atleast_2d(Xtest)
    if Xtestp is None:
        Xtestp = Xtest
    if UHtestp is None:
        UHtestp = UHtest
    fu_mean_test = UHtest @ M0                                   # :485-493
    kb_star = rbf_ard_kernel(X, Xtest, ell, s2) * (UH @ Bm @ UHtest.T)        # :536
    alpha = cholesky_solve(Y, L)                                 # :545
    mean = fu_mean_test + kb_star.T @ alpha                      # :547
    if not compute_cov:
        return mean, None, 0 * A
    kb_star_p = rbf_ard_kernel(X, Xtestp, ell, s2) * (UH @ Bm @ UHtestp.T)    # :550-552
    kb_ss = rbf_ard_kernel(Xtest, Xtestp, ell, s2) * (UHtest @ Bm @ UHtestp.T)  # :553
    v = sla.solve_triangular(L, kb_star, lower=True)             # :565 (general solve of a triangular L)
    vp = sla.solve_triangular(L, kb_star_p, lower=True)          # :575
    scalar_var = kb_ss - v.T @ vp                                # :586
    cov = np.kron(scalar_var, A)[None]                           # :602
    return mean, scalar_var, cov


# --------------------------------------------------------------------------- matrix-variate view
def custom_predict_matrix(X, UH, Y, L, A, Bm, ell, s2, M0, Xtest, Xtestp=None, compute_cov=True,
                          rand_draws2=None):
    """ControlAffineRegressorExact._custom_predict_matrix  (control_affine_model.py:983-1096).

    Returns (mean_k[b,n,1+m], A, BkXX[b,b',1+m,1+m]).  `rand_draws2` are the U[0,1)^{b(1+m)}
    vectors of the second make_psd (:1089); None adds no jitter (the pure formula A.2).
    """
    Xtest = np.atleast_2d(Xtest)
    if Xtestp is None:
        Xtestp = Xtest
    b, bp = Xtest.shape[0], Xtestp.shape[0]
    N = X.shape[0]
    m1 = UH.shape[1]
    fX_mean_test = np.broadcast_to(M0.T[None], (b, M0.shape[1], m1))          # :1022-1023
    kb_star = rbf_ard_kernel(Xtest, X, ell, s2)[:, :, None] * (UH @ Bm)[None]  # :1051  [b,N,1+m]
    Bdagger = np.stack([cholesky_solve(kb_star[i], L) for i in range(b)])      # :1053
    mean_k = fX_mean_test + np.einsum('kn,bkc->bnc', Y, Bdagger)              # :1055
    if not compute_cov:
        return mean_k, A, np.zeros((b, bp, m1, m1))
    kb_star_p = rbf_ard_kernel(Xtestp, X, ell, s2)[:, :, None] * (UH @ Bm)[None]
    # The reference reuses kb_star for both sides (it only supports Xtestp is Xtest there,
    # :1079-1088); with Xtestp given we follow the formula of its docstring.
    Bdagger_p = Bdagger if Xtestp is Xtest else np.stack(
        [cholesky_solve(kb_star_p[i], L) for i in range(bp)])
    KB = np.kron(rbf_ard_kernel(Xtest, Xtestp, ell, s2), Bm)                  # :1062-1063
    lhs = kb_star.transpose(0, 2, 1).reshape(b * m1, N)                        # :1082-1083
    rhs = Bdagger_p.transpose(1, 0, 2).reshape(N, bp * m1)                     # :1085-1086
    BkXX = KB - lhs @ rhs                                                      # :1079-1088
    if rand_draws2 is not None:
        BkXX, _, _ = make_psd(BkXX, rand_draws2)                               # :1089
    BkXX = BkXX.reshape(b, m1, bp, m1).transpose(0, 2, 1, 3)                   # :1091
    return mean_k, A, BkXX


def exact_custom_predict(X, UH, Y, L, A, Bm, ell, s2, M0, Xtest, UHtest, Xtestp=None,
                         UHtestp=None, compute_cov=True, rand_draws2=None):
    """ControlAffineRegressorExact.custom_predict  (control_affine_model.py:931-961).

    Returns (meanFXU[b,n], varFXU[b,b',n,n])."""
    mean_k, A, BkXX = custom_predict_matrix(X, UH, Y, L, A, Bm, ell, s2, M0, Xtest, Xtestp,
                                            compute_cov, rand_draws2)
    if UHtestp is None:
        UHtestp = UHtest
    meanFXU = np.einsum('bnc,bc->bn', mean_k, UHtest)                          # :954
    if not compute_cov:
        return meanFXU, np.zeros((mean_k.shape[0], BkXX.shape[1]) + A.shape)
    s = np.einsum('bc,bpcd,pd->bp', UHtest, BkXX, UHtestp)                     # :956-958
    return meanFXU, s[:, :, None, None] * A


def custom_predict_fullmat(X, UH, Y, L, A, Bm, ell, s2, M0, Xtest, rand_draws2=None):
    """ControlAffineRegressorExact.custom_predict_fullmat  (control_affine_model.py:963-980).

    Returns (vec(M_k)[b(1+m)n], kron(B_k, A)[b(1+m)n, b(1+m)n])."""
    mean_k, A, BkXX = custom_predict_matrix(X, UH, Y, L, A, Bm, ell, s2, M0, Xtest, None, True,
                                            rand_draws2)
    b, n, m1 = mean_k.shape
    meanFX = mean_k.transpose(0, 2, 1)                                         # :974
    Bk2 = BkXX.transpose(0, 2, 1, 3).reshape(b * m1, b * m1)                   # :975
    return meanFX.reshape(-1), np.kron(Bk2, A)


# --------------------------------------------------------------------------- per-step
def posterior_step(L, alpha, X, UHB, ell, s2, Bm, M0, xq, jitter2=None):
    """One query per independent instance (regime I): closed form of SURVEY Appendix A.2.

    Phi = diag(k*(x)) UH Bm;  W = L^-1 Phi;  M_k = M0^T + alpha^T Phi;  B_k = k(x,x) Bm - W^T W
    (same arithmetic as control_affine_model.py:1051-1088 with b=1).  Leading axis = instance.
    L[Bt,N,N], alpha[Bt,N,n], X[Bt,N,n], UHB[Bt,N,1+m] (=UH Bm), ell[Bt,n], s2[Bt], Bm[Bt,1+m,1+m],
    M0[Bt,1+m,n], xq[Bt,n]; jitter2[Bt,1+m] is added to diag(B_k) (the explicit second
    make_psd jitter, :1089).  Returns (Mk[Bt,n,1+m], Bk[Bt,1+m,1+m]).
    """
    Bt = L.shape[0]
    n = X.shape[2]
    m1 = UHB.shape[2]
    Mk = np.empty((Bt, n, m1))
    Bk = np.empty((Bt, m1, m1))
    for i in range(Bt):
        kstar = rbf_ard_kernel(X[i], xq[i][None], ell[i], s2[i])[:, 0]
        Phi = kstar[:, None] * UHB[i]
        W = sla.solve_triangular(L[i], Phi, lower=True)
        Mk[i] = M0[i].T + alpha[i].T @ Phi
        Bk[i] = s2[i] * Bm[i] - W.T @ W
        if jitter2 is not None:
            Bk[i] += np.diag(jitter2[i])
    return Mk, Bk


def refit_state(X, U, Xdot, Bm, ell, s2, M0, rand_draws):
    """Everything a refit produces for one instance: UH, K_b, L, Y, alpha, UHB."""
    UH = homogeneous_controls(U)
    Kb = kb_matrix(X, UH, Bm, ell, s2)
    Kbp, L, tries = make_psd(Kb, rand_draws)
    Y = residual_targets(Xdot, UH, M0)
    alpha = cholesky_solve(Y, L)
    return dict(UH=UH, Kb=Kb, Kbp=Kbp, L=L, tries=tries, Y=Y, alpha=alpha, UHB=UH @ Bm)


def chol_append(L, knew, kappa):
    """Bordered Cholesky: chol([[K,k],[k^T,kappa]]) from L=chol(K)  (no reference counterpart;
    parity is defined against full re-factorisation)."""
    N = L.shape[0]
    l = sla.solve_triangular(L, knew, lower=True)
    d2 = kappa - l @ l
    if d2 <= 0:
        raise np.linalg.LinAlgError("appended point makes K_b non positive definite")
    Lnew = np.zeros((N + 1, N + 1))
    Lnew[:N, :N] = L
    Lnew[N, :N] = l
    Lnew[N, N] = np.sqrt(d2)
    return Lnew


# --------------------------------------------------------------------------- marginal likelihood (fit)
def marginal_log_likelihood(X, UH, Y, A, Bm, ell, s2, M0, jitter, kernel="rbf"):
    """log p(vec Y) of the training set under the matrix-variate prior: covariance K_b (x) A over (points, state dims)
    (HetergeneousMatrixVariateKernel on mask-1 rows, matrix_variate_multitask_kernel.py:112-118: K11 = (H K(x)B H') (x) A,
    H K(x)B H' = k(X,X) o (UH B UH')), constant mean UH M0, no observation noise (IdentityLikelihood,
    control_affine_model.py:244).  This is what ExactMarginalLogLikelihood evaluates inside `fit`
    (control_affine_model.py:309-321) up to its 1/num_data factor and gpytorch's own jitter policy -- the arithmetic
    lives in the un-vendored gpytorch fork: PARITY UNPINNED, restated from the Gaussian log density.
      log p = -1/2 tr(A^-1 R' K_b^-1 R) - n/2 logdet K_b - N/2 logdet A - N n/2 log 2 pi,   R = Y - UH M0."""
    N, n = Y.shape
    Kb = kb_matrix(X, UH, Bm, ell, s2, kernel) + np.diag(np.asarray(jitter, dtype=np.float64))
    L = np.linalg.cholesky(Kb)
    R = Y - UH @ M0
    W = sla.solve_triangular(L, R, lower=True)
    quad = np.trace(np.linalg.solve(A, W.T @ W))
    logdetK = 2.0 * np.log(np.diag(L)).sum()
    logdetA = np.linalg.slogdet(A)[1]
    return -0.5 * quad - 0.5 * n * logdetK - 0.5 * N * logdetA - 0.5 * N * n * np.log(2.0 * np.pi)


# ------------------------------------------------------------------------------------------------
# CoGP comparator: ControlAffineRegressorVector, control_affine_model.py:1106-1330.
def rbf_linear_kernel(X1, X2, ell, s2, lin):
    """ScaleKernel(RBFKernel() + LinearKernel()) (control_affine_model.py:1121-1122): one lengthscale,
    k = s2 (exp(-1/2 |x-x'|^2/ell^2) + lin x'x')."""
    X1, X2 = np.asarray(X1, dtype=np.float64), np.asarray(X2, dtype=np.float64)
    d = (X1[:, None, :] - X2[None, :, :]) / float(np.ravel(ell)[0])
    return s2 * (np.exp(-0.5 * (d * d).sum(-1)) + lin * (X1 @ X2.T))


def cogp_kb_matrix(X, UH, Sigma, ell, s2, lin):
    """K[(i,a),(j,c)] = k(x_i,x_j) [(uh_i' (x) I_n) Sigma (uh_j (x) I_n)]_ac   (:1190-1214)."""
    N, n = X.shape
    Hb = np.kron(UH, np.eye(n))                                   # (N n, (1+m) n), torch_kron(UHtrain, In)
    S = Hb @ Sigma @ Hb.T
    return np.kron(rbf_linear_kernel(X, X, ell, s2, lin), np.ones((n, n))) * S


def cogp_refit_state(X, U, Xdot, Sigma, ell, s2, lin, M0, rand_draws):
    UH = homogeneous_controls(U)
    Kb = cogp_kb_matrix(X, UH, Sigma, ell, s2, lin)
    Kbp, L, tries = make_psd(Kb, rand_draws)
    Y = (Xdot - UH @ M0).reshape(-1)                              # vec over (sample, state) (:1252-1262)
    return dict(UH=UH, L=L, Y=Y, alpha=cholesky_solve(Y[:, None], L)[:, 0], tries=tries, Kb=Kbp)


def cogp_custom_predict_matrix(X, UH, Y, L, Sigma, ell, s2, lin, M0, Xtest, rand_draws2=None):
    """(mean_k[b,n,1+m], KkXX[b,b,(1+m)n,(1+m)n])  (:1232-1330); rand_draws2[b(1+m)n] = the make_psd draw (:1318)."""
    N, n = X.shape
    C = UH.shape[1]
    b = Xtest.shape[0]
    Hb = np.kron(UH, np.eye(n))
    HS = Hb @ Sigma                                               # (N n, (1+m) n)
    kxs = rbf_linear_kernel(Xtest, X, ell, s2, lin)               # (b, N)
    kb_star = np.repeat(kxs, n, axis=1)[:, :, None] * HS[None]    # (b, N n, (1+m) n)
    alpha = cholesky_solve(Y[:, None], L)
    mean_k = M0.T[None] + np.einsum("bkc,k->bc", kb_star, alpha[:, 0]).reshape(b, C, n).transpose(0, 2, 1)
    v = np.stack([sla.solve_triangular(L, kb_star[i], lower=True) for i in range(b)])       # (b, N n, (1+m) n)
    vb = v.transpose(1, 0, 2).reshape(N * n, b * C * n)
    KkXX = np.kron(rbf_linear_kernel(Xtest, Xtest, ell, s2, lin), Sigma) - vb.T @ vb
    if rand_draws2 is not None:
        KkXX = KkXX + np.diag(1e-5 * np.asarray(rand_draws2))
    return mean_k, KkXX.reshape(b, C * n, b, C * n).transpose(0, 2, 1, 3)


def cogp_marginal_log_likelihood(X, UH, Y, Sigma, ell, s2, lin, jitter):
    """log N(y; 0, K + diag(jitter)) of the vector-variate GP, y = vec(Y) (parity unpinned, like the MVGP fit)."""
    K = cogp_kb_matrix(X, UH, Sigma, ell, s2, lin) + np.diag(jitter)
    L = np.linalg.cholesky(K)
    y = np.asarray(Y, dtype=np.float64).reshape(-1)
    a = cholesky_solve(y[:, None], L)[:, 0]
    return -0.5 * y @ a - np.log(np.diag(L)).sum() - 0.5 * y.size * np.log(2 * np.pi)
