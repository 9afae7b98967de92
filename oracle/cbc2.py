"""Oracle: rel-degree-2 control barrier condition as a GP in u (test infrastructure).

Closed form of what the reference obtains by composing
    CBC2 = grad(L_f h)' (f + g u) + k_alpha[0] h + k_alpha[1] L_f h           (bayes_cbf/cbc2.py:26-33)
out of gp_algebra expressions (MatmulExpr :133-168, AddExpr :109-130, MulExpr :201-223,
GradientGP :319-402) and differentiating with autograd (cbc2_quadratic_terms, cbc2.py:7-23;
misc.py:268-285).  Everything is a function of the *jet* of the posterior at x:
    W = L^-1 Phi(x),  dW_d = L^-1 dPhi/dx_d,  Gram blocks W'W, dW_d'W, dW_d'dW_e,  Vw'W, Vw'dW_d.
"""
import numpy as np
import scipy.linalg as sla

from .gp_posterior import (rbf_ard_kernel, matern52_ard_kernel, matern52_ard_grad, rbf_matern52_ard_kernel,
                           rbf_matern52_ard_grad, KERNEL_KXX)

EIG_EPS = 2e-3   # gp_algebra.py:317: eigenvalues of the kernel Hessian in (-EPS, 0) are treated as rounding


def clean_hessian(H, eps=EIG_EPS, mode="reference"):
    """gp_algebra.py:384-392, restated literally.  Returns (H_clean, branch_fired).

        eigenvalues, eigenvectors = torch.eig(Hxx_k, eigenvectors=True)
        assert (eigenvalues[:, 0] > -eigeps).all()
        small_neg_eig = (eigenvalues[:, 0] > -eigeps) & (eigenvalues[:, 0] < 0)
        evalz[small_neg_eig] = 0;   Hxx_k = eigenvectors.T @ diag(evalz) @ eigenvectors

    `torch.eig` was LAPACK's GENERAL solver xGEEV (real parts of the eigenvalues, eigenvectors as columns); the harness
    that executes the reference maps it onto torch.linalg.eig (tests/golden/_refenv.py) and so does this function -- the
    SAME library call, because `V' diag(l) V` pairs eigenvalue k with ROW k of V and therefore depends on the order and
    the signs xGEEV happens to return (numpy's LAPACK build differs from torch's in ~2 % of 3x3 / 4x4 cases).
    mode="project" is the spectral projection V max(l,0) V' of the symmetric part (the non-default switch of the build)."""
    H = np.asarray(H, dtype=np.float64)
    if mode == "project":
        w, V = np.linalg.eigh(0.5 * (H + H.T))
        assert (w > -eps).all(), "Hessian must be positive definite (gp_algebra.py:386)"
        if (w < 0).any():
            return V @ np.diag(np.maximum(w, 0.0)) @ V.T, True
        return H, False
    import torch
    lam, vec = torch.linalg.eig(torch.from_numpy(H.copy()))
    evalz, eigenvectors = lam.real.numpy().copy(), vec.real.numpy()
    assert (evalz > -eps).all(), "Hessian must be positive definite (gp_algebra.py:386)"
    small_neg_eig = (evalz > -eps) & (evalz < 0)
    if small_neg_eig.any():
        evalz[small_neg_eig] = 0.0
        return eigenvectors.T @ np.diag(evalz) @ eigenvectors, True
    return H, False


def posterior_jets(L, Y, X, UHB, ell, s2, Bm, M0, x, kernel="rbf"):
    """Value and first x-derivatives of the posterior factors at one query x.

    Returns dict(Mk[n,C], dMk[n(d),n,C], Bk[C,C], G10[n(d),C,C] = dW_d'W, G11[n,n,C,C] = dW_d'dW_e).
    (control_affine_model.py:536-602 for the values; the derivatives are what autograd produces
    through k(X, x), :442-443.)
    """
    n = X.shape[1]
    if kernel == "matern52":                  # the opt-in data kernels (no reference counterpart)
        kstar = matern52_ard_kernel(X, x[None], ell, s2)[:, 0]
        dkm = matern52_ard_grad(X, x, ell, s2)
    elif kernel == "rbf_matern52":
        kstar = rbf_matern52_ard_kernel(X, x[None], ell, s2)[:, 0]
        dkm = rbf_matern52_ard_grad(X, x, ell, s2)
    else:
        kstar = rbf_ard_kernel(X, x[None], ell, s2)[:, 0]
    Phi = kstar[:, None] * UHB
    W = sla.solve_triangular(L, Phi, lower=True)
    Vw = sla.solve_triangular(L, Y, lower=True)
    dW = []
    for d in range(n):
        dk = dkm[:, d] if kernel != "rbf" else -(x[d] - X[:, d]) / ell[d] ** 2 * kstar
        dW.append(sla.solve_triangular(L, dk[:, None] * UHB, lower=True))
    Mk = M0.T + Vw.T @ W
    dMk = np.stack([Vw.T @ dW[d] for d in range(n)])
    Bk = s2 * Bm - W.T @ W
    G10 = np.stack([dW[d].T @ W for d in range(n)])
    G11 = np.stack([np.stack([dW[d].T @ dW[e] for e in range(n)]) for d in range(n)])
    return dict(Mk=Mk, dMk=dMk, Bk=Bk, G10=G10, G11=G11)


def cbc2_terms(jets, A, Bm, ell, s2, h, gh, Hh, k_alpha, u0, hessian_mode="reference", info=None, kernel="rbf"):
    """((mean_A, mean_b), (Q, p, r), mean(u0), var(u0)) of cbc2_quadratic_terms(cbc2_gp(...), x, u0).

    h, gh[n], Hh[n,n]: barrier value, gradient, Hessian at x.  The cross term C = cov(grad L_f h, f+gu)
    is frozen at u0, as in the reference (it comes out of t_jac without a graph, gp_algebra.py:395-402).
    """
    Mk, dMk, Bk, G10, G11 = (jets[k] for k in ("Mk", "dMk", "Bk", "G10", "G11"))
    n, C = Mk.shape
    m = C - 1
    e0 = np.zeros(C)
    e0[0] = 1.0
    a0 = np.concatenate([[1.0], u0])
    m0, Mt = Mk[:, 0], Mk[:, 1:]
    Agh = A @ gh
    phi0 = gh @ Agh

    def dsdz(a, ap):     # total derivative of s(z,a;z,a') w.r.t. z (both kernel arguments move)
        return np.array([-(a @ (G10[i] + G10[i].T) @ ap) for i in range(n)])

    # g = grad_x [ gh(x)' mu(x, e0) ]
    g = Hh.T @ m0 + np.array([gh @ dMk[i][:, 0] for i in range(n)])
    # H_ij = d2/dx_i dx'_j [ gh(x)'A gh(x') s(x,e0;x',e0) ] at x' = x
    s00 = Bk[0, 0]
    s_i = np.array([-G10[i][0, 0] for i in range(n)])                 # ds/dx_i (= ds/dx'_i by symmetry)
    kxx = KERNEL_KXX[kernel]                                # d2 k / dx_d dx'_d at x' = x in units of s2 / ell_d^2
    s_ij = np.array([[(kxx * s2 / ell[i] ** 2 * Bm[0, 0] if i == j else 0.0) - G11[i][j][0, 0] for j in range(n)]
                     for i in range(n)])
    HAg = Hh @ Agh
    H = (Hh @ A @ Hh) * s00 + np.outer(HAg, s_i) + np.outer(s_i, HAg) + phi0 * s_ij
    H, fired = clean_hessian(H, EIG_EPS, hessian_mode)       # the reference zeroes small negative eigenvalues (:384-392)
    if info is not None:
        info["branch_fired"] = bool(fired)
        info["H"] = H

    def Cmat(a):            # C = (d/dz [ A gh(z) s(z,a;z,e0) ])'
        sa0 = a @ Bk @ e0
        J = (A @ Hh) * sa0 + np.outer(Agh, dsdz(a, e0))      # J[k][i] = d c_k / d z_i
        return J.T
    Cu0 = Cmat(a0)
    C0 = Cmat(e0)
    dq = dsdz(e0, e0) * phi0 + s00 * 2.0 * HAg
    gAg = g @ A @ g
    gAgh = g @ Agh
    ka0, ka1 = k_alpha
    trC = np.trace(Cu0)
    ell1 = gh @ m0

    # mean(u) = g'(m0 + Mt u) + tr(C(u0)) + ka0 h + ka1 ell1
    mean_A = Mt.T @ g
    mean_b = g @ m0 + trC + ka0 * h + ka1 * ell1
    # var(u) polynomial
    Hs = H + H.T
    Q = 0.5 * Mt.T @ Hs @ Mt + gAg * Bk[1:, 1:]
    p = (Mt.T @ Hs @ m0 + 2.0 * gAg * Bk[1:, 0] + 2.0 * Mt.T @ Cu0.T @ g
         + ka1 * (2.0 * gAgh * Bk[1:, 0] + Mt.T @ dq + Mt.T @ C0 @ gh))
    r = (2.0 * trC ** 2 + m0 @ H @ m0 + gAg * Bk[0, 0] + 2.0 * m0 @ Cu0.T @ g + ka1 ** 2 * phi0 * s00
         + ka1 * (2.0 * gAgh * Bk[0, 0] + m0 @ dq + m0 @ C0 @ gh))
    mean = mean_A @ u0 + mean_b
    var = u0 @ Q @ u0 + p @ u0 + r
    return (mean_A, mean_b), (Q, p, r), mean, var
