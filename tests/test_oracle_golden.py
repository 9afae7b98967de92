"""Pin the CPU oracle against golden vectors recorded from the executed reference
(tests/golden/gen_golden.py) -- SURVEY.md 8c.  fp64, tolerance 1e-10 relative: it is the same
arithmetic, only reassociated."""
import glob
import os

import numpy as np
import pytest

from oracle import gp_posterior as gp
from oracle import cbc, unicycle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
POSTERIOR_FILES = sorted(glob.glob(os.path.join(GOLDEN, "posterior_*.npz")))
RTOL, ATOL = 1e-9, 1e-11


def close(a, b, rtol=RTOL, atol=ATOL):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_golden_files_present():
    assert len(POSTERIOR_FILES) >= 5


@pytest.mark.parametrize("path", POSTERIOR_FILES, ids=os.path.basename)
def test_posterior_matches_reference(path):
    g = np.load(path)
    X, U, Xdot = g["X"], g["U"], g["Xdot"]
    A, B, ell, s2, M0 = g["A"], g["B"], g["ell"], float(g["s2"]), g["M0"]
    st = gp.refit_state(X, U, Xdot, B, ell, s2, M0, g["jitter_rand"])
    assert st["tries"] == 1
    close(st["L"], g["L"])
    UH, Y, L = st["UH"], st["Y"], st["L"]
    Xt, Ut, Xtp, Utp = g["Xtest"], g["Utest"], g["Xtestp"], g["Utestp"]
    UHt, UHtp = gp.homogeneous_controls(Ut), gp.homogeneous_controls(Utp)
    e0 = np.zeros_like(UHt)
    e0[:, 0] = 1

    # vector-variate view (ControlAffineRegressor.custom_predict)
    mean, sv, cov = gp.custom_predict(X, UH, Y, L, A, B, ell, s2, M0, Xt, UHt)
    close(mean, g["vec_mean"])
    close(cov, g["vec_cov"])
    mean, sv, cov = gp.custom_predict(X, UH, Y, L, A, B, ell, s2, M0, Xt, UHt, Xtp, UHtp)
    close(mean, g["vec_mean_x"])
    close(cov, g["vec_cov_x"])
    mean, sv, cov = gp.custom_predict(X, UH, Y, L, A, B, ell, s2, M0, Xt, e0)
    close(mean, g["vec_mean_f"])
    close(cov, g["vec_cov_f"])
    mean, sv, cov = gp.custom_predict(X, UH, Y, L, A, B, ell, s2, M0, Xt, gp.homogeneous_controls(Ut, 0.0))
    close(mean, g["vec_mean_gu"])
    close(cov, g["vec_cov_gu"])
    # single-state views: fu_func_mean / fu_func_knl / covar_fu_f / f_func_knl (:707-818)
    mean, sv, cov = gp.custom_predict(X, UH, Y, L, A, B, ell, s2, M0, Xt[:1], UHt[:1], Xtp[:1], UHt[:1])
    close(mean[0], g["fu_mean1"])
    close(cov[0], g["fu_knl1"])
    mean, sv, cov = gp.custom_predict(X, UH, Y, L, A, B, ell, s2, M0, Xt[:1], UHt[:1], Xtp[:1], e0[:1])
    close(cov[0], g["covar_fu_f1"])
    mean, sv, cov = gp.custom_predict(X, UH, Y, L, A, B, ell, s2, M0, Xt[:1], e0[:1], Xtp[:1], e0[:1])
    close(cov[0], g["f_knl1"])

    # matrix-variate view (ControlAffineRegressorExact)
    mean_k, _, BkXX = gp.custom_predict_matrix(X, UH, Y, L, A, B, ell, s2, M0, Xt,
                                               rand_draws2=g["mat_jitter2"])
    close(mean_k, g["mat_mean_k"])
    close(BkXX, g["mat_BkXX"])
    meanFXU, varFXU = gp.exact_custom_predict(X, UH, Y, L, A, B, ell, s2, M0, Xt, UHt,
                                              rand_draws2=g["exact_jitter2"])
    close(meanFXU, g["exact_meanFXU"])
    close(varFXU, g["exact_varFXU"])
    fm, fv = gp.custom_predict_fullmat(X, UH, Y, L, A, B, ell, s2, M0, Xt, rand_draws2=g["full_jitter2"])
    close(fm, g["full_mean"])
    close(fv, g["full_var"])

    # per-step closed form == the b=1 matrix-variate call, jitter made explicit
    Mk, Bk = gp.posterior_step(L[None], st["alpha"][None], X[None], st["UHB"][None], ell[None],
                               np.array([s2]), B[None], M0[None], Xt[:1],
                               jitter2=1e-5 * g["one_jitter2"][None])
    close(Mk[0], g["one_mean_k"][0])
    close(Bk[0], g["one_BkXX"][0, 0])


def test_chol_append_equals_refactorisation():
    g = np.load(POSTERIOR_FILES[0])
    st = gp.refit_state(g["X"], g["U"], g["Xdot"], g["B"], g["ell"], float(g["s2"]), g["M0"], g["jitter_rand"])
    Kbp = st["Kbp"]
    N = Kbp.shape[0]
    Lnew = gp.chol_append(np.linalg.cholesky(Kbp[:N - 1, :N - 1]), Kbp[N - 1, :N - 1], Kbp[N - 1, N - 1])
    close(Lnew, st["L"])


def _model_at(g, x, draws_jit2):
    """(Mk, Bk, A, fhat, ghat) of the dynamics model the golden controller used at state x."""
    fhat = unicycle.ackermann_f(x)
    ghat = unicycle.ackermann_g(x, float(g["mean_L"]))
    if not bool(g["enable_learning"]):
        # AckermannDrive.fu_func_gp: knl = (uh' I uh) diag(kernel_diag_A)  (unicycle_move_to_pose.py:262-275)
        return np.zeros((3, 3)), np.eye(3), np.diag(g["kernel_diag_A"]), fhat, ghat
    X, U, Xdot = g["X"], g["U"], g["Xdot"]
    B, ell, s2, M0 = g["B"], g["ell"], float(g["s2"]), g["M0"]
    UH = gp.homogeneous_controls(U)
    L = g["L"]
    Y = gp.residual_targets(Xdot, UH, M0)
    alpha = gp.cholesky_solve(Y, L)
    Mk, Bk = gp.posterior_step(L[None], alpha[None], X[None], (UH @ B)[None], ell[None], np.array([s2]),
                               B[None], M0[None], x[None], jitter2=1e-5 * draws_jit2[None])
    return Mk[0], Bk[0], g["A"], fhat, ghat


@pytest.mark.parametrize("tag", ["fixed", "learned_N40"])
def test_unicycle_constraint_terms_match_reference(tag):
    g = np.load(os.path.join(GOLDEN, "unicycle_terms_%s.npz" % tag))
    learning = bool(g["enable_learning"])
    planner = unicycle.PiecewiseLinearPlanner(g["x0"], g["xg"], int(g["numSteps"]), float(g["dt"]),
                                              float(g["frac_time_to_reach_goal"]))
    clf = unicycle.CLFCartesian(g["Kp"])
    cbfs = unicycle.obstacles_at_mid_from_start_and_goal(g["x0"], g["xg"], tuple(g["term_weights"]))
    close(np.stack([c.center for c in cbfs]), g["obst_centers"])
    close(np.array([c.radius for c in cbfs]), g["obst_radii"])
    close(cbc.cbc1_safety_factor(float(g["max_risk"])), float(g["rho"]))
    for i, (x, t) in enumerate(zip(g["states"], g["ts"])):
        p = "s%d_" % i
        t = int(t)
        plan, dplan = planner.plan(t), planner.dot_plan(t)
        close(plan, g[p + "plan"])
        close(dplan, g[p + "dot_plan"])
        close(clf.clf(x, plan), g[p + "V"])
        close(clf.grad_clf(x, plan), g[p + "grad_V"])
        close(clf.grad_clf_wrt_goal(x, plan), g[p + "grad_V_goal"])
        close([c.cbf(x) for c in cbfs], g[p + "h"])
        close(np.stack([c.grad_cbf(x) for c in cbfs]), g[p + "grad_h"])
        draws = [g[p + "draw%d" % k] for k in range(int(g[p + "ndraws"]))]
        if learning and i == 0:
            assert draws[1].shape == (g["X"].shape[0],)      # the K_b jitter of the cached Cholesky
            draws = [draws[0]] + draws[2:]
        # per constraint: [u0, jitter2 of the differentiated evaluation, jitter2 of the returned var]
        per = 3 if learning else 1
        zero = np.zeros(3)
        # CLC: -(grad_V'(f+gu) + grad_goal_V' xdot_plan + gamma V)   (:880-899)
        j2 = draws[1] if learning else zero
        Mk, Bk, A, fhat, ghat = _model_at(g, x, j2)
        const = clf.grad_clf_wrt_goal(x, plan) @ dplan + float(g["clf_gamma"]) * clf.clf(x, plan)
        bfe, e, V, bfv, v = cbc.reldeg1_terms(Mk, Bk, A, clf.grad_clf(x, plan), const, fhat, ghat, sign=-1.0)
        for name, val in zip(("bfe", "e", "V", "bfv", "v"), (bfe, e, V, bfv, v)):
            close(val, g[p + "clc_" + name], rtol=1e-8, atol=1e-10)
        Ac, bc, cc, dc = cbc.convert_cbc_terms_to_socp_terms(bfe, e, V, bfv, v, 0)
        for name, val in zip(("A", "b", "c", "d"), (Ac, bc, cc, dc)):
            close(val, g[p + "clc_socp_" + name], rtol=1e-8, atol=1e-10)
        for k, (cbf_k, gam) in enumerate(zip(cbfs, g["cbf_gammas"])):
            j2 = draws[per * (k + 1) + 1] if learning else zero
            Mk, Bk, A, fhat, ghat = _model_at(g, x, j2)
            bfe, e, V, bfv, v = cbc.reldeg1_terms(Mk, Bk, A, cbf_k.grad_cbf(x), gam * cbf_k.cbf(x), fhat, ghat)
            for name, val in zip(("bfe", "e", "V", "bfv", "v"), (bfe, e, V, bfv, v)):
                close(val, g[p + "cbc_" + name][k], rtol=1e-8, atol=1e-10)
            Ac, bc, cc, dc = cbc.convert_cbc_terms_to_socp_terms(bfe, e, V, bfv, v, 0)
            for name, val in zip(("A", "b", "c", "d"), (Ac, bc, cc, dc)):
                close(val, g[p + "cbc_socp_" + name][k], rtol=1e-8, atol=1e-10)


CBC2_FILES = sorted(glob.glob(os.path.join(GOLDEN, "cbc2_*.npz")))


@pytest.mark.parametrize("path", CBC2_FILES, ids=os.path.basename)
def test_reldeg2_terms_match_reference(path):
    """cbc2_gp + cbc2_quadratic_terms (rel-degree 2, GradientGP by autograd in the reference) against
    the closed form from posterior jets, 4 states per file."""
    from oracle import cbc2 as oc2
    g = np.load(path)
    X, U, Xdot = g["X"], g["U"], g["Xdot"]
    A, B, ell, s2, M0 = g["A"], g["B"], g["ell"], float(g["s2"]), g["M0"]
    st = gp.refit_state(X, U, Xdot, B, ell, s2, M0, g["jitter_rand"])
    close(st["L"], g["L"])
    for i in range(len(g["xs"])):
        jets = oc2.posterior_jets(st["L"], st["Y"], X, st["UHB"], ell, s2, B, M0, g["xs"][i])
        (mA, mb), (Q, p, r), mean, var = oc2.cbc2_terms(jets, A, B, ell, s2, float(g["t_h"][i]), g["t_gh"][i],
                                                       g["t_hess"][i], g["k_alpha"], g["u0s"][i])
        for name, val in (("mean_A", mA), ("mean_b", mb), ("Q", Q), ("p", p), ("r", r), ("mean", mean), ("var", var)):
            ref = g["t_" + name][i]
            np.testing.assert_allclose(np.asarray(val).reshape(np.shape(ref)), ref, rtol=1e-9, atol=1e-11)


def test_hessian_cleanup_restated_literally_matches_reference_where_the_branch_fires():
    """gp_algebra.py:384-392 on the executed reference's own outputs: 96 hand-made Hessians (n = 1..4; 80 with an eigenvalue in
    (-1.5e-3, 0): dense, with zero rows / columns, slightly non-symmetric, two negative eigenvalues; 16 controls).  The
    oracle's literal restatement gives the reference's `eigenvectors.T @ diag(evalz) @ eigenvectors` to rounding, fires
    exactly where the reference fired -- and the spectral projection (the non-default switch) does NOT give those numbers."""
    from oracle import cbc2 as oc2
    g = np.load(os.path.join(GOLDEN, "hessclean_handmade.npz"))
    n_fired, n_proj_differs = 0, 0
    for n in (1, 2, 3, 4):
        for M, ref, fired in zip(g["M_n%d" % n], g["t_knl_n%d" % n], g["branch_fired_n%d" % n]):
            H, f = oc2.clean_hessian(M)
            assert f == bool(fired)
            np.testing.assert_allclose(H, ref, rtol=0, atol=1e-12 * max(1.0, np.abs(M).max()))
            n_fired += int(f)
            if f and n > 1:
                Hp, _ = oc2.clean_hessian(M, mode="project")
                n_proj_differs += int(np.abs(Hp - ref).max() > 1e-6)
    assert n_fired == 80 and n_proj_differs >= 50, (n_fired, n_proj_differs)


EIGFIRED_FILES = sorted(glob.glob(os.path.join(GOLDEN, "eigfired_*.npz")))


@pytest.mark.parametrize("path", EIGFIRED_FILES, ids=os.path.basename)
def test_reldeg2_terms_with_the_cleanup_branch_firing_match_reference(path):
    """cbc2_gp + cbc2_quadratic_terms of the executed reference in a state where GradientGP.knl's clean-up branch FIRES in
    every record (`branch_fired`): the Cholesky factor is the cached one of an earlier output scale `s2_L`
    (control_affine_model.py:379-385: the cache key ignores its arguments), the query runs at `s2_q[i]`.  The oracle's
    closed form with the literal clean-up holds 1e-9; raw and cleaned Hessians are pinned too."""
    from oracle import cbc2 as oc2
    g = np.load(path)
    X, U, Xdot = g["X"], g["U"], g["Xdot"]
    A, B, ell, M0 = g["A"], g["B"], g["ell"], g["M0"]
    st = gp.refit_state(X, U, Xdot, B, ell, float(g["s2_L"]), M0, g["jitter_rand"])
    close(st["L"], g["L"])
    assert g["branch_fired"].all()
    for i in range(len(g["xs"])):
        s2 = float(g["s2_q"][i])
        jets = oc2.posterior_jets(st["L"], st["Y"], X, st["UHB"], ell, s2, B, M0, g["xs"][i])
        info = {}
        (mA, mb), (Q, p, r), mean, var = oc2.cbc2_terms(jets, A, B, ell, s2, float(g["t_h"][i]), g["t_gh"][i],
                                                       g["t_hess"][i], g["k_alpha"], g["u0s"][i], info=info)
        assert info["branch_fired"]
        np.testing.assert_allclose(info["H"], g["t_Hclean"][i], rtol=0, atol=1e-9 * np.abs(g["t_Hraw"][i]).max())
        for name, val in (("mean_A", mA), ("mean_b", mb), ("Q", Q), ("p", p), ("r", r), ("mean", mean), ("var", var)):
            ref = g["t_" + name][i]
            np.testing.assert_allclose(np.asarray(val).reshape(np.shape(ref)), ref, rtol=1e-9, atol=1e-11)
        # ... and the projection gives different terms here (so the fixture does tell the two formulas apart)
        (_, _), (Qp, pp, rp), _, varp = oc2.cbc2_terms(jets, A, B, ell, s2, float(g["t_h"][i]), g["t_gh"][i],
                                                      g["t_hess"][i], g["k_alpha"], g["u0s"][i], hessian_mode="project")
        assert abs(float(varp) - float(np.ravel(g["t_var"][i])[0])) > 1e-6 * abs(float(np.ravel(g["t_var"][i])[0]))


CONTROLLER_FILES = sorted(glob.glob(os.path.join(GOLDEN, "controllers_*.npz")))


def _unpack_terms(t, m):
    o = 0
    bfe = t[o:o + m]; o += m
    e = t[o]; o += 1
    V = t[o:o + m * m].reshape(m, m); o += m * m
    bfv = t[o:o + m]; o += m
    return bfe, e, V, bfv, t[o]


def _cone_close(got, ref, ev, sign_free):
    A, b, c, d = got
    if sign_free:          # eigen fallback: rows are sqrt(lambda_a) v_a', defined up to the sign of each eigenvector
        Mg = np.column_stack([b, A[:, ev:]])
        Mr = np.column_stack([ref[1], ref[0][:, ev:]])
        close(Mg.T @ Mg, Mr.T @ Mr, rtol=1e-8, atol=1e-10)
        close(np.abs(Mg), np.abs(Mr), rtol=1e-7, atol=1e-9)
        assert np.all(A[:, :ev] == 0)
    else:
        close(A, ref[0])
        close(b, ref[1])
    close(c, ref[2])
    close(d, ref[3])


@pytest.mark.parametrize("path", CONTROLLER_FILES, ids=os.path.basename)
def test_generic_controller_cones_match_reference(path):
    """SOCPController._named_socp_constraints / QPController._qp_stability (controllers.py:396-567, 614-629)."""
    from oracle import controllers as oc
    g = np.load(path)
    m = g["urefs"].shape[1]
    fallbacks = 0
    for i in range(len(g["xs"])):
        st, sf = _unpack_terms(g["t_stab_terms"][i], m), _unpack_terms(g["t_safety_terms"][i], m)
        cons = oc.named_socp_constraints(g["urefs"][i], float(g["ctrl_reg"]), float(g["relax_weight"]), [sf],
                                         [float(g["safety_factor"])], st)
        assert [c[0] for c in cons] == ["Objective", "Safety_0 gt 0", "Stability gt 0"]
        Asq = oc._asq(sf[2], sf[3], sf[4])
        indefinite = np.linalg.eigvalsh(Asq).min() <= 0
        fallbacks += int(indefinite)
        for (name, cone), key in zip(cons, ("obj", "safety", "stab")):
            ref = tuple(g["t_%s_%s" % (key, k)][i] for k in "Abcd")
            _cone_close(cone, ref, 2, sign_free=(key == "safety" and indefinite))
        _, _, bfc, d = oc.convert_cbc_terms_to_socp_terms(*st, 1)
        close(bfc, g["t_qp_c"][i])
        close(d, g["t_qp_d"][i])
    if "N40" in path:
        assert fallbacks >= 1          # the recorded set exercises the symeig branch (:528-530)


def test_generic_controller_programs_solve_to_kkt_points():
    """The checker used for y*: the oracle's coneqp on the recorded programs satisfies the cone constraints and
    beats nearby feasible points (the reference's cvxpy/GUROBI solve is not available)."""
    from oracle import controllers as oc
    g = np.load(CONTROLLER_FILES[0])
    m = g["urefs"].shape[1]
    for i in range(len(g["xs"])):
        st, sf = _unpack_terms(g["t_stab_terms"][i], m), _unpack_terms(g["t_safety_terms"][i], m)
        u, y, sol = oc.socp_controller_control(g["urefs"][i], float(g["ctrl_reg"]), float(g["relax_weight"]), [sf],
                                               [float(g["safety_factor"])], st)
        if sol["status"] != "optimal":
            continue
        for name, (A, b, c, d) in oc.named_socp_constraints(g["urefs"][i], float(g["ctrl_reg"]), float(g["relax_weight"]),
                                                            [sf], [float(g["safety_factor"])], st):
            assert c @ y + d - np.linalg.norm(A @ y + b) > -1e-7, name
        uq, yq, solq = oc.qp_controller_control(g["urefs"][i], float(g["ctrl_reg"]), float(g["relax_weight"]), st)
        assert solq["status"] == "optimal"
        assert g["t_qp_c"][i] @ yq + g["t_qp_d"][i] > -1e-8


COGP_FILES = sorted(glob.glob(os.path.join(GOLDEN, "cogp_*.npz")))


@pytest.mark.parametrize("path", COGP_FILES, ids=os.path.basename)
def test_cogp_comparator_matches_reference(path):
    """ControlAffineRegressorVector / ...VectorDiag (control_affine_model.py:1128-1330): factor, posterior mean matrix,
    full covariance, the (x,u) prediction and the flattened fullmat output."""
    g = np.load(path)
    X, U, Xdot, Sigma, M0 = g["X"], g["U"], g["Xdot"], g["Sigma"], g["M0"]
    ell, s2, lin = g["ell"], float(g["s2"]), float(g["lin"])
    if int(g["diag"]):
        assert np.allclose(Sigma, np.diag(np.diag(Sigma)))
    st = gp.cogp_refit_state(X, U, Xdot, Sigma, ell, s2, lin, M0, g["jitter_rand"])
    assert st["tries"] == 1
    close(st["L"], g["L"])
    n, C = X.shape[1], U.shape[1] + 1
    b = g["Xtest"].shape[0]
    mean_k, KkXX = gp.cogp_custom_predict_matrix(X, st["UH"], st["Y"], st["L"], Sigma, ell, s2, lin, M0, g["Xtest"],
                                                 g["jitter2"][0])
    close(mean_k, g["mean_k"])
    close(KkXX, g["KkXX"], atol=1e-10)
    # custom_predict (:1132-1169): (uh' (x) I) blocks on both sides
    mean_k2, KkXX2 = gp.cogp_custom_predict_matrix(X, st["UH"], st["Y"], st["L"], Sigma, ell, s2, lin, M0, g["Xtest"],
                                                   g["jitter2"][1])
    UHt = gp.homogeneous_controls(g["Utest"])
    close(np.einsum("bnc,bc->bn", mean_k2, UHt), g["meanFXU"])
    blk = np.stack([np.kron(u[None], np.eye(n)) for u in UHt])
    close(np.einsum("bnk,bpkl,pml->bpnm", blk, KkXX2, blk), g["varFXU"], atol=1e-10)
    mean_k3, KkXX3 = gp.cogp_custom_predict_matrix(X, st["UH"], st["Y"], st["L"], Sigma, ell, s2, lin, M0, g["Xtest"],
                                                   g["jitter2"][2])
    close(mean_k3.transpose(0, 2, 1).reshape(-1), g["full_mean"])
    close(KkXX3.transpose(0, 2, 1, 3).reshape(b * C * n, b * C * n), g["full_var"], atol=1e-10)


# ------------------------------------------------------------------------------------------------------------------
# A pin against output of the REAL gpytorch: the committed learning run of the reference (tests/saved_learning_run.py)
def test_oracle_kernel_parameterisation_against_the_committed_learning_run():
    """(a) Before the first refit the logged covariances are the prior: Fx_var = s2 kron(B, A), Fxu_var = (uh'B uh) s2 A to
    float32 rounding, with lengthscale = scalefactor = softplus(0) -- pins the Kronecker order (B outer, A inner), the
    accessor mapping and gpytorch's initial parameterisation.  (b) After each refit the oracle's posterior on the training
    set rebuilt from the log, at the LOGGED hyper-parameters (fitted by gpytorch), reproduces the logged covariances up
    to the unknown 1e-5 rand jitters: Fx_var (queried on top of the training data: collapsed to 1e-5 ... 2e-4) to 2.5e-5
    absolute, Fxu_var (queried at the raw state, 1e-3 ... 0.2) to 2e-3 RELATIVE (6e-5 at the end of the run)."""
    import saved_learning_run as R
    from oracle import gp_posterior as ogp
    sp0 = float(ogp.softplus(0.0))
    for t in R.PRIOR_STEPS:
        hp = R.hyper(t)
        np.testing.assert_allclose(hp["ell"], sp0, rtol=1e-7)
        assert abs(hp["s2"] - sp0) < 1e-7
        xs, x, uh = R.queries(t)
        k = ogp.rbf_ard_kernel(xs[None], xs[None], hp["ell"], hp["s2"])[0, 0]
        assert abs(k - hp["s2"]) < 1e-15
        np.testing.assert_allclose(k * np.kron(hp["B"], hp["A"]), R.G["Fx_var"][t], rtol=0, atol=2e-6)
        np.testing.assert_allclose((uh @ hp["B"] @ uh) * k * hp["A"], R.G["Fxu_var"][t], rtol=2e-7, atol=2e-6)
    worst = 0.0
    for t in R.POSTERIOR_STEPS:
        hp = R.hyper(t)
        X, U = R.training_set(t)
        xs, x, uh = R.queries(t)
        N = X.shape[0]
        UH = ogp.homogeneous_controls(U)
        Kb = ogp.kb_matrix(X, UH, hp["B"], hp["ell"], hp["s2"])
        _, L, tries = ogp.make_psd(Kb, np.full((10, N), 0.5))             # the mean of the unknown rand draws
        Y = np.zeros((N, 3))                                                  # (covariances do not depend on the targets)
        M0 = np.zeros((3, 3))
        _, Fx = ogp.custom_predict_fullmat(X, UH, Y, L, hp["A"], hp["B"], hp["ell"], hp["s2"], M0, xs[None],
                                           rand_draws2=np.full((10, 3), 0.5))
        _, Fxu = ogp.exact_custom_predict(X, UH, Y, L, hp["A"], hp["B"], hp["ell"], hp["s2"], M0, x[None], uh[None],
                                          rand_draws2=np.full((10, 3), 0.5))
        prior = hp["s2"] * np.abs(np.kron(hp["B"], hp["A"])).max()
        e1 = np.abs(Fx - R.G["Fx_var"][t]).max()
        e2 = np.abs(Fxu[0, 0] - R.G["Fxu_var"][t]).max()
        fxu = np.abs(R.G["Fxu_var"][t]).max()
        worst = max(worst, e1, e2)
        # at the shift-invariant state (on top of the training data) the variance collapses to the jitter level: absolute
        assert e1 <= 2.5e-5, (t, e1, np.abs(R.G["Fx_var"][t]).max())
        assert np.abs(R.G["Fx_var"][t]).max() < (0.6 if t <= 80 else 1e-2) * prior      # the drop from the (fitted) prior
        # at the raw state (away from the training inputs [0, 0, theta]) the logged variance is 1e-3 ... 0.2: a RELATIVE pin
        assert e2 <= 3e-5 and e2 <= 2e-3 * fxu, (t, e2, fxu)
    print("learning run: worst abs deviation %.2e" % worst)
