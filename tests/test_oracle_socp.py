"""Pin the conic-solve oracle: the reference's own known answer (tests/test_optimizers.py:6-26,
28-119 of the reference), KKT residuals / an independent SLSQP solve, and the GUROBI outputs
logged in the reference's committed runs (tests/golden/saved_run_*.npz)."""
import os

import numpy as np
import pytest
from scipy.optimize import minimize

from oracle import cbc, socp, unicycle
from kat import cvxopt_doc_example

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_convert_socp_to_cvxopt_format_known_answer():
    lin, cons = cvxopt_doc_example()
    _, Gqs, hqs = socp.convert_socp_to_cvxopt_format(lin, cons)
    exp_Gqs = [np.array([[12., 13., 12.], [6., -3., -12.], [-5., -5., 6.]]),
               np.array([[3., 3., -1., 1.], [-6., -6., -9., 19.], [10., -2., -2., -3.]])]
    exp_hqs = [np.array([-12., -3., -2.]), np.array([27., 0., 3., -42.])]
    for Gq, hq, eG, eh in zip(Gqs, hqs, exp_Gqs, exp_hqs):
        np.testing.assert_allclose(Gq.T, eG)
        np.testing.assert_allclose(hq.flatten(), eh)


def test_socp_known_answer():
    lin, cons = cvxopt_doc_example()
    sol = socp.optimizer_socp(lin, cons)
    assert sol["status"] == "optimal"
    np.testing.assert_allclose(sol["x"], [-5.02, -5.77, -8.52], rtol=1e-2, atol=1e-3)
    np.testing.assert_allclose(sol["z"][:3], [1.34, -7.63e-02, -1.34], rtol=1e-2, atol=1e-3)
    np.testing.assert_allclose(sol["z"][3:], [1.02, 4.02e-01, 7.80e-01, -5.17e-01], rtol=1e-2, atol=1e-3)


def random_feasible_program(rng, K=3, m=2, rho=2.326):
    cones = []
    u_f = rng.normal(size=m)
    for k in range(K):
        Asq = rng.normal(size=(m + 1, m + 1))
        Asq = Asq @ Asq.T * rng.uniform(0.001, 1) + 1e-4 * np.eye(m + 1)
        Lc = np.linalg.cholesky(Asq)
        A_, b_ = Lc.T[:, 1:], Lc.T[:, 0]
        c_ = rng.normal(size=m) * 3
        slack = rng.uniform(0.01, 2.0) * (1 if k else rng.choice([-1, 1]))
        d_ = rho * np.linalg.norm(A_ @ u_f + b_) - c_ @ u_f + slack
        cones.append((A_, b_, c_, d_))
    return cones, rho


def test_clf_cbf_socp_kkt_and_slsqp():
    rng = np.random.default_rng(5)
    w = [0.33, 0.33, 0.33]
    for _ in range(40):
        cones, rho = random_feasible_program(rng)
        relax_mask = [1, 0, 0]
        sol = socp.clf_cbf_socp(w, [0.0, 0.0], cones, rho, relax_mask)
        assert sol["status"] == "optimal"
        y = sol["x"]

        def margin(y, k):
            A_, b_, c_, d_ = cones[k]
            return c_ @ y[:2] + d_ + relax_mask[k] * y[2] - rho * np.linalg.norm(A_ @ y[:2] + b_)
        assert min(margin(y, k) for k in range(3)) > -1e-8
        cs = [{"type": "ineq", "fun": (lambda yy, k=k: margin(yy, k))} for k in range(3)]
        rs = minimize(lambda yy: 0.33 * (yy ** 2).sum(), y + 0.01, constraints=cs, method="SLSQP",
                      options=dict(ftol=1e-15, maxiter=1000))
        if rs.status == 0:
            assert abs(rs.fun - 0.33 * (y ** 2).sum()) < 1e-7
            np.testing.assert_allclose(rs.x, y, atol=5e-5)


def test_infeasible_program_is_flagged():
    A_ = np.eye(3)[:, 1:]
    cones = [(A_, np.array([1.0, 0, 0]), np.array([1.0, 0.0]), -1.0),      # u0 - 1 >= |..| >= 1  -> u0 >= 2
             (A_, np.array([1.0, 0, 0]), np.array([-1.0, 0.0]), -1.0)]     # -u0 - 1 >= 1 -> u0 <= -2
    sol = socp.clf_cbf_socp([0.33, 0.33, 0.33], [0, 0], cones, 1.0, [0, 0])
    assert sol["status"] != "optimal"


@pytest.mark.parametrize("name", ["saved_run_mean_cbf_maxrisk0p5", "saved_run_bayes_cbf_maxrisk0p01"])
def test_saved_run_trajectories(name):
    """Fixed-kernel chance-constraint assembly + SOCP reproduces the logged GUROBI controls, and the
    Euler recursion with the true plant reproduces the logged next state (sampling.py:68-74)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    dt, T = float(g["dt"]), int(g["numSteps"])
    x0, xg = g["state_start"], g["state_goal"]
    planner = unicycle.PiecewiseLinearPlanner(x0, xg, T, dt, frac_time_to_reach_goal=0.95)
    vis_planner = unicycle.PiecewiseLinearPlanner(x0, xg, T, dt)
    clf = unicycle.CLFCartesian([0.9, 1.5, 0.0])
    cbfs = unicycle.obstacles_at_mid_from_start_and_goal(x0, xg, tuple(g["term_weights"]))
    A = np.diag(g["kernel_diag_A"])
    rho = cbc.cbc1_safety_factor(float(g["max_risk"]))
    np.testing.assert_allclose(rho, g["opt_rho"][0], rtol=1e-6)
    Mk, Bk = np.zeros((3, 3)), np.eye(3)
    checked = 0
    for t in range(0, T, 7):
        x = g["state"][t].astype(np.float64)
        plan, dplan = planner.plan(t), planner.dot_plan(t)
        # the logged plan_x is the *visualizer's* planner (default frac 0.7, unicycle_move_to_pose.py:1709-1713)
        np.testing.assert_allclose(vis_planner.plan(t), g["plan_x"][t], atol=2e-6)
        fhat, ghat = unicycle.ackermann_f(x), unicycle.ackermann_g(x, float(g["mean_L"]))
        const = clf.grad_clf_wrt_goal(x, plan) @ dplan + float(g["clf_gamma"]) * clf.clf(x, plan)
        terms = [cbc.reldeg1_terms(Mk, Bk, A, clf.grad_clf(x, plan), const, fhat, ghat, sign=-1.0)]
        for c_k, gam in zip(cbfs, g["cbf_gammas"]):
            terms.append(cbc.reldeg1_terms(Mk, Bk, A, c_k.grad_cbf(x), gam * c_k.cbf(x), fhat, ghat))
        cones = [cbc.convert_cbc_terms_to_socp_terms(*tm, 0) for tm in terms]
        sol = socp.clf_cbf_socp(g["cost_weights"], [0.0, 0.0], cones, rho, [1, 0, 0])
        assert sol["status"] == "optimal", (t, sol["status"])
        u = sol["x"][:2]
        np.testing.assert_allclose(u, g["uopt"][t], rtol=2e-3, atol=2e-3)
        value = float(np.sum(g["cost_weights"] * sol["x"] ** 2))
        np.testing.assert_allclose(value, g["opt_value"][t], rtol=1e-4, atol=1e-5)
        if t + 1 < T:
            xn = unicycle.ackermann_step(x, g["uopt"][t].astype(np.float64), dt, float(g["true_L"]))
            np.testing.assert_allclose(xn, g["state"][t + 1], atol=5e-6)
        checked += 1
    assert checked >= 25


@pytest.mark.parametrize("name", ["saved_run_mean_cbf_maxrisk0p5", "saved_run_bayes_cbf_maxrisk0p01"])
def test_control_step_restatement_reproduces_saved_run(name):
    """oracle.control_step (the end-to-end restatement the GPU tests of the fused entry point compare with) pinned on
    the reference's committed 200-step runs: logged GUROBI control and logged next state, every 5th step."""
    from oracle import control_step as ostep
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    dt, T = float(g["dt"]), int(g["numSteps"])
    x0, xg = g["state_start"], g["state_goal"]
    planner = unicycle.PiecewiseLinearPlanner(x0, xg, T, dt, frac_time_to_reach_goal=0.95)
    cbfs = unicycle.obstacles_at_mid_from_start_and_goal(x0, xg, tuple(g["term_weights"]))
    centers, radii = np.stack([c.center for c in cbfs]), np.array([c.radius for c in cbfs])
    rho = cbc.cbc1_safety_factor(float(g["max_risk"]))
    for t in range(0, T - 1, 5):
        x = g["state"][t].astype(np.float64)
        o = ostep.control_step(x, planner.plan(t), planner.dot_plan(t), np.zeros((3, 3)), np.eye(3),
                               np.diag(g["kernel_diag_A"]), [0.9, 1.5, 0.0], float(g["clf_gamma"]), centers, radii,
                               g["term_weights"], g["cbf_gammas"], float(g["mean_L"]), g["cost_weights"], [0.0, 0.0], rho,
                               dt=dt, L_true=float(g["true_L"]))
        assert o["status"] == "optimal"
        np.testing.assert_allclose(o["u"], g["uopt"][t], rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(o["x_next"], g["state"][t + 1], atol=2e-4)     # (u is logged in fp32)
    # an unsolved program takes no step (the reference raises, unicycle_move_to_pose.py:954-964)
    bad = ostep.control_step(x, planner.plan(0), planner.dot_plan(0), np.zeros((3, 3)), np.eye(3), 1e3 * np.eye(3),
                             [0.9, 1.5, 0.0], 10.0, centers, 10 * radii, g["term_weights"], g["cbf_gammas"], 1.0,
                             g["cost_weights"], [0.0, 0.0], 2.326, dt=dt, L_true=1.0)
    assert bad["status"] != "optimal" and np.array_equal(bad["x_next"], x)
