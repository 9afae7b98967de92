#!/usr/bin/env python3
"""Generate the golden vectors of tests/golden/*.npz by EXECUTING THE REFERENCE.

Runs only in the build container (needs /root/reference; see `_refenv.py` and `_shims/`).
The reference's own modules (`bayes_cbf.control_affine_model`, `gp_algebra`, `cbc2`,
`unicycle_move_to_pose`) are imported unmodified; hyper-parameters are *set*, not fitted
(fit() needs the real gpytorch); every `torch.rand` the reference draws is recorded so the
oracle / device path can replay it as an explicit input.

    python tests/golden/gen_golden.py            # rewrites tests/golden/*.npz

Only the .npz outputs travel to the GPU box.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refenv  # noqa: E402

torch = _refenv.setup()
import bayes_cbf.unicycle_move_to_pose as ump  # noqa: E402  (sets default dtype float64, :50)
import bayes_cbf.control_affine_model as cam  # noqa: E402
from bayes_cbf.cbc2 import cbc2_quadratic_terms  # noqa: E402
from bayes_cbf.planner import PiecewiseLinearPlanner  # noqa: E402


class RandRecorder:
    """Records every torch.rand(...) result drawn while active (reference make_psd :907-910,
    _clc_terms/_cbc_terms linearisation point unicycle_move_to_pose.py:894,913)."""

    def __init__(self):
        self.draws = []

    def __enter__(self):
        self._orig = torch.rand

        def rec(*a, **k):
            out = self._orig(*a, **k)
            self.draws.append(out.detach().clone().double().numpy())
            return out
        torch.rand = rec
        return self

    def __exit__(self, *exc):
        torch.rand = self._orig
        return False


def t2n(x):
    return x.detach().cpu().double().numpy()


def make_regressor(cls, n, m, N, seed, spread=1.5):
    """A regressor with set (random, seeded) hyper-parameters and a training set."""
    torch.manual_seed(seed)
    reg = cls(n, m, device='cpu')
    model = reg.model
    with torch.no_grad():
        model.input_covar.base_kernel.raw_lengthscale.copy_(0.3 * torch.randn(1, n))
        model.input_covar.raw_outputscale.copy_(0.3 * torch.randn(()))
        for bm in model.mean_module.base_means:
            bm.constant.copy_(0.2 * torch.randn(1))
    X = spread * (2 * torch.rand(N, n) - 1)
    U = torch.randn(N, m)
    # a smooth synthetic control-affine truth + small noise as targets
    Wf = torch.randn(n, n)
    Wg = torch.randn(n, n, m)
    Xdot = (torch.sin(X @ Wf.t())
            + torch.einsum('bn,knm,bm->bk', torch.cos(X), Wg, U) * 0.5
            + 1e-3 * torch.randn(N, n))
    model.set_train_data(X, U, Xdot)
    return reg, X, U, Xdot


def hyper(reg, n, m):
    A = t2n(reg.get_kernel_param('A'))
    B = t2n(reg.get_kernel_param('B'))
    ell = t2n(reg.get_kernel_param('lengthscale')).reshape(-1)
    s2 = float(t2n(reg.get_kernel_param('scalefactor')))
    consts = np.array([float(bm.constant.detach()) for bm in reg.model.mean_module.base_means])
    M0 = consts.reshape(1 + m, n)                 # matrix_variate_multitask_model.py:54-57
    return dict(A=A, B=B, ell=ell, s2=s2, M0=M0)


def gen_posterior(tag, n, m, N, b, seed, rank_one):
    out = {}
    # ---- vector-variate view: ControlAffineRegressor.custom_predict (:390-613)
    cls = cam.ControlAffineRegressorRankOne if rank_one else cam.ControlAffineRegressor
    reg, X, U, Xdot = make_regressor(cls, n, m, N, seed)
    out.update(X=t2n(X), U=t2n(U), Xdot=t2n(Xdot), **hyper(reg, n, m))
    torch.manual_seed(seed + 1)
    Xtest = 1.2 * (2 * torch.rand(b, n) - 1)
    Utest = torch.randn(b, m)
    Xtestp = 1.2 * (2 * torch.rand(b, n) - 1)
    Utestp = torch.randn(b, m)
    out.update(Xtest=t2n(Xtest), Utest=t2n(Utest), Xtestp=t2n(Xtestp), Utestp=t2n(Utestp))
    torch.manual_seed(seed + 2)
    with RandRecorder() as rr:
        mean, cov = reg.custom_predict(Xtest, Utest)
        L = reg._cache["perturbed_cholesky"]
        mean_x, cov_x = reg.custom_predict(Xtest, Utest, Xtestp_in=Xtestp, Utestp_in=Utestp)
        mean_f, cov_f = reg.custom_predict(Xtest)                       # f only: uh = e0
        mean_gu, cov_gu = reg.custom_predict(Xtest, Utest, UHfill=0)    # g(x)u only
        # single-state GP views used by gp_algebra / cbc2 (:707-818)
        fu_mean1 = reg.fu_func_mean(Utest[0], Xtest[0])
        fu_knl1 = reg.fu_func_knl(Utest[0], Xtest[0], Xtestp[0])
        covar_fu_f1 = reg.covar_fu_f(Utest[0], Xtest[0], Xtestp[0])
        f_knl1 = reg.f_func_knl(Xtest[0], Xtestp[0])
    assert len(rr.draws) == 1 and rr.draws[0].shape == (N,)
    out.update(jitter_rand=np.stack(rr.draws), L=t2n(L),
               vec_mean=t2n(mean), vec_cov=t2n(cov), vec_mean_x=t2n(mean_x), vec_cov_x=t2n(cov_x),
               vec_mean_f=t2n(mean_f), vec_cov_f=t2n(cov_f), vec_mean_gu=t2n(mean_gu),
               vec_cov_gu=t2n(cov_gu), fu_mean1=t2n(fu_mean1), fu_knl1=t2n(fu_knl1),
               covar_fu_f1=t2n(covar_fu_f1), f_knl1=t2n(f_knl1))

    # ---- matrix-variate view: ControlAffineRegressorExact (:930-1096), same data & hyper-parameters
    clsE = cam.ControlAffineRegressorExactRankOne if rank_one else cam.ControlAffineRegressorExact
    regE, XE, UE, XdotE = make_regressor(clsE, n, m, N, seed)
    assert torch.equal(XE, X) and np.allclose(hyper(regE, n, m)['B'], out['B'])
    torch.manual_seed(seed + 2)
    with RandRecorder() as rr:
        mean_k, A_, BkXX = regE._custom_predict_matrix(Xtest)
        LE = regE._cache["perturbed_cholesky"]
        meanFXU, varFXU = regE.custom_predict(Xtest, Utest)
        fullmean, fullvar = regE.custom_predict_fullmat(Xtest)
        mean_k1, _, BkXX1 = regE._custom_predict_matrix(Xtest[:1])
    assert [d.shape for d in rr.draws] == [(N,), (b * (1 + m),), (b * (1 + m),), (b * (1 + m),), (1 + m,)]
    assert np.allclose(rr.draws[0], out['jitter_rand'][0]) and np.allclose(t2n(LE), out['L'])
    out.update(mat_jitter2=rr.draws[1], mat_mean_k=t2n(mean_k), mat_BkXX=t2n(BkXX),
               exact_jitter2=rr.draws[2], exact_meanFXU=t2n(meanFXU), exact_varFXU=t2n(varFXU),
               full_jitter2=rr.draws[3], full_mean=t2n(fullmean), full_var=t2n(fullvar),
               one_jitter2=rr.draws[4], one_mean_k=t2n(mean_k1), one_BkXX=t2n(BkXX1))
    np.savez_compressed(os.path.join(HERE, 'posterior_%s.npz' % tag), **out)
    print('posterior_%s: N=%d n=%d m=%d b=%d' % (tag, N, n, m, b))


def gen_unicycle_terms(tag, N, seed, enable_learning, max_risk=0.01):
    """ControllerCLFBayesian._clc_terms / _cbcs on the unicycle (unicycle_move_to_pose.py:880-920)."""
    n, m = 3, 2
    torch.manual_seed(seed)
    x0 = torch.tensor([-3.0, -1.0, -np.pi / 4])
    xg = torch.tensor([0.0, 0.0, np.pi / 4])
    dt, numSteps = 0.01, 200
    kernel_diag_A = [1e-2, 2e-2, 3e-2]
    mean_L, Kp = 4.0, [0.9, 1.5, 0.0]
    cbf_gammas = [5.0, 5.0]
    term_weights = [0.7, 0.3]
    out = dict(x0=t2n(x0), xg=t2n(xg), dt=dt, numSteps=numSteps, mean_L=mean_L, Kp=np.array(Kp),
               cbf_gammas=np.array(cbf_gammas), term_weights=np.array(term_weights), clf_gamma=10.0,
               kernel_diag_A=np.array(kernel_diag_A), enable_learning=enable_learning,
               max_risk=max_risk, frac_time_to_reach_goal=0.95)
    if enable_learning:
        reg, X, U, Xdot = make_regressor(cam.ControlAffineRegressorExactRankOne, n, m, N, seed, spread=3.0)
        out.update(X=t2n(X), U=t2n(U), Xdot=t2n(Xdot), **hyper(reg, n, m))
    else:
        reg = None
    dyn = ump.LearnedShiftInvariantDynamics(
        dt=dt, learned_dynamics=reg,
        mean_dynamics=ump.AckermannDrive(L=mean_L, kernel_diag_A=kernel_diag_A),
        enable_learning=enable_learning)
    ctrl = ump.ControllerCLFBayesian(
        PiecewiseLinearPlanner(x0, xg, numSteps, dt, frac_time_to_reach_goal=0.95),
        coordinate_converter=lambda x, x_g: x,
        dynamics=dyn,
        clf=ump.CLFCartesian(Kp=torch.tensor(Kp)),
        cbfs=ump.obstacles_at_mid_from_start_and_goal(x0, xg, term_weights=term_weights),
        cbf_gammas=cbf_gammas, max_risk=max_risk)
    states, ts, recs = [], [], []
    torch.manual_seed(seed + 7)
    for i in range(6):
        x = torch.tensor([-3.0, -1.0, -np.pi / 4]) + torch.tensor([0.5 * i, 0.17 * i, 0.1 * i]) \
            + 0.05 * torch.randn(3)
        t = 5 + 30 * i
        state_goal = ctrl.planner.plan(t)
        rec = dict(plan=t2n(state_goal), dot_plan=t2n(ctrl.planner.dot_plan(t)))
        with RandRecorder() as rr:
            # raw terms exactly as _clc_terms / _cbc_terms obtain them (:892-894, :911-913)
            u0 = torch.rand(m)
            (bfe, e), (V, bfv, v), mean, var = cbc2_quadratic_terms(
                lambda u: ctrl._clc(x, state_goal, u, t) * -1.0, x, u0)
            rec['clc_raw'] = [t2n(z) for z in (bfe, e, V, bfv, v, mean, var)]
            rec['clc_socp'] = [t2n(z) for z in ump.ControllerCLFBayesian.convert_cbc_terms_to_socp_terms(
                bfe, e, V, bfv, v, 0)]
            rec['cbc_raw'], rec['cbc_socp'] = [], []
            for cbf, gam in zip(ctrl.cbfs, ctrl.cbf_gammas):
                u0 = torch.rand(m)
                (bfe, e), (V, bfv, v), mean, var = cbc2_quadratic_terms(
                    lambda u: ctrl._cbc(cbf, gam, x, u, t), x, u0)
                rec['cbc_raw'].append([t2n(z) for z in (bfe, e, V, bfv, v, mean, var)])
                rec['cbc_socp'].append([t2n(z) for z in
                                        ump.ControllerCLFBayesian.convert_cbc_terms_to_socp_terms(
                                            bfe, e, V, bfv, v, 0)])
            rec['h'] = [float(cbf.cbf(x)) for cbf in ctrl.cbfs]
            rec['grad_h'] = [t2n(cbf.grad_cbf(x)) for cbf in ctrl.cbfs]
            rec['V'] = float(ctrl.clf.clf_terms(x, state_goal).sum())
            rec['grad_V'] = t2n(ctrl.clf.grad_clf(x, state_goal))
            rec['grad_V_goal'] = t2n(ctrl.clf.grad_clf_wrt_goal(x, state_goal))
        rec['draws'] = rr.draws
        states.append(t2n(x))
        ts.append(t)
        recs.append(rec)
    out.update(states=np.stack(states), ts=np.array(ts))
    out['rho'] = ctrl._factor()
    out['obst_centers'] = np.stack([t2n(c.center) for c in ctrl.cbfs])
    out['obst_radii'] = np.array([float(c.radius) for c in ctrl.cbfs])
    if enable_learning:
        out['L'] = t2n(reg._cache["perturbed_cholesky"])
    for i, rec in enumerate(recs):
        p = 's%d_' % i
        out[p + 'plan'] = rec['plan']
        out[p + 'dot_plan'] = rec['dot_plan']
        for j, name in enumerate(('bfe', 'e', 'V', 'bfv', 'v', 'mean', 'var')):
            out[p + 'clc_' + name] = rec['clc_raw'][j]
            out[p + 'cbc_' + name] = np.stack([r[j] for r in rec['cbc_raw']])
        for j, name in enumerate(('A', 'b', 'c', 'd')):
            out[p + 'clc_socp_' + name] = rec['clc_socp'][j]
            out[p + 'cbc_socp_' + name] = np.stack([r[j] for r in rec['cbc_socp']])
        out[p + 'h'] = np.array(rec['h'])
        out[p + 'grad_h'] = np.stack(rec['grad_h'])
        out[p + 'V'] = rec['V']
        out[p + 'grad_V'] = rec['grad_V']
        out[p + 'grad_V_goal'] = rec['grad_V_goal']
        out[p + 'ndraws'] = len(rec['draws'])
        for k, d in enumerate(rec['draws']):
            out[p + 'draw%d' % k] = d
    np.savez_compressed(os.path.join(HERE, 'unicycle_terms_%s.npz' % tag), **out)
    print('unicycle_terms_%s: learning=%s draws/state=%s' % (
        tag, enable_learning, [len(r['draws']) for r in recs]))


def main():
    gen_posterior('n2m1_N8', n=2, m=1, N=8, b=3, seed=11, rank_one=False)
    gen_posterior('n2m1_N64', n=2, m=1, N=64, b=5, seed=12, rank_one=True)
    gen_posterior('n3m2_N8', n=3, m=2, N=8, b=2, seed=13, rank_one=True)
    gen_posterior('n3m2_N64', n=3, m=2, N=64, b=4, seed=14, rank_one=False)
    gen_posterior('n1m2_N16', n=1, m=2, N=16, b=3, seed=15, rank_one=False)
    gen_unicycle_terms('fixed', N=0, seed=21, enable_learning=False)
    gen_unicycle_terms('learned_N40', N=40, seed=22, enable_learning=True)




# ----------------------------------------------------------------------------------------------
# rel-degree-2: cbc2_gp + cbc2_quadratic_terms  (bayes_cbf/cbc2.py:7-33, gp_algebra.py:319-402)
def _h_funcs(kind, n):
    """Test barrier functions with analytic gradient / Hessian (the generator records the Hessian the
    oracle needs; the reference differentiates grad_h with autograd)."""
    if kind == "radial":          # RadialCBFRelDegree2 of bayes_cbf/pendulum.py:675-696
        import math
        dc, tc = math.pi / 8, math.pi / 4
        h = lambda x: math.cos(dc) - torch.cos(x[0] - tc)

        def gh(x):
            return torch.cat([torch.sin(x[0:1] - tc), x.new_zeros(n - 1)])

        def hess(x):
            H = x.new_zeros(n, n)
            H[0, 0] = torch.cos(x[0] - tc)
            return H
        return h, gh, hess
    Qm = torch.tensor([[1.0, 0.3, -0.2], [0.3, 0.7, 0.1], [-0.2, 0.1, 1.3]], dtype=torch.float64)[:n, :n]
    wv = torch.tensor([0.5, -0.8, 0.3], dtype=torch.float64)[:n]
    h = lambda x: 0.5 * x @ Qm @ x + torch.sin(wv @ x) - 0.2
    gh = lambda x: Qm @ x + torch.cos(wv @ x) * wv
    hess = lambda x: Qm - torch.sin(wv @ x) * torch.outer(wv, wv)
    return h, gh, hess


def gen_cbc2(tag, n, m, N, seed, kind):
    from bayes_cbf.cbc2 import cbc2_gp
    reg, X, U, Xdot = make_regressor(cam.ControlAffineRegressor, n, m, N, seed)
    out = dict(X=t2n(X), U=t2n(U), Xdot=t2n(Xdot), kind=kind, **hyper(reg, n, m))
    h, gh, hess = _h_funcs(kind, n)
    k_alpha = [1.0, 3.0]
    out["k_alpha"] = np.array(k_alpha)
    torch.manual_seed(seed + 3)
    S = 4
    xs = 0.8 * (2 * torch.rand(S, n) - 1)
    u0s = torch.rand(S, m)
    recs = {k: [] for k in ("mean_A", "mean_b", "Q", "p", "r", "mean", "var", "h", "gh", "hess")}
    with RandRecorder() as rr:
        for i in range(S):
            x, u0 = xs[i].clone(), u0s[i].clone()
            (mA, mb), (Q, p_, r_), mean, var = cbc2_quadratic_terms(
                lambda u: cbc2_gp(h, gh, reg, u, k_alpha), x, u0)
            for k, v in zip(("mean_A", "mean_b", "Q", "p", "r", "mean", "var"), (mA, mb, Q, p_, r_, mean, var)):
                recs[k].append(t2n(v))
            recs["h"].append(t2n(h(x)))
            recs["gh"].append(t2n(gh(x)))
            recs["hess"].append(t2n(hess(x)))
    assert len(rr.draws) == 1            # only the K_b jitter (vector-variate regressor)
    out.update(jitter_rand=np.stack(rr.draws), L=t2n(reg._cache["perturbed_cholesky"]), xs=t2n(xs), u0s=t2n(u0s))
    for k, v in recs.items():
        out["t_" + k] = np.stack(v)
    np.savez_compressed(os.path.join(HERE, "cbc2_%s.npz" % tag), **out)
    print("cbc2_%s: n=%d m=%d N=%d kind=%s" % (tag, n, m, N, kind))


def main_cbc2():
    gen_cbc2("pendulum_N16", n=2, m=1, N=16, seed=41, kind="radial")
    gen_cbc2("pendulum_N40", n=2, m=1, N=40, seed=42, kind="radial")
    gen_cbc2("n3m2_N24", n=3, m=2, N=24, seed=43, kind="generic")


def gen_controllers(tag, N, seed):
    """SOCPController / QPController of bayes_cbf/controllers.py on the pendulum (n=2, m=1): the named cone
    constraints the reference hands to its optimiser (:396-540, 542-567) and QPController's rows (:614-662)."""
    import math
    from bayes_cbf.controllers import SOCPController, QPController, to_numpy
    from bayes_cbf.pendulum import RadialCBFRelDegree2
    from bayes_cbf.cbc1 import RelDeg1Safety
    n, m = 2, 1
    reg, X, U, Xdot = make_regressor(cam.ControlAffineRegressor, n, m, N, seed)
    out = dict(X=t2n(X), U=t2n(U), Xdot=t2n(Xdot), **hyper(reg, n, m))
    cbf2 = RadialCBFRelDegree2(reg, dtype=torch.float64)

    class EnergyCLC(RelDeg1Safety):            # V = w^2/2 + (1 - cos theta), as a rel-degree-1 condition on the learned model
        gamma, model, max_unsafe_prob = 2.0, reg, 0.01

        def cbf(self, x):
            return 0.5 * x[1] ** 2 + (1 - torch.cos(x[0]))

        def grad_cbf(self, x):
            return torch.stack([torch.sin(x[0]), x[1]])

        def clc(self, t, u):
            return self.cbc(u) * -1.0

    clf = EnergyCLC()

    class Unsafe:
        def __init__(self):
            self.u = None

        def control(self, x, t=None):
            return self.u

    unsafe = Unsafe()
    ctrl_reg, relax_w = 1.0, 100.0
    socp = SOCPController(n, m, ctrl_reg, relax_w, reg, [cbf2], clf, unsafe, None)
    qp = QPController(n, m, ctrl_reg, relax_w, reg, [cbf2], clf, unsafe, None)
    torch.manual_seed(seed + 5)
    S = 4
    xs = 0.8 * (2 * torch.rand(S, n) - 1)
    urefs = 2 * torch.rand(S, m) - 1
    recs = {}
    with RandRecorder() as rr:
        for i in range(S):
            x, u_ref = xs[i].clone(), urefs[i].clone()
            cons = socp._named_socp_constraints(7, x, u_ref, convert_out=to_numpy, extravars=2)
            assert [c[0] for c in cons] == ["Objective", "Safety_0 gt 0", "Stability gt 0"]
            for (name, (A, b, c, d)), key in zip(cons, ("obj", "safety", "stab")):
                for k, v in zip("Abcd", (A, b, c, d)):
                    recs.setdefault("%s_%s" % (key, k), []).append(np.asarray(v, dtype=np.float64))
            (mA, mb), (Q, p_, r_), mean, var = cbc2_quadratic_terms(cbf2.cbc, x, u_ref)
            recs.setdefault("safety_terms", []).append(np.concatenate([t2n(mA).ravel(), t2n(mb).ravel(), t2n(Q).ravel(),
                                                                       t2n(p_).ravel(), t2n(r_).ravel()]))
            (mA, mb), (Q, p_, r_), mean, var = cbc2_quadratic_terms(lambda u: clf.clc(7, u), x, u_ref)
            recs.setdefault("stab_terms", []).append(np.concatenate([t2n(mA).ravel(), t2n(mb).ravel(), t2n(Q).ravel(),
                                                                     t2n(p_).ravel(), t2n(r_).ravel()]))
            bfc, d = qp._qp_stability(clf.clc, 7, x, u_ref, extravars=1)
            recs.setdefault("qp_c", []).append(t2n(bfc))
            recs.setdefault("qp_d", []).append(t2n(d))
    assert len(rr.draws) == 1            # only the K_b jitter
    out.update(jitter_rand=np.stack(rr.draws), xs=t2n(xs), urefs=t2n(urefs), ctrl_reg=ctrl_reg, relax_weight=relax_w,
               safety_factor=cbf2.safety_factor(), k_alpha=np.array(cbf2.k_alpha), clf_gamma=clf.gamma, t=7)
    for k, v in recs.items():
        out["t_" + k] = np.stack(v)
    np.savez_compressed(os.path.join(HERE, "controllers_%s.npz" % tag), **out)
    print("controllers_%s: N=%d" % (tag, N))


def main_controllers():
    gen_controllers("pendulum_N16", N=16, seed=51)
    gen_controllers("pendulum_N40", N=40, seed=52)



def gen_cogp(tag, N, b, seed, diag, n=2, m=1):
    """ControlAffineRegressorVector / ControlAffineRegVectorDiag on the pendulum shapes (n=2, m=1) and the unicycle's
    (n=3, m=2: nine task outputs): _custom_predict_matrix, custom_predict, custom_predict_fullmat
    (control_affine_model.py:1128-1330)."""
    cls = cam.ControlAffineRegVectorDiag if diag else cam.ControlAffineRegressorVector
    torch.manual_seed(seed)
    reg = cls(n, m, device='cpu')
    model = reg.model
    with torch.no_grad():
        rbf, linear = model.input_covar.base_kernel.kernels
        rbf.raw_lengthscale.copy_(0.3 * torch.randn(1, 1))
        linear.raw_variance.copy_(0.3 * torch.randn(1, 1) - 2.0)
        model.input_covar.raw_outputscale.copy_(0.3 * torch.randn(()))
        for bm in model.mean_module.base_means:
            bm.constant.copy_(0.2 * torch.randn(1))
    X = 1.5 * (2 * torch.rand(N, n) - 1)
    U = torch.randn(N, m)
    if m == 1:
        Xdot = torch.sin(X @ torch.randn(n, n).t()) + 0.5 * torch.cos(X) * U + 1e-3 * torch.randn(N, n)
    else:
        Xdot = torch.sin(X @ torch.randn(n, n).t()) + 0.5 * torch.cos(X) * (U @ torch.randn(m, n)) + 1e-3 * torch.randn(N, n)
    model.set_train_data(X, U, Xdot)
    Sigma = t2n(model.covar_module.task_covar_module.covar_matrix.evaluate())
    out = dict(X=t2n(X), U=t2n(U), Xdot=t2n(Xdot), Sigma=Sigma, ell=t2n(rbf.lengthscale).reshape(-1),
               lin=float(t2n(linear.variance)), s2=float(t2n(model.input_covar.outputscale)),
               M0=np.array([float(bm.constant.detach()) for bm in model.mean_module.base_means]).reshape(1 + m, n),
               diag=int(diag))
    torch.manual_seed(seed + 1)
    Xtest = 1.2 * (2 * torch.rand(b, n) - 1)
    Utest = torch.randn(b, m)
    with RandRecorder() as rr:
        mean_k, KkXX = reg._custom_predict_matrix(Xtest)
        L = reg._cache["perturbed_cholesky"]
        meanFXU, varFXU = reg.custom_predict(Xtest, Utest)
        fullmean, fullvar = reg.custom_predict_fullmat(Xtest)
    assert [d.shape for d in rr.draws] == [(N * n,)] + [(b * (1 + m) * n,)] * 3
    out.update(jitter_rand=rr.draws[0][None], L=t2n(L), Xtest=t2n(Xtest), Utest=t2n(Utest),
               jitter2=np.stack(rr.draws[1:]), mean_k=t2n(mean_k), KkXX=t2n(KkXX), meanFXU=t2n(meanFXU),
               varFXU=t2n(varFXU), full_mean=t2n(fullmean), full_var=t2n(fullvar))
    np.savez_compressed(os.path.join(HERE, 'cogp_%s.npz' % tag), **out)
    print('cogp_%s: N=%d b=%d diag=%s' % (tag, N, b, diag))


def main_cogp():
    gen_cogp('full_N12', N=12, b=3, seed=61, diag=False)
    gen_cogp('full_N48', N=48, b=5, seed=62, diag=False)
    gen_cogp('diag_N24', N=24, b=4, seed=63, diag=True)
    gen_cogp('full_n3m2_N10', N=10, b=3, seed=64, diag=False, n=3, m=2)
    gen_cogp('diag_n3m2_N20', N=20, b=4, seed=65, diag=True, n=3, m=2)


# ----------------------------------------------------------------------------------------------
# gp_algebra: the propagation rules themselves, node by node (bayes_cbf/gp_algebra.py:109-255, 319-402), on the trees
# the reference's own tests build (tests/test_gp_algebra.py:163-239: L1h, grad L1h, L2h, cbc2_gp) and on a tree that is
# not a safety condition (a weighted sum of two inner products with different leaves), at pairs of DIFFERENT states
def gen_gp_algebra(tag, n, m, N, seed, kind):
    from bayes_cbf.cbc2 import cbc2_gp
    from bayes_cbf.gp_algebra import DeterministicGP, GradientGP
    reg, X, U, Xdot = make_regressor(cam.ControlAffineRegressor, n, m, N, seed)
    out = dict(X=t2n(X), U=t2n(U), Xdot=t2n(Xdot), kind=kind, **hyper(reg, n, m))
    h, gh, hess = _h_funcs(kind, n)
    k_alpha = [1.0, 3.0]
    out["k_alpha"] = np.array(k_alpha)
    torch.manual_seed(seed + 7)
    S = 3
    xs = 0.8 * (2 * torch.rand(S, n) - 1)
    xps = xs + 0.3 * (2 * torch.rand(S, n) - 1)
    us = torch.randn(S, m)
    cvec = torch.randn(n)
    dvec = torch.randn(n)
    out.update(xs=t2n(xs), xps=t2n(xps), us=t2n(us), cvec=t2n(cvec), dvec=t2n(dvec))
    recs = {}

    def rec(key, val):
        recs.setdefault(key, []).append(t2n(val).astype(np.float64))

    with RandRecorder() as rr:
        for i in range(S):
            x, xp, u = xs[i].clone(), xps[i].clone(), us[i].clone()
            f_gp = reg.f_func_gp()
            fu_gp = reg.fu_func_gp(u)
            L1h = DeterministicGP(gh, shape=(n,), name="grad h").t() @ f_gp
            gL1h = GradientGP(L1h, x_shape=(n,))
            L2h = gL1h.t() @ fu_gp
            cbc2 = cbc2_gp(h, gh, reg, u, k_alpha)
            mix = (DeterministicGP(lambda z: cvec * torch.cos(z), shape=(n,), name="c").t() @ fu_gp) * 0.7 \
                + DeterministicGP(lambda z: dvec + z, shape=(n,), name="d").t() @ f_gp
            for name, e in (("L1h", L1h), ("L2h", L2h), ("cbc2", cbc2), ("mix", mix)):
                rec(name + "_mean", e.mean(x))
                rec(name + "_knl_xx", e.knl(x, x))
                rec(name + "_knl_xxp", e.knl(x, xp))
                if name != "cbc2":       # cbc2_gp makes its own fu leaf: no covariance is registered with ours (:292-300)
                    rec(name + "_covar_fu_xxp", e.covar(fu_gp, x, xp))
                rec(name + "_covar_f_xxp", e.covar(f_gp, x, xp))
            rec("gL1h_mean", gL1h.mean(x))
            rec("gL1h_knl_xx", gL1h.knl(x, x))
            rec("gL1h_knl_xxp", gL1h.knl(x, xp))
            rec("gL1h_covar_fu_xx_same", gL1h.covar(fu_gp, x, x))        # same tensor: total derivative (:395-402)
            rec("gL1h_covar_fu_xxp", gL1h.covar(fu_gp, x, xp))
            rec("gL1h_covar_f_xxp", gL1h.covar(f_gp, x, xp))
            rec("h", h(x)); rec("gh", gh(x)); rec("hess", hess(x))
    assert len(rr.draws) == 1            # only the K_b jitter (vector-variate regressor)
    out.update(jitter_rand=np.stack(rr.draws), L=t2n(reg._cache["perturbed_cholesky"]))
    for k, v in recs.items():
        out["t_" + k] = np.stack(v)
    np.savez_compressed(os.path.join(HERE, "gpalgebra_%s.npz" % tag), **out)
    print("gpalgebra_%s: n=%d m=%d N=%d kind=%s" % (tag, n, m, N, kind))


def handmade_trees(ga, P):
    """Two hand-made leaf GPs (torch callables, a registered cross-covariance), a deterministic vector function and
    trees over them that use every node type -- incl. the inner product of two RANDOM vectors.  `ga` is the algebra
    module (the reference's here, the build's in tests/test_gp_algebra_cpu.py); P holds the parameters."""
    n = P["W1"].shape[0]
    W1, W2, A1, A2, C12 = (torch.as_tensor(P[k], dtype=torch.float64) for k in ("W1", "W2", "A1", "A2", "C12"))
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
    return f, g, dict(L1=L1, gL1=gL1, L2=L2, rr=rr, mix=mix)


def gen_gp_algebra_handmade():
    import bayes_cbf.gp_algebra as rga
    torch.manual_seed(0)
    n = 3
    A1, A2 = torch.randn(n, n), torch.randn(n, n)
    P = dict(W1=t2n(torch.randn(n, n)), W2=t2n(torch.randn(n, n)), A1=t2n(A1 @ A1.t() + torch.eye(n)),
             A2=t2n(A2 @ A2.t() + torch.eye(n)), C12=t2n(0.3 * torch.randn(n, n)))
    f, g, trees = handmade_trees(rga, P)
    xs, xps = 0.5 * torch.randn(3, n), 0.5 * torch.randn(3, n)
    out = dict(P, xs=t2n(xs), xps=t2n(xps))
    for name, e in trees.items():
        for key, fn in (("mean", lambda x, xp: e.mean(x)), ("knl_xxp", lambda x, xp: e.knl(x, xp)),
                        ("knl_xx", lambda x, xp: e.knl(x, x)), ("covar_g_xxp", lambda x, xp: e.covar(g, x, xp)),
                        ("covar_f_xx_same", lambda x, xp: e.covar(f, x, x))):
            out["t_%s_%s" % (name, key)] = np.stack([t2n(fn(xs[i].clone(), xps[i].clone())) for i in range(3)])
    np.savez_compressed(os.path.join(HERE, "gpalgebra_handmade.npz"), **out)
    print("gpalgebra_handmade: n=%d, %d values" % (n, len(out)))


def main_gp_algebra():
    gen_gp_algebra("pendulum_N16", n=2, m=1, N=16, seed=71, kind="radial")
    gen_gp_algebra("n3m2_N24", n=3, m=2, N=24, seed=72, kind="generic")
    gen_gp_algebra_handmade()



# ----------------------------------------------------------------------------------------------
# The eigenvalue clean-up of GradientGP.knl at x' == x (bayes_cbf/gp_algebra.py:384-392) with the branch FIRING:
#     eigenvalues in (-2e-3, 0)  ->  Hxx_k = eigenvectors.T @ diag(evalz) @ eigenvectors      (torch.eig = xGEEV)
# No other golden family reaches it (the Hessians there are positive definite).  Two families:
#   hessclean_handmade.npz   leaf GPs with the bilinear kernel k(x, x') = x' M x (its cross Hessian is M itself), M built
#                            with one eigenvalue in (-1.5e-3, -1e-5) -- dense, with zero rows / columns (a barrier gradient
#                            with a zero component: xGEBAL's permutation), slightly non-symmetric -- and controls that do
#                            not fire; n = 1..4
#   eigfired_*.npz           the regressor path end to end (cbc2_gp + cbc2_quadratic_terms) in a state the reference can
#                            really be in: the Cholesky factor cached under a key that ignores its arguments
#                            (control_affine_model.py:379-385) and the output scale changed afterwards WITHOUT clear_cache():
#                            the posterior "kernel" k_new B - W'W is then slightly indefinite.  The scale bump is bisected
#                            per query state so that the smallest eigenvalue of the Hessian lands near -1e-3.
# Every record carries `branch_fired` (was torch.eig asked for eigenvectors) and `agree_openblas`: whether numpy's LAPACK
# build gives the same V' L V as torch's (it pairs eigenvalue k with ROW k of V, so it depends on xGEEV's order and signs).
class EigRecorder:
    """Counts the torch.eig calls of the reference that ask for eigenvectors (= the clean-up branch ran)."""

    def __enter__(self):
        self._orig = torch.eig
        self.fired = 0

        def rec(A, eigenvectors=False):
            if eigenvectors:
                self.fired += 1
            return self._orig(A, eigenvectors=eigenvectors)
        torch.eig = rec
        return self

    def __exit__(self, *exc):
        torch.eig = self._orig
        return False


def _literal_with_numpy(H, eps=2e-3):
    w, V = np.linalg.eig(H)
    ev = w.real.copy()
    small = (ev > -eps) & (ev < 0)
    if not small.any():
        return H
    ev[small] = 0.0
    return V.real.T @ np.diag(ev) @ V.real


def gen_hessclean_handmade():
    import bayes_cbf.gp_algebra as rga
    rng = np.random.RandomState(20240)
    Ms, outs, fired, agree, kinds = {}, {}, {}, {}, {}
    for n in (1, 2, 3, 4):
        Ms[n], outs[n], fired[n], agree[n], kinds[n] = [], [], [], [], []
        for t in range(24):
            A = rng.randn(n, n)
            w, V = np.linalg.eigh(A + A.T)
            w = np.abs(w) + 0.05
            kind = ("neg", "neg", "pos", "neg_zero_rows", "neg_asym", "neg_two")[t % 6]
            if kind != "pos":
                w[0] = -rng.uniform(1e-5, 1.5e-3)
            if kind == "neg_two" and n > 2:
                w[1] = -rng.uniform(1e-5, 1.5e-3)
            M = (V * w) @ V.T
            M = 0.5 * (M + M.T)
            if kind == "neg_zero_rows" and n > 1:
                for z in rng.permutation(n)[:rng.randint(1, n)]:
                    M[z, :] = 0.0
                    M[:, z] = 0.0
                if n - int((np.abs(M).sum(0) == 0).sum()) >= 1:
                    nzi = np.where(np.abs(M).sum(0) != 0)[0]
                    if len(nzi):                       # keep a small negative eigenvalue inside the non-zero block
                        wb, Vb = np.linalg.eigh(M[np.ix_(nzi, nzi)])
                        wb = np.abs(wb) + 0.05
                        wb[0] = -rng.uniform(1e-5, 1.5e-3)
                        M[np.ix_(nzi, nzi)] = (Vb * wb) @ Vb.T
            if kind == "neg_asym":
                M = M + 1e-16 * rng.randn(n, n)
            Mt = torch.tensor(M)
            leaf = rga.GaussianProcess(lambda x: x.sum(), lambda x, xp, Mt=Mt: x @ Mt @ xp, (1,), name="bilinear")
            x = torch.tensor(rng.randn(n))
            with EigRecorder() as er:
                Hc = rga.GradientGP(leaf, x_shape=(n,)).knl(x, x)
            Hc = t2n(Hc)
            Ms[n].append(M)
            outs[n].append(Hc)
            fired[n].append(er.fired > 0)
            agree[n].append(bool(np.allclose(_literal_with_numpy(M), Hc, rtol=0, atol=1e-10)))
            kinds[n].append(kind)
    out = {}
    for n in Ms:
        out.update({"M_n%d" % n: np.stack(Ms[n]), "t_knl_n%d" % n: np.stack(outs[n]), "branch_fired_n%d" % n: np.array(fired[n]),
                    "agree_openblas_n%d" % n: np.array(agree[n]), "kind_n%d" % n: np.array(kinds[n])})
    np.savez_compressed(os.path.join(HERE, "hessclean_handmade.npz"), **out)
    print("hessclean_handmade:", {n: (int(np.sum(fired[n])), int(np.sum(agree[n])), len(fired[n])) for n in Ms},
          "(fired, numpy-LAPACK agrees, cases)")


def gen_eigfired(tag, n, m, N, seed, kind, S=4, target=(-1.4e-3, -0.6e-3)):
    from bayes_cbf.cbc2 import cbc2_gp
    from bayes_cbf.gp_algebra import DeterministicGP
    from bayes_cbf.misc import t_hessian
    reg, X, U, Xdot = make_regressor(cam.ControlAffineRegressor, n, m, N, seed, spread=0.9)
    out = dict(X=t2n(X), U=t2n(U), Xdot=t2n(Xdot), kind=kind, **hyper(reg, n, m))
    out["s2_L"] = out.pop("s2")                      # the output scale the cached factor was computed with
    h, gh, hess = _h_funcs(kind, n)
    k_alpha = [1.0, 3.0]
    out["k_alpha"] = np.array(k_alpha)
    torch.manual_seed(seed + 3)
    xs = 0.5 * (2 * torch.rand(S, n) - 1)
    u0s = torch.rand(S, m)
    raw = reg.model.input_covar.raw_outputscale
    raw0 = raw.detach().clone()
    recs = {k: [] for k in ("mean_A", "mean_b", "Q", "p", "r", "mean", "var", "h", "gh", "hess", "Hraw", "Hclean")}
    s2_q, fired, agree = [], [], []
    with RandRecorder() as rr:
        reg.custom_predict(xs[:1].clone())           # the factor enters the cache here, at raw0
        L1h = DeterministicGP(gh, shape=(n,), name="grad h").t() @ reg.f_func_gp()

        def min_eig(x, bump):
            with torch.no_grad():
                raw.copy_(raw0 + bump)
            Hraw = t_hessian(L1h.knl, x.clone(), x.detach().clone())
            return float(torch.linalg.eigvalsh(0.5 * (Hraw + Hraw.t()))[0]), Hraw

        for i in range(S):
            x, u0 = xs[i].clone(), u0s[i].clone()
            lo, hi = 0.0, 0.05
            while min_eig(x, hi)[0] > target[1]:     # grow the bump until the Hessian is indefinite enough
                hi *= 2.0
                assert hi < 50.0, "no bump makes this Hessian indefinite"
            for _ in range(200):
                mid = 0.5 * (lo + hi)
                e, _ = min_eig(x, mid)
                if target[0] < e < target[1]:
                    break
                lo, hi = (mid, hi) if e >= target[1] else (lo, mid)
            else:
                raise AssertionError("bisection did not land in the target interval")
            e, Hraw = min_eig(x, mid)
            with EigRecorder() as er:
                (mA, mb), (Q, p_, r_), mean, var = cbc2_quadratic_terms(
                    lambda u: cbc2_gp(h, gh, reg, u, k_alpha), x, u0)
                from bayes_cbf.gp_algebra import GradientGP
                Hclean = GradientGP(L1h, x_shape=(n,)).knl(x, x)
            for k, v in zip(("mean_A", "mean_b", "Q", "p", "r", "mean", "var", "Hraw", "Hclean"),
                            (mA, mb, Q, p_, r_, mean, var, Hraw, Hclean)):
                recs[k].append(t2n(v))
            recs["h"].append(t2n(h(x)))
            recs["gh"].append(t2n(gh(x)))
            recs["hess"].append(t2n(hess(x)))
            s2_q.append(float(t2n(reg.get_kernel_param("scalefactor"))))
            fired.append(er.fired > 0)
            agree.append(bool(np.allclose(_literal_with_numpy(t2n(Hraw)), t2n(Hclean), rtol=0, atol=1e-9 * float(Hraw.abs().max()))))
    assert len(rr.draws) == 1            # only the K_b jitter of the one cached factor
    assert all(fired), fired
    out.update(jitter_rand=np.stack(rr.draws), L=t2n(reg._cache["perturbed_cholesky"]), xs=t2n(xs), u0s=t2n(u0s),
               s2_q=np.array(s2_q), branch_fired=np.array(fired), agree_openblas=np.array(agree))
    for k, v in recs.items():
        out["t_" + k] = np.stack(v)
    np.savez_compressed(os.path.join(HERE, "eigfired_%s.npz" % tag), **out)
    eigs = [np.linalg.eigvalsh(0.5 * (H + H.T)) for H in out["t_Hraw"]]
    print("eigfired_%s: n=%d m=%d N=%d kind=%s  s2_L=%.4f  s2_q=%s  fired=%s  numpy-LAPACK agrees=%s\n   eig(Hraw)=%s"
          % (tag, n, m, N, kind, out["s2_L"], np.round(s2_q, 4), fired, agree, [np.round(e, 5).tolist() for e in eigs]))


def main_eigfired():
    gen_hessclean_handmade()
    gen_eigfired("pendulum_N16", n=2, m=1, N=16, seed=81, kind="radial")
    gen_eigfired("n3m2_N24", n=3, m=2, N=24, seed=83, kind="generic")


# ----------------------------------------------------------------------------------------------
# bayes_cbf/misc.py helper surface (t_jac, t_hessian, get_affine_terms, get_quadratic_terms, torch_kron, epsilon,
# store_args, DynamicsModel.forward / step / F_func, ZeroDynamicsModel) and cbc2_quadratic_terms on a callable that is
# plain torch (hand-made GaussianProcess leaves): recorded from the executed reference.
def gen_misc():
    import bayes_cbf.misc as rm
    import bayes_cbf.gp_algebra as rga
    torch.manual_seed(5)
    out = {}
    n = 3
    W = torch.randn(4, n)
    x = torch.randn(n)
    fvec = lambda z: torch.tanh(W @ z) * (z @ z)
    with rm.variable_required_grad(x) as xg:
        out["jac_vec"] = t2n(rm.t_jac(fvec(xg), xg))
        out["jac_scalar"] = t2n(rm.t_jac(fvec(xg).sum(), xg))
    out.update(W=t2n(W), x=t2n(x))
    P = torch.randn(n, n)
    xp = torch.randn(n)
    f2 = lambda a, b: torch.sin(a @ P @ b) + (a * a) @ (b * b)
    out.update(P=t2n(P), xp=t2n(xp), hess=t2n(rm.t_hessian(f2, x.clone(), xp.clone())))
    Qm, pv, r0 = torch.randn(n, n), torch.randn(n), 0.7          # Qm NOT symmetric: get_quadratic_terms does not symmetrise
    quad = lambda z: z @ Qm @ z + pv @ z + r0
    aff = lambda z: pv @ z + r0
    q, l, c = rm.get_quadratic_terms(quad, x.clone())
    a, b = rm.get_affine_terms(aff, x.clone())
    out.update(Qm=t2n(Qm), pv=t2n(pv), r0=r0, quad_Q=t2n(q), quad_p=t2n(l), quad_r=t2n(c), aff_a=t2n(a), aff_b=t2n(b))
    A5, B5 = torch.randn(5, 2, 2), torch.randn(5, 3, 3)
    A0, B0 = torch.randn(2, 3), torch.randn(3, 2)
    out.update(kron_A=t2n(A5), kron_B=t2n(B5), kron_AB=t2n(rm.torch_kron(A5, B5)), kron_A0=t2n(A0), kron_B0=t2n(B0),
               kron_AB0=t2n(rm.torch_kron(A0, B0, batch_dims=0)))
    out["epsilon"] = np.array([rm.epsilon(i) for i in (0, 10, 500, 1000)] + [rm.epsilon(3, interpolate={0: 2.0, 10: 0.5})])
    out["normalize_radians"] = np.array([rm.normalize_radians(v) for v in (-7.0, -3.2, 0.0, 3.2, 9.5)])

    class Plant(rm.DynamicsModel):
        ctrl_size, state_size = 2, 3
        f_func = lambda self, X: torch.sin(X) * 0.5
        g_func = lambda self, X: torch.stack([torch.cos(X), X * 0.3], dim=-1)
    pl = Plant()
    Xb, Ub = torch.randn(4, 3), torch.randn(4, 2, 1)
    out.update(dyn_X=t2n(Xb), dyn_U=t2n(Ub), dyn_fwd_batch=t2n(pl.forward(Xb, Ub)), dyn_fwd_single=t2n(pl.forward(Xb[0], Ub[0, :, 0])),
               dyn_F=t2n(pl.F_func(Xb)))
    pl.set_init_state(Xb[1])
    s1 = pl.step(Ub[1, :, 0], 0.05)
    s2 = pl.step(Ub[2, :, 0], 0.05)
    out.update(dyn_step_x=np.stack([t2n(s1["x"]), t2n(s2["x"])]), dyn_step_xdot=np.stack([t2n(s1["xdot"]), t2n(s2["xdot"])]))
    z = rm.ZeroDynamicsModel(2, 3)
    out.update(zero_f=t2n(z.f_func(Xb)), zero_g=t2n(z.g_func(Xb)), zero_f1=t2n(z.f_func(Xb[0])), zero_g1=t2n(z.g_func(Xb[0])))

    class Store:
        @rm.store_args
        def __init__(self, a, b=2, c="see"):          # (a keyword-only default trips a bug in the reference, misc.py:64)
            self.ran = True
    st = Store(1, c="given")
    out["store_args"] = np.array([str(getattr(st, k, "<unset>")) for k in ("a", "b", "c", "ran")])
    # cbc2_quadratic_terms on a plain-torch callable u -> GP (hand-made leaves; affine mean / quadratic kernel in u)
    Pm = dict(W1=t2n(torch.randn(n, n)), A1=None)
    A1 = torch.randn(n, n)
    A1 = A1 @ A1.t() + torch.eye(n)
    G = torch.randn(n, 2)
    rbf = lambda a, b: torch.exp(-0.5 * ((a - b) ** 2).sum())

    def cbc(u):
        uh = torch.cat([torch.ones(1), u])
        f = rga.GaussianProcess(lambda z: torch.sin(W[:n] @ z) + G @ u, lambda a, b: rbf(a, b) * A1 * (uh @ uh), (n,), name="fu")
        d = rga.DeterministicGP(lambda z: torch.tanh(z), shape=(n,), name="d")
        return d.t() @ f
    u0 = torch.rand(2)
    from bayes_cbf.cbc2 import cbc2_quadratic_terms as rq
    (mA, mb), (kQ, kp, kr), mean, var = rq(cbc, x.clone(), u0)
    out.update(cbc_A1=t2n(A1), cbc_G=t2n(G), cbc_u0=t2n(u0), cbc_mean_A=t2n(mA), cbc_mean_b=t2n(mb), cbc_Q=t2n(kQ), cbc_p=t2n(kp),
               cbc_r=t2n(kr), cbc_mean=t2n(mean), cbc_var=t2n(var))
    np.savez_compressed(os.path.join(HERE, "misc_surfaces.npz"), **{k: v for k, v in out.items() if v is not None})
    print("misc_surfaces:", sorted(out))


def gen_facade():
    """Small host-side surfaces of the path, recorded from the executed reference:
    HetergeneousMatrixVariateMean.forward (matrix_variate_multitask_model.py:44-66) on observation rows, matrix rows, a
    sorted mix and raw states without a mask column; sample_generator_trajectory (sampling.py:49-75) on the Ackermann
    plant with a deterministic controller and with a controller_class; sampling_pendulum_data (pendulum.py:164-252)."""
    import math
    import bayes_cbf.sampling as rs
    import bayes_cbf.pendulum as rp
    out = {}
    n, m = 3, 2
    torch.manual_seed(77)
    reg = cam.ControlAffineRegressor(n, m, device='cpu')
    with torch.no_grad():
        for bm in reg.model.mean_module.base_means:
            bm.constant.copy_(0.5 * torch.randn(1))
    out['mean_constants'] = np.array([float(bm.constant.detach()) for bm in reg.model.mean_module.base_means])
    X5, U5 = torch.randn(5, n), torch.randn(5, m)
    _, mxu1 = reg.model.encode_from_XU(X5, U5, 1)
    _, mxu0 = reg.model.encode_from_XU(X5[:4])
    mix = torch.cat([mxu1[:3], mxu0[:2]])
    raw = torch.randn(4, n)
    mm = reg.model.mean_module
    out.update(mean_mxu1=t2n(mxu1), mean_out1=t2n(mm(mxu1)), mean_mxu0=t2n(mxu0), mean_out0=t2n(mm(mxu0)),
               mean_mix=t2n(mix), mean_outmix=t2n(mm(mix)), mean_raw=t2n(raw), mean_outraw=t2n(mm(raw)))
    # ---- rollout harness on the Ackermann plant
    ctl = lambda x, t=0: torch.stack([1.0 + 0.1 * torch.sin(x[2] + 0.05 * t), 0.3 * torch.cos(x[0]) - 0.02 * t])
    x0 = torch.tensor([-1.0, 0.4, 0.7])
    Xdot, X, U = rs.sample_generator_trajectory(ump.AckermannDrive(L=1.3), 12, dt=0.02, x0=x0, controller=ctl)
    out.update(traj_x0=t2n(x0), traj_Xdot=t2n(Xdot), traj_X=t2n(X), traj_U=t2n(U), traj_L=1.3, traj_dt=0.02)

    class Ctl:
        def __init__(self, dt=None, true_model=None):
            self.gain = 2.0 * dt * true_model.L

        def control(self, x, t=0):
            return torch.stack([self.gain * (1 + x[0] * 0), 0.1 * x[1] + 0.01 * t])
    plant = ump.AckermannDrive(L=0.7)
    Xdot, X, U = rs.sample_generator_trajectory(plant, 6, dt=0.05, x0=[0.1, -0.2, 0.3], true_model=plant, controller_class=Ctl)
    out.update(trajc_Xdot=t2n(Xdot), trajc_X=t2n(X), trajc_U=t2n(U))
    # ---- pendulum trajectory (deterministic controller: the randomised one draws torch.rand per step)
    env = rp.PendulumDynamicsModel(m=1, n=2, mass=1, gravity=10, length=1)
    pctl = lambda x, t=0: (12.0 + 2.0 * torch.sin(x[0]) + 0.5 * math.cos(0.1 * t)).reshape(1)
    dX, X, U = rp.sampling_pendulum_data(env, D=60, dt=0.05, x0=torch.tensor([5 * math.pi / 6, -0.01]), controller=pctl,
                                         visualizer=rs.VisualizerZ())
    assert float((X[1:, 0] - X[:-1, 0]).abs().max()) > 3.0             # the trajectory wraps around +-pi at least once
    out.update(pend_dX=t2n(dX), pend_X=t2n(X), pend_U=t2n(U))
    np.savez_compressed(os.path.join(HERE, 'facade_surfaces.npz'), **out)
    print('facade_surfaces:', {k: v.shape for k, v in out.items() if hasattr(v, 'shape')})


def gen_checkpoint():
    """The reference's checkpoint layout (ControlAffineRegressor.save / state_dict, control_affine_model.py:862-874 over
    ControlAffineExactGP.state_dict, :201-218), written by the executed reference:
      reference_checkpoint_n3m2_N24.pt    the pickle `reg.save(path)` wrote (plain dicts / OrderedDicts of tensors)
      reference_checkpoint_n3m2_N24.npz   what a FRESH reference regressor predicts after `load(path)` (its loader leaves the
                                          prior-mean constants at their initial zeros: the mean module's state_dict drops
                                          them, matrix_variate_multitask_model.py:68-76), what the SAVING regressor predicts
                                          (constants recorded beside it), and the recorded jitter draws of both.
    Vector-variate comparator: reference_checkpoint_vector_n2m1_N10.{pt,npz} the same way."""
    for tag, cls, n, m, N, seed in (("n3m2_N24", cam.ControlAffineRegressor, 3, 2, 24, 31),
                                    ("vector_n2m1_N10", cam.ControlAffineRegressorVector, 2, 1, 10, 32)):
        if cls is cam.ControlAffineRegressorVector:
            torch.manual_seed(seed)
            reg = cls(n, m, device='cpu')
            with torch.no_grad():
                rbf, linear = reg.model.input_covar.base_kernel.kernels
                rbf.raw_lengthscale.copy_(0.3 * torch.randn(1, 1))
                linear.raw_variance.copy_(0.3 * torch.randn(1, 1) - 2.0)
                reg.model.input_covar.raw_outputscale.copy_(0.3 * torch.randn(()))
                for bm in reg.model.mean_module.base_means:
                    bm.constant.copy_(0.2 * torch.randn(1))
            X, U = 1.5 * (2 * torch.rand(N, n) - 1), torch.randn(N, m)
            Xdot = torch.sin(X @ torch.randn(n, n).t()) + 0.5 * torch.cos(X) * U + 1e-3 * torch.randn(N, n)
            reg.model.set_train_data(X, U, Xdot)
        else:
            reg, X, U, Xdot = make_regressor(cls, n, m, N, seed)
        path = os.path.join(HERE, "reference_checkpoint_%s.pt" % tag)
        reg.save(path)
        consts = np.array([float(bm.constant.detach()) for bm in reg.model.mean_module.base_means])
        torch.manual_seed(seed + 1)
        Xtest = 1.2 * (2 * torch.rand(5, n) - 1)
        Utest = torch.randn(5, m)
        out = dict(Xtest=t2n(Xtest), Utest=t2n(Utest), mean_constants=consts, X=t2n(X), U=t2n(U), Xdot=t2n(Xdot))
        with RandRecorder() as rr:
            mean, cov = reg.custom_predict(Xtest, Utest)
        out.update(saver_mean=t2n(mean), saver_cov=t2n(cov), saver_draws=np.stack(rr.draws))
        fresh = cls(n, m, device='cpu')
        fresh.load(path)
        sd = fresh.state_dict()
        assert torch.equal(sd['model']['train_targets'], Xdot.reshape(-1))
        with RandRecorder() as rr:
            mean, cov = fresh.custom_predict(Xtest, Utest)
        out.update(loaded_mean=t2n(mean), loaded_cov=t2n(cov), loaded_draws=np.stack(rr.draws))
        np.savez_compressed(os.path.join(HERE, "reference_checkpoint_%s.npz" % tag), **out)
        print("reference_checkpoint_%s:" % tag, sorted(torch.load(path)['model']), "constants lost on load:",
              bool(np.abs(out['saver_mean'] - out['loaded_mean']).max() > 1e-6))


if __name__ == '__main__':
    if 'checkpoint' in sys.argv:
        gen_checkpoint()
        sys.exit(0)
    if 'facade' in sys.argv:
        gen_facade()
    elif 'eigfired' in sys.argv:
        main_eigfired()
    elif 'misc' in sys.argv:
        gen_misc()
    elif 'gp_algebra' in sys.argv:
        main_gp_algebra()
    elif 'cogp' in sys.argv:
        main_cogp()
    elif 'controllers' in sys.argv:
        main_controllers()
    elif 'cbc2' in sys.argv:
        main_cbc2()
    else:
        main()
        main_cbc2()
