"""Host-side surfaces of the path that need no GPU, against vectors recorded from the executed reference
(tests/golden/facade_surfaces.npz, generator: tests/golden/gen_golden.py facade):
`HetergeneousMatrixVariateMean.forward` (matrix_variate_multitask_model.py:44-66), `sample_generator_trajectory`
(sampling.py:49-75) and the pendulum trajectory (`sampling_pendulum_data`, pendulum.py:164-252)."""
import math
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "facade_surfaces.npz"))
T64 = dict(dtype=torch.float64)


def t(a):
    return torch.as_tensor(np.asarray(a), **T64)


def test_matrix_variate_mean_forward_matches_reference_on_all_row_kinds():
    from bayesian_cbf_amd.control_affine_model import CatEncoder
    from bayesian_cbf_amd.matrix_variate_multitask_model import (ConstantMean, HetergeneousMatrixVariateMean,
                                                                 SharedConstantMeans)
    n, m = 3, 2
    consts = t(G["mean_constants"])
    shared = HetergeneousMatrixVariateMean(SharedConstantMeans(lambda: consts, (1 + m) * n), CatEncoder(1, n, 1 + m), (1 + m, n))
    own = HetergeneousMatrixVariateMean(ConstantMean(dtype=torch.float64), CatEncoder(1, n, 1 + m), (1 + m, n))
    with torch.no_grad():
        for bm, c in zip(own.base_means, consts):
            bm.constant.fill_(float(c))
    for mod in (shared, own):
        for inp, out in (("mean_mxu1", "mean_out1"), ("mean_mxu0", "mean_out0"), ("mean_mix", "mean_outmix"),
                         ("mean_raw", "mean_outraw")):
            np.testing.assert_allclose(mod(t(G[inp])).detach().numpy(), G[out], rtol=0, atol=1e-14, err_msg=inp)
    # observation rows: uh' M0; matrix rows: vec(M0)
    M0 = consts.reshape(1 + m, n)
    mxu1 = t(G["mean_mxu1"])
    np.testing.assert_allclose(shared(mxu1).reshape(-1, n).numpy(), (mxu1[:, 1 + n:] @ M0).numpy(), atol=1e-14)
    assert shared.state_dict() == dict(matshape=(1 + m, n), decoder=dict(sizes=[1, n, 1 + m]))
    # unsorted masks are refused, as upstream
    bad = torch.cat([t(G["mean_mxu0"])[:1], mxu1[:2]])
    with pytest.raises(AssertionError):
        shared(torch.cat([mxu1[:1], bad]))
    # the edge of the raw-state call: a first row with x[0] == 1.0 exactly is taken for an observation row
    raw = t(G["mean_raw"]).clone()
    raw[0, 0] = 1.0
    with pytest.raises(AssertionError):
        shared(raw)


def test_regressor_mean_module_reads_the_device_paths_constants():
    from bayesian_cbf_amd.control_affine_model import ControlAffineRegressor
    reg = ControlAffineRegressor(3, 2, device="cpu", dtype=torch.float64)
    with torch.no_grad():
        reg.model.mean_constants.copy_(t(G["mean_constants"]))
    np.testing.assert_allclose(reg.mean_module(t(G["mean_mix"])).detach().numpy(), G["mean_outmix"], atol=1e-14)
    _, mxu = reg.encode_from_XU(t(G["mean_raw"]))
    assert mxu.shape == (4, 1 + 3 + 3) and float(mxu[:, 0].abs().max()) == 0 and float(mxu[:, 4:].abs().max()) == 0
    reg.set_train_data(t(G["mean_mxu1"])[:, 1:4], t(G["mean_mxu1"])[:, 5:], t(G["mean_mxu1"])[:, 1:4])
    np.testing.assert_array_equal(reg.train_inputs[0].numpy(), G["mean_mxu1"])


def test_sample_generator_trajectory_reference_surface_single_trajectory():
    from bayesian_cbf_amd.sampling import sample_generator_trajectory, Visualizer
    from bayesian_cbf_amd.unicycle_move_to_pose import AckermannDrive
    ctl = lambda x, t=0: torch.stack([1.0 + 0.1 * torch.sin(x[2] + 0.05 * t), 0.3 * torch.cos(x[0]) - 0.02 * t])
    Xdot, X, U = sample_generator_trajectory(AckermannDrive(L=float(G["traj_L"])), 12, dt=float(G["traj_dt"]),
                                             x0=t(G["traj_x0"]), controller=ctl)
    for got, key in ((Xdot, "traj_Xdot"), (X, "traj_X"), (U, "traj_U")):
        np.testing.assert_allclose(got.numpy(), G[key], rtol=0, atol=1e-6)      # the reference ran in its float32 default

    class Ctl:
        def __init__(self, dt=None, true_model=None):
            self.gain = 2.0 * dt * true_model.L

        def control(self, x, t=0):
            return torch.stack([self.gain * (1 + x[0] * 0), 0.1 * x[1] + 0.01 * t])

    seen = []

    class Rec(Visualizer):
        def setStateCtrl(self, x, u, t=0, **kw):
            seen.append((t, x.clone(), u.clone(), sorted(kw)))

    plant = AckermannDrive(L=0.7)
    Xdot, X, U = sample_generator_trajectory(plant, 6, dt=0.05, x0=[0.1, -0.2, 0.3], true_model=plant, controller_class=Ctl,
                                             visualizer=Rec())
    for got, key in ((Xdot, "trajc_Xdot"), (X, "trajc_X"), (U, "trajc_U")):
        np.testing.assert_allclose(got.numpy(), G[key], rtol=0, atol=1e-6)
    assert [s[0] for s in seen] == list(range(6)) and all(torch.equal(s[1], X[s[0]]) and torch.equal(s[2], U[s[0]]) for s in seen)
    assert seen[0][3] == []                          # a plain controller has no Bayesian model to visualise


def test_sample_generator_trajectory_batch_rows_equal_single_trajectories_cpu():
    from bayesian_cbf_amd.sampling import sample_generator_trajectory
    from bayesian_cbf_amd.unicycle_move_to_pose import AckermannDrive, CartesianDynamics
    x0 = torch.tensor([[-1.0, 0.4, 0.7], [0.3, -0.2, -1.1], [2.0, 1.0, 0.1]], **T64)
    ctl = lambda x, t=0: torch.stack([1.0 + 0.1 * torch.sin(x[..., 2] + 0.05 * t), 0.3 * torch.cos(x[..., 0]) - 0.02 * t], -1)
    for plant_cls in (lambda: AckermannDrive(L=1.3), CartesianDynamics):
        Xd, X, U = sample_generator_trajectory(plant_cls(), 9, dt=0.02, x0=x0, controller=ctl)
        assert Xd.shape == (9, 3, 3) and X.shape == (10, 3, 3) and U.shape == (9, 3, 2)
        for i in range(3):
            Xd1, X1, U1 = sample_generator_trajectory(plant_cls(), 9, dt=0.02, x0=x0[i], controller=ctl)
            np.testing.assert_allclose(X[:, i].numpy(), X1.numpy(), atol=1e-14)
            np.testing.assert_allclose(Xd[:, i].numpy(), Xd1.numpy(), atol=1e-14)
            np.testing.assert_allclose(U[:, i].numpy(), U1.numpy(), atol=1e-14)


def test_pendulum_trajectory_through_the_rollout_harness_matches_reference():
    from bayesian_cbf_amd.pendulum import PendulumDynamicsModel, sampling_pendulum, sampling_pendulum_data
    env = PendulumDynamicsModel(m=1, n=2, mass=1, gravity=10, length=1)
    pctl = lambda x, t=0: (12.0 + 2.0 * torch.sin(x[0]) + 0.5 * math.cos(0.1 * t)).reshape(1)
    x0 = torch.tensor([5 * math.pi / 6, -0.01])
    dX, X, U = sampling_pendulum_data(env, D=60, dt=0.05, x0=x0, controller=pctl)
    # (the reference integrates in float32: the wrap points coincide, values agree to float32 rounding accumulated over 60 steps)
    assert np.array_equal(np.abs(np.diff(G["pend_X"][:, 0])) > 3, np.abs(np.diff(X[:, 0].numpy())) > 3)
    np.testing.assert_allclose(X.numpy(), G["pend_X"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(U.numpy(), G["pend_U"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(dX.numpy(), G["pend_dX"], rtol=0, atol=2e-2)
    dmg, tv, th, om, uv = sampling_pendulum(env, 61, controller=pctl, x0=x0, dt=0.05)
    assert th.shape == (61,) and torch.equal(th, X[:, 0]) and 0 <= float(dmg) <= 100 and float(tv[-1]) == pytest.approx(3.0)


def test_nominal_controllers_restored_and_control_cbf_learned_default_constructible():
    """controllers.py:166-213, 269-285, 739-771: Zero / Greedy / EpsilonGreedy / NamedAffineFunc exist, and
    ControlCBFLearned() builds with its default arguments (greedy nominal controller inside the epsilon-greedy explorer)."""
    import random
    from bayesian_cbf_amd.controllers import (ControlCBFLearned, EpsilonGreedyController, GreedyController, NamedAffineFunc,
                                              ZeroController, epsilon)
    from bayesian_cbf_amd.pendulum import PendulumDynamicsModel
    from bayesian_cbf_amd.unicycle_move_to_pose import PolarDynamics
    c = ControlCBFLearned()
    assert isinstance(c.unsafe_controller, EpsilonGreedyController) and isinstance(c.unsafe_controller.base_controller, GreedyController)
    assert epsilon(0, {0: 1, 100: 0.1}) == pytest.approx(1.0) and epsilon(100, {0: 1, 100: 0.1}) == pytest.approx(0.1)
    assert epsilon(50, {0: 1, 100: 0.1}) == pytest.approx(math.sqrt(0.1))
    env = PendulumDynamicsModel()
    P, R, xg, dt = torch.tensor([[2.0, 0.3], [0.3, 1.0]], **T64), torch.tensor([[0.7]], **T64), torch.tensor([0.2, -0.1], **T64), 0.05
    g = GreedyController(env, P, R, xg, 100, dt, torch.tensor([-5.0, 5.0]))
    x = torch.tensor([1.0, 0.2], **T64)
    u = g.control(x)
    # u minimises  (1-lam) |x + f dt + G u - x_g|_P^2 + lam u' R dt u,  lam = 1/2: zero gradient
    uu = u.clone().requires_grad_(True)
    xp = x + dt * env.f_func(x[None])[0] + dt * env.g_func(x) @ uu
    cost = 0.5 * (xp - xg) @ P @ (xp - xg) + 0.5 * uu @ (R * dt) @ uu
    cost.backward()
    assert float(uu.grad.abs().max()) < 1e-12
    ub = g.control(torch.stack([x, x + 0.1]))
    assert ub.shape == (2, 1) and torch.allclose(ub[0], u)
    assert torch.equal(ZeroController(env, P, R, xg, 100, dt, None).control(x), torch.zeros(1, **T64))
    e = EpsilonGreedyController(g, 1, 100, [1, 0.1], torch.tensor([-0.05, 0.05], **T64))
    random.seed(0)
    out = torch.stack([e.control(x, t=99) for _ in range(50)])
    assert float(out.abs().max()) <= 0.05 + 1e-15                         # clipped to the control range

    class H(NamedAffineFunc):
        name = "h"
        value = lambda self, x: x[0]
        A = lambda self, x: torch.tensor([[1.0, 2.0]])
        b = lambda self, x: torch.tensor([0.5])
    assert H().__name__ == "h" and float(H()(None, torch.tensor([1.0, 1.0]))) == 2.5
    pd = PolarDynamics()
    pd.set_init_state(torch.tensor([1.0, 0.3, 0.2], **T64))
    obs = pd.step(torch.tensor([1.0, 0.5], **T64), 0.1)
    np.testing.assert_allclose(obs["xdot"].numpy(), [-math.cos(0.3), -math.sin(0.3) + 0.5, -math.sin(0.3)], atol=1e-14)


def test_hyper_cache_follows_dtype_casts_and_repointed_parameters():
    """`_hyper()` (host-side constants kept across `clear_cache()`) must follow `float_()/double_()` (reference surface,
    control_affine_model.py:625-643: module.to() bumps no version counter) and a re-pointed parameter."""
    from bayesian_cbf_amd.control_affine_model import ControlAffineRegressor
    r = ControlAffineRegressor(2, 1, device="cpu")
    assert r._hyper()["A"].dtype == r.dtype
    first = r.dtype
    other = torch.float32 if first == torch.float64 else torch.float64
    r.to(other)
    hp = r._hyper()
    assert all(v.dtype == other for v in hp.values()), {k: v.dtype for k, v in hp.items()}
    r.to(first)
    assert all(v.dtype == first for v in r._hyper().values())
    ell0 = r._hyper()["ell"].clone()
    r.model.raw_lengthscale.data = r.model.raw_lengthscale.data + 1.0        # re-pointed storage: no version bump
    assert not torch.allclose(r._hyper()["ell"], ell0)
    ell1 = r._hyper()["ell"].clone()
    r.model.raw_lengthscale.data.add_(1.0)                                   # in place through .data: invisible ...
    r.clear_cache(hyper=True)                                                # ... unless asked
    assert not torch.allclose(r._hyper()["ell"], ell1)


def test_host_side_data_kernel_shapes_match_the_oracle_and_finite_differences():
    """bayesian_cbf_amd/data_kernels.py (the prior-kernel values and derivatives the facade evaluates on the host: `_prior_knl`,
    gp_eval, the rel-degree-2 prior curvature) against the oracle's kernels and central differences, all three data kernels."""
    from bayesian_cbf_amd import data_kernels as dk
    from oracle import gp_posterior as ogp
    rng = np.random.default_rng(11)
    ell, s2 = np.array([0.7, 1.3, 0.9]), 1.6
    for kernel in dk.KINDS:
        assert kernel in ogp.DATA_KERNELS and abs(dk.kxx(kernel) - ogp.KERNEL_KXX[kernel]) < 1e-15
        for _ in range(5):
            x, xp = rng.normal(size=3), rng.normal(size=3)
            d2 = float((((x - xp) / ell) ** 2).sum())
            sh, dsh, ddsh = (float(v) for v in dk.shape_terms(kernel, torch.tensor(d2, dtype=torch.float64)))
            np.testing.assert_allclose(s2 * sh, ogp.DATA_KERNELS[kernel](x[None], xp[None], ell, s2)[0, 0], rtol=1e-13)
            d = (x - xp) / ell ** 2
            h = 1e-5
            for a in range(3):
                e = np.zeros(3); e[a] = h
                k = lambda u, v: ogp.DATA_KERNELS[kernel](u[None], v[None], ell, s2)[0, 0]
                np.testing.assert_allclose(-s2 * dsh * d[a], (k(x + e, xp) - k(x - e, xp)) / (2 * h), rtol=1e-6, atol=1e-9)   # dk/dx_a
                for b_ in range(3):
                    f = np.zeros(3); f[b_] = h
                    fd = (k(x + e, xp + f) - k(x + e, xp - f) - k(x - e, xp + f) + k(x - e, xp - f)) / (4 * h * h)
                    want = s2 * (dsh * (a == b_) / ell[a] ** 2 + ddsh * d[a] * d[b_])                                       # d2k/dx_a dx'_b
                    np.testing.assert_allclose(want, fd, rtol=2e-4, atol=2e-6)
        assert abs(float(dk.shape_terms(kernel, torch.tensor(0.0, dtype=torch.float64))[1]) - dk.kxx(kernel)) < 1e-14


def test_every_data_kernel_has_every_entry_point_the_wrappers_ask_for():
    """ops maps a data-kernel name to an entry-point suffix (`_KSUF`); every (base, kernel, dtype) the wrappers can ask for must be
    an exported, declared symbol of libbcbf (a missing twin would only show up as an AttributeError on the GPU box)."""
    from bayesian_cbf_amd import _lib, ops
    bases = ["bcbf_kb_build", "bcbf_refit", "bcbf_posterior_query", "bcbf_posterior_shared", "bcbf_posterior_jets", "bcbf_mll_grad",
             "bcbf_gp_append", "bcbf_unicycle_control_step"]
    declared = set(_lib.declared_symbols())
    for kernel in ops.DATA_KERNELS:
        for base in bases:
            for suf in ("_f32", "_f64"):
                name = base + ops._KSUF[kernel] + suf
                assert name in declared and hasattr(_lib.lib, name), name
    assert ops.DATA_KERNELS.index("rbf") == 0 and ops.DATA_KERNELS.index("matern52") == 1 and ops.DATA_KERNELS.index("rbf_matern52") == 2


# ---- checkpoint layout (ControlAffineRegressor.state_dict / save / load, control_affine_model.py:201-218, 862-874):
# the pickles were written by the executed reference's `save` (tests/golden/gen_golden.py checkpoint)
CKPT = [("n3m2_N24", "ControlAffineRegressor", 3, 2), ("vector_n2m1_N10", "ControlAffineRegressorVector", 2, 1)]
REF_MODEL_KEYS = ["matshape", "decoder", "mean_module", "task_covar", "input_covar", "covar_module", "train_inputs",
                  "train_targets"]


def _ckpt(tag):
    here = os.path.join(os.path.dirname(__file__), "golden")
    return os.path.join(here, "reference_checkpoint_%s.pt" % tag), np.load(os.path.join(here, "reference_checkpoint_%s.npz" % tag))


@pytest.mark.parametrize("tag,cls,n,m", CKPT)
def test_reference_checkpoint_loads_and_is_rewritten_key_for_key(tag, cls, n, m, tmp_path):
    import bayesian_cbf_amd.control_affine_model as cam
    path, g = _ckpt(tag)
    ref = torch.load(path)
    reg = getattr(cam, cls)(n, m, device="cpu", dtype=torch.float64)
    with torch.no_grad():
        reg.model.mean_constants.fill_(0.25)
    reg.load(path)
    np.testing.assert_array_equal(reg.Xtrain.numpy(), g["X"])
    np.testing.assert_array_equal(reg.Utrain.numpy(), g["U"])
    np.testing.assert_array_equal(reg.XdotTrain.numpy(), g["Xdot"])
    # the reference's file carries no prior-mean constants: they stay what they were (as in the reference's loader)
    assert float((reg.model.mean_constants - 0.25).abs().max()) == 0
    sd = reg.state_dict()
    assert list(sd["model"]) == REF_MODEL_KEYS and list(ref["model"]) == REF_MODEL_KEYS
    assert sorted(sd)[:2] == ["bcbf", "likelihood"] and "model" in sd and len(sd["likelihood"]) == 0 == len(ref["likelihood"])
    for mod in ("task_covar", "input_covar", "covar_module"):
        assert list(sd["model"][mod]) == list(ref["model"][mod]), mod
        for k in ref["model"][mod]:
            assert torch.equal(sd["model"][mod][k], ref["model"][mod][k]), (mod, k)
    assert sd["model"]["matshape"] == ref["model"]["matshape"] and sd["model"]["decoder"] == ref["model"]["decoder"]
    assert {k: sd["model"]["mean_module"][k] for k in ("matshape", "decoder")} == ref["model"]["mean_module"]
    assert torch.equal(sd["model"]["train_inputs"][0], ref["model"]["train_inputs"][0])
    assert torch.equal(sd["model"]["train_targets"], ref["model"]["train_targets"])
    # save -> load round trip of the façade's own file keeps the constants too
    p2 = str(tmp_path / "saved.pickle")
    reg.save(p2)
    reg2 = getattr(cam, cls)(n, m, device="cpu", dtype=torch.float64)
    reg2.load(p2)
    for (k, a), (k2, b) in zip(reg.model.state_dict().items(), reg2.model.state_dict().items()):
        assert k == k2 and torch.equal(a, b), k
    assert torch.equal(reg2.Xtrain, reg.Xtrain) and torch.equal(reg2.Utrain, reg.Utrain) and torch.equal(reg2.XdotTrain, reg.XdotTrain)
    # the layout of rounds 1-5 still loads
    reg3 = getattr(cam, cls)(n, m, device="cpu", dtype=torch.float64)
    reg3.load_state_dict(dict(model=reg.model.state_dict(), train=(reg.Xtrain, reg.Utrain, reg.XdotTrain)))
    assert torch.equal(reg3.model.mean_constants, reg.model.mean_constants) and torch.equal(reg3.Xtrain, reg.Xtrain)
    # a model of another shape or data kernel refuses the file
    with pytest.raises(ValueError):
        getattr(cam, cls)(n + 1, m, device="cpu", dtype=torch.float64).load(path)
    if cls == "ControlAffineRegressor":
        with pytest.raises(ValueError):
            cam.ControlAffineRegressor(n, m, device="cpu", dtype=torch.float64, data_kernel="matern52").load(p2)


def test_facade_checkpoint_loads_into_the_executed_reference(tmp_path):
    """Container-only (needs /root/reference): a file the façade's `save` wrote goes through the reference's own
    `load_state_dict`.  The reference runs in a child process (its harness patches torch for the old API it expects)."""
    import subprocess
    import sys
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    if not os.path.isdir("/root/reference/bayes_cbf"):
        pytest.skip("reference tree not present")
    import bayesian_cbf_amd.control_affine_model as cam
    path, g = _ckpt("n3m2_N24")
    reg = cam.ControlAffineRegressor(3, 2, device="cpu", dtype=torch.float64)
    reg.load(path)
    p2 = str(tmp_path / "saved.pickle")
    reg.save(p2)
    child = """
import sys, torch
sys.path.insert(0, %r)
import _refenv
_refenv.setup()
torch.set_default_dtype(torch.float64)
import bayes_cbf.control_affine_model as rcam
ref = rcam.ControlAffineRegressor(3, 2, device="cpu")
ref.load_state_dict(torch.load(%r))
want = torch.load(%r)["model"]
got = ref.state_dict()["model"]
for mod in ("task_covar", "input_covar", "covar_module"):
    for k in want[mod]:
        assert torch.equal(got[mod][k], want[mod][k]), (mod, k)
assert torch.equal(got["train_inputs"][0], want["train_inputs"][0]) and torch.equal(got["train_targets"], want["train_targets"])
print("reference loaded the facade checkpoint")
""" % (golden, p2, path)
    out = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "reference loaded the facade checkpoint" in out.stdout, out.stderr[-2000:]


# ---- batched hyper-parameter fit: the host-side pieces that need no GPU
def test_batched_fit_lr_schedule_is_torchs_multisteplr_and_param_layout_counts():
    """`batched_fit.lr_schedule` == torch.optim.lr_scheduler.MultiStepLR(milestones = round(f T), gamma 0.1) stepped once per iteration
    (the reference's schedule, control_affine_model.py:293-300), value for value incl. duplicate and zero milestones; the raw-parameter
    row's length is the library's (`bcbf_fit_param_count`, pure host code) for the full-rank, rank-one and diagonal parameterisations."""
    import warnings
    from bayesian_cbf_amd import batched_fit, _lib
    for T in (1, 2, 3, 4, 5, 7, 10, 33, 50, 100):
        p_ = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p_], lr=0.1)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[int(round(f * T)) for f in batched_fit.MILESTONES])
            want = []
            for _ in range(T):
                want.append(opt.param_groups[0]["lr"])
                opt.step()
                sch.step()
        assert batched_fit.lr_schedule(0.1, T) == want, T
    n, m = 3, 2
    C = 1 + m
    assert _lib.lib.bcbf_fit_param_count(n, m, n, C) == n + 1 + n * n + n + C * C + C + C * n
    assert _lib.lib.bcbf_fit_param_count(n, m, 1, 1) == n + 1 + n + n + C + C + C * n
    assert _lib.lib.bcbf_fit_param_count(n, m, 0, 0) == n + 1 + n + C + C * n
    assert _lib.lib.bcbf_fit_param_count(9, m, 1, 1) < 0 and _lib.lib.bcbf_fit_param_count(n, 4, 1, 1) < 0
    with pytest.raises(RuntimeError):
        batched_fit.BatchedHyperFit(torch.zeros(2, 37, dtype=torch.float64), n, m)          # a CPU tensor: there is no CPU path
    with pytest.raises(ValueError):
        batched_fit.BatchedHyperFit(torch.zeros(2, 36, dtype=torch.float64), n, m)          # wrong row length
