"""GPU tests of the batched hyper-parameter fit (`BatchedHyperFit`: bcbf_fit_derive, bcbf_kinv_apply, bcbf_fit_adam_step around
bcbf_refit / bcbf_trtri / bcbf_syrk_lt / bcbf_mll_grad) -- ControlAffineRegressor.fit (control_affine_model.py:268-335 of the
reference) for many models at once.  Held against (a) the CPU oracle's likelihood by finite differences, (b) the one-model
façade `fit` (torch autograd for the chain rule, torch.optim.Adam, MultiStepLR) on the same data and the same random draws."""
import numpy as np
import pytest
import torch

from _tolreport import rel_close  # noqa: F401

pytestmark = pytest.mark.gpu
DEV = "cuda"
T64 = dict(dtype=torch.float64, device=DEV)


def t(a):
    return torch.as_tensor(np.ascontiguousarray(a), **T64)


def _data(rng, Bt, N, n, m):
    X = rng.uniform(-2, 2, (Bt, N, n))
    U = rng.normal(size=(Bt, N, m))
    Wf, Wg = rng.normal(size=(Bt, n, n)), rng.normal(size=(Bt, n, n, m))
    Xdot = np.sin(np.einsum("bij,bkj->bki", Wf, X)) + 0.5 * np.einsum("bki,bijm,bkm->bkj", np.cos(X), Wg, U) + 1e-3 * rng.normal(size=(Bt, N, n))
    return X, U, Xdot


def _models(Bt, n, m, rank, seed, prior=None):
    from bayesian_cbf_amd.control_affine_model import ControlAffineRegressor
    torch.manual_seed(seed)
    regs = []
    for b in range(Bt):
        reg = ControlAffineRegressor(n, m, device=DEV, dtype=torch.float64, rank=rank, gamma_length_scale_prior=prior)
        with torch.no_grad():
            reg.model.raw_lengthscale.copy_(0.3 * torch.randn(1, n))
            reg.model.raw_outputscale.copy_(0.3 * torch.randn(()))
            reg.model.mean_constants.copy_(0.1 * torch.randn((1 + m) * n))
        regs.append(reg)
    return regs


@pytest.mark.parametrize("n,m,N,rank,prior", [(3, 2, 24, None, None), (2, 1, 40, None, None), (3, 2, 33, 1, (1e-3, 1e-3)),
                                              (2, 1, 20, 0, None), (4, 3, 17, None, None)],
                         ids=["unicycle", "pendulum", "rank-one+prior", "diag", "n4m3"])
def test_batched_value_and_gradient_vs_oracle_finite_differences(n, m, N, rank, prior):
    """loss[b] equals the oracle's -log p / (N n) (1e-9), and d loss / d theta (every raw parameter, chain rule on the device)
    equals central finite differences of it -- for three models with different data and parameters in one launch."""
    from bayesian_cbf_amd.batched_fit import BatchedHyperFit
    from oracle import gp_posterior as ogp
    import math
    Bt = 3
    rng = np.random.default_rng(5 + n)
    X, U, Xdot = _data(rng, Bt, N, n, m)
    regs = _models(Bt, n, m, rank, seed=11)
    bf = BatchedHyperFit.from_models([r.model for r in regs], gamma_length_scale_prior=prior)
    draws = rng.uniform(0.1, 0.9, (Bt, N))
    bf.jitter_rand = lambda idx, N_: t(draws)[idx]
    UHt = torch.cat([torch.ones(Bt, N, 1, **T64), t(U)], dim=2).contiguous()
    loss, grad, skip = bf.value_and_grad(t(X), UHt, t(Xdot))
    assert int(skip.sum()) == 0 and grad.shape == (Bt, bf.P)
    theta0 = bf.theta.clone()

    def oracle_loss(b, theta_row):
        hp = BatchedHyperFit(theta_row[None].contiguous(), n, m, rank=rank).derive()
        A, B = hp["A"][0].cpu().numpy(), hp["Bm"][0].cpu().numpy()
        ell, s2, M0 = hp["ell"][0].cpu().numpy(), float(hp["s2"][0]), hp["M0"][0].cpu().numpy()
        val = -ogp.marginal_log_likelihood(X[b], ogp.homogeneous_controls(U[b]), Xdot[b], A, B, ell, s2, M0, 1e-5 * draws[b]) / (N * n)
        if prior is not None:
            c, r = prior
            val -= sum(c * math.log(r) - math.lgamma(c) + (c - 1) * math.log(l) - r * l for l in ell) / (N * n)
        return val

    for b in range(Bt):
        np.testing.assert_allclose(float(loss[b]), oracle_loss(b, theta0[b]), rtol=1e-9, atol=1e-10)
        for k in range(bf.P):
            h = 1e-5
            tp, tm = theta0[b].clone(), theta0[b].clone()
            tp[k] += h
            tm[k] -= h
            fd = (oracle_loss(b, tp) - oracle_loss(b, tm)) / (2 * h)
            np.testing.assert_allclose(float(grad[b, k]), fd, rtol=3e-5, atol=3e-7, err_msg="model %d theta[%d]" % (b, k))
    # derived values are the façade's (softplus, W W' + diag softplus)
    hp = bf.derive(want_Ainv=True)
    for b, reg in enumerate(regs):
        mdl = reg.model
        for k, v in (("A", mdl.A), ("Bm", mdl.B), ("ell", mdl.lengthscale.reshape(-1)), ("s2", mdl.outputscale), ("M0", mdl.M0)):
            np.testing.assert_allclose(hp[k][b].cpu().numpy(), v.detach().cpu().numpy(), rtol=1e-13, atol=1e-15, err_msg=k)
        np.testing.assert_allclose(hp["Ainv"][b].cpu().numpy(), np.linalg.inv(mdl.A.detach().cpu().numpy()), rtol=1e-9)
        np.testing.assert_allclose(float(hp["logdetA"][b]), np.linalg.slogdet(mdl.A.detach().cpu().numpy())[1], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("n,m,N,rank,prior,iters", [(3, 2, 40, None, None, 30), (2, 1, 32, None, None, 20), (3, 2, 24, 1, (1e-3, 1e-3), 20)],
                         ids=["unicycle", "pendulum", "rank-one+prior"])
def test_batched_fit_follows_the_one_model_facade_fit(n, m, N, rank, prior, iters):
    """Four models fitted AT ONCE (seven launches per Adam iteration, optimiser state on the device) against the façade's
    one-model `fit` (torch autograd + torch.optim.Adam + MultiStepLR) run four times on the same data with the same draws:
    the loss trajectories agree to 1e-9 and the final raw parameters to 1e-8."""
    from bayesian_cbf_amd.batched_fit import BatchedHyperFit
    Bt = 4
    rng = np.random.default_rng(17 + n)
    X, U, Xdot = _data(rng, Bt, N, n, m)
    regs = _models(Bt, n, m, rank, seed=23, prior=prior)
    jd = rng.uniform(0.0, 1.0, (iters, Bt, N))
    td = rng.uniform(0.0, 1.0, (iters, Bt, N, n))
    bf = BatchedHyperFit.from_models([r.model for r in regs], gamma_length_scale_prior=prior)
    theta_init = bf.theta.clone()
    it_j, it_t = iter(range(iters)), iter(range(iters))
    bf.jitter_rand = lambda idx, N_: t(jd[next(it_j)])[idx]
    bf.target_rand = lambda Y: t(td[next(it_t)])
    bf.fit(t(X), t(U), t(Xdot), training_iter=iters, lr=0.1)
    assert int(bf.skipped.sum()) == 0
    losses = bf.losses.cpu().numpy()
    for b, reg in enumerate(regs):
        jj, tt = iter(range(iters)), iter(range(iters))
        reg.rand_fn = lambda k, b=b, jj=jj: t(jd[next(jj), b, :k])
        reg.target_rand_fn = lambda Y, b=b, tt=tt: t(td[next(tt), b])
        reg.fit(t(X[b]), t(U[b]), t(Xdot[b]), training_iter=iters, lr=0.1)
        np.testing.assert_allclose(losses[:, b], np.array(reg.fit_losses), rtol=1e-9, atol=1e-10, err_msg="loss trajectory, model %d" % b)
        ref_row = BatchedHyperFit.from_models([reg.model]).theta[0]
        np.testing.assert_allclose(bf.theta[b].cpu().numpy(), ref_row.cpu().numpy(), rtol=1e-8, atol=1e-9, err_msg="final parameters, model %d" % b)
        assert float((bf.theta[b] - theta_init[b]).abs().max()) > 0.05           # (the fit moved them)
        assert losses[-1, b] < losses[0, b]
    # rows go back into façade containers
    from bayesian_cbf_amd.control_affine_model import ControlAffineRegressor
    fresh = ControlAffineRegressor(n, m, device=DEV, dtype=torch.float64, rank=rank)
    bf.to_model(2, fresh.model)
    for (k, a), (_, b_) in zip(fresh.model.named_parameters(), regs[2].model.named_parameters()):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b_.detach().cpu().numpy(), rtol=1e-8, atol=1e-9, err_msg=k)


def test_batched_fit_retries_failed_factorisations_on_the_failed_models_only():
    """make_psd's x10 jitter retry inside the batched iteration: a model whose K_b is singular at the base jitter level (duplicated
    inputs, fp32 arithmetic) is re-factored alone at higher levels; the others' numbers are those of a batch without it."""
    from bayesian_cbf_amd.batched_fit import BatchedHyperFit
    Bt, N, n, m = 5, 48, 3, 2
    rng = np.random.default_rng(3)
    X, U, Xdot = _data(rng, Bt, N, n, m)
    X[1, 1::2], U[1, 1::2] = X[1, 0::2], U[1, 0::2]                      # model 1: every point twice
    f32 = lambda a: torch.as_tensor(a, dtype=torch.float32, device=DEV).contiguous()
    regs = _models(Bt, n, m, None, seed=29)
    mk = lambda sel: BatchedHyperFit.from_models([regs[b].model for b in sel], dtype=torch.float32)
    jd = rng.uniform(0.2, 1.0, (64, N))
    def fit(sel):
        bf = mk(sel)
        calls = []

        def draw(idx, N_):
            calls.append(idx.tolist())
            d = f32(jd[len(calls) % 64])[None].expand(idx.numel(), N_).contiguous()
            if idx.numel() == len(sel) and 1 in sel:        # base level: model 1 draws a NEGATIVE shift -> its duplicated points
                d[sel.index(1)] = -1.0                       # make K_b indefinite, whatever the rounding
            return d
        bf.jitter_rand = draw
        bf.target_rand = lambda Y: torch.zeros_like(Y)
        bf.fit(f32(X[sel]), f32(U[sel]), f32(Xdot[sel]), training_iter=3, lr=0.05)
        return bf, calls
    bf, calls = fit([0, 1, 2, 3, 4])
    assert any(c == [1] for c in calls), calls                            # a retry that drew for model 1 alone
    assert float(bf.jitter_level[1]) > float(bf.jitter_level[0])
    assert torch.isfinite(bf.losses).all() and int(bf.skipped.sum()) == 0
    assert bool(torch.isfinite(bf.theta).all())
    # the healthy models' fit does not depend on who else is in the batch (same draws: the jd rows are per call, not per model)
    bf2, calls2 = fit([0, 2, 3, 4])
    assert all(len(c) == 4 for c in calls2)


def test_kinv_apply_is_the_symmetric_product():
    from bayesian_cbf_amd import ops
    for dtype, tol in ((torch.float64, 1e-13), (torch.float32, 2e-6)):
        # (from 16 models on, N a multiple of the 16-byte vector and nt <= 4: the vector form -- 20 x 512, 17 x 256; 16 x 130 / 18 x 64 with nt = 5: the scalar form)
        for Bt, N, nt in ((3, 70, 3), (1, 512, 1), (2, 33, 8), (20, 512, 3), (17, 256, 2), (16, 130, 3), (18, 64, 5), (16, 1024, 4)):
            g = torch.Generator(device=DEV).manual_seed(N)
            S = torch.randn(Bt, N, N, dtype=dtype, device=DEV, generator=g)
            S = (S + S.transpose(1, 2)).contiguous()
            R = torch.randn(Bt, N, nt, dtype=dtype, device=DEV, generator=g)
            got = ops.kinv_apply(S, R)
            want = (S.double() @ R.double())
            assert float((got.double() - want).abs().max() / want.abs().max()) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-13), (torch.float32, 2e-6)], ids=["f64", "f32"])
def test_syrk_lt_forms_against_the_plain_product(dtype, tol):
    """bcbf_syrk_lt (K_b^-1 = Linv' Linv, Linv lower triangular: fit.hip's consumers read the full symmetric matrix) against torch's product in
    fp64, every form: one tile per wave (few models), 2 x 2 tiles per wave (batches, N not a multiple of 128), and the 128 x 128 block per workgroup
    whose operands go through an LDS ring (batches, N a multiple of 128; it skips the structurally zero part of Linv and writes the mirror image
    through LDS) -- incl. batches that are not a multiple of the eight XCDs the workgroups are dealt to.  Exactly symmetric."""
    from bayesian_cbf_amd import ops
    from bayesian_cbf_amd._lib import lib
    for Bt, N in ((2, 96), (3, 512), (16, 200), (20, 256), (17, 512), (16, 384), (33, 128)):
        g = torch.Generator(device=DEV).manual_seed(100 + N)
        Linv = torch.tril(torch.randn(Bt, N, N, dtype=dtype, device=DEV, generator=g)).contiguous()
        Kinv = torch.full((Bt, N, N), float("nan"), dtype=dtype, device=DEV)
        ops.check(getattr(lib, "bcbf_syrk_lt" + ops._suf(Linv))(ops._p(Linv), ops._p(Kinv), Bt, N, ops._stream(Linv)), "bcbf_syrk_lt")
        want = Linv.double().transpose(1, 2) @ Linv.double()
        assert bool(torch.isfinite(Kinv).all()), (Bt, N)
        assert torch.equal(Kinv, Kinv.transpose(1, 2)), (Bt, N)
        err = float((Kinv.double() - want).abs().max() / want.abs().max())
        assert err < tol * (N / 32), (Bt, N, err)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 2e-3)], ids=["f64", "f32"])
def test_trtri_on_the_matrix_cores_is_the_inverse_of_the_factor(dtype, tol):
    """bcbf_trtri for batches (trtri.hip: one wave per block column, MFMA tiles) against the dense factor the refit hands out:
    L Linv = I, zeros above the diagonal, and the same numbers as the solve-based form a single model takes (Bt < 4); N not a
    multiple of 32, one block, many blocks."""
    from bayesian_cbf_amd import ops
    from bayesian_cbf_amd._lib import lib
    from bayesian_cbf_amd.synthetic import make_instances
    for Bt, N in ((5, 70), (6, 32), (4, 17), (9, 512), (7, 300), (19, 256), (8, 480)):        # (batches that are / are not a multiple of the eight XCDs)
        p = make_instances(Bt, N, 3, 2, dtype=dtype, device=DEV, seed=N)
        jit = (p["jitter"] * (100 if dtype == torch.float32 else 1)).contiguous()
        Lop, _, info, Ld = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], jit, want_dense=True)
        assert int((info != 0).sum()) == 0
        Linv = torch.full((Bt, N, N), float("nan"), dtype=dtype, device=DEV)
        ops.check(getattr(lib, "bcbf_trtri" + ops._suf(Lop))(ops._p(Lop), ops._p(Linv), Bt, N, ops._stream(Lop)), "bcbf_trtri")
        assert bool(torch.isfinite(Linv).all())
        assert float(torch.triu(Linv, diagonal=1).abs().max()) == 0.0
        eye = torch.eye(N, dtype=torch.float64, device=DEV)
        res = (Ld.double() @ Linv.double() - eye).abs().amax(dim=(1, 2))
        cond_scale = (Linv.double().abs().amax(dim=(1, 2)) * Ld.double().abs().amax(dim=(1, 2)))
        assert float((res / cond_scale).max()) < tol, (Bt, N, float((res / cond_scale).max()))
        one = torch.empty(1, N, N, dtype=dtype, device=DEV)          # the solve-based form (a single model)
        ops.check(getattr(lib, "bcbf_trtri" + ops._suf(Lop))(ops._p(Lop[2:3].contiguous()), ops._p(one), 1, N, ops._stream(Lop)), "bcbf_trtri")
        scale = float(one.abs().max())
        assert float((one[0] - Linv[2]).abs().max()) <= (1e-10 if dtype == torch.float64 else 2e-3) * scale


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 2e-3)], ids=["f64", "f32"])
@pytest.mark.parametrize("Bt,N", [(3, 300), (5, 512), (70, 300), (66, 512), (64, 129)])
def test_mll_grad_row_form_on_half_the_pairs_equals_the_pair_form(Bt, N, dtype, tol):
    """bcbf_mll_grad's row form visits the pairs j <= i only (weight 1 on the sum of a pair's two ordered terms, 1/2 on the
    diagonal, (B + B') staged per tile, g_B = s2 (M + M')) across row chunks (N > 256), column tiles (N > 128) and column slices
    (few models); the pair-per-thread form (the entry point without a workspace: one workgroup per model, every ordered pair) is
    an independent evaluation of the same sums.  Also with a NON-symmetric B, where u_i' B u_j != u_j' B u_i."""
    from bayesian_cbf_amd import ops
    from bayesian_cbf_amd.ops import _p, _suf, _stream, lib
    from bayesian_cbf_amd.synthetic import make_instances
    n, m = 3, 2
    p = make_instances(Bt, N, n, m, dtype=dtype, device=DEV, seed=4)
    Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
    assert (info == 0).all()
    Kinv = ops.kb_inverse(Lop, N)
    R = (p["Xdot"] - p["UH"] @ p["M0"]).contiguous()
    alpha = ops.kinv_apply(Kinv, R)
    Ainv = torch.linalg.inv(p["A"]).contiguous()
    gen = torch.Generator(device=DEV).manual_seed(17)
    for sym in (True, False):
        Bm = p["Bm"] if sym else (p["Bm"] + 0.3 * torch.randn(p["Bm"].shape, dtype=dtype, device=DEV, generator=gen)).contiguous()
        rows = ops.mll_grad(Lop, alpha, Kinv, p["X"], p["UH"], R, Ainv, Bm, p["ell"], p["s2"])
        f = dict(dtype=dtype, device=DEV)
        C = m + 1
        pair = (torch.empty(Bt, n, **f), torch.empty(Bt, **f), torch.empty(Bt, C, C, **f), torch.empty(Bt, **f), torch.empty(Bt, n, n, **f),
                torch.empty(Bt, C, n, **f))
        rc = getattr(lib, "bcbf_mll_grad" + _suf(p["X"]))(_p(Lop), _p(alpha), _p(Kinv), _p(p["X"]), _p(p["UH"]), _p(R), _p(Ainv), _p(Bm),
                                                          _p(p["ell"]), _p(p["s2"]), *(_p(o) for o in pair), Bt, N, n, m, None, _stream(p["X"]))
        assert rc == 0
        torch.cuda.synchronize()
        for name, a, b in zip(("g_ell", "g_s2", "g_B", "logdetK", "RtA", "UHtA"), rows, pair):
            scale = max(float(b.abs().max()), 1e-6)
            err = float((a - b).abs().max()) / scale
            assert err <= tol, "%s (B %s): %g" % (name, "symmetric" if sym else "not symmetric", err)
