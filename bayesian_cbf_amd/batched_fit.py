"""Hyper-parameter fit of MANY control-affine GPs at once (regime I at the reference's own workload: every refit of
`LearnedShiftInvariantDynamics` is `learned_dynamics.fit(..., training_iter=100)`, unicycle_move_to_pose.py:364-386 ->
ControlAffineRegressor.fit, control_affine_model.py:268-335 -- 100 Adam steps of the marginal likelihood, one model).

`BatchedHyperFit` holds the reference's RAW parameters of Bt models as rows of one device array `theta[Bt, P]` (layout:
include/bcbf.h, bcbf_fit_derive) and runs an Adam iteration of all of them as seven launches of libbcbf --
derive, refit, trtri, syrk_lt, kinv_apply, mll_grad, adam_step -- with the optimiser state on the device.  The host draws the
random numbers the reference draws (make_psd's jitter, the 1 + 1e-6 rand target perturbation, :318-321, :899-921) and looks
at the factorisation's `info` once per iteration (the x10 jitter retry, on the failed models only).

The one-model façade (`ControlAffineRegressor.fit`: torch autograd for the chain rule, torch.optim.Adam, MultiStepLR) is kept
as it is: it is the independent implementation the tests hold this one against (same draws -> same loss trajectory).
"""
import math

import torch

from . import ops

MILESTONES = (0.3, 0.6, 0.8, 0.9)          # MultiStepLR milestones as fractions of training_iter (:293-300), gamma 0.1


def lr_schedule(lr, training_iter):
    """lr of iteration 0 .. training_iter-1 under MultiStepLR(milestones=round(f T), gamma=0.1) stepped once per iteration
    (a milestone that occurs twice counts twice, as torch's Counter does)."""
    from collections import Counter
    ms = Counter(int(round(f * training_iter)) for f in MILESTONES)
    out, cur = [], lr
    for it in range(training_iter):
        if it in ms:
            cur = cur * 0.1 ** ms[it]                 # (the chainable form torch uses: successive products, not lr * 0.1^k)
        out.append(cur)
    return out


def _inv_softplus(v):
    v = v.double()
    return torch.where(v > 30, v, torch.log(torch.expm1(v)))


class BatchedHyperFit:
    def __init__(self, theta, x_dim, u_dim, rank=None, gamma_length_scale_prior=None):
        self.n, self.m = int(x_dim), int(u_dim)
        self.rA = self.n if rank is None else int(rank)
        self.rB = 1 + self.m if rank is None else int(rank)
        self.P = ops.fit_param_count(self.n, self.m, self.rA, self.rB)
        if theta.dim() != 2 or theta.shape[1] != self.P:
            raise ValueError("theta must be [Bt, %d] for n=%d m=%d ranks (%d, %d)" % (self.P, self.n, self.m, self.rA, self.rB))
        if not theta.is_cuda:
            raise RuntimeError("the batched fit runs in libbcbf on a ROCm GPU; there is no CPU path")
        self.theta = theta.contiguous()
        self.gamma_length_scale_prior = gamma_length_scale_prior
        self.mom1 = torch.zeros_like(self.theta)
        self.mom2 = torch.zeros_like(self.theta)
        self.steps_done = 0
        dev, dt = theta.device, theta.dtype
        # the reference's draws, replaceable (tests replay recorded ones): jitter_rand(idx[k], N) -> [k, N] in [0,1),
        # target_rand(Y) -> like Y
        self.jitter_rand = lambda idx, N: torch.rand(idx.numel(), N, dtype=dt, device=dev)
        self.target_rand = torch.rand_like
        self.jitter_level = None          # per model: the level its last factorisation succeeded at
        self.losses = None
        self.skipped = None

    # ---- parameter layout (bcbf.h) --------------------------------------------------------------------------------------
    def _offsets(self):
        n, C, rA, rB = self.n, 1 + self.m, self.rA, self.rB
        o = dict(ell=0, s2=n, Wa=n + 1)
        o["va"] = o["Wa"] + n * rA
        o["Wb"] = o["va"] + n
        o["vb"] = o["Wb"] + C * rB
        o["M0"] = o["vb"] + C
        return o

    @classmethod
    def from_models(cls, models, gamma_length_scale_prior=None, dtype=None, device=None):
        """Rows from parameter containers of the façade (`ControlAffineRegressor.model`: KernelParams)."""
        m0 = models[0]
        C, n = m0.matshape
        rows = [torch.cat([m.raw_lengthscale.detach().reshape(-1), m.raw_outputscale.detach().reshape(-1),
                           m.A_covar_factor.detach().reshape(-1), m.A_raw_var.detach().reshape(-1),
                           m.B_covar_factor.detach().reshape(-1), m.B_raw_var.detach().reshape(-1),
                           m.mean_constants.detach().reshape(-1)]) for m in models]
        theta = torch.stack(rows).to(dtype=dtype or m0.raw_lengthscale.dtype, device=device or m0.raw_lengthscale.device)
        rank = None if (m0.A_covar_factor.shape[1] == n and m0.B_covar_factor.shape[1] == C) else m0.A_covar_factor.shape[1]
        return cls(theta, n, C - 1, rank=rank, gamma_length_scale_prior=gamma_length_scale_prior)

    def to_model(self, b, model):
        """Write row b back into a façade parameter container (in place)."""
        o, n, C = self._offsets(), self.n, 1 + self.m
        th = self.theta[b]
        with torch.no_grad():
            model.raw_lengthscale.copy_(th[o["ell"]:o["ell"] + n].reshape(1, n))
            model.raw_outputscale.copy_(th[o["s2"]])
            model.A_covar_factor.copy_(th[o["Wa"]:o["va"]].reshape(n, self.rA))
            model.A_raw_var.copy_(th[o["va"]:o["Wb"]])
            model.B_covar_factor.copy_(th[o["Wb"]:o["vb"]].reshape(C, self.rB))
            model.B_raw_var.copy_(th[o["vb"]:o["M0"]])
            model.mean_constants.copy_(th[o["M0"]:])
        return model

    @classmethod
    def from_values(cls, A, Bm, ell, s2, M0, dtype=None):
        """Rows whose derived values are the given A[Bt,n,n], Bm[Bt,C,C], ell[Bt,n], s2[Bt], M0[Bt,C,n] (full-rank factors =
        Cholesky factors, a tiny diagonal -- `ControlAffineRegressor.set_kernel_params`)."""
        Bt, n = ell.shape
        C = Bm.shape[1]

        def fac(S):
            dev = S.device
            S = S.double().cpu()                          # (k x k factorisations, k <= 8: on the host, like set_kernel_params)
            eps = 1e-10 * S.diagonal(dim1=1, dim2=2).mean(dim=1)
            L = torch.linalg.cholesky(S - eps[:, None, None] * torch.eye(S.shape[1], dtype=S.dtype))
            return L.reshape(Bt, -1).to(dev), _inv_softplus(eps[:, None].expand(Bt, S.shape[1])).to(dev)
        Wa, va = fac(A)
        Wb, vb = fac(Bm)
        theta = torch.cat([_inv_softplus(ell), _inv_softplus(s2).reshape(Bt, 1), Wa, va, Wb, vb, M0.double().reshape(Bt, -1)], dim=1)
        return cls(theta.to(dtype or ell.dtype).contiguous(), n, C - 1)

    def derive(self, want_Ainv=False):
        """dict(ell, s2, A, Bm, M0[, Ainv, logdetA]) at the current parameters."""
        return ops.fit_derive(self.theta, self.n, self.m, self.rA, self.rB, want_Ainv=want_Ainv)

    # ---- one likelihood evaluation ---------------------------------------------------------------------------------------
    def _factor(self, X, UH, hp, max_tries=10):
        """make_psd's schedule per model (:899-921): jitter = level * rand(N), x10 and a fresh draw on a failed pivot -- the
        retries run on the failed models only (a gathered sub-batch).  Inside one fit a model starts one level below the level
        that last worked, never below 1e-5 (as the façade's `neg_mll_backward`).  Returns (Lop, still_bad[Bt] int32)."""
        Bt, N, _ = X.shape
        dev = X.device
        if self.jitter_level is None:
            level = torch.full((Bt,), 1e-5, dtype=X.dtype, device=dev)
        else:
            level = torch.clamp(self.jitter_level / 10, min=1e-5)
        everyone = torch.arange(Bt, device=dev)
        jit = (level[:, None] * self.jitter_rand(everyone, N)).contiguous()
        Lop, _, info, _ = ops.refit(X, UH, hp["Bm"], hp["ell"], hp["s2"], jit)
        bad = info != 0
        for ntry in range(1, max_tries):
            idx = bad.nonzero().flatten()                     # the iteration's look at the device (one per jitter level)
            if idx.numel() == 0:
                break
            level[idx] = level[idx] * 10
            sub = lambda t: t.index_select(0, idx).contiguous()
            jit_s = (level[idx][:, None] * self.jitter_rand(idx, N)).contiguous()
            Ls, _, info_s, _ = ops.refit(sub(X), sub(UH), sub(hp["Bm"]), sub(hp["ell"]), sub(hp["s2"]), jit_s)
            Lop.index_copy_(0, idx, Ls)
            bad = torch.zeros_like(bad)
            bad[idx] = info_s != 0
        self.jitter_level = level
        return Lop, bad.to(torch.int32)

    def value_and_grad(self, X, UH, Y, step=0, lr=0.0):
        """loss[Bt] (and, step >= 1, one Adam update) at the current parameters; Y = the (perturbed) targets."""
        Bt, N, n = X.shape
        hp = self.derive(want_Ainv=True)
        Lop, skip = self._factor(X, UH, hp)
        R = (Y - (UH.unsqueeze(-1) * hp["M0"].unsqueeze(1)).sum(2)).contiguous()
        Kinv = ops.kb_inverse(Lop, N)
        alpha = ops.kinv_apply(Kinv, R)
        sums = ops.mll_grad(Lop, alpha, Kinv, X, UH, R, hp["Ainv"], hp["Bm"], hp["ell"], hp["s2"])
        loss, grad = ops.fit_adam_step(self.theta, self.mom1, self.mom2, sums, hp["Ainv"], hp["logdetA"], N, self.n, self.m, self.rA,
                                       self.rB, step, lr, skip=skip, gamma_prior=self.gamma_length_scale_prior, want_grad=step == 0)
        return loss, grad, skip

    def fit(self, X, U, Xdot, training_iter=100, lr=0.1):
        """`training_iter` Adam steps on -log p(Y_b) / (N n) of every model b (X[Bt,N,n], U[Bt,N,m], Xdot[Bt,N,n]; the models
        are independent).  Records `losses[training_iter, Bt]` (NaN where a model's factorisation failed after ten jitter
        levels: that model skipped the step) and `skipped[Bt]` (count)."""
        ops._chk(self.theta, X, U, Xdot)
        Bt, N, n = X.shape
        if Bt != self.theta.shape[0] or n != self.n or U.shape[2] != self.m:
            raise ValueError("data [%d, %d, %d] / [.., %d] does not fit %d models of n=%d m=%d" % (Bt, N, n, U.shape[2], self.theta.shape[0], self.n, self.m))
        UH = torch.cat([torch.ones_like(U[..., :1]), U], dim=-1).contiguous()
        self.jitter_level = None
        self.mom1.zero_()                                   # a fit() is a NEW optimiser, as in the reference (:290-300)
        self.mom2.zero_()
        self.steps_done = 0
        losses = torch.empty(training_iter, Bt, dtype=self.theta.dtype, device=self.theta.device)
        skipped = torch.zeros(Bt, dtype=torch.int32, device=self.theta.device)
        for it, lr_it in enumerate(lr_schedule(lr, training_iter)):
            Y = (Xdot * (1 + 1e-6 * self.target_rand(Xdot))).contiguous()          # :318-321
            self.steps_done += 1
            loss, _, skip = self.value_and_grad(X, UH, Y, step=self.steps_done, lr=lr_it)
            losses[it] = loss
            skipped += skip
        self.losses, self.skipped = losses, skipped
        return self
