"""Closed-loop learning schedule shared by `LearnedShiftInvariantDynamics` (unicycle_move_to_pose.py:340-386) and
`MeanAdjustedModel` (controllers.py:336-378).

What the reference does: the controller hands every visited (x_t, u_t) to `train`; every `train_every_n_steps` calls
the regressor is REFIT FROM SCRATCH on the whole buffer -- finite-difference targets (x_{t+1} - x_t) / dt minus the
prior-mean dynamics, a random subsample when the buffer exceeds `max_train`, `training_iter` Adam steps on the
hyper-parameters, a fresh O(N^3) factorisation.  Between two refits the model does not see new data.

What this class adds on the same schedule (both off by default = the reference's behaviour, sample for sample):

* ``hyper_refit_every = k``: only every k-th scheduled refit re-optimises the hyper-parameters and refactorises;
  at the other scheduled points the observations gathered since the last one enter the regressor through
  `append_data` (`bcbf_gp_append`: bordered Cholesky, O(N^2) per observation).  With unchanged hyper-parameters
  that IS the refactorisation on the same points (tests compare it with the oracle's from-scratch refit).
* ``online_update = True``: every observation enters as soon as its finite-difference target exists (the call after
  it was visited), so the model the controller queries is never stale -- until the buffer outgrows `max_train`: from
  then on the regressor holds a random subsample, which is only re-drawn at the scheduled points.

When the training set would exceed `max_train` the window is re-drawn as the reference draws it (random subsample of
the whole buffer, `subsample`) and factored from scratch -- a bordered factor cannot drop rows.

* ``window = W`` (instead of `max_train`; needs one of the two options above): the model holds the MOST RECENT samples, a
  sliding window that moves in steps of 32 (the packed factor's block size): between W and W + 31 samples; when the next
  one would make it W + 32 the 32 oldest leave and the window is factored from scratch at the current hyper-parameters
  (`fit(..., training_iter=0)`: one matrix-core refit per 32 observations -- cheaper than a rank-32 update of the trailing
  factor would be, and exactly the from-scratch factor; `ops.ReservedGP(window=...)` is the batched form).  No random
  re-draw ever happens: the controller's model is always the newest data.
"""
import torch


class OnlineLearner:
    def __init__(self, regressor, residual_targets, dt, train_every_n_steps, max_train, training_iter, subsample,
                 enable_learning=True, hyper_refit_every=1, online_update=False, transform=None, window=None):
        """residual_targets(X[k,n], U[k,m], Xdot[k,n]) -> the regressor's targets (Xdot minus the prior mean);
        subsample(count, max_train) -> index tensor (the reference's two classes draw it differently);
        transform(X) -> regressor inputs (e.g. the shift-invariant wrapper)."""
        self.regressor, self.residual_targets, self.dt = regressor, residual_targets, dt
        self.train_every_n_steps, self.max_train, self.training_iter = train_every_n_steps, max_train, training_iter
        self.subsample, self.enable_learning = subsample, enable_learning
        self.hyper_refit_every, self.online_update = max(1, int(hyper_refit_every)), online_update
        self.transform = transform or (lambda X: X)
        self.window = None if window is None else int(window)
        self.lo = 0                    # window mode: the regressor holds buffer samples [lo, n_in_model)
        self.Xtrain, self.Utrain = [], []
        self.n_scheduled = 0           # scheduled refit points seen so far
        self.n_in_model = None         # the regressor holds exactly buffer samples [0, n_in_model); None: a subsample
        self.has_been_trained_once = False

    # -- buffer
    def _samples(self, lo, hi):
        """Buffer samples lo..hi-1 as (X, U, finite-difference Xdot); needs x_{hi} in the buffer."""
        X = torch.stack([x.reshape(-1) for x in self.Xtrain[lo:hi + 1]])
        U = torch.stack([u.reshape(-1) for u in self.Utrain[lo:hi]])
        return X[:-1], U, (X[1:] - X[:-1]) / self.dt

    BLOCK = 32

    def _window_lo(self, count):
        """Start of the sliding window for `count` samples: the largest multiple of 32 that keeps at most W + 31 of them."""
        over = count - self.window
        return 0 if over < self.BLOCK else (over // self.BLOCK) * self.BLOCK

    def _refit_from_scratch(self, training_iter):
        count = len(self.Xtrain) - 1
        if count <= 0:
            return
        if self.window is not None:
            self.lo = self._window_lo(count)
            X, U, Xdot = self._samples(self.lo, count)
            self.regressor.fit(self.transform(X), U, self.residual_targets(X, U, Xdot), training_iter=training_iter)
            self.n_in_model = count
            self.has_been_trained_once = True
            return
        X, U, Xdot = self._samples(0, count)
        Y = self.residual_targets(X, U, Xdot)
        self.n_in_model = count
        if self.max_train is not None and count > self.max_train:
            idx = self.subsample(count, self.max_train).to(X.device)
            X, U, Y = X[idx], U[idx], Y[idx]
            self.n_in_model = None
        self.regressor.fit(self.transform(X), U, Y, training_iter=training_iter)
        self.has_been_trained_once = True

    def _append_new(self):
        """Buffer samples the regressor has not seen yet -> append_data.  False if the window must be re-drawn."""
        count = len(self.Xtrain) - 1
        if self.n_in_model is None or not self.has_been_trained_once:
            return False
        if self.window is not None:
            if self._window_lo(count) != self.lo:
                return False                        # the window moves on by a block: factor it from scratch (hyper-parameters kept)
        elif self.max_train is not None and count > self.max_train:
            return False
        if count > self.n_in_model:
            X, U, Xdot = self._samples(self.n_in_model, count)
            self.regressor.append_data(self.transform(X), U, self.residual_targets(X, U, Xdot))
            self.n_in_model = count
        return True

    def observe(self, xi, uopt):
        """`train(xi, uopt)` of the reference classes."""
        nbuf = len(self.Xtrain)
        scheduled = nbuf > 0 and nbuf % int(self.train_every_n_steps) == 0 and self.enable_learning
        if scheduled:
            hyper = self.n_scheduled % self.hyper_refit_every == 0
            self.n_scheduled += 1
            if hyper or not self._append_new():
                self._refit_from_scratch(self.training_iter if hyper else 0)
        elif self.online_update and self.enable_learning and self.has_been_trained_once and self.n_in_model is not None:
            # (n_in_model None: the regressor holds a random subsample of a buffer beyond max_train -- new points then wait
            #  for the next SCHEDULED re-draw; re-drawing and refactorising at every control step would be a full O(N^3)
            #  refit per step on a window that changes at random)
            if not self._append_new():
                self._refit_from_scratch(0)        # the window overflowed just now: one re-draw
        self.Xtrain.append(xi.detach())
        self.Utrain.append(uopt.detach())
