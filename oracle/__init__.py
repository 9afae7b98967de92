"""CPU oracle for the Bayesian-CBF hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain numpy/scipy (fp64 by default) restatement of the reference's algorithm for the
GP-posterior + CBF/CLF chance-constraint + conic-solve path.  Every function cites the
reference file:line it follows (paths relative to the upstream checkout).

Who may import this package: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` -- as the checker / the timed CPU baseline only.  Nothing under
`bayesian_cbf_amd/` imports it; the product path fails loudly when the HIP library is
missing instead of falling back to this code.

Parity pin: `tests/golden/*.npz` were produced by executing the reference's own
`bayes_cbf` modules in the build container (`tests/golden/gen_golden.py`; third-party
gpytorch/tensorboard/kwplus replaced by the stand-ins in `tests/golden/_shims`).
`tests/test_oracle_golden.py` checks this oracle against every one of them.
Third-party arithmetic that is NOT in the reference tree and is restated from its
published definition: gpytorch's RBF-ARD/ScaleKernel/IndexKernel parameterisation
(`gp_posterior.rbf_ard_kernel`, `index_kernel_covar`) and cvxopt's `coneqp`
(`socp.coneqp`); see DESIGN.md section "Oracle".
"""
