"""Constants of bayes_cbf/gp_algebra.py used by the oracle."""
EIG_EPS = 2e-3   # gp_algebra.py:317: eigenvalues of the kernel Hessian in (-EPS, 0) are treated as rounding
