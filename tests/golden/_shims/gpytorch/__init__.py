"""Minimal stand-in for the gpytorch fork the reference pins (requirements.txt:3).
Only what the reference's hand-written prediction path touches has behaviour."""
from . import kernels, means, models, lazy, likelihoods, mlls, priors, settings, distributions, utils  # noqa
