import torch.distributions as base_distributions  # noqa
from torch.distributions import MultivariateNormal  # noqa
