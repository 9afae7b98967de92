"""Mirror of bayes_cbf/cbc1.py: rel-degree-1 probabilistic safety condition."""
from .cbc2 import CBCExpr, cbc1_safety_factor  # noqa: F401


class RelDeg1Safety:
    """cbc1.py:17-52: subclasses provide gamma, model, max_unsafe_prob, cbf(x), grad_cbf(x);
    cbc(u) = grad_cbf(x)' (f + g u)(x) + gamma cbf(x) as a GP in x."""

    def cbc(self, u0):
        return CBCExpr(1, self.cbf, self.grad_cbf, self.model, u0, gamma=self.gamma)

    def safety_factor(self):
        return cbc1_safety_factor(self.max_unsafe_prob)
