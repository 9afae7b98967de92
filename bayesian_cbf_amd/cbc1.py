"""Mirror of bayes_cbf/cbc1.py: rel-degree-1 probabilistic safety condition."""
from .cbc2 import cbc1_safety_factor  # noqa: F401
from .gp_algebra import DeterministicGP


class RelDeg1Safety:
    """cbc1.py:17-52: subclasses provide gamma, model, max_unsafe_prob, cbf(x), grad_cbf(x);
    cbc(u) = grad_cbf(x)' (f + g u)(x) + gamma cbf(x) as a GP in x -- the reference's expression, lowered onto
    bcbf_posterior_query + bcbf_cbc_terms by gp_algebra."""

    def cbc(self, u0):
        h_gp = DeterministicGP(lambda x: self.gamma * self.cbf(x), shape=(1,), name="h(x)")
        grad_h_gp = DeterministicGP(self.grad_cbf, shape=(self.model.state_size,), name="grad h(x)")
        fu_gp = self.model.fu_func_gp(u0)
        return grad_h_gp.t() @ fu_gp + h_gp

    def safety_factor(self):
        return cbc1_safety_factor(self.max_unsafe_prob)
