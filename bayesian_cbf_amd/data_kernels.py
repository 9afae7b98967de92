"""The data kernels of the library on the host (torch), for the few places that evaluate a PRIOR kernel value or derivative
outside the device kernels (`_prior_knl`, `gp_eval`, the rel-degree-2 façade): k(x, x') = s2 * shape(d2),
d2 = sum_d ((x_d - x'_d) / ell_d)^2 -- csrc/bcbf_common.h: kernel_shape.

    "rbf"           exp(-d2 / 2)                                     the reference's ScaleKernel(RBFKernel(ard)), control_affine_model.py:164-171
    "matern52"      (1 + a + a^2 / 3) exp(-a),  a = sqrt(5 d2)       opt-in
    "rbf_matern52"  the product of the two, one set of length scales  opt-in ("RBF x Matern", BASELINE.json north_star)

The opt-in kernels have no reference counterpart (the reference has no Matern kernel): parity unpinned."""
import torch

KINDS = ("rbf", "matern52", "rbf_matern52")


def shape_terms(kernel, d2):
    """(shape, dshape, ddshape) at squared scaled distance d2 (tensor): dshape = -2 d shape / d(d2), ddshape = 2 d dshape / d(d2).
    With d_d = (x_d - x'_d) / ell_d^2:  dk/dx_d = -s2 dshape d_d,  d2k / dx_d dx'_e = s2 (dshape delta_de / ell_d^2 + ddshape d_d d_e)."""
    if kernel == "rbf":
        e = torch.exp(-0.5 * d2)
        return e, e, -e
    a = torch.sqrt(5.0 * d2)
    if kernel == "matern52":
        e = torch.exp(-a)
        return (1.0 + a + a * a / 3.0) * e, 5.0 / 3.0 * (1.0 + a) * e, -25.0 / 3.0 * e
    if kernel == "rbf_matern52":
        e = torch.exp(-a - 0.5 * d2)
        poly, dpoly = 1.0 + a + 5.0 / 3.0 * d2, 5.0 / 3.0 * (1.0 + a)
        return poly * e, (poly + dpoly) * e, -(38.0 / 3.0 + 13.0 / 3.0 * a + 5.0 / 3.0 * d2) * e
    raise ValueError("data kernel %r: one of %s" % (kernel, KINDS))


def kxx(kernel):
    """d2 shape / dx_d dx'_d at x' = x in units of 1 / ell_d^2 (= dshape(0))."""
    return {"rbf": 1.0, "matern52": 5.0 / 3.0, "rbf_matern52": 8.0 / 3.0}[kernel]
