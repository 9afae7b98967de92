"""Known-answer inputs shared by the CPU and GPU tests."""
import numpy as np


def cvxopt_doc_example():
    linear_objective = np.array([-2., 1., 5.])
    A = [np.array([[-13., 3., 5.], [-12., 12., -6.]]),
         np.array([[-3., 6., 2.], [1., 9., 2.], [-1., -19., 3.]])]
    b = [np.array([-3., -2.]), np.array([0., 3., -42.])]
    c = [np.array([-12., -6., 5.]), np.array([-3., 6., -10.])]
    d = [np.array(-12.), np.array(27.)]
    return linear_objective, list(zip(("1", "2"), zip(A, b, c, d)))
