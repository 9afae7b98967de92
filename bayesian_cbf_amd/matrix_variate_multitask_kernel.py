"""Mirror of bayes_cbf/matrix_variate_multitask_kernel.py: the prior covariance of the matrix-variate GP over mixed
train-type rows (mask 1: one state x_i with its homogeneous control uh_i, n outputs F(x_i) uh_i) and test-type rows
(mask 0: the full matrix F(x_j), (1+m) n outputs), for encoded inputs MXU = [mask, x, uh]
(`HetergeneousMatrixVariateKernel.forward / mask_dependent_covar / kernel1 / kernel2 / correlation_kernel_12`, :99-204):

    K11 = (H1 (K (x) B) H2') (x) A      K22 = ((K (x) B)) (x) A      K12 = (H1 (K (x) B)) (x) A,    H = blockdiag(uh_i')

The reference composes gpytorch lazy tensors.  Here the data-kernel factor is ONE launch of `bcbf_kb_build` on the
expanded row set -- a mask-1 row contributes one row with its uh, a mask-0 row contributes 1+m rows with the unit
vectors e_p, so that k(x, x') (uh' B uh') is exactly the wanted block entry (H (K (x) B) H' = K o (UH B UH'), SURVEY
8a) -- and the Kronecker factor A is applied when the dense matrix is handed back.  The hot path never forms it
(`ControlAffineRegressor` works on K_b alone); this class exists for code written against the kernel module itself.
"""
import torch

from . import ops
from .control_affine_model import CatEncoder  # noqa: F401  (the decoder type, as upstream)


class MatrixVariateIndexKernel:
    """vec(F) ~ N(M, V (x) U): U = A [n,n] over state dimensions, V = B [(1+m),(1+m)] over controls (:18-47)."""

    def __init__(self, U, V):
        self.U, self.V = U, V
        self.matshape = (U.shape[-1], V.shape[-1])

    @property
    def covar_matrix(self):
        return torch.kron(self.V, self.U)


class HetergeneousMatrixVariateKernel:
    def __init__(self, task_covar_module, lengthscale, outputscale, decoder):
        """task_covar_module: MatrixVariateIndexKernel(A, B); lengthscale[n], outputscale: the ARD-RBF data kernel
        ScaleKernel(RBFKernel(ard_num_dims=n)) of the reference's model (control_affine_model.py:164-171);
        decoder: CatEncoder(1, n, 1+m)."""
        self.task_covar_module, self.decoder = task_covar_module, decoder
        self.lengthscale, self.outputscale = lengthscale, outputscale

    @property
    def num_tasks(self):
        n, C = self.task_covar_module.matshape
        return n * C

    @staticmethod
    def _split(M):
        """Rows are sorted: mask-1 rows first (:137-150)."""
        Ms = M[..., 0]
        idx = torch.nonzero(Ms - torch.ones_like(Ms))
        end = int(idx.min()) if idx.numel() else Ms.shape[-1]
        assert bool((Ms[end:] == 0).all()), "mask-1 (train) rows must precede mask-0 (test) rows"
        return end

    def _expand(self, mxu):
        """(X'[R,n], UH'[R,C], rows per input) of the expanded row set."""
        M, X, UH = self.decoder.decode(mxu)
        end = self._split(M)
        C = UH.shape[-1]
        eye = torch.eye(C, dtype=X.dtype, device=X.device)
        Xe = torch.cat([X[:end], X[end:].repeat_interleave(C, dim=0)])
        UHe = torch.cat([UH[:end], eye.repeat(X.shape[0] - end, 1)])
        return Xe.contiguous(), UHe.contiguous(), end

    def num_outputs_per_input(self, mxu1, mxu2):
        M1, X1, _ = self.decoder.decode(mxu1)
        end = self._split(M1)
        n = X1.shape[-1]
        return (end * n + (M1.shape[-2] - end) * self.num_tasks) / M1.shape[-2]

    def forward(self, mxu1, mxu2, diag=False, **params):
        """Dense covariance between the outputs of the rows of mxu1 and of mxu2: train-type rows contribute n outputs
        each, test-type rows (1+m) n, in the reference's order (row, [control,] state dimension)."""
        assert not torch.isnan(mxu1).any() and not torch.isnan(mxu2).any()
        if not mxu1.is_cuda:
            raise RuntimeError("the kernel blocks are built by libbcbf on a ROCm device (no CPU path)")
        A, B = self.task_covar_module.U.to(mxu1), self.task_covar_module.V.to(mxu1)
        X1, UH1, _ = self._expand(mxu1)
        X2, UH2, _ = self._expand(mxu2)
        R1 = X1.shape[0]
        X = torch.cat([X1, X2])[None].contiguous()
        UH = torch.cat([UH1, UH2])[None].contiguous()
        f = dict(dtype=mxu1.dtype, device=mxu1.device)           # (as_tensor of a Python float alone would be fp32)
        ell = torch.as_tensor(self.lengthscale, **f).reshape(1, -1).contiguous()
        s2 = torch.as_tensor(self.outputscale, **f).reshape(1).contiguous()
        Kb = ops.kb_build(X, UH, B[None].contiguous(), ell, s2)[0]           # k(x, x') (uh' B uh') on the stacked rows
        res = torch.kron(Kb[:R1, R1:].contiguous(), A)                      # (.) (x) A
        return res.diagonal() if diag else res

    __call__ = forward
