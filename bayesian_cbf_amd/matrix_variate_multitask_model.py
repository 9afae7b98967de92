"""bayes_cbf/matrix_variate_multitask_model.py: the constant matrix-variate prior mean of F(x) = [f(x) g(x)].

`HetergeneousMatrixVariateMean.forward(MXU)` (:44-66) serves the two related processes with one object: rows whose mask
column is 1 are observations  xdot = F(x)' uh  and get  M0' uh  (n numbers per row); rows with mask 0 are the matrix
F(x)' itself and get vec(M0) ((1+m) n numbers per row); mask-1 rows come first.  The hot path never calls this class --
`bcbf_potrs` forms  Y = Xdot - UH M0  on the device and `bcbf_posterior_*` add M0' back -- it is the reference's
container for code that builds or reads MXU rows, with the reference's `state_dict` keys (:68-76).

Upstream's `custom_predict` hands it raw states X [b, n] WITHOUT the mask column (control_affine_model.py:485, 527, 1033);
the decoder then reads column 0 of x as the mask, finds it different from 1 and returns vec(M0) per row -- which is
what those call sites reshape.  That behaviour falls out of the same decode-and-split here, including its edge: a
batch whose first rows have x[0] == 1.0 EXACTLY is taken for observation rows and fails the sortedness assertion
(or, if every row has it, the mean1 product on an empty UH), as upstream."""
import copy

import torch


def prod(L):
    """matrix_variate_multitask_kernel.py `prod`."""
    out = 1
    for v in L:
        out *= v
    return out


class ConstantMean(torch.nn.Module):
    """gpytorch.means.ConstantMean (0.3.x): one learnable scalar, broadcast over the batch."""

    def __init__(self, dtype=None):
        super().__init__()
        self.constant = torch.nn.Parameter(torch.zeros(1, dtype=dtype or torch.get_default_dtype()))

    def forward(self, x):
        return self.constant.expand(x.shape[:-1])


class SharedConstantMeans:
    """`num_tasks` constant means that read entry t of ONE parameter vector (the façade keeps the (1+m) n mean constants
    of a regressor in a single tensor, control_affine_model.KernelParams.mean_constants)."""

    def __init__(self, getter, num_tasks):
        self.getter, self.num_tasks = getter, num_tasks

    def __len__(self):
        return self.num_tasks

    def __iter__(self):
        for t in range(self.num_tasks):
            yield (lambda x, t=t: self.getter()[t].to(x).expand(x.shape[:-1]))


class HetergeneousMatrixVariateMean(torch.nn.Module):
    """mean_module: one mean (replicated prod(matshape) times, as gpytorch's MultitaskMean does), a list of
    prod(matshape) means, or a `SharedConstantMeans`.  decoder: CatEncoder(1, n, 1+m).  matshape = (1+m, n)."""

    def __init__(self, mean_module, decoder, matshape, **kwargs):
        super().__init__()
        num_tasks = prod(matshape)
        if isinstance(mean_module, SharedConstantMeans):
            object.__setattr__(self, "base_means", mean_module)
        else:
            means = list(mean_module) if isinstance(mean_module, (list, tuple)) else [mean_module]
            if len(means) == 1:
                means = means + [copy.deepcopy(means[0]) for _ in range(num_tasks - 1)]
            if len(means) != num_tasks:
                raise RuntimeError("base_means should be a list of means of length either 1 or num_tasks")
            self.base_means = torch.nn.ModuleList(means)
        self.num_tasks = num_tasks
        self.decoder = decoder
        self.matshape = tuple(matshape)

    def mean1(self, UH, mu):
        """Observation rows: uh' M0 per row, flattened [D n]."""
        return (UH.unsqueeze(-2) @ mu).reshape(-1)

    def mean2(self, mu):
        """Matrix rows: vec(M0) per row, flattened [D (1+m) n]."""
        return mu.reshape(-1)

    def forward(self, MXU):
        assert not torch.isnan(MXU).any()
        Ms, _, UH = self.decoder.decode(MXU)
        assert Ms.size(-1) == 1
        Ms = Ms[..., 0]
        rows = Ms.size(-1)
        other = torch.nonzero(Ms != 1)
        first0 = int(other.min()) if other.numel() else rows           # observation rows come first
        mu = torch.stack([sub(MXU) for sub in self.base_means], dim=-1)
        assert not torch.isnan(mu).any()
        mu = mu.reshape(-1, *self.matshape)
        pieces = []
        if first0 != 0:
            assert (Ms[..., first0:] == 0).all(), "mask column must be sorted: observation rows (1) before matrix rows (0)"
            pieces.append(self.mean1(UH[..., :first0, :], mu[:first0]))
        if first0 != rows:
            pieces.append(self.mean2(mu[first0:]))
        return pieces[0] if len(pieces) == 1 else torch.cat(pieces)

    def state_dict(self, *args, **kwargs):
        return dict(matshape=self.matshape, decoder=self.decoder.state_dict())

    def load_state_dict(self, state_dict, *args, **kwargs):
        self.matshape = tuple(state_dict.pop("matshape"))
        self.decoder.load_state_dict(state_dict["decoder"])
