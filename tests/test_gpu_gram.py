"""GPU tests of bcbf_gram (the Gram of whitened cross-covariances on the matrix cores: the reference's `v.t() @ vp` and
`kb_star' Bdagger`, control_affine_model.py:586, 1079-1088) and bcbf_predict_fullmat (query -> Gram -> assembly in one host call,
custom_predict_fullmat :963-980).  The façade paths that use them are held to the reference's golden vectors in test_gpu_facade.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-13), (torch.float32, 3e-6)], ids=["f64", "f32"])
def test_gram_equals_the_plain_contraction(dtype, tol):
    from bayesian_cbf_amd import ops
    g = torch.Generator(device=DEV).manual_seed(3)
    for b, bp, Np, C in ((400, 400, 512, 2), (37, 5, 96, 3), (1, 1, 32, 1), (70, 33, 160, 1), (9, 11, 64, 9), (20, 20, 1056, 4)):
        W = torch.randn(b, Np, C, dtype=dtype, device=DEV, generator=g)
        Wp = torch.randn(bp, Np, C, dtype=dtype, device=DEV, generator=g)
        want = torch.einsum("bkc,pkd->bpcd", W.double(), Wp.double())
        got = ops.gram(W, Wp)
        assert got.shape == (b, bp, C, C)
        assert float((got.double() - want).abs().max() / want.abs().max()) < tol * Np ** 0.5, (b, bp, Np, C)
        if b == bp:                                              # the symmetric form: half the tiles, mirrored
            want_s = torch.einsum("bkc,pkd->bpcd", W.double(), W.double())
            got_s = ops.gram(W)
            assert float((got_s.double() - want_s).abs().max() / want_s.abs().max()) < tol * Np ** 0.5
            assert torch.equal(got_s, got_s.permute(1, 0, 3, 2))     # exactly symmetric: the mirror tile is the same accumulator
    with pytest.raises(ValueError):
        ops.gram(torch.zeros(2, 32, 2, dtype=dtype, device=DEV), torch.zeros(2, 64, 2, dtype=dtype, device=DEV))


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 1e-4)], ids=["f64", "f32"])
@pytest.mark.parametrize("kernel", ["rbf", "matern52"])
def test_predict_fullmat_is_query_gram_assemble(dtype, tol, kernel):
    """One host call == the three entry points called one by one (same launches, same buffers' contents): Mk, BkXX and the
    Kronecker form, with and without the make_psd jitter."""
    from bayesian_cbf_amd import ops
    from bayesian_cbf_amd.synthetic import make_instances
    N, n, m, b = 200, 2, 1, 45
    p = make_instances(1, N, n, m, dtype=dtype, device=DEV, seed=8)
    Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"], kernel=kernel)
    assert int(info[0]) == 0
    Vw, _ = ops.potrs(Lop, p["Xdot"], p["UH"], p["M0"], want_alpha=False)
    g = torch.Generator(device=DEV).manual_seed(1)
    Xq = (torch.rand(b, n, dtype=dtype, device=DEV, generator=g) * 2 - 2).contiguous()
    jit = (1e-5 * torch.rand(b * (1 + m), dtype=dtype, device=DEV, generator=g)).contiguous()
    A = p["A"][0].contiguous()
    for j in (None, jit):
        Mk, BkXX, Kron = ops.predict_fullmat(Lop, Vw, p["X"], UHB, p["ell"], p["s2"], p["Bm"], p["M0"], A, Xq, j, want_BkXX=True, kernel=kernel)
        Mk2, _, W = ops.posterior_query(Lop, Vw, p["X"], UHB, p["ell"], p["s2"], p["Bm"], p["M0"], Xq, shared=True, want_W=True, kernel=kernel)
        G = torch.einsum("bkc,pkd->bpcd", W, W).contiguous()
        BkXX2, Kron2 = ops.predict_assemble(G, Xq, Xq, p["ell"].reshape(-1), p["s2"], p["Bm"][0].contiguous(), A, j, want_BkXX=True, want_kron=True,
                                            kernel=kernel)
        scale = float((p["s2"][0] * p["Bm"][0].abs().max()))
        assert torch.equal(Mk, Mk2)
        assert float((BkXX - BkXX2).abs().max()) <= tol * scale
        assert float((Kron - Kron2).abs().max()) <= tol * scale * float(A.abs().max())
