"""csrc/geev_small.h -- the n <= 4 port of LAPACK's general eigen-solver path that the device uses for the reference's
Hessian clean-up (gp_algebra.py:384-392) -- compiled for the HOST (tests/geev_host.cpp, g++) and compared with the
LAPACK builds on this machine: torch's (the one the reference runs on here) and numpy's.

What is asserted, and why not more: `eigenvectors.T @ diag(evalz) @ eigenvectors` depends on the ORDER of xGEEV's
eigenvalues and the SIGN of each eigenvector.  Up to n = 2 no QR sweep is involved and every implementation agrees in
every case.  For an active 3x3 / 4x4 block the number of sweeps before a deflation hinges on rounding-level residues,
one sweep more flips the sign of two Schur vectors -- torch's MKL and numpy's OpenBLAS disagree with EACH OTHER in ~2 % of
such matrices, so no third implementation can agree with both.  There the port must agree in the large majority and every
disagreement must be exactly a sign pattern D H D, D = diag(+-1) (same eigenvalues, same order, flipped eigenvectors)."""
import ctypes
import itertools
import os
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
PD = ctypes.POINTER(ctypes.c_double)


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("geev") / "geev_host.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "geev_host.cpp")], check=True)
    lib = ctypes.CDLL(so)
    lib.geev_small_clean.argtypes = [ctypes.c_int, PD, ctypes.c_double, ctypes.c_int]
    lib.geev_small_eig.argtypes = [ctypes.c_int, PD, PD, PD]

    class H:
        @staticmethod
        def clean(M, eps=2e-3, mode=0):
            n = M.shape[0]
            buf = np.ascontiguousarray(M, dtype=np.float64).copy()
            rc = lib.geev_small_clean(n, buf.ctypes.data_as(PD), eps, mode)
            return rc, buf

        @staticmethod
        def eig(M):
            n = M.shape[0]
            A = np.ascontiguousarray(M, dtype=np.float64).copy()
            w, V = np.zeros(n), np.zeros((n, n))
            rc = lib.geev_small_eig(n, A.ctypes.data_as(PD), w.ctypes.data_as(PD), V.ctypes.data_as(PD))
            return rc, w, V
    return H


def literal(M, eig, eps=2e-3):
    """gp_algebra.py:384-392 with a given general eigen-solver `eig(M) -> (w, V)`."""
    w, V = eig(M)
    ev = np.real(w).copy()
    assert (ev > -eps).all()
    small = (ev > -eps) & (ev < 0)
    if not small.any():
        return False, M
    ev[small] = 0.0
    V = np.real(V)
    return True, V.T @ np.diag(ev) @ V


def torch_eig(M):
    w, V = torch.linalg.eig(torch.from_numpy(np.array(M, dtype=np.float64)))
    return w.numpy(), V.numpy()


def sign_pattern_explains(H, Href, tol):
    n = H.shape[0]
    return any(np.abs(np.diag(d) @ Href @ np.diag(d) - H).max() <= tol for d in itertools.product([1.0, -1.0], repeat=n))


def test_port_against_the_executed_reference_fixture(host):
    """tests/golden/hessclean_handmade.npz (GradientGP.knl of the executed reference on 96 hand-made Hessians): status and
    result for every n <= 2 case and every n >= 3 case on which the two LAPACK builds agree; sign pattern otherwise."""
    g = np.load(os.path.join(GOLDEN, "hessclean_handmade.npz"))
    same = explained = 0
    for n in (1, 2, 3, 4):
        for M, ref, fired, stable in zip(g["M_n%d" % n], g["t_knl_n%d" % n], g["branch_fired_n%d" % n], g["agree_openblas_n%d" % n]):
            rc, H = host.clean(M)
            assert (rc in (4, 6)) == bool(fired), (n, rc, fired)
            tol = 1e-10 * max(1.0, np.abs(M).max())
            if np.abs(H - ref).max() <= tol:
                same += 1
                continue
            assert n >= 3 and sign_pattern_explains(H, ref, tol), (n, M, H, ref)
            explained += 1
    assert same >= 90 and same + explained == 96, (same, explained)


@pytest.mark.parametrize("n", [1, 2, 3, 4])
def test_port_against_both_lapack_builds_on_random_near_psd_matrices(host, n):
    rng = np.random.RandomState(100 + n)
    cnt = dict(same_torch=0, same_numpy=0, explained=0, builds_disagree=0, fired=0, total=0)
    for t in range(1500):
        A = rng.randn(n, n)
        w, V = np.linalg.eigh(A + A.T)
        w = np.abs(w) + 0.02 * (1 + np.arange(n))                 # distinct eigenvalues: eigenvectors well defined
        if t % 4:
            w[0] = -rng.uniform(1e-6, 1.8e-3)
        M = (V * w) @ V.T
        M = 0.5 * (M + M.T)
        if t % 5 == 1:
            M = M + 1e-16 * rng.randn(n, n)                        # the asymmetry autograd leaves behind
        if t % 3 == 0 and n > 1:
            z = rng.randint(n)
            M[z, :] = 0.0
            M[:, z] = 0.0
            w2 = np.linalg.eigvalsh(M)
            if np.min(np.diff(w2)) < 1e-6 or w2[0] <= -2e-3:      # a repeated (zero) eigenvalue: skip, nothing is defined
                continue
        cnt["total"] += 1
        f_t, H_t = literal(M, torch_eig)
        f_n, H_n = literal(M, np.linalg.eig)
        rc, H = host.clean(M)
        assert f_t == f_n == (rc in (4, 6)), (M, rc)
        if not f_t:
            assert rc == 0 and np.array_equal(H, M)
            continue
        cnt["fired"] += 1
        tol = 1e-9 * max(1.0, np.abs(M).max())
        a, b = np.abs(H - H_t).max() <= tol, np.abs(H - H_n).max() <= tol
        cnt["same_torch"] += int(a)
        cnt["same_numpy"] += int(b)
        cnt["builds_disagree"] += int(np.abs(H_t - H_n).max() > tol)
        if not a:
            assert n >= 3, "n <= 2 has no QR sweep: must agree always"
            assert sign_pattern_explains(H, H_t, tol), (M, H, H_t)
            cnt["explained"] += 1
        # the projection is a different matrix whenever n > 1 (the switch is not a no-op)
    print(n, cnt)
    assert cnt["fired"] > 700
    if n <= 2:
        assert cnt["same_torch"] == cnt["same_numpy"] == cnt["fired"] and cnt["builds_disagree"] == 0
    else:
        assert cnt["same_torch"] >= 0.93 * cnt["fired"] and cnt["same_numpy"] >= 0.93 * cnt["fired"], cnt


def test_eigenpairs_are_eigenpairs_and_status_codes(host):
    rng = np.random.RandomState(7)
    for n in (1, 2, 3, 4):
        for _ in range(200):
            A = rng.randn(n, n)
            M = A + A.T
            rc, w, V = host.eig(M)
            assert rc == 0
            np.testing.assert_allclose(M @ V, V * w, rtol=0, atol=1e-12 * max(1.0, np.abs(M).max()) * 50)
            np.testing.assert_allclose(np.linalg.norm(V, axis=0), 1.0, rtol=1e-13)
            np.testing.assert_allclose(np.sort(w), np.linalg.eigvalsh(M), rtol=0, atol=1e-12 * np.abs(M).max() * 20)
    M = np.diag([1.0, -5e-3, 2.0])
    rc, H = host.clean(M)
    assert rc == 1 and np.array_equal(H, M)                        # the reference's assert: nothing is touched
    rc, H = host.clean(np.diag([1.0, -5e-4, 2.0]), mode=1)
    assert rc == 4 and np.allclose(H, np.diag([1.0, 0.0, 2.0]))
    rot = np.array([[0.0, 1.0], [-1.0, 0.0]])                      # complex pair: the general solver's path is refused
    assert host.eig(rot)[0] == 2


@pytest.mark.parametrize("n", [2, 3, 4])
def test_branch_decision_on_the_boundaries_follows_the_general_eigenvalues(host, n):
    """Which branch runs (status 0 untouched / 4 cleaned / 1 the reference's assert) is decided from the real parts of the
    GENERAL eigen-solver's eigenvalues of H itself, as gp_algebra.py:385-387 does -- also for an H that is non-symmetric by
    rounding, right next to the -EPS assert boundary and with an eigenvalue within 1e-15 of zero (where the eigenvalues of the
    symmetric part can land on the other side).  Compared with the reference's statements on torch.linalg.eig; cases on which
    torch's and numpy's LAPACK builds themselves put an eigenvalue on different sides are skipped (nothing is defined)."""
    rng = np.random.RandomState(900 + n)
    EPS = 2e-3
    checked = dict(assert_=0, fired=0, untouched=0, skipped=0)
    for t in range(1200):
        Q, _ = np.linalg.qr(rng.randn(n, n))
        w = np.abs(rng.randn(n)) + 0.05 * (1 + np.arange(n))
        kind = t % 6
        if kind == 0:
            w[0] = -EPS * (1 + rng.uniform(1e-9, 1e-3))            # just beyond the assert boundary
        elif kind == 1:
            w[0] = -EPS * (1 - rng.uniform(1e-9, 1e-3))            # just inside it: cleaned
        elif kind == 2:
            w[0] = rng.uniform(-1e-15, 1e-15)                       # an eigenvalue that rounds to either side of zero
        elif kind == 3:
            w[0] = -rng.uniform(1e-14, 1e-12)
        elif kind == 4:
            w[0] = rng.uniform(1e-14, 1e-12)
        else:
            w[0] = -rng.uniform(1e-6, 1.9e-3)
        M = (Q * w) @ Q.T
        if t % 2:
            M = M + 1e-16 * np.abs(M).max() * rng.randn(n, n)      # asymmetric by rounding (what autograd leaves)
        ev_t = np.real(torch_eig(M)[0])
        ev_n = np.real(np.linalg.eig(M)[0])
        side = lambda ev: ((ev <= -EPS).any(), (ev < 0).any())
        if side(ev_t) != side(ev_n):
            checked["skipped"] += 1
            continue
        rc, H = host.clean(M)
        bad, neg = side(ev_t)
        # (any solver's eigenvalues carry a rounding error of a few eps |M|: a case whose deciding eigenvalue is within
        #  5e-16 |M| of the boundary may legitimately land on the other side -- only then is a disagreement accepted; the
        #  1e-14 .. 1e-12 kinds and both sides of -EPS at relative distance >= 1e-9 are asserted strictly)
        margin = min(np.abs(ev_t + EPS).min(), np.abs(ev_t).min())
        if bad:
            ok = rc == 1 and np.array_equal(H, M)
            checked["assert_"] += 1
        elif neg:
            ok = rc in (4, 6)
            checked["fired"] += 1
        else:
            ok = rc == 0 and np.array_equal(H, M)
            checked["untouched"] += 1
        assert ok or margin < 5e-16 * max(1.0, np.abs(M).max()), (kind, rc, ev_t, M)
    print(n, checked)
    assert checked["assert_"] > 100 and checked["fired"] > 300 and checked["untouched"] > 100


def test_non_finite_input_terminates_with_the_assert_status(host):
    """A NaN / Inf entry: the reference's eigenvalues are NaN and its assert fails -> status 1, matrix untouched -- and the
    port RETURNS (xGEBAL's balancing loop spins forever on NaN: every comparison is false; on the device that is a hung GPU)."""
    import multiprocessing as mp
    for bad in (np.nan, np.inf, -np.inf):
        for n in (1, 2, 3, 4):
            M = np.eye(n) + 0.1
            M[n - 1, 0] = bad
            M[0, n - 1] = bad
            rc, H = host.clean(M)
            assert rc == 1
            np.testing.assert_array_equal(H, M)
            rc, H = host.clean(M, mode=1)
            assert rc == 1
            assert host.eig(M)[0] in (0, 1, 2)        # (whatever it reports, it comes back)
