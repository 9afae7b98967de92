"""Thin host wrappers over the C ABI: torch tensors in, torch tensors out, on the caller's HIP
stream.  PyTorch is only the allocator / stream provider here; all arithmetic is in libbcbf."""
import ctypes

import os

import torch

from . import _lib
from ._lib import lib, check

_SUF = {torch.float32: "_f32", torch.float64: "_f64"}


def _suf(t):
    try:
        return _SUF[t.dtype]
    except KeyError:
        raise TypeError("libbcbf supports float32 / float64 tensors, got %s" % t.dtype)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _chk(*ts):
    ref = ts[0]
    if not ref.is_cuda:
        raise RuntimeError("libbcbf kernels need ROCm device tensors (got a CPU tensor); there is no CPU path")
    for t in ts:
        if t is None:
            continue
        if t.device != ref.device:
            raise RuntimeError("all tensors must live on the same device")
        if t.dtype not in (ref.dtype, torch.int32):
            raise TypeError("mixed dtypes: %s vs %s" % (t.dtype, ref.dtype))
        if not t.is_contiguous():
            raise RuntimeError("libbcbf needs contiguous row-major tensors")


def lop_elems(N, dtype):
    return int(getattr(lib, "bcbf_lop_elems" + _SUF[dtype])(N))


def hbm_read_probe(buf, launches=10, warm=2):
    """Measured read-only HBM ceiling of this device in GB/s over the caller's (large, resident) buffer `buf`:
    `launches` timed launches of `bcbf_hbm_read_probe` (16-byte non-temporal loads, nothing written), each between its own
    pair of HIP events on the current stream.  Returns dict(best_gbs, mean_gbs, bytes, launches)."""
    if not buf.is_cuda or not buf.is_contiguous():
        raise RuntimeError("hbm_read_probe needs a contiguous device buffer")
    nbytes = buf.numel() * buf.element_size()
    sink = torch.zeros(256, dtype=torch.float32, device=buf.device)
    got = ctypes.c_size_t(0)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
    for _ in range(warm):
        check(lib.bcbf_hbm_read_probe(_p(buf), nbytes, _p(sink), ctypes.byref(got), _stream(buf)), "bcbf_hbm_read_probe")
    for e0, e1 in evs:
        e0.record()
        check(lib.bcbf_hbm_read_probe(_p(buf), nbytes, _p(sink), ctypes.byref(got), _stream(buf)), "bcbf_hbm_read_probe")
        e1.record()
    torch.cuda.synchronize(buf.device)
    ms = [e0.elapsed_time(e1) for e0, e1 in evs]
    return dict(best_gbs=got.value / (min(ms) * 1e-3) / 1e9, mean_gbs=got.value / (sum(ms) / len(ms) * 1e-3) / 1e9,
                bytes=int(got.value), launches=launches)


DATA_KERNELS = ("rbf", "matern52", "rbf_matern52")       # (index = the library's kernel_kind; the last two are opt-in)
_KSUF = {"rbf": "", "matern52": "_matern52", "rbf_matern52": "_rbfm52"}    # entry-point suffix of a data kernel


def kb_build(X, UH, Bm, ell, s2, jitter=None, lin=None, kernel="rbf"):
    """K_b[Bt,N,N]  (control_affine_model.py:370-372 + make_psd diagonal :907-910); lin[Bt] adds the linear part of
    the CoGP comparator's data kernel, k = s2 (exp(..) + lin x'x') (:1121-1122).  kernel="matern52": the opt-in
    Matern-5/2 data kernel (bcbf.h; parity unpinned -- the reference has no Matern kernel)."""
    _chk(X, UH, Bm, ell, s2, jitter, lin)
    Bt, N, n = X.shape
    m = UH.shape[2] - 1
    Kb = torch.empty(Bt, N, N, dtype=X.dtype, device=X.device)
    if kernel not in DATA_KERNELS:
        raise ValueError("data kernel %r: one of %s" % (kernel, DATA_KERNELS))
    if kernel != "rbf":
        if lin is not None:
            raise ValueError("the opt-in data kernels have no linear part")
        check(getattr(lib, "bcbf_kb_build" + _KSUF[kernel] + _suf(X))(_p(X), _p(UH), _p(Bm), _p(ell), _p(s2), _p(jitter), _p(Kb),
                                                                      Bt, N, n, m, _stream(X)), "bcbf_kb_build" + _KSUF[kernel])
        return Kb
    if lin is not None:
        check(getattr(lib, "bcbf_kb_build_rbflin" + _suf(X))(_p(X), _p(UH), _p(Bm), _p(ell), _p(s2), _p(lin), _p(jitter),
                                                             _p(Kb), Bt, N, n, m, _stream(X)), "bcbf_kb_build_rbflin")
        return Kb
    check(getattr(lib, "bcbf_kb_build" + _suf(X))(_p(X), _p(UH), _p(Bm), _p(ell), _p(s2), _p(jitter), _p(Kb),
                                                  Bt, N, n, m, _stream(X)), "bcbf_kb_build")
    return Kb


def _kern(base, kernel):
    if kernel not in DATA_KERNELS:
        raise ValueError("kernel %r: one of %s" % (kernel, DATA_KERNELS))
    return base + _KSUF[kernel]


def refit(X, UH, Bm, ell, s2, jitter=None, want_dense=False, out=None, kernel="rbf"):
    """Fused K_b build + Cholesky + packing.  Returns (Lop[Bt,E], UHB[Bt,N,C], info[Bt], Ldense|None).
    out = (Lop, UHB, info): write into the caller's buffers (a closed loop that has bound their addresses).
    kernel="matern52": the opt-in Matern-5/2 data kernel (bcbf_refit_matern52)."""
    _chk(X, UH, Bm, ell, s2, jitter)
    Bt, N, n = X.shape
    C = UH.shape[2]
    if out is not None:
        Lop, UHB, info = out
        _chk(X, Lop, UHB, info)
        assert Lop.shape == (Bt, lop_elems(N, X.dtype)) and UHB.shape == (Bt, N, C) and info.shape == (Bt,)
    else:
        Lop = torch.empty(Bt, lop_elems(N, X.dtype), dtype=X.dtype, device=X.device)
        UHB = torch.empty(Bt, N, C, dtype=X.dtype, device=X.device)
        info = torch.empty(Bt, dtype=torch.int32, device=X.device)
    Ld = torch.empty(Bt, N, N, dtype=X.dtype, device=X.device) if want_dense else None
    check(getattr(lib, _kern("bcbf_refit", kernel) + _suf(X))(_p(X), _p(UH), _p(Bm), _p(ell), _p(s2), _p(jitter), _p(Lop), _p(UHB),
                                                              _p(Ld), _p(info), Bt, N, n, C - 1, _stream(X)), "bcbf_refit")
    return Lop, UHB, info, Ld


def refit_retry(X, UH, Bm, ell, s2, jitter, Lop, UHB, prev_info, info, kernel="rbf"):
    """Factor again ONLY the instances with prev_info[b] != 0 (bcbf_refit_retry; no host round trip), with the jitter the
    caller has raised for them; writes `info` (a different buffer than prev_info)."""
    _chk(X, UH, Bm, ell, s2, jitter, Lop, UHB, prev_info, info)
    Bt, N, n = X.shape
    if kernel != "rbf":
        _kern("", kernel)
        check(getattr(lib, "bcbf_refit_retry_kind" + _suf(X))(_p(X), _p(UH), _p(Bm), _p(ell), _p(s2), _p(jitter), _p(Lop), _p(UHB),
                                                              _p(prev_info), _p(info), Bt, N, n, UH.shape[2] - 1,
                                                              DATA_KERNELS.index(kernel), _stream(X)), "bcbf_refit_retry_kind")
        return info
    check(getattr(lib, "bcbf_refit_retry" + _suf(X))(_p(X), _p(UH), _p(Bm), _p(ell), _p(s2), _p(jitter), _p(Lop), _p(UHB),
                                                     _p(prev_info), _p(info), Bt, N, n, UH.shape[2] - 1, _stream(X)), "bcbf_refit_retry")
    return info


def refit_with_retries(X, UH, Bm, ell, s2, jitter, out, levels=3, scratch=None, level=None, counts=None, kernel="rbf"):
    """bcbf_refit followed by `levels` unconditional retry launches (x10 jitter on the instances that failed: make_psd's
    schedule, control_affine_model.py:903-919) -- nothing waits for the host.  out = (Lop, UHB, info); `jitter` is raised IN
    PLACE for the failed instances (it is the record of what every point was factored with), and so is `level[Bt]` (the
    instance's jitter level, kept by callers that start the next refit near the level that worked).  counts[levels + 1]
    (optional, int64 on the device): += the instances each launch had to factor.  Returns info (0, or the pivot of an instance
    that failed every level)."""
    Lop, UHB, info = out
    refit(X, UH, Bm, ell, s2, jitter, out=(Lop, UHB, info), kernel=kernel)
    if counts is not None:
        counts[0] += X.shape[0]
    other = torch.empty_like(info) if scratch is None else scratch
    cur, nxt = info, other
    for k in range(levels):
        fac = torch.where(cur != 0, 10.0, 1.0).to(jitter.dtype)
        jitter.mul_(fac[:, None])
        if level is not None:
            level.mul_(fac)
        if counts is not None:
            counts[k + 1] += (cur != 0).sum()
        refit_retry(X, UH, Bm, ell, s2, jitter, Lop, UHB, cur, nxt, kernel=kernel)
        cur, nxt = nxt, cur
    if cur is not info:
        info.copy_(cur)
    return info


def gram(W, Wp=None, out=None):
    """G[b,bp,C,C] = einsum("bkc,pkd->bpcd", W, Wp) on the matrix cores (bcbf_gram); Wp None = W (symmetric: half the tiles)."""
    Wp = W if Wp is None else Wp
    _chk(W, Wp, out)
    b, Np, C = W.shape
    bp = Wp.shape[0]
    if Wp.shape[1] != Np or Wp.shape[2] != C:
        raise ValueError("gram: W %s vs Wp %s" % (tuple(W.shape), tuple(Wp.shape)))
    G = torch.empty(b, bp, C, C, dtype=W.dtype, device=W.device) if out is None else out
    check(getattr(lib, "bcbf_gram" + _suf(W))(_p(W), _p(Wp), _p(G), b, bp, Np, C, _stream(W)), "bcbf_gram")
    return G


def predict_fullmat(Lop, Vw, X, UHB, ell, s2, Bm, M0, A, Xq, jitter=None, want_BkXX=False, want_kron=True, kernel="rbf"):
    """query -> Gram -> assembly in ONE host call (bcbf_predict_fullmat).  Returns (Mk[b,n,C], BkXX | None, Kron | None)."""
    _chk(Lop, Vw, X, UHB, ell, s2, Bm, M0, A, Xq, jitter)
    if X.shape[0] != 1:
        raise ValueError("predict_fullmat: GP tensors of ONE model (leading axis 1)")
    N, n = X.shape[1], X.shape[2]
    C, b = UHB.shape[2], Xq.shape[0]
    f = dict(dtype=X.dtype, device=X.device)
    Np = (N + 31) // 32 * 32
    Mk, Bk, W, G = torch.empty(b, n, C, **f), torch.empty(b, C, C, **f), torch.empty(b, Np, C, **f), torch.empty(b, b, C, C, **f)
    BkXX = torch.empty(b, b, C, C, **f) if want_BkXX else None
    Kron = torch.empty(b * C * n, b * C * n, **f) if want_kron else None
    check(getattr(lib, "bcbf_predict_fullmat" + _suf(X))(_p(Lop), _p(Vw), _p(X), _p(UHB), _p(ell), _p(s2), _p(Bm), _p(M0), _p(A), _p(Xq),
                                                         _p(jitter), _p(Mk), _p(Bk), _p(W), _p(G), _p(BkXX), _p(Kron), b, N, n, C - 1,
                                                         DATA_KERNELS.index(kernel), _stream(X)), "bcbf_predict_fullmat")
    return Mk, BkXX, Kron


def potrf(Kb, want_dense=False):
    """Cholesky of caller-supplied SPD matrices (torch.linalg.cholesky, control_affine_model.py:911)."""
    _chk(Kb)
    Bt, N, _ = Kb.shape
    Lop = torch.empty(Bt, lop_elems(N, Kb.dtype), dtype=Kb.dtype, device=Kb.device)
    info = torch.empty(Bt, dtype=torch.int32, device=Kb.device)
    Ld = torch.empty(Bt, N, N, dtype=Kb.dtype, device=Kb.device) if want_dense else None
    check(getattr(lib, "bcbf_potrf" + _suf(Kb))(_p(Kb), _p(Lop), _p(Ld), _p(info), Bt, N, _stream(Kb)), "bcbf_potrf")
    return Lop, info, Ld


def potrs(Lop, Xdot, UH, M0, want_alpha=True, out_Vw=None):
    """Vw = L^-1 (Xdot - UH M0), alpha = K_b^-1 (Xdot - UH M0)  (control_affine_model.py:525-545)."""
    _chk(Lop, Xdot, UH, M0, out_Vw)
    Bt, N, n = Xdot.shape
    m = UH.shape[2] - 1
    Vw = torch.empty(Bt, N, n, dtype=Xdot.dtype, device=Xdot.device) if out_Vw is None else out_Vw
    assert Vw.shape == (Bt, N, n)
    alpha = torch.empty_like(Vw) if want_alpha else None
    check(getattr(lib, "bcbf_potrs" + _suf(Xdot))(_p(Lop), _p(Xdot), _p(UH), _p(M0), _p(Vw), _p(alpha),
                                                  Bt, N, n, m, _stream(Xdot)), "bcbf_potrs")
    return Vw, alpha


def chol_append(Lop, knew, kappa, N):
    _chk(Lop, knew, kappa)
    Bt = Lop.shape[0]
    out = torch.empty(Bt, lop_elems(N + 1, Lop.dtype), dtype=Lop.dtype, device=Lop.device)
    info = torch.empty(Bt, dtype=torch.int32, device=Lop.device)
    check(getattr(lib, "bcbf_chol_append" + _suf(Lop))(_p(Lop), _p(knew), _p(kappa), _p(out), _p(info), Bt, N,
                                                       _stream(Lop)), "bcbf_chol_append")
    return out, info


GP_APPEND_STREAM_MIN_N = int(os.environ.get("BCBF_APPEND_STREAM_MIN_N", "384"))


def gp_append(Lop, Vw, X, UHB, ell, s2, Bm, M0, x_new, uh_new, xdot_new, jitter_new=None, kernel="rbf"):
    """Online update: one observation per instance enters the GP without refactorisation.
    Returns (Lop', Vw'[Bt,N+1,n], X'[Bt,N+1,n], UHB'[Bt,N+1,C], info).  The operator is updated in place
    (the returned Lop' IS Lop) while N+1 stays inside the same 32-row padding, re-packed otherwise.
    From N = GP_APPEND_STREAM_MIN_N on the forward solve l = L^-1 k runs on the streaming posterior kernel
    (bcbf_gp_append_stream: W = L^-1 Phi(x_new) at the HBM roofline, l = W uh_new)."""
    _chk(Lop, Vw, X, UHB, ell, s2, Bm, M0, x_new, uh_new, xdot_new, jitter_new)
    Bt, N, n = X.shape
    C = UHB.shape[2]
    f = dict(dtype=X.dtype, device=X.device)
    same_pad = (N + 31) // 32 == (N + 32) // 32
    Lout = Lop if same_pad else torch.empty(Bt, lop_elems(N + 1, X.dtype), **f)
    Vw2, X2, UHB2 = torch.empty(Bt, N + 1, n, **f), torch.empty(Bt, N + 1, n, **f), torch.empty(Bt, N + 1, C, **f)
    info = torch.empty(Bt, dtype=torch.int32, device=X.device)
    if kernel != "rbf" and N >= GP_APPEND_STREAM_MIN_N:  # opt-in kernels, large N: the forward solve on that kind's streaming kernel
        _kern("bcbf_gp_append", kernel)
        Np = (N + 31) // 32 * 32
        Ww, Mkw, Bkw = torch.empty(Bt, Np, C, **f), torch.empty(Bt, n, C, **f), torch.empty(Bt, C, C, **f)
        check(getattr(lib, "bcbf_gp_append_stream_kind" + _suf(X))(
            _p(Lop), _p(Vw), _p(X), _p(UHB), _p(ell), _p(s2), _p(Bm), _p(M0), _p(x_new), _p(uh_new), _p(xdot_new),
            _p(jitter_new), _p(Lout), _p(Vw2), _p(X2), _p(UHB2), _p(info), _p(Ww), _p(Mkw), _p(Bkw), Bt, N, n, C - 1,
            DATA_KERNELS.index(kernel), _stream(X)), "bcbf_gp_append_stream_kind")
        return Lout, Vw2, X2, UHB2, info
    if kernel != "rbf":                                # opt-in kernels: the simple forward solve
        check(getattr(lib, _kern("bcbf_gp_append", kernel) + _suf(X))(
            _p(Lop), _p(Vw), _p(X), _p(UHB), _p(ell), _p(s2), _p(Bm), _p(M0), _p(x_new), _p(uh_new), _p(xdot_new),
            _p(jitter_new), _p(Lout), _p(Vw2), _p(X2), _p(UHB2), _p(info), Bt, N, n, C - 1, _stream(X)), "bcbf_gp_append")
        return Lout, Vw2, X2, UHB2, info
    if N >= GP_APPEND_STREAM_MIN_N:
        Np = (N + 31) // 32 * 32
        Ww, Mkw, Bkw = torch.empty(Bt, Np, C, **f), torch.empty(Bt, n, C, **f), torch.empty(Bt, C, C, **f)
        check(getattr(lib, "bcbf_gp_append_stream" + _suf(X))(
            _p(Lop), _p(Vw), _p(X), _p(UHB), _p(ell), _p(s2), _p(Bm), _p(M0), _p(x_new), _p(uh_new), _p(xdot_new),
            _p(jitter_new), _p(Lout), _p(Vw2), _p(X2), _p(UHB2), _p(info), _p(Ww), _p(Mkw), _p(Bkw), Bt, N, n, C - 1,
            _stream(X)), "bcbf_gp_append_stream")
        return Lout, Vw2, X2, UHB2, info
    check(getattr(lib, "bcbf_gp_append" + _suf(X))(_p(Lop), _p(Vw), _p(X), _p(UHB), _p(ell), _p(s2), _p(Bm), _p(M0),
                                                   _p(x_new), _p(uh_new), _p(xdot_new), _p(jitter_new), _p(Lout),
                                                   _p(Vw2), _p(X2), _p(UHB2), _p(info), Bt, N, n, C - 1, _stream(X)),
          "bcbf_gp_append")
    return Lout, Vw2, X2, UHB2, info


class ReservedGP:
    """Capacity-reserving storage of Bt independent GPs for the online path (bcbf.h: bcbf_gp_reserve,
    bcbf_posterior_query_reserved, bcbf_gp_append_reserved).  Built from a fitted state of N points
    (`refit` / `potrs` outputs); `append` enters one observation per instance IN PLACE -- no allocation, no copy of
    the per-instance arrays, no re-pack of the operator -- until N reaches `capacity`; `grow(capacity)` re-reserves.
    The work buffers of the forward solve are allocated once.  `N` is the live size; `Lop`, `Vw`, `X`, `UHB` are the
    reserved buffers ([Bt, lop_elems(capacity)], [Bt, capacity, .]).

    window = W (with the raw data of the initial points: UH, Xdot, jitter): a SLIDING WINDOW over the most recent points at
    the granularity of the packed layout's 32-row blocks (the reference keeps a random subsample of at most
    `max_train` points and refactorises from scratch, unicycle_move_to_pose.py:373-384, controllers.py:348-352).  The live
    size runs between W and W + 31; the append that would make it W + 32 first drops the OLDEST 32 points
    (`drop_oldest_block`): the remaining rows move up and the factor of the window is recomputed from the data by the
    batched refit kernel.  Why a refit and not the rank-32 update L22' L22'' = L22 L22' + L21 L21' of the trailing factor: the
    update is 32 dependent sweeps of Givens-type row operations over the whole trailing factor (2 * 32 N^2 flop = 17 MFLOP at
    N = 512, latency bound, plus re-inverting every diagonal block of the packed layout), the refit is N^3 / 3 = 45 MFLOP on
    the matrix cores at 18-50 TFLOP/s -- 0.5 ms for 256 windows of 512 points in fp64, once per 32 appends of 0.47 ms each --
    and it leaves EXACTLY the factor a from-scratch refit of the window gives (no accumulation over wrap-arounds).
    Needs capacity >= W + 32.  `drop` = k (default 32) forgets the k oldest points at a time instead (the live size then
    runs between W and W + k - 1, capacity >= W + k): the reference's refit cadence `train_every_n_steps` = 40
    (unicycle_move_to_pose.py:340-386) is `drop=40`.  What this is and is not: a REFIT SCHEDULE on in-place storage, not a
    down-date kernel -- between two drops the appends are true rank-one updates in place, the drop itself recomputes the
    window's factor from the data.
    A failed factorisation of the window (after `max_tries` jitter levels) is reported: `drop_info[Bt]` holds the last
    drop's per-instance pivot index, `drop_failures` counts instances over all drops, and the next `append` returns an
    `info` that carries it (OR-ed in as a negative value) so a caller that only watches `append`'s result sees it."""

    BLOCK = 32

    TAIL_MAX = 64            # tail rows the tail step holds (bcbf_gp_tail_step: tcap <= 64)

    def __init__(self, Lop, Vw, X, UHB, ell, s2, Bm, M0, capacity, A=None, window=None, UH=None, Xdot=None, jitter=None,
                 drop=None, tail=False, retry_levels=None, factor_dtype=None, min_jitter_level=1e-5, kernel="rbf"):
        """kernel: the data kernel of the state ("rbf"; opt-in "matern52", "rbf_matern52": the `*_kind` entry points of bcbf.h --
        queries, appends, the tail step and the window refits -- bcbf_refit_retry_kind with retry_levels -- evaluate it).
        factor_dtype (retry_levels mode; e.g. float64 for an fp32 model): the window refits factor in that precision and ROUND the
        operator / UH B / Vw into the buffers the passes read (the packed layout is the same element for element) -- the passes then
        add cond(L) eps, not cond(K_b) eps.  min_jitter_level: floor of the per-instance level (fp32 passes cannot resolve a posterior
        variance below ~ sqrt(cond K_b) eps32 of the prior: 1e-3 for fp32 passes on fp64 factors).
        retry_levels = k (window mode): the window refit of a drop never waits for the host -- bcbf_refit followed by k
        unconditional bcbf_refit_retry launches (x10 jitter on the instances that failed), into a second operator buffer that
        is swapped in (no allocation per drop).  `drop_info` then holds the last level's info; `drop_failures` is counted only
        when asked for (`count_drop_failures()`: one host read).  None: ten levels with a look at the device per level."""
        _chk(Lop, Vw, X, UHB, ell, s2, Bm, M0)
        _kern("", kernel)
        self.kernel, self._kind = kernel, DATA_KERNELS.index(kernel)
        self.retry_levels = retry_levels
        self.factor_dtype = factor_dtype if (factor_dtype is not None and factor_dtype != X.dtype) else None
        self.min_jitter_level = float(min_jitter_level)
        if self.factor_dtype is not None and retry_levels is None:
            raise ValueError("factor_dtype: mixed precision is built on the host-free window refit (retry_levels=...)")
        self._alt = None
        # retry_levels mode: the jitter LEVEL of every instance (make_psd starts at 1e-5 and goes x10 per failure; a window refit here
        # starts one level below the one that last worked) -- `jitter_level` is also what a caller scales a new point's draw with
        self.jitter_level = torch.full((X.shape[0],), float(min_jitter_level), dtype=X.dtype, device=X.device)
        self.retry_counts = torch.zeros((retry_levels or 0) + 1, dtype=torch.int64, device=X.device)
        self.Bt, self.N, self.n = X.shape
        self.C = UHB.shape[2]
        self.ell, self.s2, self.Bm, self.M0, self.A = ell, s2, Bm, M0, A
        self.capacity = 0
        self.window = None if window is None else int(window)
        self.drops = 0
        self.drop = self.BLOCK if drop is None else int(drop)
        self.drop_failures = 0
        self.drop_info = None
        self._fill(Lop, Vw, X, UHB, self.N, int(capacity))
        if self.window is not None:
            if UH is None or Xdot is None:
                raise ValueError("a sliding window refits from the data: pass UH and Xdot (and the jitter) of the initial points")
            if self.drop < 1 or self.capacity < self.window + self.drop or self.N > self.window + self.drop - 1:
                raise ValueError("window %d needs capacity >= %d and at most %d initial points" % (self.window, self.window + self.drop,
                                                                                                   self.window + self.drop - 1))
            _chk(UH, Xdot, jitter)
            f = dict(dtype=X.dtype, device=X.device)
            # raw rows of the live points (the reserved arrays hold derived quantities: UH B, L^-1 (Xdot - UH M0))
            self._rUH, self._rY = torch.zeros(self.Bt, self.capacity, self.C, **f), torch.zeros(self.Bt, self.capacity, self.n, **f)
            self._rJ = torch.zeros(self.Bt, self.capacity, **f)
            self._rUH[:, :self.N], self._rY[:, :self.N] = UH, Xdot
            if jitter is not None:
                self._rJ[:, :self.N] = jitter
        # tail=True (window mode): the points observed since the last window refit are kept as contiguous ROWS of a bordered factor
        # beside the window's own (bcbf_gp_tail_step) instead of being written into the operator's columns one element at a time
        self.tail = bool(tail)
        self.N0, self.t = self.N, 0
        if self.tail and self.window is None:
            # GROWTH with a tail (no window): the appends since the last commit are rows of a bordered factor beside the reserved
            # operator; every 32nd append commits them as one whole block row (bcbf_gp_tail_commit: full-line writes into the
            # column layout).  The live size then is N0 (committed, a multiple of 32) + t.
            if self.N % self.BLOCK or self.n > 4:
                raise ValueError("tail=True without a window starts from a multiple of %d points (and n <= 4)" % self.BLOCK)
            f = dict(dtype=X.dtype, device=X.device)
            Npc = (self.capacity + 31) // 32 * 32
            self._tcap = self.BLOCK
            self._Rb = torch.zeros(self.Bt, self._tcap, Npc, **f)
            self._Rinv = torch.zeros(self.Bt, self._tcap, self._tcap, **f)
            self._Wfull = torch.empty(self.Bt, Npc, self.C + 1, **f)
            self._sw = torch.empty(self.Bt, 1 + self.n, **f)
            self._ones = torch.ones(self.Bt, self.C, **f)
            self._Lcap = self.capacity
            self._rUH = self._rY = self._rJ = None
        elif self.tail:
            # rows the tail must hold: until the first window refit window + drop - N_init of them, `drop` after every refit
            # (N0 = window, t = 0 there) -- whichever is larger (N_init > window makes the first period the shorter one)
            self._tcap = max(self.drop, self.window + self.drop - self.N)
            if self._tcap > self.TAIL_MAX or self.n > 4:
                raise ValueError("tail=True holds at most %d points between two window refits (needs %d here) and n <= 4"
                                 % (self.TAIL_MAX, self._tcap))
            f = dict(dtype=X.dtype, device=X.device)
            Npc = (self.capacity + 31) // 32 * 32
            self._Rb = torch.zeros(self.Bt, self._tcap, Npc, **f)
            self._Rinv = torch.zeros(self.Bt, self._tcap, self._tcap, **f)
            self._Wfull = torch.empty(self.Bt, Npc, self.C + 1, **f)
            self._sw = torch.empty(self.Bt, 1 + self.n, **f)
            self._ones = torch.ones(self.Bt, self.C, **f)
            self._Lcap = self.capacity           # what the operator is laid out for: the reservation now, the window after a refit

    def _fill(self, Lop, Vw, X, UHB, N, capacity, cap_in=0, reuse=False):
        if capacity < N:
            raise ValueError("capacity %d < %d live points" % (capacity, N))
        f = dict(dtype=X.dtype, device=X.device)
        Bt, n, C = self.Bt, self.n, self.C
        if reuse:                                     # re-lay a packed state out INTO the buffers this object already owns
            Lr, Vr, Xr, Ur = self.Lop, self.Vw, self.X, self.UHB
        else:
            Lr = torch.empty(Bt, lop_elems(capacity, X.dtype), **f)
            Vr, Xr, Ur = torch.empty(Bt, capacity, n, **f), torch.empty(Bt, capacity, n, **f), torch.empty(Bt, capacity, C, **f)
        check(getattr(lib, "bcbf_gp_reserve" + _suf(X))(_p(Lop), _p(Vw), _p(X), _p(UHB), _p(Lr), _p(Vr), _p(Xr), _p(Ur), Bt, N,
                                                        cap_in, capacity, n, C - 1, _stream(X)), "bcbf_gp_reserve")
        if reuse:
            return
        self.Lop, self.Vw, self.X, self.UHB, self.capacity = Lr, Vr, Xr, Ur, capacity
        Npc = (capacity + 31) // 32 * 32
        self._Ww, self._Mkw, self._Bkw = torch.empty(Bt, Npc, C, **f), torch.empty(Bt, n, C, **f), torch.empty(Bt, C, C, **f)
        self.info = torch.empty(Bt, dtype=torch.int32, device=X.device)

    def grow(self, capacity):
        """Re-reserve for a larger capacity: ONE copy of the state (geometric growth amortises it to O(1) per append)."""
        if capacity <= self.capacity:
            return self
        if self.window is not None:
            raise RuntimeError("a windowed ReservedGP keeps its capacity (window + 32 suffices)")
        self._fill(self.Lop, self.Vw, self.X, self.UHB, self.N, int(capacity), cap_in=self.capacity)
        return self

    def drop_oldest_block(self, max_tries=10):
        """Window mode: forget the `drop` (default 32) oldest points.  The raw rows move up, the window's factor and whitened targets are
        recomputed from them (`refit` with the jitter every point ENTERED with -- an instance whose factorisation fails
        retries with its jitter x10, as make_psd does, control_affine_model.py:903-919) and laid out into the reserved
        buffers this object already owns.  Returns info[Bt] of the last factorisation (0 = fine)."""
        if self.window is None:
            raise RuntimeError("drop_oldest_block needs a ReservedGP built with window=...")
        k = self.drop
        N2 = self.N - k
        if N2 < 1:
            raise RuntimeError("nothing would be left")
        X = self.X[:, k:self.N].contiguous()
        UH, Y, J = self._rUH[:, k:self.N].contiguous(), self._rY[:, k:self.N].contiguous(), self._rJ[:, k:self.N].contiguous()
        if self.retry_levels is not None:
            # no host round trip, no allocation: the refit and its retries write the operator buffer that is NOT being read
            f = dict(dtype=X.dtype, device=X.device)
            if self._alt is None or self._alt[0].shape[1] != lop_elems(N2, X.dtype):
                self._alt = (torch.empty(self.Bt, lop_elems(N2, X.dtype), **f), torch.empty(self.Bt, N2, self.C, **f),
                             torch.empty(self.Bt, dtype=torch.int32, device=X.device), torch.empty(self.Bt, dtype=torch.int32, device=X.device))
            Lop, UHB, info, scratch = self._alt
            # a FRESH draw for every point of the window at the instance's level, as make_psd draws one per factorisation (:907-910)
            # (raising the points' stored jitter x10 per failed refit instead would compound from window to window)
            # (starting at the level that last worked; one level down every `level_decay_every`-th refit -- 1: make_psd's restart, one
            #  level below, at every refit)
            every = getattr(self, "level_decay_every", 1)
            if every and (self.drops + 1) % every == 0:
                self.jitter_level.div_(10).clamp_(min=self.min_jitter_level)
            J = (self.jitter_level[:, None] * torch.rand(self.Bt, N2, **f)).contiguous()
            Vw_wide = None
            if self.factor_dtype is not None:
                # factor in the wide precision on the rows cast up; round the results into the buffers the passes read
                wd = self.factor_dtype
                fw = dict(dtype=wd, device=X.device)
                if getattr(self, "_wide", None) is None or self._wide[0].shape[1] != lop_elems(N2, wd):
                    self._wide = (torch.empty(self.Bt, lop_elems(N2, wd), **fw), torch.empty(self.Bt, N2, self.C, **fw))
                up = lambda t_: t_.to(wd)
                Xd, UHd, Yd, Jd, lvl = up(X), up(UH), up(Y), up(J), up(self.jitter_level)
                refit_with_retries(Xd, UHd, up(self.Bm), up(self.ell), up(self.s2), Jd, (self._wide[0], self._wide[1], info),
                                   levels=self.retry_levels, scratch=scratch, level=lvl, counts=self.retry_counts, kernel=self.kernel)
                Vw_wide, _ = potrs(self._wide[0], Yd, UHd, up(self.M0), want_alpha=False)
                Lop.copy_(self._wide[0]); UHB.copy_(self._wide[1]); J.copy_(Jd); self.jitter_level.copy_(lvl)
            else:
                refit_with_retries(X, UH, self.Bm, self.ell, self.s2, J, (Lop, UHB, info), levels=self.retry_levels, scratch=scratch,
                                   level=self.jitter_level, counts=self.retry_counts, kernel=self.kernel)
            if self.tail:
                self._alt = (self.Lop if self.Lop.shape == Lop.shape else None, UHB, info, scratch)    # the buffer now read is the next drop's target
                if self._alt[0] is None:
                    self._alt = None
        else:
            for ntry in range(max_tries):
                Lop, UHB, info, _ = refit(X, UH, self.Bm, self.ell, self.s2, J, kernel=self.kernel)
                bad = info != 0
                if not bool(bad.any()):                   # (one host round trip per drop: ~30 us against the refit's milliseconds;
                    break                                 #  speculative jitter levels would cost two more refits per drop instead)
                if ntry + 1 < max_tries:
                    J = torch.where(bad[:, None], J * 10, J)
            else:
                # still failing after max_tries levels: the instance's window is laid out as the (garbage) factor the kernel left;
                # say so where callers look -- drop_info, the failure count, and the next append's info
                self.drop_failures += int(bad.sum())
        self.drop_info = info
        if self.retry_levels is not None and self.factor_dtype is not None:
            Vw = Vw_wide.to(X.dtype)
        else:
            Vw, _ = potrs(Lop, Y, UH, self.M0, want_alpha=False)
        self._rUH[:, :N2], self._rY[:, :N2], self._rJ[:, :N2] = UH, Y, J
        if self.tail:
            # the window's operator is only read until the next refit: the packed one the refit wrote serves as it is (no re-layout
            # into the reservation, 1.3 ms of a 4.9 ms window refit at 4096 x 472); the arrays keep their reserved rows
            self.Lop, self._Lcap = Lop, N2
            self.Vw[:, :N2], self.X[:, :N2], self.UHB[:, :N2] = Vw, X, UHB
        else:
            self._fill(Lop, Vw, X, UHB, N2, self.capacity, reuse=True)
        self.N = self.N0 = N2
        self.t = 0
        self.drops += 1
        return info

    def count_drop_failures(self):
        """Host read of the last window refit's info (retry_levels mode keeps it on the device): instances still failing."""
        return 0 if self.drop_info is None else int((self.drop_info != 0).sum())

    def _tail_step(self, xq, x_new, uh_new, xdot_new, jitter_new, Mk, Bk, do_append):
        # (the pointers of this object's own buffers are converted once per window: a closed loop on part batches makes this call
        #  thousands of times and the host side of it is what bounds four part batches)
        st = self.__dict__.get("_tail_static")
        if st is None or st[0] is not self.Lop:
            st = self._tail_static = (self.Lop, getattr(lib, ("bcbf_gp_tail_step_kind" if self._kind else "bcbf_gp_tail_step") + _suf(self.X)),
                                      (_p(self.Lop), _p(self.Vw), _p(self.X), _p(self.UHB), _p(self.ell), _p(self.s2), _p(self.Bm), _p(self.M0)),
                                      (_p(self._Rb), _p(self._Rinv), _p(self.info), _p(self._Wfull), _p(self._sw)),
                                      (_p(self._rUH), _p(self._rY), _p(self._rJ)))
        check(st[1](*st[2], _p(xq), _p(x_new), _p(uh_new), _p(xdot_new), _p(jitter_new), *st[3], _p(Mk), _p(Bk), *st[4], self.Bt,
                    self.N0, self.t, self._tcap, self.capacity, self._Lcap, self.n, self.C - 1, int(do_append),
                    *((self._kind,) if self._kind else ()), _stream(self.X)),
              "bcbf_gp_tail_step")

    def posterior(self, xq, jitter2=None, want_W=False, out=None):
        """(Mk[Bt,n,C], Bk[Bt,C,C]) (+ W[Bt,Np,C]) at one query per instance on the live points."""
        _chk(self.X, xq, jitter2)
        f = dict(dtype=self.X.dtype, device=self.X.device)
        if out is None:
            Mk, Bk = torch.empty(self.Bt, self.n, self.C, **f), torch.empty(self.Bt, self.C, self.C, **f)
        else:
            Mk, Bk = out
        if self.tail:
            if jitter2 is not None or want_W:
                raise NotImplementedError("tail=True: the posterior at one query per instance, no jitter, no W")
            self._tail_step(xq, xq, self._ones, None, None, Mk, Bk, False)
            return Mk, Bk
        W = torch.empty(self.Bt, (self.N + 31) // 32 * 32, self.C, **f) if want_W else None
        check(getattr(lib, ("bcbf_posterior_query_reserved_kind" if self._kind else "bcbf_posterior_query_reserved") + _suf(self.X))(
            _p(self.Lop), _p(self.Vw), _p(self.X), _p(self.UHB), _p(self.ell), _p(self.s2), _p(self.Bm), _p(self.M0), _p(xq),
            _p(jitter2), _p(Mk), _p(Bk), _p(W), self.Bt, self.N, self.capacity, self.n, self.C - 1,
            *((self._kind,) if self._kind else ()), _stream(self.X)),
            "bcbf_posterior_query_reserved")
        return (Mk, Bk, W) if want_W else (Mk, Bk)

    def append(self, x_new, uh_new, xdot_new, jitter_new=None, query=None, out=None):
        """One observation per instance, in place.  Returns info[Bt] (0, or N+1 where the new pivot was not positive: that
        instance gained a neutral point -- retrying is the caller's business, as with `gp_append`).
        query[Bt,n] (with out = (Mk, Bk), or allocated): the posterior at `query` on the points BEFORE the append, computed
        on the same pass over the factors as the append's forward solve -- a "posterior, then append" step for the traffic
        of one; then returns (info, Mk, Bk)."""
        if self.window is not None and self.N >= self.window + self.drop:
            raise RuntimeError("window mode: the live size is already window + %d (drop_oldest_block failed?)" % self.drop)
        if self.N >= self.capacity:
            raise RuntimeError("ReservedGP is full (%d points): reserve a larger capacity" % self.capacity)
        _chk(self.X, x_new, uh_new, xdot_new, jitter_new, query)
        if self.tail:
            if self.t >= self._tcap:
                raise RuntimeError("tail=True: %d points since the last window refit -- drop_oldest_block first" % self.t)
            f = dict(dtype=self.X.dtype, device=self.X.device)
            if query is not None:
                Mk, Bk = out if out is not None else (torch.empty(self.Bt, self.n, self.C, **f), torch.empty(self.Bt, self.C, self.C, **f))
            else:
                Mk, Bk = self._Mkw, self._Bkw
            self._tail_step(x_new if query is None else query, x_new, uh_new, xdot_new, jitter_new, Mk, Bk, True)
            self.t += 1
            if self.window is None and self.t == self._tcap:
                # growth: the 32 tail rows become block row N0 / 32 of the reserved operator (one launch, full-line writes)
                check(getattr(lib, "bcbf_gp_tail_commit" + _suf(self.X))(_p(self.Lop), _p(self._Rb), _p(self._Rinv), self.Bt, self.N0, self.t,
                                                                          self._tcap, self.capacity, _stream(self.X)), "bcbf_gp_tail_commit")
                self.N0 += self._tcap
                self.t = 0
            return self._appended(query, Mk, Bk)
        Mk = Bk = None
        if query is not None:
            f = dict(dtype=self.X.dtype, device=self.X.device)
            Mk, Bk = out if out is not None else (torch.empty(self.Bt, self.n, self.C, **f), torch.empty(self.Bt, self.C, self.C, **f))
            if self._Ww.shape[0] != 2 * self.Bt:                      # work buffers for two queries per instance
                self._Ww = torch.empty(2 * self.Bt, *self._Ww.shape[1:], **f)
                self._Mkw, self._Bkw = torch.empty(2 * self.Bt, self.n, self.C, **f), torch.empty(2 * self.Bt, self.C, self.C, **f)
        head = (_p(self.Lop), _p(self.Vw), _p(self.X), _p(self.UHB), _p(self.ell), _p(self.s2), _p(self.Bm), _p(self.M0),
                _p(x_new), _p(uh_new), _p(xdot_new), _p(jitter_new), _p(self.info), _p(self._Ww), _p(self._Mkw), _p(self._Bkw),
                _p(query), _p(Mk), _p(Bk))
        tail = (self.Bt, self.N, self.capacity, self.n, self.C - 1, _stream(self.X))
        if self._kind:
            raw = (_p(self._rUH), _p(self._rY), _p(self._rJ)) if self.window is not None else (None, None, None)
            check(getattr(lib, "bcbf_gp_append_reserved_kind" + _suf(self.X))(*head, *raw, *tail[:-1], self._kind, tail[-1]),
                  "bcbf_gp_append_reserved_kind")
        elif self.window is not None:
            # window mode: the same launches also record the point's RAW row (uh, xdot, jitter) -- neutral (0, 0, unit pivot)
            # where the new pivot failed, which is what the in-place path holds -- for the next refit of the window
            check(getattr(lib, "bcbf_gp_append_reserved_raw" + _suf(self.X))(*head, _p(self._rUH), _p(self._rY), _p(self._rJ), *tail),
                  "bcbf_gp_append_reserved_raw")
        else:
            check(getattr(lib, "bcbf_gp_append_reserved" + _suf(self.X))(*head, *tail), "bcbf_gp_append_reserved")
        return self._appended(query, Mk, Bk)

    def _appended(self, query, Mk, Bk):
        self.N += 1
        info = self.info
        if self.window is not None and self.N >= self.window + self.drop:
            dinfo = self.drop_oldest_block()          # AFTER the append: the posterior the caller asked for saw every point
            if self.drop_failures:                    # a window that could not be factored: carried in the returned info (< 0)
                info = torch.where(dinfo != 0, -dinfo.abs() - 1, info)
        return info if query is None else (info, Mk, Bk)

    def live(self):
        """Views of the live rows: (Vw[Bt,N,n], X[Bt,N,n], UHB[Bt,N,C]) (strided: not inputs of the packed-layout kernels)."""
        return self.Vw[:, :self.N], self.X[:, :self.N], self.UHB[:, :self.N]


def kb_inverse(Lop, N, gemm=True):
    """Dense K_b^-1 [Bt,N,N] from the packed factor (fit path only).  gemm=True: L^-1 from bcbf_trtri (the forward half of
    the solve) and K_b^-1 = L^-T L^-1 on the matrix cores (bcbf_syrk_lt: 32 x 32 tiles of the triangular product);
    gemm=False: bcbf_potri (forward + backward solves of the identity: its backward half is latency bound, 2 ms at
    N = 512 for one model against 0.3 ms this way)."""
    _chk(Lop)
    Bt = Lop.shape[0]
    Kinv = torch.empty(Bt, N, N, dtype=Lop.dtype, device=Lop.device)
    if gemm:
        Linv = torch.empty(Bt, N, N, dtype=Lop.dtype, device=Lop.device)
        check(getattr(lib, "bcbf_trtri" + _suf(Lop))(_p(Lop), _p(Linv), Bt, N, _stream(Lop)), "bcbf_trtri")
        check(getattr(lib, "bcbf_syrk_lt" + _suf(Lop))(_p(Linv), _p(Kinv), Bt, N, _stream(Lop)), "bcbf_syrk_lt")
        return Kinv
    check(getattr(lib, "bcbf_potri" + _suf(Lop))(_p(Lop), _p(Kinv), Bt, N, _stream(Lop)), "bcbf_potri")
    return Kinv


_MLL_WORK = {}


def _mll_work(Bt, N, m, device):
    """Workspace of bcbf_mll_grad's split form (partial sums; reused between the iterations of a fit).  One buffer per
    (device, HIP stream): fits on different streams / threads never share partial sums, and a buffer is only ever
    replaced by a LARGER one on its own stream, where stream order puts the free behind the launches that read it (the
    launches go through ctypes, so torch's allocator sees no other stream using the block)."""
    nbytes = int(lib.bcbf_mll_grad_work_bytes(Bt, N, m))
    if nbytes == 0:
        return None
    key = (str(device), int(torch.cuda.current_stream(device).cuda_stream))
    buf = _MLL_WORK.get(key)
    if buf is None or buf.numel() * 8 < nbytes:
        buf = _MLL_WORK[key] = torch.empty(nbytes // 8, dtype=torch.float64, device=device)
    return buf


def mll_grad(Lop, alpha, Kinv, X, UH, R, Ainv, Bm, ell, s2, lin=None, kernel="rbf"):
    """O(N^2) sums of the marginal-log-likelihood gradient (bcbf.h K12).  Returns
    (g_ell[Bt,n], g_s2[Bt], g_B[Bt,C,C], logdetK[Bt], RtA[Bt,nt,nt], UHtA[Bt,C,nt]) (+ g_lin[Bt] when `lin` is given:
    RBF + Linear data kernel, nt = R.shape[2] target columns)."""
    _chk(Lop, alpha, Kinv, X, UH, R, Ainv, Bm, ell, s2, lin)
    Bt, N, n = X.shape
    C = UH.shape[2]
    f = dict(dtype=X.dtype, device=X.device)
    if lin is not None:
        nt = R.shape[2]
        g_ell, g_s2, g_B, g_lin = torch.empty(Bt, n, **f), torch.empty(Bt, **f), torch.empty(Bt, C, C, **f), torch.empty(Bt, **f)
        logdet, RtA, UHtA = torch.empty(Bt, **f), torch.empty(Bt, nt, nt, **f), torch.empty(Bt, C, nt, **f)
        check(getattr(lib, "bcbf_mll_grad_rbflin" + _suf(X))(
            _p(Lop), _p(alpha), _p(Kinv), _p(X), _p(UH), _p(R), _p(Ainv), _p(Bm), _p(ell), _p(s2), _p(lin), _p(g_ell),
            _p(g_s2), _p(g_lin), _p(g_B), _p(logdet), _p(RtA), _p(UHtA), Bt, N, n, C - 1, nt,
            _p(_mll_work(Bt, N, C - 1, X.device)), _stream(X)), "bcbf_mll_grad_rbflin")
        return g_ell, g_s2, g_B, logdet, RtA, UHtA, g_lin
    g_ell, g_s2, g_B = torch.empty(Bt, n, **f), torch.empty(Bt, **f), torch.empty(Bt, C, C, **f)
    logdet, RtA, UHtA = torch.empty(Bt, **f), torch.empty(Bt, n, n, **f), torch.empty(Bt, C, n, **f)
    check(getattr(lib, _kern("bcbf_mll_grad", kernel) + _suf(X))(_p(Lop), _p(alpha), _p(Kinv), _p(X), _p(UH), _p(R), _p(Ainv), _p(Bm),
                                                  _p(ell), _p(s2), _p(g_ell), _p(g_s2), _p(g_B), _p(logdet), _p(RtA),
                                                  _p(UHtA), Bt, N, n, C - 1, _p(_mll_work(Bt, N, C - 1, X.device)),
                                                  _stream(X)), "bcbf_mll_grad")
    return g_ell, g_s2, g_B, logdet, RtA, UHtA


def kinv_apply(Kinv, R, out=None):
    """alpha[Bt,N,nt] = Kinv R for the dense symmetric K_b^-1 (bcbf_kinv_apply: the fit iteration's `Kinv @ R` without a
    library GEMM)."""
    _chk(Kinv, R, out)
    Bt, N, nt = R.shape
    alpha = torch.empty_like(R) if out is None else out
    check(getattr(lib, "bcbf_kinv_apply" + _suf(R))(_p(Kinv), _p(R), _p(alpha), Bt, N, nt, _stream(R)), "bcbf_kinv_apply")
    return alpha


def fit_param_count(n, m, rA=None, rB=None):
    """Raw parameters per model of the batched fit (bcbf.h: theta's layout)."""
    P = int(lib.bcbf_fit_param_count(n, m, n if rA is None else rA, 1 + m if rB is None else rB))
    if P < 0:
        raise ValueError("unsupported fit shape n=%d m=%d ranks %s %s" % (n, m, rA, rB))
    return P


def fit_derive(theta, n, m, rA, rB, want_Ainv=True):
    """theta[Bt,P] -> dict(ell, s2, A, Bm, M0[, Ainv, logdetA]) (bcbf_fit_derive)."""
    _chk(theta)
    Bt, C = theta.shape[0], 1 + m
    f = dict(dtype=theta.dtype, device=theta.device)
    out = dict(ell=torch.empty(Bt, n, **f), s2=torch.empty(Bt, **f), A=torch.empty(Bt, n, n, **f), Bm=torch.empty(Bt, C, C, **f),
               M0=torch.empty(Bt, C, n, **f))
    if want_Ainv:
        out.update(Ainv=torch.empty(Bt, n, n, **f), logdetA=torch.empty(Bt, **f))
    check(getattr(lib, "bcbf_fit_derive" + _suf(theta))(_p(theta), _p(out["ell"]), _p(out["s2"]), _p(out["A"]), _p(out["Bm"]),
                                                        _p(out["M0"]), _p(out.get("Ainv")), _p(out.get("logdetA")), Bt, n, m, rA, rB,
                                                        _stream(theta)), "bcbf_fit_derive")
    return out


def fit_adam_step(theta, mom1, mom2, sums, Ainv, logdetA, N, n, m, rA, rB, step, lr, betas=(0.9, 0.999), eps=1e-8, skip=None,
                  gamma_prior=None, want_grad=False):
    """One Adam update of every model from mll_grad's `sums` = (g_ell, g_s2, g_B, logdetK, RtA, UHtA) (bcbf_fit_adam_step).
    Returns (loss[Bt], grad[Bt,P] | None); step = 0: value and gradient only."""
    g_ell, g_s2, g_B, logdetK, RtA, UHtA = sums[:6]
    _chk(theta, mom1, mom2, g_ell, g_s2, g_B, logdetK, RtA, UHtA, Ainv, logdetA, skip)
    Bt = theta.shape[0]
    loss = torch.empty(Bt, dtype=theta.dtype, device=theta.device)
    grad = torch.empty_like(theta) if want_grad else None
    prior = None if gamma_prior is None else (ctypes.c_double * 2)(float(gamma_prior[0]), float(gamma_prior[1]))
    check(getattr(lib, "bcbf_fit_adam_step" + _suf(theta))(
        _p(theta), _p(mom1), _p(mom2), _p(g_ell), _p(g_s2), _p(g_B), _p(logdetK), _p(RtA), _p(UHtA), _p(Ainv), _p(logdetA), _p(skip),
        _p(loss), _p(grad), Bt, N, n, m, rA, rB, int(step), float(lr), float(betas[0]), float(betas[1]), float(eps), prior,
        _stream(theta)), "bcbf_fit_adam_step")
    return loss, grad


def posterior_step(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2=None, out=None):
    """(Mk[Bt,n,C], Bk[Bt,C,C]) at one query per instance  (control_affine_model.py:1051-1091, b=1)."""
    _chk(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2)
    Bt, N, n = X.shape
    C = UHB.shape[2]
    if out is None:
        Mk = torch.empty(Bt, n, C, dtype=X.dtype, device=X.device)
        Bk = torch.empty(Bt, C, C, dtype=X.dtype, device=X.device)
    else:
        Mk, Bk = out
    check(getattr(lib, "bcbf_posterior_step" + _suf(X))(_p(Lop), _p(Vw), _p(X), _p(UHB), _p(ell), _p(s2), _p(Bm),
                                                        _p(M0), _p(xq), _p(jitter2), _p(Mk), _p(Bk), Bt, N, n, C - 1,
                                                        _stream(X)), "bcbf_posterior_step")
    return Mk, Bk


def posterior_query(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2=None, shared=True, want_W=False, lin=None, kernel="rbf"):
    """b queries against one shared GP (shared=True; GP tensors carry a leading axis of 1) or one query per
    instance.  Returns (Mk[b,n,C], Bk[b,C,C], W[b,Np,C] | None).  lin: linear part of the data kernel (CoGP);
    kernel="matern52": the opt-in Matern-5/2 data kernel (streaming kernel)."""
    _chk(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2, lin)
    N, n = X.shape[1], X.shape[2]
    C = UHB.shape[2]
    b = xq.shape[0]
    if shared and X.shape[0] != 1:
        raise ValueError("shared query: GP tensors must have a leading axis of 1")
    Mk = torch.empty(b, n, C, dtype=X.dtype, device=X.device)
    Bk = torch.empty(b, C, C, dtype=X.dtype, device=X.device)
    Np = (N + 31) // 32 * 32
    W = torch.empty(b, Np, C, dtype=X.dtype, device=X.device) if want_W else None
    if kernel not in DATA_KERNELS:
        raise ValueError("data kernel %r: one of %s" % (kernel, DATA_KERNELS))
    if kernel != "rbf":
        if lin is not None:
            raise ValueError("the opt-in data kernels have no linear part")
        check(getattr(lib, "bcbf_posterior_query" + _KSUF[kernel] + _suf(X))(
            _p(Lop), _p(Vw), _p(X), _p(UHB), _p(ell), _p(s2), _p(Bm), _p(M0), _p(xq), _p(jitter2), _p(Mk), _p(Bk), _p(W),
            1 if shared else 0, b, N, n, C - 1, _stream(X)), "bcbf_posterior_query" + _KSUF[kernel])
        return Mk, Bk, W
    if lin is not None:
        check(getattr(lib, "bcbf_posterior_query_rbflin" + _suf(X))(
            _p(Lop), _p(Vw), _p(X), _p(UHB), _p(ell), _p(s2), _p(lin), _p(Bm), _p(M0), _p(xq), _p(jitter2), _p(Mk), _p(Bk),
            _p(W), 1 if shared else 0, b, N, n, C - 1, _stream(X)), "bcbf_posterior_query_rbflin")
        return Mk, Bk, W
    check(getattr(lib, "bcbf_posterior_query" + _suf(X))(_p(Lop), _p(Vw), _p(X), _p(UHB), _p(ell), _p(s2), _p(Bm),
                                                         _p(M0), _p(xq), _p(jitter2), _p(Mk), _p(Bk), _p(W),
                                                         1 if shared else 0, b, N, n, C - 1, _stream(X)),
          "bcbf_posterior_query")
    return Mk, Bk, W


def posterior_shared(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2=None, want_W=False, kernel="rbf"):
    """b queries against one GP on the matrix cores (GP tensors carry a leading axis of 1; fp64: N <= 512, n <= 4).
    posterior_query(shared=True) routes here for b >= 16; this entry forces the MFMA kernel for any b.
    kernel="matern52": the opt-in Matern-5/2 data kernel (bcbf_posterior_shared_matern52)."""
    if kernel not in DATA_KERNELS:
        raise ValueError("kernel must be one of %s" % (DATA_KERNELS,))
    _chk(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, jitter2)
    if X.dtype == torch.float64 and (X.shape[1] > 512 or X.shape[2] > 4):
        raise ValueError("posterior_shared in fp64 holds the solution in registers: N <= 512, n <= 4 (posterior_query streams the rest)")
    if X.shape[0] != 1:
        raise ValueError("shared query: GP tensors must have a leading axis of 1")
    N, n = X.shape[1], X.shape[2]
    C = UHB.shape[2]
    b = xq.shape[0]
    Mk = torch.empty(b, n, C, dtype=X.dtype, device=X.device)
    Bk = torch.empty(b, C, C, dtype=X.dtype, device=X.device)
    Np = (N + 31) // 32 * 32
    W = torch.empty(b, Np, C, dtype=X.dtype, device=X.device) if want_W else None
    fn = getattr(lib, "bcbf_posterior_shared" + _KSUF[kernel] + _suf(X))
    check(fn(_p(Lop), _p(Vw), _p(X), _p(UHB), _p(ell), _p(s2), _p(Bm), _p(M0), _p(xq),
             _p(jitter2), _p(Mk), _p(Bk), _p(W), b, N, n, C - 1, _stream(X)), "bcbf_posterior_shared")
    return Mk, Bk, W


def posterior_jets(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq, shared=False, want_W=False, kernel="rbf"):
    """(Mk, Bk, G[b,CT,CT], Mj[b,n,CT]) with CT = (1+m)(1+n): value + first x-derivative jets
    (replaces autograd through custom_predict in GradientGP, gp_algebra.py:340-402).  want_W: also
    Wj[b,Np,CT] = L^-1 [Phi, dPhi/dx_d] (fifth return value), for derivative kernels between different states."""
    _chk(Lop, Vw, X, UHB, ell, s2, Bm, M0, xq)
    N, n = X.shape[1], X.shape[2]
    C = UHB.shape[2]
    b = xq.shape[0]
    CT = C * (1 + n)
    f = dict(dtype=X.dtype, device=X.device)
    Mk, Bk = torch.empty(b, n, C, **f), torch.empty(b, C, C, **f)
    G, Mj = torch.empty(b, CT, CT, **f), torch.empty(b, n, CT, **f)
    Wj = torch.empty(b, (N + 31) // 32 * 32, CT, **f) if want_W else None
    check(getattr(lib, _kern("bcbf_posterior_jets", kernel) + _suf(X))(
        _p(Lop), _p(Vw), _p(X), _p(UHB), _p(ell), _p(s2), _p(Bm), _p(M0), _p(xq), _p(Mk), _p(Bk), _p(G), _p(Mj), _p(Wj),
        1 if shared else 0, b, N, n, C - 1, _stream(X)), "bcbf_posterior_jets")
    return (Mk, Bk, G, Mj, Wj) if want_W else (Mk, Bk, G, Mj)


def predict_assemble(G, Xq, Xqp, ell, s2, Bm, A=None, jitter=None, want_BkXX=True, want_kron=False, kernel="rbf"):
    """Predictive covariance of a query set against ONE GP in one launch, from the Gram G[b, b', 1+m, 1+m] = W_b' W'_b' of its
    whitened cross-covariances (`posterior_query(..., want_W=True)`; `torch.einsum("bnc,pnd->bpcd", W, Wp)`):
    BkXX[b, b', 1+m, 1+m] = k(x_b, x'_b') Bm - G (+ `jitter`[b (1+m)] on its diagonal: make_psd, control_affine_model.py:1089)
    and / or kron(Bk2, A) [b (1+m) n, b' (1+m) n] (`custom_predict_fullmat`, :963-980).
    Xq[b, n], Xqp[b', n], ell[n], s2[1], Bm[1+m, 1+m], A[n, n]."""
    _chk(G, Xq, Xqp, ell, s2, Bm, A, jitter)
    if kernel not in DATA_KERNELS:
        raise ValueError("kernel %r: one of %s" % (kernel, DATA_KERNELS))
    b, bp, C, _ = G.shape
    n = Xq.shape[1]
    if G.shape[3] != C or Xq.shape[0] != b or Xqp.shape[0] != bp or Xqp.shape[1] != n:
        raise ValueError("predict_assemble: inconsistent shapes")
    if want_kron and A is None:
        raise ValueError("predict_assemble: the Kronecker product needs A")
    BkXX = torch.empty(b, bp, C, C, dtype=G.dtype, device=G.device) if want_BkXX else None
    Kron = torch.empty(b * C * n, bp * C * n, dtype=G.dtype, device=G.device) if want_kron else None
    check(getattr(lib, "bcbf_predict_assemble" + _suf(G))(
        _p(G), _p(Xq), _p(Xqp), _p(ell), _p(s2), _p(Bm), _p(A) if A is not None else None,
        _p(jitter) if jitter is not None else None, _p(BkXX) if want_BkXX else None, _p(Kron) if want_kron else None,
        b, bp, n, C - 1, DATA_KERNELS.index(kernel), _stream(G)), "bcbf_predict_assemble")
    return BkXX, Kron


HESSIAN_MODES = {"reference": 0, "project": 1}


def clean_hessian(H, eigeps=2e-3, mode="reference"):
    """The eigenvalue clean-up GradientGP.knl(x, x) applies to its Hessian (gp_algebra.py:384-392) on a batch H[b,n,n]
    (n <= 4) on the device.  mode "reference": `eigenvectors.T @ diag(evalz) @ eigenvectors` with the general solver's
    eigenvectors (csrc/geev_small.h); "project": spectral projection of the symmetric part.  Returns (H_clean, status):
    status 0 untouched, 1 an eigenvalue <= -eigeps (H returned unchanged; the reference asserts), 4 / 6 cleaned."""
    _chk(H)
    b, n, _ = H.shape
    out = torch.empty_like(H)
    status = torch.empty(b, dtype=torch.int32, device=H.device)
    check(getattr(lib, "bcbf_clean_hessian" + _suf(H))(_p(H), _p(out), _p(status), b, n, float(eigeps),
                                                       HESSIAN_MODES[mode], _stream(H)), "bcbf_clean_hessian")
    return out, status


def cbc2_terms(Mk, Bk, G, Mj, A, Bm, ell, s2, h, gh, Hh, kalpha, u0, hessian_mode="reference", kernel="rbf"):
    """Rel-degree-2 terms (cbc2_gp + cbc2_quadratic_terms, cbc2.py:7-33).  Returns
    ((mean_A[b,m], mean_b[b]), (Q[b,m,m], p[b,m], r[b]), mean[b], var[b], status[b]); status 1 = the reference's
    positive-definiteness assert fails, 4 / 6 = the Hessian clean-up ran (`clean_hessian`), 0 otherwise."""
    _chk(Mk, Bk, G, Mj, A, Bm, ell, s2, h, gh, Hh, kalpha, u0)
    b, n, C = Mk.shape
    m = C - 1
    out = torch.empty(b, m + 1 + m * m + m + 1 + 2, dtype=Mk.dtype, device=Mk.device)
    status = torch.empty(b, dtype=torch.int32, device=Mk.device)
    check(getattr(lib, "bcbf_cbc2_terms" + _suf(Mk))(_p(Mk), _p(Bk), _p(G), _p(Mj), _p(A), _p(Bm), _p(ell), _p(s2),
                                                     _p(h), _p(gh), _p(Hh), _p(kalpha), _p(u0), _p(out), _p(status),
                                                     b, n, m, HESSIAN_MODES[hessian_mode], DATA_KERNELS.index(kernel),
                                                     _stream(Mk)), "bcbf_cbc2_terms")
    o = 0
    mean_A = out[:, o:o + m]; o += m
    mean_b = out[:, o]; o += 1
    Q = out[:, o:o + m * m].reshape(b, m, m); o += m * m
    p = out[:, o:o + m]; o += m
    r = out[:, o]; o += 1
    return (mean_A, mean_b), (Q, p, r), out[:, o], out[:, o + 1], status


def terms_width(m):
    return m + 1 + m * m + m + 1


def cone_width(m):
    return (m + 1) * m + (m + 1) + m + 1


def cbc_terms(Mk, Bk, A, grad, cst, sign, fhat, ghat, want_terms=True, out=None):
    """Rel-degree-1 constraint terms + cone form (cbc2.py:7-23, unicycle_move_to_pose.py:837-916)."""
    _chk(Mk, Bk, A, grad, cst, sign, fhat, ghat)
    Bt, K, n = grad.shape
    m = ghat.shape[2]
    if out is None:
        terms = torch.empty(Bt, K, terms_width(m), dtype=Mk.dtype, device=Mk.device) if want_terms else None
        cones = torch.empty(Bt, K, cone_width(m), dtype=Mk.dtype, device=Mk.device)
        cstatus = torch.empty(Bt, K, dtype=torch.int32, device=Mk.device)
    else:
        terms, cones, cstatus = out
    check(getattr(lib, "bcbf_cbc_terms" + _suf(Mk))(_p(Mk), _p(Bk), _p(A), _p(grad), _p(cst), _p(sign), _p(fhat),
                                                    _p(ghat), _p(terms), _p(cones), _p(cstatus), Bt, K, n, m,
                                                    _stream(Mk)), "bcbf_cbc_terms")
    return terms, cones, cstatus


def unpack_terms(terms, m):
    o = 0
    bfe = terms[..., o:o + m]; o += m
    e = terms[..., o]; o += 1
    V = terms[..., o:o + m * m].reshape(*terms.shape[:-1], m, m); o += m * m
    bfv = terms[..., o:o + m]; o += m
    v = terms[..., o]
    return bfe, e, V, bfv, v


def unpack_cones(cones, m):
    o = 0
    A = cones[..., o:o + (m + 1) * m].reshape(*cones.shape[:-1], m + 1, m); o += (m + 1) * m
    b = cones[..., o:o + m + 1]; o += m + 1
    c = cones[..., o:o + m]; o += m
    d = cones[..., o]
    return A, b, c, d


def pack_cones(A, b, c, d):
    lead = A.shape[:-2]
    return torch.cat([A.reshape(*lead, -1), b, c, d.reshape(*lead, 1)], dim=-1).contiguous()


def socp(w, r, cones, relax_mask, rho, max_iters=100, out=None):
    """The CLF-CBF program of ControllerCLFBayesian.control (unicycle_move_to_pose.py:926-953).
    Returns (y[Bt,m+1] = [u, relax], status[Bt], iters[Bt])."""
    _chk(w, r, cones, relax_mask, rho)
    Bt, K, _ = cones.shape
    m = r.shape[1]
    if K > 4:                                  # more cones than the quad kernel's four lanes: the generic solver
        return _socp_generic(w, r, cones, relax_mask, rho, max_iters, out)
    if out is None:
        y = torch.empty(Bt, m + 1, dtype=w.dtype, device=w.device)
        status = torch.empty(Bt, dtype=torch.int32, device=w.device)
        iters = torch.empty(Bt, dtype=torch.int32, device=w.device)
    else:
        y, status, iters = out
    check(getattr(lib, "bcbf_socp" + _suf(w))(_p(w), _p(r), _p(cones), _p(relax_mask), _p(rho), _p(y), _p(status),
                                              _p(iters), Bt, K, m, max_iters, _stream(w)), "bcbf_socp")
    return y, status, iters


def _socp_generic(w, r, cones, relax_mask, rho, max_iters=100, out=None):
    """The CLF-CBF program with more than four cones (e.g. a third obstacle), through `bcbf_coneqp_f64`:
    min sum w_i (u_i - r_i)^2 + w_m relax^2  s.t.  c_k'u + d_k + relax_mask_k relax >= rho |A_k u + b_k|
    as  min 1/2 y'P y + q'y, G y + s = h, s in Q^{m+2} x ... (rows [-c', -relax_mask; -rho A, 0], h = [d; rho b]).
    Row assembly is host-side tensor plumbing; the solve is the library's (fp64)."""
    Bt, K, _ = cones.shape
    m = r.shape[1]
    f64 = dict(dtype=torch.float64, device=w.device)
    A, b, c, d = (t.to(torch.float64) for t in unpack_cones(cones, m))
    nv, D = m + 1, m + 2
    rho64 = rho.to(torch.float64)
    G = torch.zeros(Bt, K, D, nv, **f64)
    G[:, :, 0, :m] = -c
    G[:, :, 0, m] = -relax_mask.to(torch.float64)[None, :]
    G[:, :, 1:, :m] = -rho64[:, None, None, None] * A
    h = torch.cat([d[..., None], rho64[:, None, None] * b], dim=-1)
    P = torch.diag_embed(2.0 * w.to(torch.float64)).contiguous()
    q = torch.zeros(Bt, nv, **f64)
    q[:, :m] = -2.0 * w[:, :m].to(torch.float64) * r.to(torch.float64)
    x, status, iters = coneqp(P, q, G.reshape(Bt, K * D, nv).contiguous(), h.reshape(Bt, K * D).contiguous(), 0, [D] * K,
                              max_iters=max_iters)
    y = x.to(w.dtype)
    if out is not None:
        out[0].copy_(y); out[1].copy_(status); out[2].copy_(iters)
        return out
    return y, status, iters


def cbc_socp(Mk, Bk, A, grad, cst, sign, fhat, ghat, w, r, relax_mask, rho, max_iters=100, want_terms=False):
    """cbc_terms + socp in one launch (four lanes per instance).  Returns (y, status, iters, cones, cstatus, terms)."""
    _chk(Mk, Bk, A, grad, cst, sign, fhat, ghat, w, r, relax_mask, rho)
    Bt, K, n = grad.shape
    m = ghat.shape[2]
    if K > 4:                                  # two launches + the generic solver (see `socp`)
        terms, cones, cstatus = cbc_terms(Mk, Bk, A, grad, cst, sign, fhat, ghat, want_terms=want_terms)
        y, status, iters = _socp_generic(w, r, cones, relax_mask, rho, max_iters)
        status = torch.where((cstatus != 0).any(dim=1), torch.full_like(status, 3), status)
        return y, status, iters, cones, cstatus, terms
    f = dict(dtype=Mk.dtype, device=Mk.device)
    terms = torch.empty(Bt, K, terms_width(m), **f) if want_terms else None
    cones = torch.empty(Bt, K, cone_width(m), **f)
    cstatus = torch.empty(Bt, K, dtype=torch.int32, device=Mk.device)
    y = torch.empty(Bt, m + 1, **f)
    status = torch.empty(Bt, dtype=torch.int32, device=Mk.device)
    iters = torch.empty(Bt, dtype=torch.int32, device=Mk.device)
    check(getattr(lib, "bcbf_cbc_socp" + _suf(Mk))(_p(Mk), _p(Bk), _p(A), _p(grad), _p(cst), _p(sign), _p(fhat),
                                                   _p(ghat), _p(w), _p(r), _p(relax_mask), _p(rho), _p(terms),
                                                   _p(cones), _p(cstatus), _p(y), _p(status), _p(iters), Bt, K, n, m,
                                                   max_iters, _stream(Mk)), "bcbf_cbc_socp")
    return y, status, iters, cones, cstatus, terms


def coneqp(P, q, G, h, l, qdims, max_iters=100):
    """Generic small cone QP, fp64 (optimizers.py:42-116)."""
    _chk(P, q, G, h)
    if P.dtype != torch.float64:
        raise TypeError("coneqp is fp64 only")
    Bt, nv = q.shape
    qd = (ctypes.c_int * max(1, len(qdims)))(*qdims)
    x = torch.empty(Bt, nv, dtype=P.dtype, device=P.device)
    status = torch.empty(Bt, dtype=torch.int32, device=P.device)
    iters = torch.empty(Bt, dtype=torch.int32, device=P.device)
    check(lib.bcbf_coneqp_f64(_p(P), _p(q), _p(G), _p(h), nv, l, qd, len(qdims), _p(x), _p(status), _p(iters), Bt,
                              max_iters, _stream(P)), "bcbf_coneqp")
    return x, status, iters


def controller_cones(terms, u_ref, kinds, factors=None, ctrl_reg=1.0, relax_weight=1.0, extravars=2, objective=True):
    """Rows (G[Bt,Kt,nv], h[Bt,Kt], qdims, l, cstatus) of the SOCPController / QPController program over
    y = [extravars.., u] (controllers.py:396-540, 614-662) from packed quadratic terms[Bt,K,T]."""
    K = len(kinds)
    if K:
        _chk(terms)
    ref = terms if K else u_ref
    Bt = ref.shape[0]
    m = u_ref.shape[1] if u_ref is not None else None
    if m is None:
        T = terms.shape[2]
        m = next(mm for mm in range(1, 8) if terms_width(mm) == T)
    kd = (ctypes.c_int * max(1, K))(*kinds)
    fc = (ctypes.c_double * max(1, K))(*([1.0] * K if factors is None else [float(f) for f in factors]))
    Kt = lib.bcbf_controller_cones_rows(kd, K, m, int(bool(objective)))
    if Kt < 0:
        raise ValueError("bad constraint kinds %r" % (kinds,))
    nv = extravars + m
    G = torch.empty(Bt, Kt, nv, dtype=torch.float64, device=ref.device)
    h = torch.empty(Bt, Kt, dtype=torch.float64, device=ref.device)
    cstatus = torch.zeros(Bt, max(K, 1), dtype=torch.int32, device=ref.device)
    check(getattr(lib, "bcbf_controller_cones" + _suf(ref))(
        _p(terms if K else None), _p(u_ref.contiguous() if u_ref is not None else None), kd, fc, float(ctrl_reg),
        float(relax_weight), extravars, int(bool(objective)), _p(G), _p(h), _p(cstatus), Bt, K, m, _stream(ref)),
        "bcbf_controller_cones")
    l = sum(1 for k in kinds if k == 2)
    qdims = ([m + 2] if objective else []) + [m + 2 for k in kinds if k != 2]
    return G, h, qdims, l, cstatus[:, :K]


def unicycle_constraints(x, plan, dot_plan, Kp, clf_gamma, centers, radii, tw, gammas, L_mean, out=None):
    """CLC row + obstacle rows: (grad[Bt,1+Kob,3], cst[Bt,1+Kob], fhat[Bt,3], ghat[Bt,3,2])."""
    _chk(x, plan, dot_plan, Kp, centers, radii, tw, gammas)
    Bt = x.shape[0]
    Kob = 0 if centers is None else centers.shape[1]
    if out is None:
        grad = torch.empty(Bt, 1 + Kob, 3, dtype=x.dtype, device=x.device)
        cst = torch.empty(Bt, 1 + Kob, dtype=x.dtype, device=x.device)
        fhat = torch.empty(Bt, 3, dtype=x.dtype, device=x.device)
        ghat = torch.empty(Bt, 3, 2, dtype=x.dtype, device=x.device)
    else:
        grad, cst, fhat, ghat = out
    check(getattr(lib, "bcbf_unicycle_constraints" + _suf(x))(
        _p(x), _p(plan), _p(dot_plan), _p(Kp), clf_gamma, _p(centers), _p(radii), _p(tw), _p(gammas), L_mean,
        _p(grad), _p(cst), _p(fhat), _p(ghat), Bt, Kob, _stream(x)), "bcbf_unicycle_constraints")
    return grad, cst, fhat, ghat


def unicycle_step(x, u, dt, L_true):
    """In-place explicit Euler step of the Ackermann plant (unicycle_move_to_pose.py:277-282)."""
    _chk(x, u)
    check(getattr(lib, "bcbf_unicycle_step" + _suf(x))(_p(x), _p(u), dt, L_true, x.shape[0], _stream(x)),
          "bcbf_unicycle_step")
    return x


def rollout_stats(cst, y, status, w, gammas, min_h, cost, fails):
    """Safety bookkeeping of one closed-loop step (bcbf_rollout_stats): min_h, cost, fails updated in place."""
    _chk(cst, y, w, gammas, min_h, cost)
    Bt, Kob = cst.shape[0], cst.shape[1] - 1
    check(getattr(lib, "bcbf_rollout_stats" + _suf(cst))(_p(cst), _p(y), _p(status), _p(w), _p(gammas), _p(min_h), _p(cost),
                                                         _p(fails), Bt, Kob, y.shape[1], _stream(cst)), "bcbf_rollout_stats")


def _control_step_args(gp, task, ws, x):
    """Argument checks shared by the two forms below: returns (gp with every key present, A[Bt,n,n], N, shared)."""
    Bt = x.shape[0]
    if gp.get("Lop") is None:              # fixed-kernel model: ws["Mk"], ws["Bk"] are inputs, gp carries only A
        _chk(x, gp["A"], task["plan"], ws["y"], ws["Mk"], ws["Bk"])
        N, shared = 0, False
        gp = dict(gp, Lop=None, Vw=None, X=None, UHB=None, ell=None, s2=None, Bm=None, M0=None)
    else:
        _chk(x, gp["Lop"], gp["Vw"], gp["X"], gp["UHB"], task["plan"], ws["y"])
        N = gp["X"].shape[1]
        shared = gp["X"].shape[0] == 1 and Bt > 1
        if not shared and gp["X"].shape[0] != Bt:
            raise ValueError("GP tensors must carry a leading axis of 1 (shared model) or Bt")
    A = gp["A"]
    if A.shape[0] != Bt:                     # per-instance kernel matrix A for the fused terms+SOCP kernel
        if ws.get("A_shared_src") is not A:
            ws["A_shared"], ws["A_shared_src"] = A.expand(Bt, *A.shape[1:]).contiguous(), A
        A = ws["A_shared"]
    return gp, A, N, shared


def unicycle_control_step(gp, task, ws, x, dt=0.0, L_true=1.0, L_mean=1.0, clf_gamma=10.0, max_iters=100,
                          ev_start=None, ev_stop=None):
    """One control step for a batch of unicycle instances in ONE host call
    (ControllerCLFBayesian.control, unicycle_move_to_pose.py:926-995).

    gp:   dict(Lop, Vw, X, UHB, ell, s2, Bm, M0, A), or dict(A) alone for the fixed-kernel model (then ws['Mk'], ws['Bk']
    are inputs: 0 and I);  task: dict(plan, dot_plan, Kp, centers, radii, tw, gammas,
    w, r, sign, relax_mask, rho);  ws: dict of workspaces (grad, cst, fhat, ghat, Mk, Bk, cones, cstatus, y,
    status, iters) from `control_workspace`.  x[Bt,3] is advanced in place when dt > 0.  Returns ws['y'].
    GP tensors with a leading axis of 1 and Bt > 1 = one learned model shared by all instances (Monte-Carlo
    rollouts of a fixed model): the posterior runs as a shared query (fp32: the matrix-core kernel)."""
    if task["centers"].shape[1] + 1 > 4:       # more constraints than the fused kernel's four lanes: composed path
        return _unicycle_control_step_composed(gp, task, ws, x, dt, L_true, L_mean, clf_gamma, max_iters)
    return unicycle_control_step_prepare(gp, task, ws, x, dt, L_true, L_mean, clf_gamma, max_iters)(ev_start, ev_stop)


def _unicycle_control_step_composed(gp, task, ws, x, dt, L_true, L_mean, clf_gamma, max_iters):
    """The control step as separate entry points (task rows -> posterior -> terms -> generic cone solver -> plant step on
    the solved instances): for programs with more than four constraints."""
    gp, A, N, shared = _control_step_args(gp, task, ws, x)
    unicycle_constraints(x, task["plan"], task["dot_plan"], task["Kp"], clf_gamma, task["centers"], task["radii"],
                         task["tw"], task["gammas"], L_mean, out=(ws["grad"], ws["cst"], ws["fhat"], ws["ghat"]))
    kernel = gp.get("kernel", "rbf")
    if gp["Lop"] is not None:
        if shared or kernel != "rbf":
            Mk, Bk, _ = posterior_query(gp["Lop"], gp["Vw"], gp["X"], gp["UHB"], gp["ell"], gp["s2"], gp["Bm"], gp["M0"], x,
                                        shared=shared, kernel=kernel)
            ws["Mk"].copy_(Mk); ws["Bk"].copy_(Bk)
        else:
            posterior_step(gp["Lop"], gp["Vw"], gp["X"], gp["UHB"], gp["ell"], gp["s2"], gp["Bm"], gp["M0"], x,
                           out=(ws["Mk"], ws["Bk"]))
    y, status, iters, cones, cstatus, _ = cbc_socp(ws["Mk"], ws["Bk"], A, ws["grad"], ws["cst"], task["sign"], ws["fhat"],
                                                   ws["ghat"], task["w"], task["r"], task["relax_mask"], task["rho"],
                                                   max_iters=max_iters)
    ws["y"].copy_(y); ws["status"].copy_(status); ws["iters"].copy_(iters)
    ws["cones"].copy_(cones); ws["cstatus"].copy_(cstatus)
    if dt > 0:
        u = torch.where((status == 0)[:, None], y[:, :2], torch.zeros_like(y[:, :2])).contiguous()
        unicycle_step(x, u, dt, L_true)        # unsolved instances: zero displacement (they keep their state)
    return ws["y"]


def unicycle_control_step_prepare(gp, task, ws, x, dt=0.0, L_true=1.0, L_mean=1.0, clf_gamma=10.0, max_iters=100,
                                  stream=None, observe=None):
    """Bind every argument of `unicycle_control_step` once and return `step(ev_start=None, ev_stop=None)`.
    A closed loop calls the same entry point with the same buffers thousands of times; converting ~40 tensors to
    pointers per call costs more host time than the two launches take on the device for small batches.  The tensors
    must keep their storage (update them in place); the closure keeps them alive.
    stream: a fixed torch stream for both launches (default: the current stream at call time).
    observe = dict(xq=[Bt,3] | None, xq_next=[Bt,3] | None, shift_invariant=True, advance_plan=False): the loop LEARNS FROM ITSELF
    (bcbf_unicycle_control_step_observe; RBF models): the posterior is queried at `xq`, and the solve / plant launch writes this
    step's observation row where the call says -- `step(ev_start, ev_stop, obs=(obs_x, obs_uh, obs_y, ld))`, three-column
    tensors whose row b * ld is instance b's (ld = 1: [Bt,3] tensors) -- and the next query into `xq_next`."""
    gp, A, N, shared = _control_step_args(gp, task, ws, x)
    Bt = x.shape[0]
    Kob = task["centers"].shape[1]
    kernel = gp.get("kernel", "rbf")                       # data kernel of the learned model: "rbf" (the reference's) | "matern52"
    if kernel not in DATA_KERNELS:
        raise ValueError("gp['kernel'] must be one of %s" % (DATA_KERNELS,))
    if observe is not None and kernel != "rbf":
        raise NotImplementedError("the observing control step is built for the reference's RBF data kernel")
    fn = getattr(lib, ("bcbf_unicycle_control_step_observe" if observe is not None else "bcbf_unicycle_control_step" + _KSUF[kernel]) + _suf(x))
    head = (_p(gp["Lop"]), _p(gp["Vw"]), _p(gp["X"]), _p(gp["UHB"]), _p(gp["ell"]), _p(gp["s2"]), _p(gp["Bm"]),
            _p(gp["M0"]), _p(A), _p(x), _p(task["plan"]), _p(task["dot_plan"]), _p(task["Kp"]), clf_gamma,
            _p(task["centers"]), _p(task["radii"]), _p(task["tw"]), _p(task["gammas"]), L_mean, _p(task["w"]),
            _p(task["r"]), _p(task["sign"]), _p(task["relax_mask"]), _p(task["rho"]), _p(ws["grad"]), _p(ws["cst"]),
            _p(ws["fhat"]), _p(ws["ghat"]), _p(ws["Mk"]), _p(ws["Bk"]), _p(ws["cones"]), _p(ws["cstatus"]), _p(ws["y"]),
            _p(ws["status"]), _p(ws["iters"]), dt, L_true, Bt, N, Kob, max_iters, 1 if shared else 0)
    keep = (dict(gp), dict(task), dict(ws), x, A, stream, observe)      # the pointers above are only valid while these live
    dev, y = x.device, ws["y"]
    fixed = ctypes.c_void_p(stream.cuda_stream) if stream is not None else None

    if observe is not None:
        xq, xq_next = observe.get("xq"), observe.get("xq_next")
        _chk(x, xq, xq_next)
        p_xq, p_next = _p(xq), _p(xq_next)
        si = (1 if observe.get("shift_invariant", True) else 0) | (2 if observe.get("advance_plan", False) else 0)   # the entry's `flags`

        def step(ev_start=None, ev_stop=None, obs=None):
            ev0 = ctypes.c_void_p(ev_start.cuda_event) if ev_start is not None else None
            ev1 = ctypes.c_void_p(ev_stop.cuda_event) if ev_stop is not None else None
            if obs is None:
                o = (None, None, None, 1)
            else:
                o = (_p(obs[0]), _p(obs[1]), _p(obs[2]), int(obs[3]))
            rc = fn(*head, p_xq, *o, p_next, si, ev0, ev1,
                    fixed if fixed is not None else ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            if rc:
                check(rc, "bcbf_unicycle_control_step_observe")
            return y
        step.keep = keep
        return step

    def step(ev_start=None, ev_stop=None):
        ev0 = ctypes.c_void_p(ev_start.cuda_event) if ev_start is not None else None
        ev1 = ctypes.c_void_p(ev_stop.cuda_event) if ev_stop is not None else None
        rc = fn(*head, ev0, ev1, fixed if fixed is not None else ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc:
            check(rc, "bcbf_unicycle_control_step")
        return y
    step.keep = keep
    return step


class ConcurrentControlLoop:
    """The unicycle control step for a batch split into `parts` part batches, each on its OWN HIP stream (posterior
    launch, then solve launch, in stream order).  Instances never interact, so this is the same computation as
    `unicycle_control_step` on the whole batch -- every instance takes one control step per `step()` -- but the device
    runs one part's solve launch (task rows + terms + SOCP + plant step: latency bound, one wave per CU) beside another
    part's posterior stream (HBM bound), and a part's posterior fills the tail of the other's.  At the BASELINE config the
    serialized solve launch is 15-19 % of a single-stream step.  BASELINE config, ms per step: 1 part 0.408, 2 parts
    0.358, 3 parts 0.314, 4 parts 0.309 -- PROVIDED the runtime has a hardware queue per part stream: ROCm maps the
    streams of a process onto GPU_MAX_HW_QUEUES (default 4) queues, the null stream included, and two streams that share
    a queue serialize (4 parts at the default: 0.446).  Export GPU_MAX_HW_QUEUES=8 before HIP initialises for four parts
    (bench.py does), use three otherwise.  From three parts on the solves are hidden completely and the staggered
    posterior launches stream at a higher rate than one big launch (0.90 of the HBM peak against 0.84).  (An event-chained variant -- all posterior launches on one
    stream, solves on side streams -- was measured and rejected: each cross-stream dependency costs ~10 us on this
    stack, which ate the whole gain.)  gp / task tensors with a leading axis of Bt are sliced per part; `x` [Bt,3] is
    advanced in place.

    step(events=None): events = [(ev_start, ev_stop)] per part brackets that part's posterior kernel on its stream.
    Results: `y`, `status`, `iters` ([Bt,...] buffers); call `synchronize()` before reading them on another stream."""

    GP_INSTANCE_KEYS = ("Lop", "Vw", "X", "UHB", "ell", "s2", "Bm", "M0")       # + "A" when it carries the batch axis
    TASK_INSTANCE_KEYS = ("x", "x0", "xg", "plan", "dot_plan", "centers", "radii", "w", "r", "rho")

    def __init__(self, gp, task, x, parts=2, dt=0.0, L_true=1.0, L_mean=1.0, clf_gamma=10.0, max_iters=100):
        Bt = x.shape[0]
        if not 1 <= parts <= Bt:
            raise ValueError("cannot split a batch of %d into %d parts" % (Bt, parts))
        dev = x.device
        # contiguous part batches, the first Bt % parts of them one instance longer (4096 -> 1366 + 1365 + 1365)
        base, rem = divmod(Bt, parts)
        self.bounds = [(c * base + min(c, rem), (c + 1) * base + min(c + 1, rem)) for c in range(parts)]
        self.parts, self.x, self.device = parts, x, dev
        Kob = task["centers"].shape[1]
        self.ws = control_workspace(Bt, Kob, x.dtype, dev)
        self.y, self.status, self.iters = self.ws["y"], self.ws["status"], self.ws["iters"]
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
        self._steps = []
        shared_model = gp.get("Lop") is not None and gp["X"].shape[0] == 1 and Bt > 1
        cur = torch.cuda.current_stream(dev)
        A = gp["A"]                                  # [Bt,n,n], or [1,n,n] = one kernel matrix for every instance
        a_key = ("A",) if (A.dim() == 3 and A.shape[0] == Bt and Bt > 1) else ()
        gp_keys = a_key if shared_model else self.GP_INSTANCE_KEYS + a_key
        for c in range(parts):
            sl = slice(*self.bounds[c])
            # per-instance tensors are named, not inferred from their shape (with a small batch a global task tensor --
            # Kp[3], sign[3], tw[Kob], gammas[Kob], relax_mask[3] -- can have a leading dimension equal to Bt)
            def cut(k, v, keys):
                if not (torch.is_tensor(v) and k in keys):
                    return v
                if v.dim() == 0 or v.shape[0] != Bt:
                    raise ValueError("%s: expected a per-instance tensor with leading dimension %d, got %s" % (k, Bt, tuple(v.shape)))
                return v[sl]
            gpc = {k: cut(k, v, gp_keys) for k, v in gp.items()}
            taskc = {k: cut(k, v, self.TASK_INSTANCE_KEYS) for k, v in task.items()}
            wsc = {k: v[sl] for k, v in self.ws.items()}
            self._steps.append(unicycle_control_step_prepare(
                gpc, taskc, wsc, x[sl], dt=dt, L_true=L_true, L_mean=L_mean, clf_gamma=clf_gamma, max_iters=max_iters,
                stream=self.streams[c]))
            self.streams[c].wait_stream(cur)       # whatever produced the inputs on the current stream happens first

    def step(self, events=None):
        for c, st in enumerate(self._steps):
            if events is None:
                st()
            else:
                st(events[c][0], events[c][1])

    def synchronize(self):
        for s in self.streams:
            s.synchronize()


def control_workspace(Bt, Kob, dtype, device, n=3, m=2):
    K = 1 + Kob
    f = dict(dtype=dtype, device=device)
    i = dict(dtype=torch.int32, device=device)
    return dict(grad=torch.empty(Bt, K, n, **f), cst=torch.empty(Bt, K, **f), fhat=torch.empty(Bt, n, **f),
                ghat=torch.empty(Bt, n, m, **f), Mk=torch.empty(Bt, n, 1 + m, **f), Bk=torch.empty(Bt, 1 + m, 1 + m, **f),
                cones=torch.empty(Bt, K, cone_width(m), **f), cstatus=torch.empty(Bt, K, **i),
                y=torch.empty(Bt, m + 1, **f), status=torch.empty(Bt, **i), iters=torch.empty(Bt, **i))
