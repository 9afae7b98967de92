"""ctypes binding of libbcbf.so (the C ABI of include/bcbf.h).  No fallbacks: if the HIP library
is missing or does not export a symbol, importing this module raises."""
import ctypes
import os

import torch  # noqa: F401  (loads the ROCm runtime first so libbcbf binds to the same libamdhip64)

_HERE = os.path.dirname(os.path.abspath(__file__))
# BCBF_LIB_PATH: development hook of the tuning tools (a variant build of the same sources); the product is the in-tree library
LIB_PATH = os.environ.get("BCBF_LIB_PATH") or os.path.join(_HERE, "libbcbf.so")

c_int, c_void_p, c_size_t = ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
c_float, c_double = ctypes.c_float, ctypes.c_double
P = c_void_p

# name -> (restype, argtypes); "{T}" entries are instantiated for f32 (c_float) and f64 (c_double)
_SIGS = {
    "bcbf_version": (c_int, []),
    "bcbf_last_error": (ctypes.c_char_p, []),
    "bcbf_hbm_read_probe": (c_int, [P, c_size_t, P, ctypes.POINTER(c_size_t), P]),
    "bcbf_lop_elems_f32": (c_size_t, [c_int]),
    "bcbf_lop_elems_f64": (c_size_t, [c_int]),
    "bcbf_posterior_shared_f32": (c_int, [P] * 13 + [c_int, c_int, c_int, c_int, P]),
    "bcbf_posterior_shared_f64": (c_int, [P] * 13 + [c_int, c_int, c_int, c_int, P]),
    "bcbf_posterior_shared_matern52_f32": (c_int, [P] * 13 + [c_int, c_int, c_int, c_int, P]),
    "bcbf_posterior_shared_rbfm52_f32": (c_int, [P] * 13 + [c_int, c_int, c_int, c_int, P]),
    "bcbf_posterior_shared_matern52_f64": (c_int, [P] * 13 + [c_int, c_int, c_int, c_int, P]),
    "bcbf_posterior_shared_rbfm52_f64": (c_int, [P] * 13 + [c_int, c_int, c_int, c_int, P]),
    "bcbf_controller_cones_rows": (c_int, [ctypes.POINTER(c_int), c_int, c_int, c_int]),
    "bcbf_controller_cones_f32": (c_int, [P, P, ctypes.POINTER(c_int), ctypes.POINTER(c_double), c_double, c_double, c_int, c_int, P, P, P, c_int, c_int, c_int, P]),
    "bcbf_controller_cones_f64": (c_int, [P, P, ctypes.POINTER(c_int), ctypes.POINTER(c_double), c_double, c_double, c_int, c_int, P, P, P, c_int, c_int, c_int, P]),
    "bcbf_mll_grad_work_bytes": (c_size_t, [c_int, c_int, c_int]),
    "bcbf_fit_param_count": (c_int, [c_int, c_int, c_int, c_int]),
    "bcbf_coneqp_f64": (c_int, [P, P, P, P, c_int, c_int, ctypes.POINTER(c_int), c_int, P, P, P, c_int, c_int, P]),
}
_TSIGS = {
    "bcbf_kb_build": [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "bcbf_kb_build_matern52": [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "bcbf_kb_build_rbfm52": [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "bcbf_posterior_query_matern52": [P] * 13 + [c_int, c_int, c_int, c_int, c_int, P],
    "bcbf_posterior_query_rbfm52": [P] * 13 + [c_int, c_int, c_int, c_int, c_int, P],
    "bcbf_refit": [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "bcbf_refit_retry": [P] * 10 + [c_int, c_int, c_int, c_int, P],
    "bcbf_refit_retry_kind": [P] * 10 + [c_int, c_int, c_int, c_int, c_int, P],
    "bcbf_gram": [P, P, P, c_int, c_int, c_int, c_int, P],
    "bcbf_predict_fullmat": [P] * 17 + [c_int] * 5 + [P],
    "bcbf_unicycle_control_step_observe": [P] * 13 + ["T"] + [P] * 4 + ["T"] + [P] * 16 + ["T", "T", c_int, c_int, c_int, c_int, c_int,
                                           P, P, P, P, c_int, P, c_int, P, P, P],
    "bcbf_potrf": [P, P, P, P, c_int, c_int, P],
    "bcbf_potrs": [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "bcbf_chol_append": [P, P, P, P, P, c_int, c_int, P],
    "bcbf_gp_append": [P] * 17 + [c_int, c_int, c_int, c_int, P],
    "bcbf_gp_append_stream": [P] * 20 + [c_int, c_int, c_int, c_int, P],
    "bcbf_gp_reserve": [P] * 8 + [c_int, c_int, c_int, c_int, c_int, c_int, P],
    "bcbf_posterior_query_reserved": [P] * 13 + [c_int, c_int, c_int, c_int, c_int, P],
    "bcbf_gp_append_reserved": [P] * 19 + [c_int, c_int, c_int, c_int, c_int, P],
    "bcbf_gp_append_reserved_raw": [P] * 22 + [c_int, c_int, c_int, c_int, c_int, P],
    "bcbf_gp_tail_step": [P] * 23 + [c_int] * 9 + [P],
    # the online entry points with kernel_kind (opt-in data kernels)
    "bcbf_gp_append_stream_kind": [P] * 20 + [c_int, c_int, c_int, c_int, c_int, P],
    "bcbf_posterior_query_reserved_kind": [P] * 13 + [c_int, c_int, c_int, c_int, c_int, c_int, P],
    "bcbf_gp_append_reserved_kind": [P] * 22 + [c_int, c_int, c_int, c_int, c_int, c_int, P],
    "bcbf_gp_tail_step_kind": [P] * 23 + [c_int] * 10 + [P],
    "bcbf_gp_tail_commit": [P, P, P, c_int, c_int, c_int, c_int, c_int, P],
    "bcbf_potri": [P, P, c_int, c_int, P],
    "bcbf_trtri": [P, P, c_int, c_int, P],
    "bcbf_mll_grad": [P] * 16 + [c_int, c_int, c_int, c_int, P, P],
    "bcbf_syrk_lt": [P, P, c_int, c_int, P],
    "bcbf_fit_derive": [P] * 8 + [c_int] * 5 + [P],
    "bcbf_fit_adam_step": [P] * 14 + [c_int] * 7 + [c_double] * 4 + [ctypes.POINTER(c_double), P],
    "bcbf_kinv_apply": [P, P, P, c_int, c_int, c_int, P],
    "bcbf_kb_build_rbflin": [P] * 8 + [c_int, c_int, c_int, c_int, P],
    "bcbf_posterior_query_rbflin": [P] * 14 + [c_int, c_int, c_int, c_int, c_int, P],
    "bcbf_mll_grad_rbflin": [P] * 18 + [c_int, c_int, c_int, c_int, c_int, P, P],
    "bcbf_posterior_step": [P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "bcbf_posterior_query": [P] * 13 + [c_int, c_int, c_int, c_int, c_int, P],
    "bcbf_posterior_jets": [P] * 14 + [c_int, c_int, c_int, c_int, c_int, P],
    "bcbf_cbc2_terms": [P] * 15 + [c_int, c_int, c_int, c_int, c_int, P],
    "bcbf_refit_matern52": [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "bcbf_refit_rbfm52": [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "bcbf_posterior_jets_matern52": [P] * 14 + [c_int, c_int, c_int, c_int, c_int, P],
    "bcbf_posterior_jets_rbfm52": [P] * 14 + [c_int, c_int, c_int, c_int, c_int, P],
    "bcbf_mll_grad_matern52": [P] * 16 + [c_int, c_int, c_int, c_int, P, P],
    "bcbf_mll_grad_rbfm52": [P] * 16 + [c_int, c_int, c_int, c_int, P, P],
    "bcbf_gp_append_matern52": [P] * 17 + [c_int, c_int, c_int, c_int, P],
    "bcbf_gp_append_rbfm52": [P] * 17 + [c_int, c_int, c_int, c_int, P],
    "bcbf_clean_hessian": [P, P, P, c_int, c_int, c_double, c_int, P],
    "bcbf_predict_assemble": [P] * 10 + [c_int] * 5 + [P],
    "bcbf_cbc_terms": [P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "bcbf_socp": [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "bcbf_cbc_socp": [P] * 18 + [c_int, c_int, c_int, c_int, c_int, P],
    "bcbf_unicycle_constraints": [P, P, P, P, "T", P, P, P, P, "T", P, P, P, P, c_int, c_int, P],
    "bcbf_unicycle_step": [P, P, "T", "T", c_int, P],
    "bcbf_rollout_stats": [P] * 8 + [c_int, c_int, c_int, P],
    "bcbf_unicycle_control_step": [P] * 13 + ["T"] + [P] * 4 + ["T"] + [P] * 16 + ["T", "T", c_int, c_int, c_int, c_int, c_int, P, P, P],
    "bcbf_unicycle_control_step_matern52": [P] * 13 + ["T"] + [P] * 4 + ["T"] + [P] * 16 + ["T", "T", c_int, c_int, c_int, c_int, c_int, P, P, P],
    "bcbf_unicycle_control_step_rbfm52": [P] * 13 + ["T"] + [P] * 4 + ["T"] + [P] * 16 + ["T", "T", c_int, c_int, c_int, c_int, c_int, P, P, P],
}


def declared_symbols():
    names = list(_SIGS)
    for base in _TSIGS:
        names += [base + "_f32", base + "_f64"]
    return names


def _build_if_missing():
    """A fresh checkout has no libbcbf.so (it is a build artefact): compile it once, under a file lock so that the
    ranks of a multi-process launch do not race.  No hipcc -> the ImportError below; there is no CPU fallback."""
    import fcntl
    with open(os.path.join(_HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not os.path.exists(LIB_PATH):
                from . import build as _build
                _build.build()
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _load():
    if not os.path.exists(LIB_PATH):
        try:
            _build_if_missing()
        except Exception as err:          # noqa: BLE001  (hipcc missing / compile error: report both facts)
            raise ImportError(
                "libbcbf.so not found at %s and building it failed (%s) -- build it with "
                "`python -m bayesian_cbf_amd.build` (hipcc, --offload-arch=gfx950).  There is no CPU fallback."
                % (LIB_PATH, err))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    for base, args in _TSIGS.items():
        for suf, ct in (("_f32", c_float), ("_f64", c_double)):
            fn = getattr(lib, base + suf)
            fn.restype = c_int
            fn.argtypes = [ct if a == "T" else a for a in args]
    return lib


lib = _load()


class BcbfError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        msg = lib.bcbf_last_error().decode() if rc == -2 else "invalid argument"
        raise BcbfError("%s failed (rc=%d): %s" % (what, rc, msg))
