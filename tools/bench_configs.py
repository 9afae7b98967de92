#!/usr/bin/env python3
"""Timings of the non-headline BASELINE configs on one GPU (development / DESIGN.md numbers).
C2: batched kernel build + Cholesky + posterior, N=256, n=2, m=1, batch 1024, fp64.
C3 pieces in fp64.  C5: growing N with chol_append."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances


from _timing import timeit          # warms the clocks up first (tools/_timing.py)


def config(Bt, N, n, m, dtype, name):
    p = make_instances(Bt, N, n, m, dtype=dtype, device="cuda", seed=5)
    isz = p["X"].element_size()
    out = {"config": name, "batch": Bt, "N": N, "n": n, "m": m, "dtype": str(dtype)}
    t = timeit(lambda: ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"]))
    Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
    assert int((info != 0).sum()) == 0 or os.environ.get("BCBF_ABLATION"), "Cholesky failed"      # ablation builds factor garbage
    flops = Bt * (N ** 3 / 3.0)
    out["refit_ms"] = t
    out["refit_TFLOPs"] = flops / (t * 1e-3) / 1e12
    out["refit_write_GBs"] = Bt * ops.lop_elems(N, dtype) * isz / (t * 1e-3) / 1e9
    # roofline of the refit (SURVEY 8d, K1+K2): matrix-core bound -- N^3/3 flop per instance against the dense MFMA peak of
    # the dtype; algorithmic bytes = inputs in (N (n + 1 + m + 1) values) + the packed factor out, per instance
    peak = 78.6 if isz == 8 else 157.3
    refit_bytes = Bt * isz * (N * (n + 2 + m) + N * (N + 1) // 2)
    out["roofline"] = {"refit": dict(bound="mfma", kernel=os.environ.get("BCBF_REFIT_KERNEL_NAME", "refit_*_kernel<%s> (form chosen by the launcher)" % ("double" if isz == 8 else "float")),
                                     algorithmic_flops_per_launch=flops, achieved=out["refit_TFLOPs"], peak=peak, unit="TFLOP/s",
                                     frac=out["refit_TFLOPs"] / peak, algorithmic_bytes_per_launch=refit_bytes, traffic=None,
                                     kernel_ms=t)}
    t = timeit(lambda: ops.potrs(Lop, p["Xdot"], p["UH"], p["M0"], want_alpha=False))
    Vw, _ = ops.potrs(Lop, p["Xdot"], p["UH"], p["M0"], want_alpha=False)
    out["potrs_ms"] = t
    t = timeit(lambda: ops.posterior_step(Lop, Vw, p["X"], UHB, p["ell"], p["s2"], p["Bm"], p["M0"], p["xq"]), reps=20)
    by = Bt * isz * (N * (N + 1) // 2 + 2 * N * n + N * (1 + m))
    out["posterior_ms"] = t
    out["posterior_GBs_algorithmic"] = by / (t * 1e-3) / 1e9
    out["roofline"]["posterior"] = dict(bound="hbm", kernel="posterior_step_kernel<%s, %d, ...>" % ("double" if isz == 8 else "float", 1 + m),
                                        algorithmic_bytes_per_launch=by, achieved=out["posterior_GBs_algorithmic"], peak=8000.0,
                                        unit="GB/s", frac=out["posterior_GBs_algorithmic"] / 8000.0, traffic=None, kernel_ms=t)
    print(json.dumps(out))


CONFIGS = {"C2": (1024, 256, 2, 1, torch.float64, "C2"), "C3f64": (4096, 512, 3, 2, torch.float64, "C3 in fp64"),
           "C3": (4096, 512, 3, 2, torch.float32, "C3"), "N1024f64": (1024, 1024, 3, 2, torch.float64, "N=1024 fp64"),
           # the wide fp64 instantiations of the streaming kernel (m = 3: <double, 4, 4>; n > 4: <double, 3, 8>, <double, 4, 8>), next
           # to the same sizes at (n, m) = (3, 2) in "C3 in fp64": two columns per pipeline stage since round 5 (at four they spilled)
           "m3f64": (4096, 512, 3, 3, torch.float64, "fp64, m = 3 (posterior_step_kernel<double, 4, 4>)"),
           "n6f64": (4096, 512, 6, 2, torch.float64, "fp64, n = 6 (posterior_step_kernel<double, 3, 8>)"),
           "n6m3f64": (4096, 512, 6, 3, torch.float64, "fp64, n = 6, m = 3 (posterior_step_kernel<double, 4, 8>: 28 B scratch)")}
DEFAULT = ("C2", "C3f64", "C3", "N1024f64")

if __name__ == "__main__":
    for key in (sys.argv[1:] or list(DEFAULT)):          # e.g. `bench_configs.py C2` for a counter pass of one config
        config(*CONFIGS[key])
