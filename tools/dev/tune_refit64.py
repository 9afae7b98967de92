#!/usr/bin/env python3
"""Variant timing of the fp64 MFMA refit kernel (development tool): build / run."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VDIR = os.path.join(ROOT, "tools", "_variants")
CSRC = os.path.join(ROOT, "bayesian_cbf_amd", "csrc")
VARIANTS = {}
VARIANTS["default"] = []
VARIANTS["nofactor"] = ["-DBCBF_R64_ABL_NOFACTOR"]
def build():
    os.makedirs(VDIR, exist_ok=True)
    ps = []
    for name, fl in VARIANTS.items():
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
               "-I" + CSRC] + fl + [os.path.join(CSRC, "refit_mfma64.hip"), os.path.join(CSRC, "common.hip"), "-o", os.path.join(VDIR, "r64_" + name + ".so")]
        ps.append(subprocess.Popen(cmd, stderr=subprocess.DEVNULL))
        if len(ps) >= 6:
            assert all(p.wait() == 0 for p in ps); ps = []
    assert all(p.wait() == 0 for p in ps)
def run():
    import torch
    sys.path.insert(0, ROOT)
    from bayesian_cbf_amd import ops
    from bayesian_cbf_amd.synthetic import make_instances
    P = ctypes.c_void_p
    for (Bt, N, n, m) in ((1024, 256, 2, 1), (4096, 512, 3, 2)):
        p = make_instances(Bt, N, n, m, dtype=torch.float64, device="cuda", seed=5)
        C = m + 1
        Lop = torch.empty(Bt, ops.lop_elems(N, torch.float64), dtype=torch.float64, device="cuda")
        UHB = torch.empty(Bt, N, C, dtype=torch.float64, device="cuda"); info = torch.empty(Bt, dtype=torch.int32, device="cuda")
        q = lambda t: P(t.data_ptr())
        for name in VARIANTS:
            lib = ctypes.CDLL(os.path.join(VDIR, "r64_" + name + ".so"))
            def call():
                rc = lib.bcbf_refit_mfma_f64(q(p["X"]), q(p["UH"]), q(p["Bm"]), q(p["ell"]), q(p["s2"]), q(p["jitter"]), None, q(Lop), q(UHB), None, q(info),
                                             Bt, N, n, m, P(torch.cuda.current_stream().cuda_stream)); assert rc == 0
            call(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): call()
            e1.record(); torch.cuda.synchronize()
            print("N=%4d %-10s %.3f ms  fails=%d" % (N, name, e0.elapsed_time(e1) / 3, int((info != 0).sum())))
if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
