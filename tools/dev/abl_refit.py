#!/usr/bin/env python3
"""Ablation timing of the MFMA refit kernel (development tool): build / run."""
import ctypes, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VDIR = os.path.join(ROOT, "tools", "_variants")
CSRC = os.path.join(ROOT, "bayesian_cbf_amd", "csrc")
VARIANTS = {"full": []}
for ks in (2, 4, 8):
    for occ in (3, 4):
        VARIANTS["t1_k%d_o%d" % (ks, occ)] = ["-DBCBF_R32_MAXT=1", "-DBCBF_R32_KS=%d" % ks, "-DBCBF_R32_OCC=%d" % occ]
def build():
    os.makedirs(VDIR, exist_ok=True)
    ps = []
    for name, fl in VARIANTS.items():
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
               "-I" + CSRC] + fl + [os.path.join(CSRC, "refit_mfma.hip"), os.path.join(CSRC, "common.hip"), "-o", os.path.join(VDIR, "rm_" + name + ".so")]
        ps.append(subprocess.Popen(cmd))
    assert all(p.wait() == 0 for p in ps)
def run():
    import torch
    sys.path.insert(0, ROOT)
    from bayesian_cbf_amd import ops
    from bayesian_cbf_amd.synthetic import make_instances
    P = ctypes.c_void_p
    p = make_instances(4096, 512, 3, 2, dtype=torch.float32, device="cuda", seed=1)
    Bt, N, n = p["X"].shape; C = 3
    Lop = torch.empty(Bt, ops.lop_elems(N, torch.float32), dtype=torch.float32, device="cuda")
    UHB = torch.empty(Bt, N, C, dtype=torch.float32, device="cuda"); info = torch.empty(Bt, dtype=torch.int32, device="cuda")
    q = lambda t: P(t.data_ptr())
    for name in VARIANTS:
        lib = ctypes.CDLL(os.path.join(VDIR, "rm_" + name + ".so"))
        def call():
            rc = lib.bcbf_refit_mfma_f32(q(p["X"]), q(p["UH"]), q(p["Bm"]), q(p["ell"]), q(p["s2"]), q(p["jitter"]), None, q(Lop), q(UHB), None, q(info),
                                         Bt, N, n, C - 1, P(torch.cuda.current_stream().cuda_stream)); assert rc == 0
        call(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): call()
        torch.cuda.synchronize()
        print("%-9s %.2f ms" % (name, (time.perf_counter() - t0) / 3 * 1e3))
if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
