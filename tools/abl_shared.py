#!/usr/bin/env python3
"""Ablation timing of the shared-GP MFMA kernel (development tool): build / run."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, "tools", "_variants")
CSRC = os.path.join(ROOT, "bayesian_cbf_amd", "csrc")
VARIANTS = {"full": [], "prof_nn": ["-DBCBF_PSH_PROFILE", "-DBCBF_PSH_ABL_NOLOAD", "-DBCBF_PSH_ABL_NOOFF"], "noload": ["-DBCBF_PSH_ABL_NOLOAD"], "profile": ["-DBCBF_PSH_PROFILE"], "nooff": ["-DBCBF_PSH_ABL_NOOFF"],
            "noload_nooff": ["-DBCBF_PSH_ABL_NOLOAD", "-DBCBF_PSH_ABL_NOOFF"]}
def build():
    os.makedirs(VDIR, exist_ok=True)
    ps = []
    for name, fl in VARIANTS.items():
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
               "-I" + CSRC, "-mllvm", "-amdgpu-mfma-vgpr-form"] + fl + [os.path.join(CSRC, "posterior_shared.hip"), os.path.join(CSRC, "common.hip"), "-o", os.path.join(VDIR, "psh_" + name + ".so")]
        ps.append(subprocess.Popen(cmd))
    assert all(p.wait() == 0 for p in ps)
def run():
    import torch
    sys.path.insert(0, ROOT)
    from bayesian_cbf_amd import ops
    from bayesian_cbf_amd.synthetic import make_instances
    P = ctypes.c_void_p
    N, n, m = 512, 3, 2
    p = make_instances(1, N, n, m, dtype=torch.float32, device="cuda", seed=1)
    Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
    Vw, _ = ops.potrs(Lop, p["Xdot"], p["UH"], p["M0"])
    q = lambda t: P(t.data_ptr())
    for b in (4, 4096):
        xq = (p["X"][0, torch.randint(0, N, (b,), device="cuda")] + 0.3 * torch.randn(b, n, device="cuda")).contiguous()
        Mk = torch.empty(b, n, m + 1, device="cuda"); Bk = torch.empty(b, m + 1, m + 1, device="cuda")
        for name in VARIANTS:
            lib = ctypes.CDLL(os.path.join(VDIR, "psh_" + name + ".so"))
            def call():
                rc = lib.bcbf_posterior_shared_f32(q(Lop), q(Vw), q(p["X"]), q(UHB), q(p["ell"]), q(p["s2"]), q(p["Bm"]), q(p["M0"]), q(xq), None,
                                                   q(Mk), q(Bk), None, b, N, n, m, P(torch.cuda.current_stream().cuda_stream)); assert rc == 0
            for _ in range(3): call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): call()
            e1.record(); torch.cuda.synchronize()
            print("b=%6d %-9s %8.1f us" % (b, name, e0.elapsed_time(e1) * 100))
            if name.startswith("prof"):
                print("   cycles staging / loop / whole kernel / diag steps / first steps:", Bk.flatten()[:5].tolist())
if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
