#!/usr/bin/env python3
"""Kernel-tuning harness for the posterior-step kernel (development tool, not product code).

  build : python tools/dev/tune_posterior.py build          (here; hipcc cross-compiles)
  run   : python tools/dev/tune_posterior.py run            (on the GPU box; interleaved rounds, one process)

Each variant is the same source compiled with different -D knobs into tools/_variants/<name>.so.
"""
import ctypes
import itertools
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VDIR = os.path.join(ROOT, "tools", "_variants")
CSRC = os.path.join(ROOT, "bayesian_cbf_amd", "csrc")

VARIANTS = {"base": []}
for u in (2, 4, 8):
    for w in (1, 2, 3, 4):
        VARIANTS["u%d_w%d" % (u, w)] = ["-DBCBF_PS_UNR=%d" % u, "-DBCBF_PS_WAVES=%d" % w]


def build():
    os.makedirs(VDIR, exist_ok=True)
    procs = []
    for name, flags in VARIANTS.items():
        out = os.path.join(VDIR, name + ".so")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + flags + [
               os.path.join(CSRC, "posterior_step.hip"), os.path.join(CSRC, "posterior_shared.hip"), os.path.join(CSRC, "common.hip"), "-mllvm", "-amdgpu-mfma-vgpr-form", "-o", out,
               "-Rpass-analysis=kernel-resource-usage"]
        procs.append((name, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
    for name, p in procs:
        _, err = p.communicate()
        if p.returncode != 0:
            print(err)
            raise SystemExit("build failed: " + name)
        lines = err.splitlines()
        for i, l in enumerate(lines):
            if "Function Name" in l and "IfLi3ELi4ELi0" in l:
                info = " ".join(x.split("remark:")[1].strip().replace(" [-Rpass-analysis=kernel-resource-usage]", "")
                                for x in lines[i + 1:i + 12] if ("VGPRs:" in x or "ScratchSize" in x or "Occupancy" in x))
                print(name, info)


def run():
    import torch
    sys.path.insert(0, ROOT)
    from bayesian_cbf_amd import ops
    from bayesian_cbf_amd.synthetic import make_instances
    f64 = "f64" in sys.argv
    Bt, N, n, m = (1024, 256, 2, 1) if "c2" in sys.argv else (4096, 512, 3, 2)
    dt = torch.float64 if f64 else torch.float32
    p = make_instances(Bt, N, n, m, dtype=dt, device="cuda", seed=1234)
    Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
    Vw, _ = ops.potrs(Lop, p["Xdot"], p["UH"], p["M0"], want_alpha=False)
    Mk_ref, Bk_ref = ops.posterior_step(Lop, Vw, p["X"], UHB, p["ell"], p["s2"], p["Bm"], p["M0"], p["xq"])
    torch.cuda.synchronize()
    libs = {}
    for name in VARIANTS:
        path = os.path.join(VDIR, name + ".so")
        if os.path.exists(path):
            libs[name] = ctypes.CDLL(path)
    P = ctypes.c_void_p
    st = P(torch.cuda.current_stream().cuda_stream)
    Mk = torch.empty_like(Mk_ref)
    Bk = torch.empty_like(Bk_ref)

    def call(lib):
        rc = getattr(lib, "bcbf_posterior_step_f64" if f64 else "bcbf_posterior_step_f32")(P(Lop.data_ptr()), P(Vw.data_ptr()), P(p["X"].data_ptr()), P(UHB.data_ptr()),
                                         P(p["ell"].data_ptr()), P(p["s2"].data_ptr()), P(p["Bm"].data_ptr()),
                                         P(p["M0"].data_ptr()), P(p["xq"].data_ptr()), None, P(Mk.data_ptr()),
                                         P(Bk.data_ptr()), Bt, N, n, m, st)
        assert rc == 0
    bytes_alg = (8 if f64 else 4) * (N * (N + 1) // 2 + N * (2 * n + 1 + m)) * Bt
    times = {k: [] for k in libs}
    for rnd in range(4):
        for name, lib in libs.items():
            call(lib)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call(lib)
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / 10)
            if rnd == 0:
                err = float((Bk - Bk_ref).abs().max()), float((Mk - Mk_ref).abs().max())
                print(name, "max diff vs shipped kernel: Bk %.2e Mk %.2e" % err)
    for name, ts in times.items():
        ts = sorted(ts)
        med = ts[len(ts) // 2]
        print("%-8s median %.1f us  min %.1f us  -> %.0f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
            name, med * 1e3, ts[0] * 1e3, bytes_alg / (med * 1e-3) / 1e9, bytes_alg / (med * 1e-3) / 8e12 * 100))


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
