#!/usr/bin/env python3
"""Tuning harness of the jets instantiations of the posterior kernel (rel-degree-2 path; development tool).

  build : python tools/dev/tune_jets.py build      (here; hipcc cross-compiles; prints registers / scratch per variant)
  run   : python tools/dev/tune_jets.py run [f64]  (on the GPU box; interleaved rounds in one process)

Variants = columns per pipeline stage x occupancy target (-DBCBF_PJ_UNR32/64, -DBCBF_PJ_WAVES32/64) in tools/_variants/jets_*.so;
results are checked against the fp64 jets of the shipped library."""
import ctypes, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VDIR = os.path.join(ROOT, "tools", "_variants")
CSRC = os.path.join(ROOT, "bayesian_cbf_amd", "csrc")
VARIANTS = {"u%d_w%d" % (u, w): ["-DBCBF_PJ_UNR32=%d" % u, "-DBCBF_PJ_WAVES32=%d" % w, "-DBCBF_PJ_UNR64=%d" % u, "-DBCBF_PJ_WAVES64=%d" % w]
            for u, w in ((2, 1), (2, 2), (4, 1), (4, 2), (8, 1))}


def build():
    os.makedirs(VDIR, exist_ok=True)
    procs = []
    for name, flags in VARIANTS.items():
        out = os.path.join(VDIR, "jets_" + name + ".so")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
               "-I" + CSRC] + flags + [os.path.join(CSRC, f) for f in ("posterior_step.hip", "posterior_shared.hip", "common.hip")] + \
              [os.path.join(ROOT, "tools", "probe", "variant_stubs.hip")] + \
              ["-mllvm", "-amdgpu-mfma-vgpr-form", "-o", out, "-Rpass-analysis=kernel-resource-usage"]
        procs.append((name, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
    for name, p in procs:
        _, err = p.communicate()
        if p.returncode != 0:
            print(err[-3000:])
            raise SystemExit("build failed: " + name)
        lines = err.splitlines()
        for i, l in enumerate(lines):
            for tag, key in (("f32 n3m2", "IfLi3ELi4ELi3E"), ("f32 n2m1", "IfLi2ELi4ELi2E"), ("f64 n3m2", "IdLi3ELi4ELi3E"), ("f64 n2m1", "IdLi2ELi4ELi2E")):
                if "Function Name" in l and key in l:
                    blob = " ".join(lines[i + 1:i + 14])
                    g = lambda pat: re.search(pat, blob).group(1)
                    print("%-6s %s: vgpr %s agpr %s scratch %s occupancy %s" % (name, tag, g(r" VGPRs: (\d+)"), g(r"AGPRs: (\d+)"),
                          g(r"ScratchSize \[bytes/lane\]: (\d+)"), g(r"Occupancy \[waves/SIMD\]: (\d+)")))


def run():
    import torch
    sys.path.insert(0, ROOT)
    from bayesian_cbf_amd import ops
    from bayesian_cbf_amd.synthetic import make_instances
    f64 = "f64" in sys.argv
    dt = torch.float64 if f64 else torch.float32
    libs = {n_: ctypes.CDLL(os.path.join(VDIR, "jets_" + n_ + ".so")) for n_ in VARIANTS if os.path.exists(os.path.join(VDIR, "jets_" + n_ + ".so"))}
    P = ctypes.c_void_p
    for (Bt, N, n, m) in ((4096, 512, 3, 2), (4096, 512, 2, 1)) if not f64 else ((2048, 512, 3, 2), (1024, 256, 2, 1)):
        p = make_instances(Bt, N, n, m, dtype=dt, device="cuda", seed=3)
        Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
        Vw, _ = ops.potrs(Lop, p["Xdot"], p["UH"], p["M0"], want_alpha=False)
        ref = ops.posterior_jets(Lop, Vw, p["X"], UHB, p["ell"], p["s2"], p["Bm"], p["M0"], p["xq"])
        outs = [torch.empty_like(t) for t in ref]
        st = P(torch.cuda.current_stream().cuda_stream)

        def call(lib):
            rc = getattr(lib, "bcbf_posterior_jets_f64" if f64 else "bcbf_posterior_jets_f32")(
                *[P(t.data_ptr()) for t in (Lop, Vw, p["X"], UHB, p["ell"], p["s2"], p["Bm"], p["M0"], p["xq"])],
                *[P(t.data_ptr()) for t in outs], None, 0, Bt, N, n, m, st)
            assert rc == 0, rc
        by = Bt * p["X"].element_size() * (N * (N + 1) // 2 + 2 * N * n + N * (1 + m))
        times = {k: [] for k in libs}
        times["shipped"] = []
        shipped = lambda: ops.posterior_jets(Lop, Vw, p["X"], UHB, p["ell"], p["s2"], p["Bm"], p["M0"], p["xq"])
        for rnd in range(4):
            for name in list(libs) + ["shipped"]:
                fn = shipped if name == "shipped" else (lambda: call(libs[name]))
                fn(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record(); torch.cuda.synchronize()
                times[name].append(e0.elapsed_time(e1) / 10)
                if rnd == 0 and name != "shipped":
                    print(name, "max rel diff vs shipped:", ["%.1e" % float((a - b).abs().max() / b.abs().max()) for a, b in zip(outs, ref)])
        for name, ts in times.items():
            med = sorted(ts)[len(ts) // 2]
            print("N=%d n=%d m=%d %-8s median %.1f us  min %.1f us -> %.0f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
                N, n, m, name, med * 1e3, min(ts) * 1e3, by / (med * 1e-3) / 1e9, by / (med * 1e-3) / 8e12 * 100))


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
