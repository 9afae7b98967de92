#!/usr/bin/env python3
"""Development: per-section cycle counts / timings of the one-wave-per-instance fp64 refit for a set of -D variants.

  build : python tools/dev/prof_refit_wave.py build     (here; hipcc cross-compiles into tools/_variants/)
  run   : python tools/dev/prof_refit_wave.py run       (on the GPU box)

Variants named *_prof carry -DBCBF_RW64_PROF (cycle counters in Ldense[b][0][1..6])."""
import ctypes, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VDIR = os.path.join(ROOT, "tools", "_variants")
CSRC = os.path.join(ROOT, "bayesian_cbf_amd", "csrc")
VARIANTS = {
    "rw_base": [],
    "rw_prof": ["-DBCBF_RW64_PROF"],
}
if os.environ.get("BCBF_RW_SWEEP"):
    VARIANTS.update({"rw_occ2_ks4": ["-DBCBF_RW64_OCC=2", "-DBCBF_RW64_KS=4"], "rw_occ2_ks2": ["-DBCBF_RW64_OCC=2", "-DBCBF_RW64_KS=2"], "rw_ks4": ["-DBCBF_RW64_KS=4"],
                     "rw_wpb2": ["-DBCBF_RW64_WPB=2"], "rw_wpb4": ["-DBCBF_RW64_WPB=4"]})


def build():
    os.makedirs(VDIR, exist_ok=True)
    procs = []
    for name, flags in VARIANTS.items():
        out = os.path.join(VDIR, name + ".so")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + flags + [
               os.path.join(CSRC, f) for f in ("refit_wave64.hip", "refit_mfma64.hip", "common.hip")] + ["-o", out]
        procs.append((name, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
    for name, p in procs:
        _, err = p.communicate()
        if p.returncode != 0:
            print(err[-3000:])
            raise SystemExit("build failed: " + name)
        print("built", name)


def run():
    import torch
    sys.path.insert(0, ROOT)
    from bayesian_cbf_amd.synthetic import make_instances
    os.environ["BCBF_REFIT_WAVE"] = "1"
    P = ctypes.c_void_p
    shapes = [(1024, 256, 2, 1), (4096, 256, 2, 1), (4096, 512, 3, 2)]
    data = {}
    for Bt, N, n, m in shapes:
        p = make_instances(Bt, N, n, m, dtype=torch.float64, device="cuda", seed=5)
        E = (N + 31) // 32 * 32
        E = E * (E + 2) // 2 + 32 * E
        data[(Bt, N, n, m)] = (p, torch.empty(Bt, E, dtype=torch.float64, device="cuda"),
                               torch.empty(Bt, N, m + 1, dtype=torch.float64, device="cuda"),
                               torch.empty(Bt, N, N, dtype=torch.float64, device="cuda") if Bt * N * N * 8 < 12e9 else None,
                               torch.empty(Bt, dtype=torch.int32, device="cuda"))
    for name in VARIANTS:
        path = os.path.join(VDIR, name + ".so")
        if not os.path.exists(path):
            continue
        lib = ctypes.CDLL(path)
        fn = lib.bcbf_refit_mfma_f64
        fn.restype = ctypes.c_int
        for (Bt, N, n, m), (p, Lop, UHB, Ld, info) in data.items():
            prof = name.endswith("_prof")
            args = [P(p[k].data_ptr()) for k in ("X", "UH", "Bm", "ell", "s2", "jitter")] + [None, P(Lop.data_ptr()), P(UHB.data_ptr()),
                    P(Ld.data_ptr()) if (prof and Ld is not None) else None, P(info.data_ptr()), Bt, N, n, m, None]
            for _ in range(2):
                rc = fn(*args)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                rc = fn(*args)
            e1.record()
            torch.cuda.synchronize()
            row = dict(variant=name, batch=Bt, N=N, rc=rc, ms=e0.elapsed_time(e1) / 5, fails=int((info != 0).sum()))
            if prof and Ld is not None:
                c = Ld[:, 0, 1:7].mean(dim=0).cpu().tolist()
                names = ["stage", "kb_values", "update_stream", "factor", "inverse", "panel"]
                row["cycles"] = {k: round(v) for k, v in zip(names, c)}
                row["cycles_total"] = round(sum(c))
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    build() if "build" in sys.argv else run()
