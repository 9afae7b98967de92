#!/usr/bin/env python3
"""Development: per-section cycle counts of the fp32 one-wave refit (plain and super-panel form) from a -DBCBF_RW64_PROF build.
   build: bash tools/build_variant.sh prof refit_wave64.hip -DBCBF_RW64_PROF      run (GPU box): python tools/dev/prof_refit32.py"""
import os, sys, json
os.environ["BCBF_LIB_PATH"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_variants", "libbcbf_prof.so")
os.environ["BCBF_REFIT_WAVE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
names = ["stage", "kb_values", "update_stream", "factor", "diag_rest", "panel"]
for Bt, N in ((1024, 512), (4096, 512), (1024, 1024)):
    p = make_instances(Bt, N, 3, 2, dtype=torch.float32, device="cuda", seed=5)
    for sup in ("0", "1"):
        for occ in ("1", "2"):
            os.environ["BCBF_RW32_SUPER_FORCE"], os.environ["BCBF_RW32_OCC"] = sup, occ
            for _ in range(3):
                Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
            e1.record()
            torch.cuda.synchronize()
            Np = (N + 31) // 32 * 32
            off = Np * (Np + 2) // 2 + 32 * 31
            c = Lop[:, off:off + 6].double().mean(dim=0).cpu().tolist()
            tot = sum(c)
            print(json.dumps(dict(batch=Bt, N=N, super=int(sup), occ=int(occ), ms=round(e0.elapsed_time(e1) / 5, 4), fails=int((info != 0).sum()),
                                  cycles_total=round(tot), share={k: round(v / tot, 3) for k, v in zip(names, c)})), flush=True)
