"""Development: refit time against the number of instances in flight (is the launch sensitive to the working set / MALL?)."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
for dt, N, batches in ((torch.float64, 256, (128, 256, 512, 768, 1024, 1536, 2048, 4096)), (torch.float32, 512, (512, 1024, 2048, 4096, 8192))):
    for Bt in batches:
        n, m = (2, 1) if N <= 256 else (3, 2)
        p = make_instances(Bt, N, n, m, dtype=dt, device="cuda", seed=5)
        out = None
        for _ in range(3):
            out = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"], out=out[:3])
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        isz = 8 if dt == torch.float64 else 4
        print(json.dumps(dict(dtype=str(dt), N=N, batch=Bt, ms=round(ms, 4), us_per_instance=round(ms / Bt * 1e3, 3), factor_set_MB=round(Bt * ops.lop_elems(N, dt) * isz / 1e6, 1),
                              tflops=round(Bt * N ** 3 / 3 / ms / 1e9, 2))), flush=True)
