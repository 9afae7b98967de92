"""Development: bcbf_trtri / bcbf_syrk_lt / bcbf_refit alone at the batched fit's shapes (HIP events)."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayesian_cbf_amd import ops
from bayesian_cbf_amd._lib import lib
from bayesian_cbf_amd.synthetic import make_instances
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for dt, Bt, N in ((torch.float32, 4096, 512), (torch.float64, 4096, 512), (torch.float32, 1024, 256), (torch.float32, 256, 512)):
    p = make_instances(Bt, N, 3, 2, dtype=dt, device="cuda", seed=1)
    Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"] * 100)
    Linv = torch.empty(Bt, N, N, dtype=dt, device="cuda"); Kinv = torch.empty_like(Linv)
    t_tr = ev(lambda: ops.check(getattr(lib, "bcbf_trtri" + ops._suf(Lop))(ops._p(Lop), ops._p(Linv), Bt, N, ops._stream(Lop)), "trtri"))
    t_sy = ev(lambda: ops.check(getattr(lib, "bcbf_syrk_lt" + ops._suf(Lop))(ops._p(Linv), ops._p(Kinv), Bt, N, ops._stream(Lop)), "syrk"))
    fl = Bt * N ** 3 / 3
    print(json.dumps(dict(dtype=str(dt), batch=Bt, N=N, trtri_ms=round(t_tr, 3), trtri_tflops=round(fl / t_tr / 1e9, 1), syrk_ms=round(t_sy, 3), syrk_tflops=round(fl / t_sy / 1e9, 1))), flush=True)
