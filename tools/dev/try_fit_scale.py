"""Development: cost of one batched Adam iteration of the marginal likelihood at C3 scale, and the learning loop with a fit per refit."""
import sys, os, json, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from bayesian_cbf_amd.batched_fit import BatchedHyperFit
from bayesian_cbf_amd.synthetic import make_instances
for Bt, N, dt in ((256, 512, torch.float64), (1024, 512, torch.float64), (4096, 512, torch.float64), (4096, 512, torch.float32), (1024, 256, torch.float64)):
    p = make_instances(Bt, N, 3, 2, dtype=dt, device="cuda", seed=1)
    bf = BatchedHyperFit.from_values(p["A"], p["Bm"], p["ell"], p["s2"], p["M0"])
    bf.fit(p["X"], p["U"], p["Xdot"], training_iter=12)       # (fp32: the per-instance jitter levels settle over the first iterations -- retries until then)
    el = None
    for _ in range(2):                                             # (second of two timed calls: the first still grows the allocator's pools when retries gather sub-batches)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bf.fit(p["X"], p["U"], p["Xdot"], training_iter=10)
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 10
    print(json.dumps(dict(batch=Bt, N=N, dtype=str(dt), ms_per_adam_iteration=el * 1e3, loss0=float(bf.losses[0].mean()), loss9=float(bf.losses[-1].mean()),
                          skipped=int(bf.skipped.sum()), level_max=float(bf.jitter_level.max()))), flush=True)
