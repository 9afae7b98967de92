"""Development: where one batched Adam iteration (4096 x 512) spends GPU time outside the bcbf kernels (torch profiler table)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from bayesian_cbf_amd.batched_fit import BatchedHyperFit
from bayesian_cbf_amd.synthetic import make_instances
from torch.profiler import profile, ProfilerActivity
dt = torch.float32 if (len(sys.argv) < 2 or sys.argv[1] == "f32") else torch.float64
p = make_instances(4096, 512, 3, 2, dtype=dt, device="cuda", seed=1)
bf = BatchedHyperFit.from_values(p["A"], p["Bm"], p["ell"], p["s2"], p["M0"])
bf.fit(p["X"], p["U"], p["Xdot"], training_iter=12)
bf.fit(p["X"], p["U"], p["Xdot"], training_iter=10)
torch.cuda.synchronize(); t0 = time.perf_counter()
bf.fit(p["X"], p["U"], p["Xdot"], training_iter=10)
torch.cuda.synchronize(); print("ms per iteration %.2f" % ((time.perf_counter() - t0) * 100))
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    bf.fit(p["X"], p["U"], p["Xdot"], training_iter=10)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=70))
