"""Development: the online loop's fused append + query call in a plain python loop -- same objects, wall clock -- with the loop's
ingredients added one at a time (argv: hwq = GPU_MAX_HW_QUEUES=8 before torch loads)."""
import os, sys, time
if "hwq" in sys.argv:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.rollouts import learning_closed_loop
out, final = learning_closed_loop(4096, 512, 40, 40, warmup=0, dtype=torch.float32, device="cuda", seed=1234)
rgp, p, x, ws = final["rgp"], final["p"], final["x"], final["ws"]
print("loop pass_ms", round(out["shares"]["pass_ms_per_step"], 4), "N", rgp.N, "hwq", os.environ.get("GPU_MAX_HW_QUEUES"), flush=True)
obs = [t.transpose(0, 1).contiguous() for t in (p["X"], p["UH"], p["Xdot"], p["jitter"])]
N0 = rgp.N
def run(tag, body, reps=120):
    for t in range(10):
        body(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(reps):
        body(t)
    torch.cuda.synchronize()
    print("%-60s %.4f ms per call" % (tag, (time.perf_counter() - t0) / reps * 1e3), flush=True)
def fixed_row(t):
    rgp.append(obs[0][500], obs[1][500], obs[2][500], obs[3][500], query=x, out=(ws["Mk"], ws["Bk"])); rgp.N = N0
def varying_row(t):
    k = 472 + t % 40
    rgp.append(obs[0][k], obs[1][k], obs[2][k], obs[3][k], query=x, out=(ws["Mk"], ws["Bk"])); rgp.N = N0
def growing(t):
    k = 472 + t % 39
    if t % 39 == 0:
        rgp.N = N0
    rgp.append(obs[0][k], obs[1][k], obs[2][k], obs[3][k], query=x, out=(ws["Mk"], ws["Bk"]))
E = lambda: torch.cuda.Event(enable_timing=True)
evs = [[E(), E(), E()] for _ in range(200)]
def with_events(t):
    e = evs[t]
    e[0].record()
    varying_row(t)
    e[1].record()
    e[2].record()
solve = ops.unicycle_control_step_prepare(dict(A=p["A"]), {k: v for k, v in final.items() if False} or None, ws, x) if False else None
def with_bool_accumulate(t):
    varying_row(t)
    acc.__iadd__(rgp.info != 0)
acc = torch.zeros(4096, dtype=torch.int32, device="cuda")
run("fixed obs row, fixed N", fixed_row)
run("varying obs row + 3 event records per call", with_events)
run("varying obs row + (info != 0) accumulate", with_bool_accumulate)
run("varying obs row, fixed N", varying_row)
run("varying obs row, N growing 472..510 (no drop)", growing)
rgp.N = N0
run("fixed obs row, fixed N (again)", fixed_row)
