"""Development: the online loop's fused append + query kernel IN the loop (first 80 launches: two windows of the schedule) and the
same call on the same object in a plain loop afterwards (N growing 472 -> 511 in both), one process -- for a counter pass
(tools/dev/diag_clock.sh) that compares cycles, clock and bytes per launch between the two.
CAREFUL when reading its output: resetting `rgp.N` leaves the diagonal block's rows of the first round in place, so from the second
round on every pivot fails and the row kernel writes no operator row -- rounds 2 and 3 are the pass WITHOUT the previous append's
dirty lines (0.35 ms), round 1 and the loop are the real thing (0.45 ms).  That difference is the finding (DESIGN_NOTES, round 5)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bayesian_cbf_amd.rollouts import learning_closed_loop
out, final = learning_closed_loop(4096, 512, 80, 40, warmup=0, dtype=torch.float32, device="cuda", seed=1234)
rgp, p, x, ws = final["rgp"], final["p"], final["x"], final["ws"]
print("loop pass_ms", round(out["shares"]["pass_ms_per_step"], 4), "N", rgp.N, flush=True)
obs = [t.transpose(0, 1).contiguous() for t in (p["X"], p["UH"], p["Xdot"], p["jitter"])]
torch.cuda.synchronize()
N0 = 472
for rep in range(3):
    rgp.N = N0
    for k in range(472, 511):
        rgp.append(obs[0][k], obs[1][k], obs[2][k], obs[3][k], query=x, out=(ws["Mk"], ws["Bk"]))
torch.cuda.synchronize()
