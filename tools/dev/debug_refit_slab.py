import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
Bt, N, n, m = 8, 512, 3, 2
p = make_instances(Bt, N, n, m, dtype=torch.float32, device="cuda", seed=7)
def run(slab):
    if slab: os.environ["BCBF_REFIT_SLAB"] = "1"
    else: os.environ.pop("BCBF_REFIT_SLAB", None)
    out = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])[:3]
    torch.cuda.synchronize()
    return [o.clone().cpu() for o in out]
a, s = run(False), run(True)
Np = 512; NB = 32
def lop_base(j):
    J, c = j // NB, j % NB
    return NB * J * Np - 512 * J * (J + 1) + c * (Np - NB * (J + 1)) - NB * (J + 1)
offd = Np * Np // 2 - 16 * Np
print("info", s[2].tolist())
for J in range(0, 6):
    for I in range(J + 1, 16, 5):
        d = 0.0; mx = 0.0
        for c in range(32):
            b0 = lop_base(J * NB + c) + I * NB
            d = max(d, (a[0][0, b0:b0 + 32] - s[0][0, b0:b0 + 32]).abs().max().item()); mx = max(mx, a[0][0, b0:b0+32].abs().max().item())
        print("tile", I, J, "max diff", d, "max", mx)
    bd = offd + 544 * J
    print("dinv", J, (a[0][0, bd:bd + 528] - s[0][0, bd:bd + 528]).abs().max().item(), a[0][0, bd:bd+528].abs().max().item())
