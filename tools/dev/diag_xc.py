"""Development: the fused query + append-column kernel against the plain query kernel on the same storage (fp32, 4096 instances)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesian_cbf_amd import ops
from bayesian_cbf_amd.synthetic import make_instances
from _timing import timeit
Bt, dt = 4096, torch.float32
CASES = [(int(a.split(",")[0]), int(a.split(",")[1])) for a in sys.argv[1:]] or [(512, 2), (512, 3), (480, 2), (496, 2)]
for N, m in CASES:
    p = make_instances(Bt, N + 1, 3, m, dtype=dt, device="cuda", seed=3)
    cut = lambda t: t[:, :N].contiguous()
    Lop, UHB, info, _ = ops.refit(cut(p["X"]), cut(p["UH"]), p["Bm"], p["ell"], p["s2"], cut(p["jitter"]))
    Vw, _ = ops.potrs(Lop, cut(p["Xdot"]), cut(p["UH"]), p["M0"], want_alpha=False)
    X = cut(p["X"])
    t_plain = timeit(lambda: ops.posterior_step(Lop, Vw, X, UHB, p["ell"], p["s2"], p["Bm"], p["M0"], p["xq"]))
    out = dict(N=N, m=m, plain_ms=t_plain)
    if m == 2:
        cap = int(os.environ.get("DIAG_CAP", "0")) or (512 if N < 512 else 544)
        rgp = ops.ReservedGP(Lop, Vw, X, UHB, p["ell"], p["s2"], p["Bm"], p["M0"], cap)
        out["capacity"] = cap
        t_res = timeit(lambda: rgp.posterior(p["xq"]))
        xn, uhn, xdn, jn = (p[k][:, N].contiguous() for k in ("X", "UH", "Xdot", "jitter"))
        def fused():
            rgp.append(xn, uhn, xdn, jn, query=p["xq"])
            rgp.N = N                      # (same N again: the row is overwritten)
        t_fused = timeit(fused)
        def app():
            rgp.append(xn, uhn, xdn, jn)
            rgp.N = N
        t_app = timeit(app)
        out.update(reserved_query_ms=t_res, fused_append_query_ms=t_fused, append_only_ms=t_app)
    print(json.dumps(out), flush=True)
