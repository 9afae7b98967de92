#!/usr/bin/env python3
"""Time line of the two waves of one instance in the two-waves-per-instance refit (development).

    tools/build_variant.sh rptrace refit_wave64.hip -DBCBF_RP_TRACE
    BCBF_LIB_PATH=tools/_variants/libbcbf_rptrace.so python tools/trace_refit_pair.py [f32|f64] [batch] [N]

Prints, per block column, where the streamer (values / wait for the solver / update stream / hand-off) and the solver
(wait for the diagonal tile / factor + invert / stores / per panel tile: wait, solve, hand-off) of workgroup 0 spent
their time, in microseconds."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bayesian_cbf_amd import ops, _lib
from bayesian_cbf_amd.synthetic import make_instances

DT = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else torch.float64
Bt = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
N = int(sys.argv[3]) if len(sys.argv) > 3 else 256
os.environ["BCBF_REFIT_WAVE"] = "1"; os.environ["BCBF_REFIT_PAIR"] = "1"
n, m = (2, 1) if N <= 256 else (3, 2)
p = make_instances(Bt, N, n, m, dtype=DT, device="cuda", seed=5)
for _ in range(20):
    ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
torch.cuda.synchronize()
buf = np.zeros((2, 4096), dtype=np.int64)
rc = _lib.lib.bcbf_debug_rp_trace(ctypes.c_void_p(buf.ctypes.data)); assert rc == 0
nb = (N + 31) // 32
t0 = min(buf[0][0], buf[1][0])
us = lambda v: (v - t0) / 100.0
so, st = buf[0], buf[1]
i0 = i1 = 0
tot = dict(values=0.0, s_wait=0.0, update=0.0, s_pub=0.0, d_wait=0.0, factor=0.0, store=0.0, p_wait=0.0, panel=0.0)
print("col | streamer: start  values  wait  update  publish (per tile, us) | solver: diag-wait factor store | panels: wait solve+publish")
for J in range(nb):
    srow = []
    for I in range(J, nb):
        a = [us(st[i1 + k]) for k in range(5)]; i1 += 5
        srow.append("%d:%.1f v%.1f w%.1f u%.1f p%.1f" % (I, a[0], a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[3]))
        tot["values"] += a[1] - a[0]; tot["s_wait"] += a[2] - a[1]; tot["update"] += a[3] - a[2]; tot["s_pub"] += a[4] - a[3]
    c = [us(so[i0 + k]) for k in range(4)]; i0 += 4
    tot["d_wait"] += c[1] - c[0]; tot["factor"] += c[2] - c[1]; tot["store"] += c[3] - c[2]
    prow = []
    for I in range(J + 1, nb):
        q = [us(so[i0 + k]) for k in range(3)]; i0 += 3
        prow.append("%d:w%.1f s%.1f" % (I, q[1] - q[0], q[2] - q[1]))
        tot["p_wait"] += q[1] - q[0]; tot["panel"] += q[2] - q[1]
    print("J=%d S[%s]\n     V[@%.1f wait %.1f factor %.1f store %.1f | %s]" % (J, "  ".join(srow), c[0], c[1] - c[0], c[2] - c[1], c[3] - c[2], " ".join(prow)))
print("end: streamer %.1f us, solver %.1f us" % (us(st[i1 - 1]), us(so[i0 - 1])))
print({k: round(v, 1) for k, v in tot.items()})
