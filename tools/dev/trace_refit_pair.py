#!/usr/bin/env python3
"""Time line of the two waves of one instance in the two-waves-per-instance refit (development).

    tools/build_variant.sh rptrace refit_wave64.hip -DBCBF_RP_TRACE
    BCBF_LIB_PATH=tools/_variants/libbcbf_rptrace.so python tools/dev/trace_refit_pair.py [f32|f64] [batch] [N]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
tot = dict(c_prep=0.0, c_wait_last=0.0, factor=0.0, c_store=0.0, b_first=0.0, b_wait=0.0, b_solve=0.0, b_rest=0.0)
print("J | chain: @start prepare wait+last-update factor store | bulk: @start first-S' wait solve+deliver rest")
for J in range(nb):
    a = [us(buf[0][5 * J + k]) for k in range(5)]
    line = "J=%d  chain @%.1f prep %.1f wait+last %.1f factor %.1f store %.1f" % (J, a[0], a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[3])
    tot["c_prep"] += a[1] - a[0]; tot["c_wait_last"] += a[2] - a[1]; tot["factor"] += a[3] - a[2]; tot["c_store"] += a[4] - a[3]
    if J + 1 < nb:
        q = [us(buf[1][5 * J + k]) for k in range(5)]
        line += "   | bulk @%.1f first %.1f wait %.1f solve %.1f rest %.1f" % (q[0], q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3])
        tot["b_first"] += q[1] - q[0]; tot["b_wait"] += q[2] - q[1]; tot["b_solve"] += q[3] - q[2]; tot["b_rest"] += q[4] - q[3]
    print(line)
print("end: chain %.1f us, bulk %.1f us" % (us(buf[0][5 * nb - 1]), us(buf[1][5 * (nb - 1) - 1])))
print({k: round(float(v), 1) for k, v in tot.items()})
# inside the first diagonal tile's factor + inverse (first launch): stamps (code << 56 | time)
st = buf[0][4000:4040]
if st[0]:
    tm = [(int(v) >> 56, (int(v) & ((1 << 56) - 1)) / 100.0) for v in st if v]
    names = {0: "publish", 1: "pivots4x4", 2: "row+publish", 3: "rank4", 4: "inverse-step"}
    prev = tm[0][1]
    agg = {}
    row = []
    for code, t in tm[1:]:
        row.append("%s %.2f" % (names[code], t - prev)); agg[names[code]] = agg.get(names[code], 0.0) + t - prev; prev = t
    print("diagonal tile, first call (us between stamps):", " | ".join(row))
    print({k: round(v, 2) for k, v in agg.items()})
