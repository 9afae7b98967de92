import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bayesian_cbf_amd import ops, _lib
from bayesian_cbf_amd.synthetic import make_instances
lib = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libbcbf.so"))
os.environ["BCBF_REFIT_SLAB"] = "1"
names = ["stage+barrierA", "values", "slab loop", "factor J0 (wave 0)", "barrier B + solve J0", "barrier C + inner update", "factor J1 wait (D)", "solve J1", "fence + barrier E"]
for Bt in (1, 512, 4096):
    p = make_instances(Bt, 512, 3, 2, dtype=torch.float32, device="cuda", seed=3)
    ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"]); torch.cuda.synchronize()
    out = (ctypes.c_longlong * 16)()
    lib.bcbf_debug_rs_prof(out, 1)
    ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"]); torch.cuda.synchronize()
    lib.bcbf_debug_rs_prof(out, 1)
    tot = sum(out[:9])
    print("Bt", Bt, "total us", tot / 100.0)
    for k in range(9): print("   %-28s %8.1f us" % (names[k], out[k] / 100.0))
