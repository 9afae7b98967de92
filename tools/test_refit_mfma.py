import ctypes, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bayesian_cbf_amd import ops
from bayesian_cbf_amd._lib import lib
from bayesian_cbf_amd.synthetic import make_instances
P = ctypes.c_void_p
lib.bcbf_refit_mfma_f32.restype = ctypes.c_int
def refit_mfma(X, UH, Bm, ell, s2, jitter, want_dense=False, Kdense=None):
    Bt, N, n = X.shape; C = UH.shape[2]
    Lop = torch.empty(Bt, ops.lop_elems(N, X.dtype), dtype=X.dtype, device=X.device)
    UHB = torch.empty(Bt, N, C, dtype=X.dtype, device=X.device)
    info = torch.empty(Bt, dtype=torch.int32, device=X.device)
    Ld = torch.empty(Bt, N, N, dtype=X.dtype, device=X.device) if want_dense else None
    p = lambda t: None if t is None else P(t.data_ptr())
    rc = lib.bcbf_refit_mfma_f32(p(X), p(UH), p(Bm), p(ell), p(s2), p(jitter), p(Kdense), p(Lop), p(UHB), p(Ld), p(info),
                                 Bt, N, n, C - 1, P(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    return Lop, UHB, info, Ld
for (Bt, N, n, m) in [(4, 64, 2, 1), (4, 100, 3, 2), (8, 512, 3, 2), (2, 1024, 3, 3)]:
    p = make_instances(Bt, N, n, m, dtype=torch.float32, device="cuda", seed=N)
    L1, U1, i1, D1 = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"], want_dense=True)
    L2, U2, i2, D2 = refit_mfma(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"], want_dense=True)
    torch.cuda.synchronize()
    Kb = ops.kb_build(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"]).double()
    rec1 = (D1.double() @ D1.double().transpose(1, 2) - Kb).abs().max().item()
    rec2 = (D2.double() @ D2.double().transpose(1, 2) - Kb).abs().max().item()
    print("Bt=%d N=%d: info %s %s | recon err VALU %.2e MFMA %.2e | Ldense diff %.2e | UHB diff %.2e | Lop diff %.2e (max |Lop| %.1f)" % (
        Bt, N, i1.tolist()[:3], i2.tolist()[:3], rec1, rec2, (D1 - D2).abs().max().item(), (U1 - U2).abs().max().item(),
        (L1 - L2).abs().max().item(), L1.abs().max().item()))
    # dense-input variant
    L3, _, i3, D3 = refit_mfma(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], None, want_dense=True, Kdense=Kb.float().contiguous())
    print("   from dense: info %s recon err %.2e" % (i3.tolist()[:3], (D3.double() @ D3.double().transpose(1, 2) - Kb).abs().max().item()))
p = make_instances(4096, 512, 3, 2, dtype=torch.float32, device="cuda", seed=1)
for name, fn in (("VALU", lambda: ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])),
                 ("MFMA", lambda: refit_mfma(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"]))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%s refit 4096x512: %.2f ms  (%.1f TFLOP/s of N^3/3), info ok %s" % (name, dt * 1e3, 4096 * 512**3 / 3 * 2 / dt / 1e12, bool((out[2] == 0).all())))
