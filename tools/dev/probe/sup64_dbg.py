import os, sys, torch, numpy as np, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
if len(sys.argv) > 1:
    from bayesian_cbf_amd import ops
    from bayesian_cbf_amd.synthetic import make_instances
    os.environ["BCBF_REFIT_WAVE"] = "1"
    p = make_instances(4, 1024, 3, 2, dtype=torch.float64, device="cuda", seed=5)
    Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
    np.save(sys.argv[1], Lop[0].cpu().numpy())
    print("info", info.tolist())
else:
    subprocess.run([sys.executable, __file__, "/tmp/a.npy"])
    subprocess.run([sys.executable, __file__, "/tmp/b.npy"], env=dict(os.environ, BCBF_RW64_SUPER_FORCE="0"))
    a, b = np.load("/tmp/a.npy"), np.load("/tmp/b.npy")
    Np = 1024
    bad = np.nonzero(~(np.abs(a - b) <= 1e-9 * (1 + np.abs(b))))[0]
    print("differing entries", len(bad), "first", bad[:10])
    # packed layout: column c holds rows 32*(c//32+1).. : offset of column c
    def base(c):
        K = c // 32
        # columns of block K have length Np - 32 (K+1)
        off = 0
        for k in range(K):
            off += 32 * (Np - 32 * (k + 1))
        return off + (c - 32 * K) * (Np - 32 * (K + 1))
    offs = np.array([base(c) for c in range(Np)])
    for e in bad[:10]:
        c = np.searchsorted(offs, e, side="right") - 1
        print("entry", e, "col", c, "row", 32 * (c // 32 + 1) + e - offs[c] if c < Np - 32 else "(dinv area)", a[e], b[e])
