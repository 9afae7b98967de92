"""Development: first runs of rollouts.self_learning_closed_loop (small + C3 scale)."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from bayesian_cbf_amd.rollouts import self_learning_closed_loop, final_model_vs_fp64_refit
for kw in (dict(Bt=24, max_train=120, steps=48, refit_every=24, parts=3, dtype=torch.float64, schedule="reference"),
           dict(Bt=24, max_train=120, steps=48, refit_every=24, parts=3, dtype=torch.float64, schedule="online_tail", mid_period_steps=7),
           dict(Bt=24, max_train=120, steps=48, refit_every=24, parts=3, dtype=torch.float32, schedule="reference"),
           dict(Bt=24, max_train=120, steps=48, refit_every=24, parts=3, dtype=torch.float32, schedule="online_tail", mid_period_steps=7),
           dict(Bt=4096, max_train=512, steps=200, refit_every=40, parts=4, dtype=torch.float32, schedule="reference"),
           dict(Bt=4096, max_train=512, steps=200, refit_every=40, parts=4, dtype=torch.float32, schedule="reference", stagger=False),
           dict(Bt=4096, max_train=512, steps=200, refit_every=40, parts=4, dtype=torch.float32, schedule="reference", shift_invariant=False),
           dict(Bt=4096, max_train=512, steps=200, refit_every=40, parts=4, dtype=torch.float32, schedule="online_tail"),
           dict(Bt=4096, max_train=512, steps=200, refit_every=40, parts=2, dtype=torch.float32, schedule="online_tail"),
           dict(Bt=4096, max_train=512, steps=200, refit_every=40, parts=4, dtype=torch.float32, schedule="reference", retry_levels=5)):
    try:
        rep, final = self_learning_closed_loop(**kw)
        chk = final_model_vs_fp64_refit(final)
        rep["final_vs_fp64_refit_on_device"] = chk
        rep.pop("roofline", None)
        print(json.dumps(rep))
    except Exception as e:
        import traceback; traceback.print_exc()
        print("FAILED", kw, repr(e)[:300])
    sys.stdout.flush()
