#!/usr/bin/env python3
"""BASELINE configs[0] end to end: pendulum.learn_dynamics_matrix_vector (reference pendulum.py:1052-1271) -- simulate,
fit MVGP and CoGP on a random subset, evaluate on the 20x20 grid, log in the reference's event-file tags and print the
variance-weighted learning errors (published for N=200: CoGP 3.436 vs MVGP 0.659,
docs/saved-runs/learn_matrix_vector_v1.5.2-2-g31b30e1/vector_matrix_learning_error.txt)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bayesian_cbf_amd import tblog
from bayesian_cbf_amd.pendulum import learn_dynamics_matrix_vector_exp


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else "/tmp/learn_matrix_vector"      # ~20 MB of event file per run
    rows = []
    for N in (64, 200):
        for seed in range(3):
            torch.manual_seed(seed)
            np.random.seed(seed)
            logger = tblog.TBLogger(["learn_matrix_vector", "N%d" % N, "seed%d" % seed], runs_dir=out_dir)
            t0 = time.perf_counter()
            res = learn_dynamics_matrix_vector_exp(max_train=N, logger=logger, dtype=torch.float64)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            logger.summary_writer.close()
            back = tblog.load_tensorboard_scalars(logger.summary_writer.path)           # the log reads back
            assert back["log_learned_model/matrix/Fx/FX_learned"][0][1].shape == (20, 20, 2, 2)
            rows.append(dict(N_train=N, seed=seed, seconds=el, error_matrix=res["matrix"][2], error_vector=res["vector"][2],
                             fit_loss_matrix=[res["matrix"][0].fit_losses[0], res["matrix"][0].fit_losses[-1]],
                             fit_loss_vector=[res["vector"][0].fit_losses[0], res["vector"][0].fit_losses[-1]]))
            print(json.dumps(rows[-1]))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/learn_matrix_vector.jsonl", "w") as f:
        f.write("\n".join(json.dumps(r) for r in rows) + "\n")
    return rows


if __name__ == "__main__":
    main()
