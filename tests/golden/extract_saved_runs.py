#!/usr/bin/env python3
"""Extract the logged trajectories of the reference's committed runs into small .npz fixtures.

docs/saved-runs/unicycle_move_to_pose_fixed_mean_cbf_collides_{v1.2.3,1209-1257}/ hold TensorBoard
event files written by the reference's Logger (unicycle_move_to_pose.py:1288-1311) with, per
step, `vis/state[3]`, `vis/uopt[2]`, `vis/plan_x[3]`, `opt/value`, `opt/rho`, `vis/cbc_value`
(the optimiser was GUROBI through cvxpy).  These are data files of the reference (not source);
the parser below is a ~60-line TFRecord + protobuf-varint reader (no tensorboard needed).
Build-container only:  python tests/golden/extract_saved_runs.py
"""
import glob
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("BCBF_REFERENCE", "/root/reference")
RUNS = {
    "saved_run_mean_cbf_maxrisk0p5": "unicycle_move_to_pose_fixed_mean_cbf_collides_v1.2.3",
    "saved_run_bayes_cbf_maxrisk0p01": "unicycle_move_to_pose_fixed_mean_cbf_collides_1209-1257",
}


def varint(buf, i):
    shift = val = 0
    while True:
        b = buf[i]
        i += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, i
        shift += 7


def fields(buf):
    """Yield (field_number, wire_type, value) of one protobuf message."""
    i = 0
    while i < len(buf):
        key, i = varint(buf, i)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, i = varint(buf, i)
        elif wt == 1:
            v = buf[i:i + 8]
            i += 8
        elif wt == 2:
            ln, i = varint(buf, i)
            v = buf[i:i + ln]
            i += ln
        elif wt == 5:
            v = buf[i:i + 4]
            i += 4
        else:
            raise ValueError("wire type %d" % wt)
        yield fno, wt, v


def records(path):
    with open(path, "rb") as f:
        data = f.read()
    i = 0
    while i + 12 <= len(data):
        (ln,) = struct.unpack("<Q", data[i:i + 8])
        yield data[i + 12:i + 12 + ln]
        i += 12 + ln + 4


def parse_events(path):
    out = {}
    for rec in records(path):
        step, summary = 0, None
        for fno, wt, v in fields(rec):
            if fno == 2 and wt == 0:
                step = v
            elif fno == 5 and wt == 2:
                summary = v
        if summary is None:
            continue
        for fno, wt, val in fields(summary):
            if fno != 1:
                continue
            tag, simple, tensor = None, None, None
            for f2, w2, v2 in fields(val):
                if f2 == 1:
                    tag = v2.decode()
                elif f2 == 2 and w2 == 5:
                    simple = struct.unpack("<f", v2)[0]
                elif f2 == 8:
                    floats = []
                    for f3, w3, v3 in fields(v2):
                        if f3 == 5 and w3 == 2:
                            floats.extend(struct.unpack("<%df" % (len(v3) // 4), v3))
                        elif f3 == 5 and w3 == 5:
                            floats.append(struct.unpack("<f", v3)[0])
                    tensor = np.array(floats, dtype=np.float32)
            if tag is not None:
                out.setdefault(tag, {})[step] = simple if tensor is None else tensor
    return out


def write_event_slice(nrec=150):
    """First `nrec` raw records of one committed run's event file (a data file of the reference): the byte-exact
    fixture of tests/test_tblog_cpu.py for the log writer / reader."""
    rdir = os.path.join(REFERENCE, "docs", "saved-runs", RUNS["saved_run_bayes_cbf_maxrisk0p01"])
    src = glob.glob(os.path.join(rdir, "events.out.tfevents.*"))[0]
    data = open(src, "rb").read()
    i = 0
    for _ in range(nrec):
        (ln,) = struct.unpack("<Q", data[i:i + 8])
        i += 16 + ln
    with open(os.path.join(HERE, "reference_events_slice.tfevents"), "wb") as f:
        f.write(data[:i])
    print("reference_events_slice.tfevents: %d records, %d bytes" % (nrec, i))


def main():
    write_event_slice()
    for name, d in RUNS.items():
        rdir = os.path.join(REFERENCE, "docs", "saved-runs", d)
        ev = parse_events(glob.glob(os.path.join(rdir, "events.out.tfevents.*"))[0])
        cfg = json.load(open(os.path.join(rdir, "config.json")))
        steps = sorted(ev["vis/state"])
        out = dict(steps=np.array(steps),
                   state=np.stack([ev["vis/state"][s] for s in steps]),
                   uopt=np.stack([ev["vis/uopt"][s] for s in steps]),
                   plan_x=np.stack([ev["vis/plan_x"][s] for s in steps]),
                   opt_value=np.array([ev["opt/value"].get(s, np.nan) for s in steps], dtype=np.float32),
                   opt_rho=np.array([ev["opt/rho"].get(s, np.nan) for s in steps], dtype=np.float32),
                   dt=cfg["dt"], numSteps=cfg["numSteps"],
                   cbf_gammas=np.array(cfg["cbf_gammas"]), term_weights=np.array(cfg["cbfs"]["term_weights"]),
                   mean_L=cfg["mean_dynamics_gen"]["L"], kernel_diag_A=np.array(cfg["mean_dynamics_gen"]["kernel_diag_A"]),
                   true_L=cfg["true_dynamics_gen"]["L"], clf_gamma=cfg["controller_class"]["clf_gamma"],
                   cost_weights=np.array(cfg["controller_class"]["cost_weights"]),
                   max_risk=cfg["controller_class"]["max_risk"],
                   state_start=np.array(cfg["state_start"], dtype=np.float64),
                   state_goal=np.array(cfg["state_goal"], dtype=np.float64))
        if "vis/cbc_value" in ev:
            out["cbc_value"] = np.stack([np.atleast_1d(ev["vis/cbc_value"].get(s, np.nan)) for s in steps])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: np.shape(v) for k, v in out.items() if np.ndim(v)}, "tags:", sorted(ev)[:12])


LEARNING_RUN = "unicycle_move_to_pose_fixed_learning_helps_avoid_getting_stuck_v1.6.3-1-g5fa08e8"


def main_learning():
    """The committed learning run that also logs the hyper-parameters per step (written with the REAL gpytorch, cvxpy and
    GUROBI): per step t = 0..199 the state, the control, the kernel parameters as the reference's accessors return them
    (`get_kernel_param`: A, B, lengthscale, scalefactor, unicycle_move_to_pose.py:975-978) and the logged covariances
    `Fx_var` = custom_predict_fullmat(x_t)[1] ([9, 9] = kron(B_k, A)) and `Fxu_var` = fu_func_gp(u_t).knl(x_t, x_t) ([3, 3])
    (:979-982).  Data of the reference, float32 as logged."""
    rdir = os.path.join(REFERENCE, "docs", "saved-runs", LEARNING_RUN)
    ev = parse_events(glob.glob(os.path.join(rdir, "events.out.tfevents.*"))[0])
    cfg = json.load(open(os.path.join(rdir, "config.json")))
    steps = sorted(ev["vis/state"])
    st = lambda tag, shape: np.stack([np.asarray(ev[tag][s], dtype=np.float32).reshape(shape) for s in steps])
    out = dict(steps=np.array(steps), state=st("vis/state", (3,)), uopt=st("vis/uopt", (2,)), xtp1=st("vis/xtp1", (3,)),
               knl_A=st("vis/knl_A", (3, 3)), knl_B=st("vis/knl_B", (3, 3)), knl_lengthscale=st("vis/knl_lengthscale", (3,)),
               knl_scalefactor=st("vis/knl_scalefactor", ()), Fx_var=st("vis/Fx_var", (9, 9)), Fxu_var=st("vis/Fxu_var", (3, 3)),
               dt=cfg["dt"], numSteps=cfg["numSteps"], train_every_n_steps=cfg["train_every_n_steps"],
               mean_L=cfg["mean_dynamics_gen"]["L"], true_L=cfg["true_dynamics_gen"]["L"],
               max_risk=cfg["controller_class"]["max_risk"])
    np.savez_compressed(os.path.join(HERE, "saved_run_learning_v1p6p3.npz"), **out)
    print("saved_run_learning_v1p6p3", {k: np.shape(v) for k, v in out.items() if np.ndim(v)})


if __name__ == "__main__":
    if "learning" in sys.argv:
        sys.exit(main_learning())
    main()
    sys.exit(main_learning())
