"""Run-log format: the event-file writer / reader against a slice of a run the reference committed
(docs/saved-runs/unicycle_move_to_pose_fixed_mean_cbf_collides_1209-1257; first 150 records, a data file) and the
trajectory fixture extracted from the same run."""
import os

import numpy as np
import pytest

from bayesian_cbf_amd import tblog

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SLICE = os.path.join(GOLDEN, "reference_events_slice.tfevents")


def test_crc32c_known_answers():
    assert tblog.crc32c(b"") == 0
    assert tblog.crc32c(b"123456789") == 0xE3069283          # the CRC-32C check value
    assert tblog.crc32c(b"\x00" * 32) == 0x8A9136AA           # RFC 3720 B.4


def test_reference_records_reencode_byte_for_byte():
    """Every record the reference's SummaryWriter / make_tensor_summary wrote (misc.py:320-335, 394-405) is reproduced
    by the encoder, framing and both masked CRCs included."""
    raw = open(SLICE, "rb").read()
    out, n, tags = b"", 0, set()
    for rec in tblog.read_records(SLICE, check_crc=True):
        ev = tblog.decode_event(rec)
        assert "other_fields" not in ev
        enc = tblog.encode_event(ev["wall_time"], ev["step"], ev.get("file_version"), ev.get("tag"),
                                 ev.get("simple_value"), ev.get("tensor"))
        assert enc == rec
        out += tblog.frame_record(enc)
        n += 1
        tags.add(ev.get("tag"))
    assert n == 150 and out == raw
    assert {"vis/state", "vis/uopt", "vis/plan_x", "opt/value", "opt/rho"} <= tags


def test_reader_agrees_with_the_extracted_trajectory_fixture():
    g = np.load(os.path.join(GOLDEN, "saved_run_bayes_cbf_maxrisk0p01.npz"))
    by_tag = tblog.load_tensorboard_scalars(SLICE)
    for t, state in by_tag["vis/state"]:
        np.testing.assert_array_equal(state, g["state"][list(g["steps"]).index(t)])
    for t, u in by_tag["vis/uopt"]:
        np.testing.assert_array_equal(u, g["uopt"][list(g["steps"]).index(t)])
    steps = [t for t, _ in by_tag["opt/value"]]
    assert steps == sorted(steps) and len(steps) >= 5
    for t, v in by_tag["opt/value"]:
        assert v == pytest.approx(float(g["opt_value"][list(g["steps"]).index(t)]), rel=1e-6)


def test_write_read_round_trip_and_playback(tmp_path):
    class Plan:
        def plan(self, t):
            return np.array([0.1 * t, -0.2 * t, 0.0])

    log = tblog.TBLogger(["unit", "test"], runs_dir=str(tmp_path))
    log.write_config(dict(state_start=[0.0, 0.0, 0.0], state_goal=[1.0, 1.0, 0.5], numSteps=4, dt=0.01))
    rl = tblog.RolloutLogger(Plan(), 0.01, log)
    rng = np.random.default_rng(0)
    states, us = rng.normal(size=(4, 3)).astype(np.float32), rng.normal(size=(4, 2)).astype(np.float32)
    for t in range(4):
        rl.add_info(t, "rho", 2.5 + t)
        rl.add_info(t, "grid", np.arange(6, dtype=np.float32).reshape(2, 3) * t)
        rl.setStateCtrl(states[t], us[t], t)
        log.add_scalars("opt", dict(value=float(t) ** 2), t)
    log.summary_writer.close()
    run = tblog.playback_logfile(log.experiment_logs_dir)
    assert run["config"]["numSteps"] == 4 and list(run["steps"]) == [0, 1, 2, 3]
    np.testing.assert_array_equal(run["state"], states)
    np.testing.assert_array_equal(run["uopt"], us)
    np.testing.assert_array_equal(run["info"]["grid"][3], np.arange(6, dtype=np.float32).reshape(2, 3) * 3)
    assert run["info"]["rho"][2].shape == () and float(run["info"]["rho"][2]) == 4.5
    by_tag = tblog.load_tensorboard_scalars(run["events_file"])
    # the reference's reader returns `simple_value or tensor`: a logged 0.0 scalar comes back as the (absent) tensor
    assert [v for _, v in by_tag["opt/value"]][1:] == [1.0, 4.0, 9.0]
    first = next(tblog.read_records(run["events_file"]))
    assert tblog.decode_event(first)["file_version"] == "brain.Event:2"


def test_corrupt_record_is_detected(tmp_path):
    raw = bytearray(open(SLICE, "rb").read())
    raw[40] ^= 0xFF
    p = tmp_path / "bad.tfevents"
    p.write_bytes(bytes(raw))
    with pytest.raises(ValueError):
        list(tblog.read_records(str(p)))
