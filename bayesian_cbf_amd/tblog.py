"""Run logs in the reference's on-disk format: TensorBoard event files + config.json.

The reference logs every closed-loop step through `torch.utils.tensorboard.SummaryWriter` (scalars) and hand-made
`Summary(tensor=TensorProto(DT_FLOAT, float_val, shape))` records (misc.py:320-359, 362-405;
unicycle_move_to_pose.py:1257-1311) and plays runs back from them (`playback_logfile`, :1421-1453).  This module
writes and reads the same records without tensorboard / tensorflow / protobuf: a ~100-line TFRecord framing
(length, masked CRC-32C, payload, masked CRC-32C) + protobuf wire encoder/decoder for the three messages involved
(Event, Summary.Value, TensorProto).  Files written here load in TensorBoard and in the reference's
`load_tensorboard_scalars`; files written by the reference load here (tests/test_tblog_cpu.py re-encodes a slice of a
committed reference run byte for byte).

Plotting / animation of the reference's Visualizer are out of scope; `playback_logfile` returns the arrays."""
import glob
import json
import os
import os.path as osp
import socket
import struct
import time
from abc import ABC, abstractmethod

import numpy as np

CONFIG_FILE_BASENAME = "config.json"          # unicycle_move_to_pose.py:60

# ------------------------------------------------------------------------------------------------ CRC-32C + framing
_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def frame_record(payload):
    head = struct.pack("<Q", len(payload))
    return head + struct.pack("<I", masked_crc(head)) + payload + struct.pack("<I", masked_crc(payload))


def read_records(path, check_crc=True):
    with open(path, "rb") as f:
        data = f.read()
    i = 0
    while i + 12 <= len(data):
        (ln,) = struct.unpack("<Q", data[i:i + 8])
        if i + 12 + ln + 4 > len(data):
            break                                           # truncated tail (a run that was killed)
        payload = data[i + 12:i + 12 + ln]
        if check_crc:
            (c1,) = struct.unpack("<I", data[i + 8:i + 12])
            (c2,) = struct.unpack("<I", data[i + 12 + ln:i + 16 + ln])
            if c1 != masked_crc(data[i:i + 8]) or c2 != masked_crc(payload):
                raise ValueError("corrupt record at byte %d of %s" % (i, path))
        yield payload
        i += 16 + ln


# ------------------------------------------------------------------------------------------------ protobuf wire format
def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(fno, wt):
    return _varint((fno << 3) | wt)


def _ld(fno, payload):
    return _key(fno, 2) + _varint(len(payload)) + payload


def _read_varint(buf, i):
    shift = val = 0
    while True:
        b = buf[i]
        i += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, i
        shift += 7


def _fields(buf):
    i = 0
    while i < len(buf):
        key, i = _read_varint(buf, i)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _read_varint(buf, i)
        elif wt == 1:
            v, i = buf[i:i + 8], i + 8
        elif wt == 2:
            ln, i = _read_varint(buf, i)
            v, i = buf[i:i + ln], i + ln
        elif wt == 5:
            v, i = buf[i:i + 4], i + 4
        else:
            raise ValueError("unsupported wire type %d" % wt)
        yield fno, wt, v


def encode_tensor_proto(arr):
    """TensorProto{dtype=DT_FLOAT(1), tensor_shape{dim{size}}, float_val packed}  (misc.py:320-326)."""
    arr = np.asarray(arr, dtype=np.float32)
    dims = b"".join(_ld(2, _key(1, 0) + _varint(s)) for s in arr.shape)
    out = _key(1, 0) + _varint(1) + _ld(2, dims)
    flat = arr.reshape(-1)
    if flat.size:
        out += _ld(5, struct.pack("<%df" % flat.size, *flat.tolist()))
    return out


def encode_event(wall_time, step=0, file_version=None, tag=None, simple_value=None, tensor=None):
    """Event{wall_time, step, file_version | summary{value{tag, simple_value | tensor}}}."""
    out = _key(1, 1) + struct.pack("<d", wall_time)
    if step:
        out += _key(2, 0) + _varint(step)
    if file_version is not None:
        out += _ld(3, file_version.encode())
    if tag is not None:
        val = _ld(1, tag.encode())
        if tensor is not None:
            val += _ld(8, encode_tensor_proto(tensor))
        else:
            val += _key(2, 5) + struct.pack("<f", simple_value)
        out += _ld(5, _ld(1, val))
    return out


def decode_event(rec):
    """-> dict(wall_time, step, file_version | tag + simple_value | tensor[np.float32 with its logged shape])."""
    ev = dict(wall_time=0.0, step=0)
    for fno, wt, v in _fields(rec):
        if fno == 1 and wt == 1:
            ev["wall_time"] = struct.unpack("<d", v)[0]
        elif fno == 2 and wt == 0:
            ev["step"] = v
        elif fno == 3 and wt == 2:
            ev["file_version"] = v.decode()
        elif fno == 5 and wt == 2:
            for f1, w1, val in _fields(v):
                if f1 != 1:
                    continue
                for f2, w2, v2 in _fields(val):
                    if f2 == 1:
                        ev["tag"] = v2.decode()
                    elif f2 == 2 and w2 == 5:
                        ev["simple_value"] = struct.unpack("<f", v2)[0]
                    elif f2 == 8:
                        floats, shape = [], []
                        for f3, w3, v3 in _fields(v2):
                            if f3 == 5 and w3 == 2:
                                floats.extend(struct.unpack("<%df" % (len(v3) // 4), v3))
                            elif f3 == 5 and w3 == 5:
                                floats.append(struct.unpack("<f", v3)[0])
                            elif f3 == 2 and w3 == 2:
                                for f4, w4, v4 in _fields(v3):
                                    if f4 == 2:
                                        size = [x for fn, _, x in _fields(v4) if fn == 1]
                                        shape.append(size[0] if size else 0)
                        ev["tensor"] = np.array(floats, dtype=np.float32).reshape(shape)
                    else:
                        ev.setdefault("other_fields", []).append(f2)
        else:
            ev.setdefault("other_fields", []).append(fno)
    return ev


# ------------------------------------------------------------------------------------------------ writer / reader
class EventFileWriter:
    """events.out.tfevents.<time>.<host>.<pid>.<n> in `logdir`, first record file_version 'brain.Event:2'."""
    _count = 0

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        name = "events.out.tfevents.%010d.%s.%d.%d" % (int(time.time()), socket.gethostname(), os.getpid(),
                                                        EventFileWriter._count)
        EventFileWriter._count += 1
        self.path = osp.join(logdir, name)
        self._f = open(self.path, "ab")
        self._f.write(frame_record(encode_event(time.time(), file_version="brain.Event:2")))
        self._f.flush()

    def add_scalar(self, tag, value, step):
        self._f.write(frame_record(encode_event(time.time(), step=int(step), tag=tag, simple_value=float(value))))

    def add_tensor(self, tag, array, step):
        self._f.write(frame_record(encode_event(time.time(), step=int(step), tag=tag, tensor=np.asarray(array))))

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def stream_tensorboard_scalars(event_file):
    """(step, tag, value) per record; value = the scalar, or the tensor reshaped to its logged shape (misc.py:343-352)."""
    for rec in read_records(event_file):
        ev = decode_event(rec)
        if "tag" not in ev:
            continue
        value = ev.get("simple_value") or ev.get("tensor", ev.get("simple_value"))
        yield ev["step"], ev["tag"], value


def load_tensorboard_scalars(event_file):
    """{tag: [(step, value), ...]}  (misc.py:355-359)."""
    groupby_tag = dict()
    for t, tag, value in stream_tensorboard_scalars(event_file):
        groupby_tag.setdefault(tag, []).append((t, value))
    return groupby_tag


def _to_numpy(v):
    return v.detach().cpu().double().numpy() if hasattr(v, "detach") else np.asarray(v)


# ------------------------------------------------------------------------------------------------ Logger API (misc.py:362-405)
class Logger(ABC):
    @property
    @abstractmethod
    def experiment_logs_dir(self):
        return "/tmp"

    @abstractmethod
    def add_scalars(self, tag, var_dict, t):
        pass

    @abstractmethod
    def add_tensors(self, tag, var_dict, t):
        pass


class NoLogger(Logger):
    experiment_logs_dir = "/tmp"

    def add_scalars(self, tag, var_dict, t):
        pass

    def add_tensors(self, tag, var_dict, t):
        pass


class TBLogger(Logger):
    """Scalars and float tensors under `<runs_dir>/<exp_tags joined by _>_<version>` (misc.py:386-405)."""

    def __init__(self, exp_tags, runs_dir="data/runs", version="amd"):
        self.exp_tags, self.runs_dir = list(exp_tags), runs_dir
        self.exp_dir = osp.join(runs_dir, "_".join(self.exp_tags + [version]))
        self.summary_writer = EventFileWriter(self.exp_dir)

    @property
    def experiment_logs_dir(self):
        return self.exp_dir

    def add_scalars(self, tag, var_dict, t):
        for k, v in var_dict.items():
            self.summary_writer.add_scalar("/".join((tag, k)), float(_to_numpy(v)), t)
        self.summary_writer.flush()

    def add_tensors(self, tag, var_dict, t):
        for k, v in var_dict.items():
            self.summary_writer.add_tensor("/".join((tag, k)), _to_numpy(v), t)
        self.summary_writer.flush()

    def write_config(self, config):
        """config.json next to the event file (unicycle_move_to_pose.py:1760-1764)."""
        with open(osp.join(self.exp_dir, CONFIG_FILE_BASENAME), "w") as f:
            json.dump(config, f, indent=1, default=lambda o: _to_numpy(o).tolist())


class RolloutLogger:
    """Per-step log of a closed loop, the tags of the reference's unicycle Logger (unicycle_move_to_pose.py:1257-1311):
    vis/state, vis/uopt, vis/plan_x and whatever `add_info(t, key, value)` attached to the step (numbers / arrays)."""

    def __init__(self, planner, dt, tblogger):
        self.planner, self.dt, self.tblog = planner, dt, tblogger
        self.info = dict()

    def add_info(self, t, key, value):
        self.info.setdefault(t, dict())[key] = value

    def setStateCtrl(self, state, uopt, t=None, **kw):
        self.tblog.add_tensors("vis", dict(state=state, uopt=uopt, plan_x=self.planner.plan(t)), t)
        keep = {k: v for k, v in self.info.get(t, {}).items()
                if hasattr(v, "detach") or isinstance(v, (float, int, np.ndarray, np.floating))}
        if keep:
            self.tblog.add_tensors("vis", keep, t)

    @staticmethod
    def load_visualizer(events_file):
        """Yield (t, state, uopt, info) per logged step (:1347-1372)."""
        by_step = dict()
        for t, tag, value in stream_tensorboard_scalars(events_file):
            by_step.setdefault(t, dict())[tag] = value
        for t in sorted(by_step):
            rec = by_step[t]
            if "vis/state" not in rec or "vis/uopt" not in rec:
                continue
            info = {k.split("/", 1)[1]: v for k, v in rec.items() if k not in ("vis/state", "vis/uopt")}
            yield t, rec["vis/state"], rec["vis/uopt"], info


def playback_logfile(events_dir):
    """Read a run directory (config.json + newest event file) back: dict(config, steps, state[T,n], uopt[T,m],
    plan_x[T,n] if logged, info{tag: [T,...]})  -- the data half of unicycle_move_to_pose.py:1421-1453."""
    config = json.load(open(osp.join(events_dir, CONFIG_FILE_BASENAME)))
    events_file = max(glob.glob(osp.join(events_dir, "*tfevents*")), key=lambda f: os.stat(f).st_mtime)
    steps, states, uopts, infos = [], [], [], dict()
    for t, state, uopt, info in RolloutLogger.load_visualizer(events_file):
        steps.append(t)
        states.append(state)
        uopts.append(uopt)
        for k, v in info.items():
            infos.setdefault(k, dict())[t] = v
    return dict(config=config, events_file=events_file, steps=np.array(steps), state=np.stack(states),
                uopt=np.stack(uopts), info=infos)
