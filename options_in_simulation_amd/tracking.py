"""Run artefacts in the reference's formats, without TensorBoard installed.

The reference's trainers log the arena-0 diagnostics and ``infos["episode"]`` through skrl's
``track_data`` into TensorBoard event files — tags ``Reward_comp_env0 / <name>`` and
``Info / <key>``, one scalar per write interval = the mean of the values tracked since the last
write (src/my_trainers/base_KA_DDPG.py:184-262, src/my_agents/base.py:192-210) — and its
artefact CLIs read those files back (src/options_sim/cli/artifacts.py:23-30 → the
``reward_components_env0_timeseries.csv`` each shipped run directory holds). This module writes
both from a run of this environment:

* ``EventFileWriter``: a dependency-free writer of the TFRecord / ``tensorflow.Event`` wire
  format (the published formats: length + masked CRC-32C framing; protobuf fields
  Event{wall_time=1, step=2, file_version=3, summary=5}, Summary{value=1},
  Value{tag=1, simple_value=2}) — the files load in TensorBoard's ``EventAccumulator``.
* ``Env0Tracker``: the trainer's tracking block as a callable to put after ``env.step``.
* ``read_events``: a reader of the same subset (tests; and so that the CSV can be rebuilt from
  an event file exactly as the reference CLI does).
"""
from __future__ import annotations

import csv
import os
import socket
import struct
import time
from collections import defaultdict
from typing import Dict, Iterator, List, Optional, Tuple

# ------------------------------------------------------------------ CRC-32C (Castagnoli) ---
_CRC_TABLE: List[int] = []


def _crc_table() -> List[int]:
    if not _CRC_TABLE:
        for n in range(256):
            c = n
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            _CRC_TABLE.append(c)
    return _CRC_TABLE


def crc32c(data: bytes) -> int:
    t = _crc_table()
    c = 0xFFFFFFFF
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data: bytes) -> int:
    """the TFRecord mask: rotate right by 15 and add a constant"""
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------------ protobuf subset --------
def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _field_bytes(num: int, payload: bytes) -> bytes:
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def encode_event(wall_time: float, step: int, scalars: Optional[Dict[str, float]] = None,
                 file_version: Optional[str] = None) -> bytes:
    ev = _varint((1 << 3) | 1) + struct.pack("<d", wall_time) + _varint((2 << 3) | 0) + _varint(step)
    if file_version is not None:
        ev += _field_bytes(3, file_version.encode())
    if scalars:
        summary = b""
        for tag, value in scalars.items():
            val = _field_bytes(1, tag.encode()) + _varint((2 << 3) | 5) + struct.pack("<f", float(value))
            summary += _field_bytes(1, val)
        ev += _field_bytes(5, summary)
    return ev


def _read_varint(buf: bytes, i: int) -> Tuple[int, int]:
    shift = n = 0
    while True:
        b = buf[i]
        i += 1
        n |= (b & 0x7F) << shift
        if not b & 0x80:
            return n, i
        shift += 7


def _fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    i = 0
    while i < len(buf):
        key, i = _read_varint(buf, i)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _read_varint(buf, i)
        elif wt == 1:
            v, i = buf[i:i + 8], i + 8
        elif wt == 5:
            v, i = buf[i:i + 4], i + 4
        elif wt == 2:
            n, i = _read_varint(buf, i)
            v, i = buf[i:i + n], i + n
        else:
            raise ValueError(f"unsupported wire type {wt}")
        yield num, wt, v


def read_events(path: str) -> List[Tuple[int, float, Dict[str, float]]]:
    """[(step, wall_time, {tag: value})] of an event file; verifies both CRCs of every record."""
    out = []
    with open(path, "rb") as f:
        data = f.read()
    i = 0
    while i < len(data):
        header = data[i:i + 8]
        (length,) = struct.unpack("<Q", header)
        (hcrc,) = struct.unpack("<I", data[i + 8:i + 12])
        if hcrc != masked_crc(header):
            raise ValueError("corrupt record header")
        payload = data[i + 12:i + 12 + length]
        (pcrc,) = struct.unpack("<I", data[i + 12 + length:i + 16 + length])
        if pcrc != masked_crc(payload):
            raise ValueError("corrupt record payload")
        i += 16 + length
        step, wall, scalars = 0, 0.0, {}
        for num, wt, v in _fields(payload):
            if num == 1 and wt == 1:
                (wall,) = struct.unpack("<d", v)
            elif num == 2 and wt == 0:
                step = v
            elif num == 5 and wt == 2:
                for n2, w2, val in _fields(v):
                    if n2 == 1 and w2 == 2:
                        tag, sv = None, None
                        for n3, w3, x in _fields(val):
                            if n3 == 1 and w3 == 2:
                                tag = x.decode()
                            elif n3 == 2 and w3 == 5:
                                (sv,) = struct.unpack("<f", x)
                        if tag is not None and sv is not None:
                            scalars[tag] = sv
        out.append((step, wall, scalars))
    return out


class EventFileWriter:
    """``events.out.tfevents.<time>.<host>`` in ``logdir`` (torch's SummaryWriter naming)."""

    def __init__(self, logdir: str, filename_suffix: str = ""):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, f"events.out.tfevents.{int(time.time())}.{socket.gethostname()}{filename_suffix}")
        self._f = open(self.path, "wb")
        self._record(encode_event(time.time(), 0, file_version="brain.Event:2"))

    def _record(self, payload: bytes) -> None:
        header = struct.pack("<Q", len(payload))
        self._f.write(header + struct.pack("<I", masked_crc(header)) + payload + struct.pack("<I", masked_crc(payload)))

    def add_scalar(self, tag: str, value: float, step: int) -> None:
        self._record(encode_event(time.time(), int(step), {tag: float(value)}))

    def add_scalars(self, scalars: Dict[str, float], step: int) -> None:
        self._record(encode_event(time.time(), int(step), scalars))

    def flush(self) -> None:
        self._f.flush()

    def close(self) -> None:
        if not self._f.closed:
            self._f.close()


# task.observed_components key -> tag suffix of "Reward_comp_env0 / ..." (base_KA_DDPG.py:184-254)
ENV0_TAGS = {
    "env0_desired_v": "desired_v", "env0_current_v": "current_v", "env0_desired_omega": "desired_omega",
    "env0_current_omega": "current_omega", "env0_v_delta": "v_delta", "env0_omega_delta": "omega_delta",
    "env0_linear_acc": "linear_acc", "env0_angular_acc": "angular_acc", "env0_episode_count": "episode_count",
    "env0_torque_apl_fl": "torque_fl", "env0_torque_apl_rl": "torque_rl", "env0_torque_apl_fr": "torque_fr",
    "env0_torque_apl_rr": "torque_rr", "env0_exp_left": "exp_torque_l", "env0_exp_right": "exp_torque_r",
    "env0_policy_torque_fl": "policy_torqs_fl", "env0_policy_torque_rl": "policy_torqs_rl",
    "env0_policy_torque_fr": "policy_torqs_fr", "env0_policy_torque_rr": "policy_torqs_rr",
    "env0_perc_r1": "env0_perc_r1", "env0_perc_r2": "env0_perc_r2", "env0_perc_r3": "env0_perc_r3",
    "env0_perc_dense": "env0_perc_dense", "env0_perc_sparse": "env0_perc_sparse",
    "env0_perc_observed": "env0_perc_observed", "env0_perc_guiding": "env0_perc_guiding",
}
# columns of reward_components_env0_timeseries.csv (artifacts.py:23-30, :96-108)
CSV_COLUMNS = ("v_delta", "desired_v", "current_v", "omega_delta", "desired_omega", "current_omega")


class Env0Tracker:
    """The tracking block of ``Trainer.single_agent_train`` for one ``VecEnv``: call
    ``track(timestep, infos)`` after every ``env.step``; every ``write_interval`` timesteps the
    means since the last write go to the event file (skrl's ``write_tracking_data``) and one row
    is appended to ``reward_components_env0_timeseries.csv``."""

    def __init__(self, env, logdir: str, write_interval: int = 800, environment_info: str = "episode"):
        self.env = env
        self.logdir = logdir
        self.write_interval = int(write_interval)
        self.environment_info = environment_info
        self.writer = EventFileWriter(logdir)
        self.tracking_data: Dict[str, List[float]] = defaultdict(list)
        self.csv_path = os.path.join(logdir, "reward_components_env0_timeseries.csv")
        with open(self.csv_path, "w", newline="") as f:
            csv.writer(f).writerow(("step",) + CSV_COLUMNS)

    def track_data(self, tag: str, value: float) -> None:
        self.tracking_data[tag].append(float(value))

    def track(self, timestep: int, infos: Optional[dict] = None) -> None:
        oc = self.env.task.observed_components
        for key, name in ENV0_TAGS.items():
            self.track_data(f"Reward_comp_env0 / {name}", oc[key])
        if infos and self.environment_info in infos:  # base_KA_DDPG.py:259-262
            for k, v in infos[self.environment_info].items():
                if hasattr(v, "numel") and v.numel() == 1:
                    self.track_data(f"Info / {k}", v.item())
        if timestep > 1 and self.write_interval > 0 and not timestep % self.write_interval:
            self.write_tracking_data(timestep)

    def write_tracking_data(self, timestep: int) -> None:
        row = {}
        for k, v in self.tracking_data.items():  # base.py:200-206
            val = min(v) if k.endswith("(min)") else max(v) if k.endswith("(max)") else sum(v) / len(v)
            self.writer.add_scalar(k, val, timestep)
            row[k] = val
        self.tracking_data.clear()
        self.writer.flush()
        # the CLI builds this file from the event file, whose simple_value is a float32: the same rounding here
        with open(self.csv_path, "a", newline="") as f:
            csv.writer(f).writerow([timestep] + [_as_f32(row.get(f"Reward_comp_env0 / {c}", float("nan"))) for c in CSV_COLUMNS])

    def close(self) -> None:
        self.writer.close()


def _as_f32(v: float) -> float:
    """the value an event file holds for v (Summary.Value.simple_value is a float32)"""
    return struct.unpack("<f", struct.pack("<f", v))[0]


def write_timeseries_csv(event_path: str, out_csv: Optional[str] = None) -> str:
    """``reward_components_env0_timeseries.csv`` rebuilt from an event file, as the reference's CLI does it
    (src/options_sim/cli/artifacts.py:68-108: _collect_timeseries + _write_csv): one row per step that carries any
    of the six tags, NaN where a tag has no value at that step."""
    steps, data = timeseries_from_events(event_path)
    if out_csv is None:
        out_csv = os.path.join(os.path.dirname(event_path), "reward_components_env0_timeseries.csv")
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(("step",) + CSV_COLUMNS)
        for i, s_ in enumerate(steps):
            w.writerow([s_] + [data[c][i] for c in CSV_COLUMNS])
    return out_csv


def timeseries_from_events(path: str) -> Tuple[List[int], Dict[str, List[float]]]:
    """what artifacts.py:_collect_timeseries builds from an event file: step-indexed columns"""
    table: Dict[int, Dict[str, float]] = {}
    for step, _, scalars in read_events(path):
        for tag, v in scalars.items():
            for c in CSV_COLUMNS:
                if tag in (f"Reward_comp_env0 / {c}", f"Reward_comp_env0/{c}", f"reward_comp_env0/{c}"):
                    table.setdefault(step, {})[c] = v
    steps = sorted(table)
    return steps, {c: [table[s].get(c, float("nan")) for s in steps] for c in CSV_COLUMNS}


# ---- eval artefact: tracking error vs commanded speed (src/options_sim/cli/eval_artifacts.py) ----
TRACKING_ERR_TAGS = ("Info / rew_Tracking error", "Info/rew_Tracking error", "tracking_error")
DESIRED_VEL_TAGS = ("Info / rew_Desired velocity", "Info/rew_Desired velocity", "desired_velocity")


def write_tracking_error_vs_speed(event_path: str, out_csv: Optional[str] = None, smooth: int = 0,
                                  vmin: float = 1.0, vmax: float = 2.0) -> Tuple[List[float], List[float]]:
    """``tracking_error_vs_speed.csv`` (columns ``speed_m_s,tracking_error_m_s``) from an eval
    run's event file, as eval_artifacts.py:120-160 builds it: the desired-velocity series
    left-joined with the tracking-error series on the step (a vmin→vmax ramp if the former is
    missing), optional trailing moving average over the error."""
    series: Dict[str, List[Tuple[int, float]]] = defaultdict(list)
    for step, _, scalars in read_events(event_path):
        for tag, v in scalars.items():
            series[tag].append((step, v))
    te = next((series[t] for t in TRACKING_ERR_TAGS if series.get(t)), None)
    dv = next((series[t] for t in DESIRED_VEL_TAGS if series.get(t)), None)
    if not te:
        raise ValueError(f"no tracking-error series in {event_path} (tags tried: {TRACKING_ERR_TAGS})")
    if dv:
        by_step = dict(te)
        xs = [v for s, v in dv if s in by_step]
        ys = [by_step[s] for s, _ in dv if s in by_step]
    else:
        n = len(te)
        xs = [vmin + (vmax - vmin) * i / max(n - 1, 1) for i in range(n)]
        ys = [v for _, v in te]
    if smooth and smooth > 1:
        out, window, total = [], [], 0.0
        for v in ys:
            window.append(v); total += v
            if len(window) > smooth:
                total -= window.pop(0)
            out.append(total / len(window))
        ys = out
    if out_csv is None:
        out_csv = os.path.join(os.path.dirname(event_path), "tracking_error_vs_speed.csv")
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["speed_m_s", "tracking_error_m_s"])
        w.writerows(zip(xs, ys))
    return xs, ys
