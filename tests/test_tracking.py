"""Run artefacts in the reference's formats: TFRecord / tensorflow.Event
writer and reader without TensorBoard, the trainer's env-0 tracking block, the
reward_components_env0_timeseries.csv of src/options_sim/cli/artifacts.py.

Pinned: CRC-32C against the RFC 3720 B.4 vectors; the CSV header against the file the
reference ships (my_runs/.../reward_components_env0_timeseries.csv: same columns, same order).
The event-file bytes cannot be checked against TensorBoard itself (not installed here, no
event file in the reference tree): writer and reader are checked against each other, against
the published record framing (length, masked crc of the length, payload, masked crc), and the
payloads are decoded by Google's protobuf runtime against TensorFlow's published Event / Summary
schema (test_event_bytes_parse_with_the_protobuf_runtime).
"""
import csv
import os
import struct

import numpy as np
import pytest

from options_in_simulation_amd import tracking as tr


def test_crc32c_known_answers():
    assert tr.crc32c(b"123456789") == 0xE3069283
    assert tr.crc32c(bytes(32)) == 0x8A9136AA            # RFC 3720 B.4
    assert tr.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert tr.crc32c(bytes(range(32))) == 0x46DD794E
    assert tr.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    c = tr.crc32c(b"abc")
    assert tr.masked_crc(b"abc") == ((((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF)


def test_event_file_round_trip_and_framing(tmp_path):
    w = tr.EventFileWriter(str(tmp_path))
    w.add_scalar("Reward_comp_env0 / v_delta", 0.25, 800)
    w.add_scalars({"Info / rew_Final reward": -1.5, "Info / terrain_level": 0.0}, 1600)
    w.add_scalar("big step", 3.0, 2 ** 40 + 7)
    w.close()
    assert os.path.basename(w.path).startswith("events.out.tfevents.")
    ev = tr.read_events(w.path)
    assert len(ev) == 4 and ev[0][2] == {} and ev[0][0] == 0            # the file_version record
    assert ev[1][0] == 800 and ev[1][2] == {"Reward_comp_env0 / v_delta": 0.25}
    assert ev[2][0] == 1600 and ev[2][2] == {"Info / rew_Final reward": -1.5, "Info / terrain_level": 0.0}
    assert ev[3][0] == 2 ** 40 + 7
    with open(w.path, "rb") as f:
        raw = f.read()
    (n,) = struct.unpack("<Q", raw[:8])
    assert raw[12:12 + n].endswith(b"brain.Event:2") and raw[12] == 0x09  # field 1, fixed64 wall_time first
    # a flipped payload bit is detected
    bad = bytearray(raw); bad[14] ^= 1
    p = tmp_path / "bad"; p.write_bytes(bytes(bad))
    with pytest.raises(ValueError):
        tr.read_events(str(p))


def test_csv_header_matches_the_reference_artefact():
    assert ("step",) + tr.CSV_COLUMNS == ("step", "v_delta", "desired_v", "current_v", "omega_delta", "desired_omega",
                                           "current_omega")  # header of the shipped CSV / artifacts.py:96
    assert len(tr.ENV0_TAGS) == 26 and tr.ENV0_TAGS["env0_torque_apl_fl"] == "torque_fl"  # base_KA_DDPG.py:229-254


class _FakeTask:
    def __init__(self):
        self.t = 0

    @property
    def observed_components(self):
        d = {k: float(self.t) for k in tr.ENV0_TAGS}
        d["env0_desired_v"] = 1.5
        d["env0_episode_count"] = self.t // 99
        return d


class _FakeEnv:
    task = _FakeTask()


def test_env0_tracker_writes_interval_means_and_csv(tmp_path):
    torch = pytest.importorskip("torch")
    env = _FakeEnv()
    trk = tr.Env0Tracker(env, str(tmp_path), write_interval=10)
    for t in range(1, 31):
        env.task.t = t
        infos = {"episode": {"rew_Final reward": torch.tensor(float(-t)), "vector": torch.zeros(3)}} if t % 5 == 0 else {}
        trk.track(t, infos)
    trk.close()
    ev = [e for e in tr.read_events(trk.writer.path) if e[2]]
    by_step = {}
    for step, _, sc in ev:
        by_step.setdefault(step, {}).update(sc)
    assert sorted(by_step) == [10, 20, 30]
    assert by_step[10]["Reward_comp_env0 / current_v"] == pytest.approx(5.5)     # mean of 1..10
    assert by_step[30]["Reward_comp_env0 / current_v"] == pytest.approx(25.5)    # mean of 21..30 (cleared per write)
    assert by_step[20]["Reward_comp_env0 / desired_v"] == pytest.approx(1.5)
    assert by_step[10]["Info / rew_Final reward"] == pytest.approx(-7.5)         # mean of -5, -10
    assert "Info / vector" not in by_step[10]                                    # only one-element values (:261)
    with open(trk.csv_path) as f:
        rows = list(csv.reader(f))
    assert rows[0] == ["step", "v_delta", "desired_v", "current_v", "omega_delta", "desired_omega", "current_omega"]
    assert [int(r[0]) for r in rows[1:]] == [10, 20, 30] and float(rows[1][3]) == pytest.approx(5.5)
    steps, cols = tr.timeseries_from_events(trk.writer.path)                     # the route artifacts.py takes
    assert steps == [10, 20, 30] and cols["current_v"] == pytest.approx([5.5, 15.5, 25.5])
    assert cols["desired_v"] == pytest.approx([float(r[2]) for r in rows[1:]])


@pytest.mark.gpu
def test_tracker_on_the_real_env(tmp_path):
    torch = pytest.importorskip("torch")
    from options_in_simulation_amd import VecEnv
    env = VecEnv(num_envs=32, device="cuda:0", variant="KAMMA_medium", seed=3)
    trk = tr.Env0Tracker(env, str(tmp_path), write_interval=50)
    states, infos = env.reset()
    for t in range(1, 101):
        states, rew, term, trunc, infos = env.step(torch.rand((32, 4), device="cuda"))
        trk.track(t, infos)
    trk.close()
    steps, cols = tr.timeseries_from_events(trk.writer.path)
    assert steps == [50, 100]
    assert all(np.isfinite(cols[c]).all() for c in tr.CSV_COLUMNS)
    # expert phase (gamma ~ 1): the tracked speed approaches the command, v_delta = desired - current on average
    assert cols["desired_v"][0] > 0.9 and abs(cols["desired_v"][0] - cols["current_v"][0] - cols["v_delta"][0]) < 1e-4
    tags = set()
    for _, _, sc in tr.read_events(trk.writer.path):
        tags |= set(sc)
    assert {"Reward_comp_env0 / torque_fl", "Reward_comp_env0 / env0_perc_guiding", "Info / rew_Final reward",
            "Info / gamma_1 action assist", "Info / terrain_level"} <= tags
    env.close()


def test_tracking_error_vs_speed_csv(tmp_path):
    """eval_artifacts.py: join on step, trailing moving average, header of the shipped CSV
    (my_runs/eval_kamma_4d/.../tracking_error_vs_speed.csv)"""
    w = tr.EventFileWriter(str(tmp_path))
    for i, s in enumerate(range(100, 1100, 100)):
        w.add_scalar("Info / rew_Desired velocity", 1.0 + 0.1 * i, s)
        if s != 500:  # a step without a tracking-error point drops out of the join
            w.add_scalar("Info / rew_Tracking error", float(i), s)
    w.close()
    xs, ys = tr.write_tracking_error_vs_speed(w.path, smooth=3)
    assert len(xs) == 9 and xs[0] == pytest.approx(1.0) and xs[4] == pytest.approx(1.5)
    raw = [0, 1, 2, 3, 5, 6, 7, 8, 9]
    assert ys == pytest.approx([np.mean(raw[max(0, k - 2):k + 1]) for k in range(9)])
    with open(tmp_path / "tracking_error_vs_speed.csv") as f:
        rows = list(csv.reader(f))
    assert rows[0] == ["speed_m_s", "tracking_error_m_s"] and len(rows) == 10
    w2 = tr.EventFileWriter(str(tmp_path / "b"))
    w2.add_scalar("Info / rew_Tracking error", 0.5, 1); w2.add_scalar("Info / rew_Tracking error", 0.7, 2)
    w2.close()
    xs, ys = tr.write_tracking_error_vs_speed(w2.path)  # no desired-velocity tag: the 1 -> 2 m/s ramp
    assert xs == [1.0, 2.0] and ys == pytest.approx([0.5, 0.7])
    w3 = tr.EventFileWriter(str(tmp_path / "c")); w3.close()
    with pytest.raises(ValueError):
        tr.write_tracking_error_vs_speed(w3.path)


def test_event_bytes_parse_with_the_protobuf_runtime(tmp_path):
    """The event files are written without TensorBoard (not importable here), and until now only the
    module's own reader had read them back. This parses the records with Google's protobuf runtime against
    descriptors built from TensorFlow's published schema (tensorflow/core/util/event.proto: Event{wall_time = 1
    double, step = 2 int64, file_version = 3 string, summary = 5}; framework/summary.proto: Summary{value = 1
    repeated}, Value{tag = 1 string, simple_value = 2 float}): an independent decoder for the wire format that
    EventAccumulator (src/options_sim/cli/artifacts.py:68-92) reads. The TFRecord framing is checked by hand:
    uint64 length, masked crc32c of the length, payload, masked crc32c of the payload."""
    import struct
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="ois_event_check.proto", package="oisck", syntax="proto3")
    val = fd.message_type.add(name="Value")
    val.field.add(name="tag", number=1, type=descriptor_pb2.FieldDescriptorProto.TYPE_STRING, label=1)
    val.field.add(name="simple_value", number=2, type=descriptor_pb2.FieldDescriptorProto.TYPE_FLOAT, label=1)
    summ = fd.message_type.add(name="Summary")
    summ.field.add(name="value", number=1, type=descriptor_pb2.FieldDescriptorProto.TYPE_MESSAGE, label=3, type_name=".oisck.Value")
    ev = fd.message_type.add(name="Event")
    ev.field.add(name="wall_time", number=1, type=descriptor_pb2.FieldDescriptorProto.TYPE_DOUBLE, label=1)
    ev.field.add(name="step", number=2, type=descriptor_pb2.FieldDescriptorProto.TYPE_INT64, label=1)
    ev.field.add(name="file_version", number=3, type=descriptor_pb2.FieldDescriptorProto.TYPE_STRING, label=1)
    ev.field.add(name="summary", number=5, type=descriptor_pb2.FieldDescriptorProto.TYPE_MESSAGE, label=1, type_name=".oisck.Summary")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    Event = message_factory.GetMessageClass(pool.FindMessageTypeByName("oisck.Event"))

    w = tr.EventFileWriter(str(tmp_path))
    w.add_scalar("Reward_comp_env0 / v_delta", 0.1645, 800)
    w.add_scalars({"Info / rew_Final reward": -36.5, "Info / gamma_1 action assist": 0.9984}, 1600)
    w.add_scalar("Reward / Total reward (mean)", -3.0e5, 2 ** 40)  # int64 step beyond 32 bits
    w.close()
    path = [os.path.join(tmp_path, f) for f in os.listdir(tmp_path) if "tfevents" in f][0]
    data = open(path, "rb").read()
    events, i = [], 0
    while i < len(data):
        (n,) = struct.unpack("<Q", data[i:i + 8])
        assert struct.unpack("<I", data[i + 8:i + 12])[0] == tr.masked_crc(data[i:i + 8])
        payload = data[i + 12:i + 12 + n]
        assert struct.unpack("<I", data[i + 12 + n:i + 16 + n])[0] == tr.masked_crc(payload)
        e = Event()
        e.ParseFromString(payload)          # raises on a malformed message
        from google.protobuf import unknown_fields
        assert len(unknown_fields.UnknownFieldSet(e)) == 0  # every byte belonged to a field of the schema
        events.append(e)
        i += 16 + n
    assert events[0].file_version == "brain.Event:2" and events[0].wall_time > 1e9
    got = [(e.step, {v.tag: v.simple_value for v in e.summary.value}) for e in events[1:]]
    assert got[0] == (800, {"Reward_comp_env0 / v_delta": np.float32(0.1645)})
    assert got[1][0] == 1600 and got[1][1]["Info / rew_Final reward"] == np.float32(-36.5)
    assert got[1][1]["Info / gamma_1 action assist"] == np.float32(0.9984)
    assert got[2] == (2 ** 40, {"Reward / Total reward (mean)": np.float32(-3.0e5)})
