"""C++ trainer shell (trainer/train.cc): keeps the reference's CLI / YAML surface over the C ABI.
CPU: builds, fails loudly without a GPU, validates configs.  GPU: trains the debug config and writes a
TensorBoard event file whose TFRecord framing (masked CRC32C) and tags are checked here."""
import os
import struct
import subprocess

import pytest

from conftest import ROOT

TRAIN = os.path.join(ROOT, "trainer", "train")


@pytest.fixture(scope="module")
def trainer():
    from __graft_entry__ import build
    build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "trainer")])
    return TRAIN


def _crc32c(data):
    table = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        table.append(c)
    c = 0xFFFFFFFF
    for b in data:
        c = table[(c ^ b) & 255] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked(data):
    c = _crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def read_events(path):
    """yield raw Event payloads, checking both CRCs of every TFRecord"""
    with open(path, "rb") as f:
        while True:
            hdr = f.read(8)
            if not hdr:
                return
            (n,) = struct.unpack("<Q", hdr)
            assert struct.unpack("<I", f.read(4))[0] == _masked(hdr)
            data = f.read(n)
            assert struct.unpack("<I", f.read(4))[0] == _masked(data)
            yield data


def test_usage_and_cpu_failure(trainer, tmp_path):
    r = subprocess.run([trainer], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([trainer, "rom.bin", str(tmp_path / "tb" / "run.log"), str(tmp_path), "grp",
                            os.path.join(ROOT, "trainer", "configs", "debug.yaml")], capture_output=True, text=True)
        assert r.returncode == 1 and "no CPU fallback" in r.stderr


def test_config_validation(trainer, tmp_path):
    cfg = tmp_path / "bad.yaml"
    cfg.write_text("total_environments: 8\nhorizon: 5\nnum_mini_batches: 3\n")
    r = subprocess.run([trainer, "rom.bin", str(tmp_path / "x.log"), str(tmp_path), "g", str(cfg)],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "divisible by num_mini_batches" in r.stderr  # train.h:140-143
    r = subprocess.run([trainer, "rom.bin", str(tmp_path / "x.log"), str(tmp_path), "g", str(tmp_path / "nope.yaml")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open config" in r.stderr


def test_orthogonal_init_property(trainer, tmp_path):
    """layer_init (src/bin/train.cc:212-228): orthogonal weights with gain sqrt(2) (convs, fc), 0.01 (action head), 1
    (value head), zero biases.  Every weight has rows <= cols, so W W^T = gain^2 I.  (Bitwise parity with libtorch's
    LAPACK-QR init is not a goal; the property is.)  No GPU: the trainer dumps its initial parameters."""
    import numpy as np
    cfg = tmp_path / "c.yaml"
    cfg.write_text("total_environments: 8\nhidden_size: 64\naction_size: 6\nhorizon: 8\nnum_mini_batches: 4\n"
                   "deterministic: true\n")
    dump = tmp_path / "init.bin"
    r = subprocess.run([trainer, "rom.bin", str(tmp_path / "x.log"), str(tmp_path), "g", str(cfg)],
                       capture_output=True, text=True, env=dict(os.environ, ALEPPO_TRAINER_DUMP_INIT=str(dump)))
    assert r.returncode == 0, r.stderr
    p = np.fromfile(dump, np.float32)
    H, A = 64, 6
    shapes = [(32, 256, 2 ** 0.5), (64, 512, 2 ** 0.5), (64, 576, 2 ** 0.5), (H, 3136, 2 ** 0.5), (A, H, 0.01),
              (1, H, 1.0)]
    assert p.size == sum(r_ * c + r_ for r_, c, _ in shapes)
    o = 0
    for rows, cols, gain in shapes:
        w = p[o:o + rows * cols].reshape(rows, cols).astype(np.float64)
        b = p[o + rows * cols:o + rows * cols + rows]
        o += rows * cols + rows
        np.testing.assert_allclose(w @ w.T, gain * gain * np.eye(rows), atol=1e-5 * max(1.0, gain * gain))
        assert (b == 0).all()
    # deterministic: true -> seed 42 (train.cc:293-318,354-355): the same parameters every time
    dump2 = tmp_path / "init2.bin"
    subprocess.run([trainer, "rom.bin", str(tmp_path / "x.log"), str(tmp_path), "g", str(cfg)], capture_output=True,
                   env=dict(os.environ, ALEPPO_TRAINER_DUMP_INIT=str(dump2)), check=True)
    assert (np.fromfile(dump2, np.float32) == p).all()


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16+fp16planes+graph", "fp32+rawframes"])
def test_trains_debug_config_and_writes_event_file(trainer, tmp_path, precision):
    cfg = tmp_path / "debug.yaml"
    txt = open(os.path.join(ROOT, "trainer", "configs", "debug.yaml")).read().replace("num_rollouts: 10",
                                                                                      "num_rollouts: 4")
    if "graph" in precision:  # the optional keys: half-precision rollout planes, the update as a captured graph
        txt = txt.replace("cuda_graph: false", "cuda_graph: true") + "rollout_precision: fp16\n"
    if "rawframes" in precision:  # N2: raw 210x160 frame pairs from the emulator threads, preprocessing on the device
        txt += "device_preprocess: true\n"
    cfg.write_text(txt.replace("precision: fp32", f"precision: {precision.split('+')[0]}"))
    log = tmp_path / "tb" / "run.log"
    os.makedirs(log.parent)
    profile = tmp_path / "profile.json"
    r = subprocess.run([trainer, "breakout.bin", str(log), str(tmp_path), "grp", str(cfg), str(profile)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Rollout 4 of 4" in r.stdout and "Success" in r.stdout
    # total_steps counts only non-start slots (rollout.cc:225,266): every slot is a step or a start slot; start slots are
    # the E initial ones + one per finished episode, minus the ones still pending when the run ends
    import json
    import re
    mo = re.search(r"steps (\d+) episodes (\d+) pending_starts (\d+) slots (\d+)", r.stdout)
    steps, episodes, pending, slots = map(int, mo.groups())
    assert slots == 5 * 8 * 32 and steps == slots - 8 - episodes + pending and episodes > 0
    # [profile] (train.cc:409-419,459-462): host spans of the C-ABI calls + device time per kernel class
    prof = json.load(open(profile))
    names = {e["name"] for e in prof["traceEvents"]}
    assert {"aleppo_act", "aleppo_step", "aleppo_finish_rollout", "aleppo_train", "log_data"} <= names
    assert sum(e["name"] == "aleppo_train" for e in prof["traceEvents"]) == 4
    classes = {d["kernel_class"] for d in prof["device_kernel_classes"]}
    assert {"gae", "head", "adam", "infer_head"} <= classes and all(d["avg_ms"] > 0 for d in prof["device_kernel_classes"])
    files = [f for f in os.listdir(log.parent) if f.startswith("run.tfevents.")]
    assert len(files) == 1
    payloads = list(read_events(str(log.parent / files[0])))
    assert b"brain.Event:2" in payloads[0]
    blob = b"".join(payloads)
    for tag in (b"mean_loss", b"mean_clipped_gradient", b"mean_value_loss", b"mean_entropy", b"mean_ratio",
                b"learning_rate", b"clipped_gradients", b"losses", b"clipped_losses", b"value_losses", b"entropies",
                b"ratios", b"advantages", b"returns", b"episode_returns", b"episode_lengths"):
        assert tag in blob, tag
    # hparams session record (logger.add_hparams, train.cc:72-105,:389): plugin name, tag, group and every key
    hp = payloads[1]
    assert b"_hparams_/session_start_info" in hp and b"hparams" in hp and b"grp" in hp
    for key in (b"total_environments", b"hidden_size", b"action_size", b"horizon", b"max_steps", b"frame_stack",
                b"learning_rate", b"clip_param", b"value_loss_coef", b"entropy_coef", b"num_epochs", b"mini_batch_size",
                b"num_mini_batches", b"gae_discount", b"gae_lambda", b"max_gradient_norm", b"num_rollouts",
                b"num_workers", b"worker_batch_size", b"frame_skip", b"max_return", b"record_observation",
                b"record_video", b"cuda_graph", b"deterministic"):
        assert key in hp, key
    assert len(payloads) >= 2 + 4 * 14


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["bf16", "fp32+rawframes"])
def test_slot_ahead_loop_trains_to_the_same_parameters(trainer, tmp_path, variant):
    """`slot_ahead: true` (default; aleppo_arm_step before the emulator threads run, aleppo_release_step after) against
    `slot_ahead: false` (aleppo_step): same actions, same episodes, bit-identical parameters after 3 rollouts + updates"""
    import re
    import numpy as np
    txt = open(os.path.join(ROOT, "trainer", "configs", "debug.yaml")).read().replace("num_rollouts: 10", "num_rollouts: 3")
    if "rawframes" in variant:
        txt += "device_preprocess: true\n"
    txt = txt.replace("precision: fp32", f"precision: {variant.split('+')[0]}")
    out = {}
    for ahead in ("true", "false"):
        d = tmp_path / ahead
        os.makedirs(d / "tb")
        cfg = d / "debug.yaml"
        cfg.write_text(txt + f"slot_ahead: {ahead}\n")
        dump = d / "final.bin"
        r = subprocess.run([trainer, "breakout.bin", str(d / "tb" / "run.log"), str(d), "grp", str(cfg)], capture_output=True,
                           text=True, timeout=300, env=dict(os.environ, ALEPPO_TRAINER_DUMP_FINAL=str(dump)))
        assert r.returncode == 0, r.stderr
        mo = re.search(r"steps (\d+) episodes (\d+) pending_starts (\d+) slots (\d+)", r.stdout)
        out[ahead] = (tuple(map(int, mo.groups())), np.fromfile(dump, np.float32))
    assert out["true"][0] == out["false"][0]
    assert out["true"][0][1] > 0
    np.testing.assert_array_equal(out["true"][1], out["false"][1])
