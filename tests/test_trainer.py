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


@pytest.mark.gpu
@pytest.mark.parametrize("name,E,T,epochs,M", [("v0", 8, 128, 4, 4), ("v1", 4096, 5, 1, 16)])
def test_trains_the_reference_configs(trainer, tmp_path, name, E, T, epochs, M):
    """The reference's REAL hyper-parameter sets (configs/v0.yaml: 8 envs x T = 128, 4 epochs x 4 minibatches of 256;
    configs/v1.yaml: 4096 envs x T = 5, 1 epoch x 16 minibatches of 1280, 32 workers popping 8 environments at a time) through
    the trainer, every key as shipped except num_rollouts (3 instead of 9760 / 1000000): the loop of train.cc:420-458 runs,
    the step accounting of rollout.cc:225,266 holds, every log_data scalar is finite, the parameters moved."""
    import re
    import numpy as np
    src = open(os.path.join(ROOT, "trainer", "configs", f"{name}.yaml")).read()
    txt, n = re.subn(r"(?m)^num_rollouts: \d+", "num_rollouts: 3", src)
    assert n == 1
    for key, val in (("total_environments", E), ("horizon", T), ("num_epochs", epochs), ("num_mini_batches", M),
                     ("mini_batch_size", E * T // M)):  # the shipped file IS the reference's shape
        assert re.search(rf"(?m)^{key}: {val}\s*$", txt), key
    cfg = tmp_path / f"{name}.yaml"
    cfg.write_text(txt)
    os.makedirs(tmp_path / "tb")
    init, final = tmp_path / "init.bin", tmp_path / "final.bin"
    subprocess.run([trainer, "breakout.bin", str(tmp_path / "x.log"), str(tmp_path), "g", str(cfg)], capture_output=True,
                   env=dict(os.environ, ALEPPO_TRAINER_DUMP_INIT=str(init)), check=True)
    r = subprocess.run([trainer, "breakout.bin", str(tmp_path / "tb" / "run.log"), str(tmp_path), name, str(cfg)],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, ALEPPO_TRAINER_DUMP_FINAL=str(final)))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Rollout 3 of 3" in r.stdout and "Success" in r.stdout
    mo = re.search(r"steps (\d+) episodes (\d+) pending_starts (\d+) slots (\d+)", r.stdout)
    steps, episodes, pending, slots = map(int, mo.groups())
    assert slots == 4 * E * T and steps == slots - E - episodes + pending
    p0, p1 = np.fromfile(init, np.float32), np.fromfile(final, np.float32)
    assert p0.shape == p1.shape and np.isfinite(p1).all()
    assert np.abs(p1 - p0).max() > 1e-4  # three annealed updates of lr 2.5e-4 moved the weights
    files = [f for f in os.listdir(tmp_path / "tb") if f.startswith("run.tfevents.")]
    assert len(files) == 1
    payloads = list(read_events(str(tmp_path / "tb" / files[0])))
    scalars = {}
    for ev in payloads[2:]:  # simple_value scalars: tag string followed by field 2 (fixed32 float)
        for tag in (b"mean_loss", b"mean_clipped_gradient", b"mean_value_loss", b"mean_entropy", b"mean_ratio",
                    b"learning_rate"):
            i = ev.find(b"\n" + bytes([len(tag)]) + tag + b"\x15")
            if i >= 0:
                j = i + 2 + len(tag) + 1
                scalars.setdefault(tag, []).append(struct.unpack("<f", ev[j:j + 4])[0])
    for tag in (b"mean_loss", b"mean_clipped_gradient", b"mean_value_loss", b"mean_entropy", b"mean_ratio", b"learning_rate"):
        assert len(scalars.get(tag, [])) == 3, (tag, scalars.keys())
        assert np.isfinite(scalars[tag]).all(), (tag, scalars[tag])
    np.testing.assert_allclose(scalars[b"learning_rate"], [2.5e-4 * (1 - i / 3) for i in range(3)], rtol=1e-6)
    assert all(0 < e <= np.log(4) + 1e-3 for e in scalars[b"mean_entropy"])  # 4-action policy (train.cc:36)


# ------------------------------------------------------------------ host side without a GPU: ThreadSanitizer + rank rendezvous
TRAIN_TSAN = os.path.join(ROOT, "trainer", "train_tsan")


@pytest.fixture(scope="module")
def trainer_tsan():
    """trainer/train.cc built with -fsanitize=thread against tests/stub/aleppo_stub.cc (a host-only stand-in for the
    library that touches the caller's buffers where the device work would; TEST INFRASTRUCTURE)"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "trainer"), "train_tsan"])
    return TRAIN_TSAN


def _debug_yaml(tmp_path, name, **over):
    txt = open(os.path.join(ROOT, "trainer", "configs", "debug.yaml")).read()
    for k, v in over.items():
        import re
        txt, n = re.subn(rf"(?m)^{k}: .*$", f"{k}: {v}", txt)
        if n == 0:
            txt += f"{k}: {v}\n"
    cfg = tmp_path / name
    cfg.write_text(txt)
    return str(cfg)


@pytest.mark.parametrize("ahead,raw,workers", [("true", "false", 4), ("false", "false", 4), ("true", "true", 3)])
def test_trainer_host_side_is_tsan_clean(trainer_tsan, tmp_path, ahead, raw, workers):
    """SURVEY 5 "TSan-clean handoff" (the reference's own loop races benignly at rollout.cc:303-313): worker pool, index
    queue, the frame / episode-start buffers the workers fill and the released step reads, the action buffer the workers
    read - several workers, slot_ahead on and off, raw frame pairs - with no ThreadSanitizer report."""
    cfg = _debug_yaml(tmp_path, "d.yaml", num_rollouts=3, num_workers=workers, slot_ahead=ahead, device_preprocess=raw)
    r = subprocess.run([trainer_tsan, "rom.bin", str(tmp_path / "run.log"), str(tmp_path), "g", cfg], capture_output=True,
                       text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66"))
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Rollout 3 of 3" in r.stdout and "Success" in r.stdout


def _spawn_rank(trainer_tsan, tmp_path, cfg, rank, world, port, extra=None):
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    env.update(extra or {})
    return subprocess.Popen([trainer_tsan, "rom.bin", str(tmp_path / "tb" / "run.log"), str(tmp_path), "g", cfg],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)


def test_trainer_rank_rendezvous_over_the_id_file(trainer_tsan, tmp_path):
    """SURVEY 8e (no reference counterpart, src/bin/train.cc:336-345 is single-device): two trainer processes with RANK /
    WORLD_SIZE take their environment shards and agree on rank 0's communicator id through `<log>.rcclid.<MASTER_PORT>` -
    written atomically, removed again once every rank has joined; a leftover of a crashed launch with the same name is
    neither read (too old) nor kept (rank 0 removes it first); a rank whose rank 0 never comes gives up with a message."""
    import re
    import time
    os.makedirs(tmp_path / "tb")
    cfg = _debug_yaml(tmp_path, "d.yaml", num_rollouts=2, num_workers=2)
    port = 29517
    idfile = str(tmp_path / "tb" / "run.log") + f".rcclid.{port}"
    with open(idfile, "wb") as f:  # a crashed earlier launch left its id behind, ten minutes ago
        f.write(b"\xee" * 128)
    old = time.time() - 600
    os.utime(idfile, (old, old))
    ranks = [_spawn_rank(trainer_tsan, tmp_path, cfg, 1, 2, port)]
    time.sleep(0.5)  # rank 1 is already polling when rank 0 starts: it must not take the leftover
    ranks.insert(0, _spawn_rank(trainer_tsan, tmp_path, cfg, 0, 2, port))
    outs = [p.communicate(timeout=600) for p in ranks]
    for p, (so, se) in zip(ranks, outs):
        assert p.returncode == 0, se[-2000:]
        assert "ThreadSanitizer" not in se, se[-3000:]
        assert "Success" in so
    ids = [re.search(r"stub comm_init rank (\d) of 2 id ([0-9a-f]{32})", so).groups() for so, _ in outs]
    assert ids[0][0] == "0" and ids[1][0] == "1"
    assert ids[0][1] == ids[1][1] and ids[0][1] != "ee" * 16  # rank 0's fresh id, not the leftover
    assert "environments [0, 4)" in outs[0][0] and "environments [4, 8)" in outs[1][0]  # 8 environments, 2 ranks
    assert not os.path.exists(idfile) and not os.path.exists(idfile + ".tmp")
    # rank 1 of a launch whose rank 0 never arrives: a bounded wait and a message that names the file
    p = _spawn_rank(trainer_tsan, tmp_path, cfg, 1, 2, port + 1, {"ALEPPO_RENDEZVOUS_TIMEOUT_S": "1"})
    so, se = p.communicate(timeout=120)
    assert p.returncode == 1 and "timed out" in se and f".rcclid.{port + 1}" in se
    # bad RANK / WORLD_SIZE combinations are refused before anything is created
    p = _spawn_rank(trainer_tsan, tmp_path, cfg, 2, 2, port + 2)
    so, se = p.communicate(timeout=120)
    assert p.returncode == 1 and "RANK must be in [0, WORLD_SIZE)" in se
    p = _spawn_rank(trainer_tsan, tmp_path, cfg, 0, 3, port + 3)
    so, se = p.communicate(timeout=120)
    assert p.returncode == 1 and "divisible by WORLD_SIZE" in se
