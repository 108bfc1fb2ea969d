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


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_trains_debug_config_and_writes_event_file(trainer, tmp_path, precision):
    cfg = tmp_path / "debug.yaml"
    txt = open(os.path.join(ROOT, "trainer", "configs", "debug.yaml")).read().replace("num_rollouts: 10",
                                                                                      "num_rollouts: 4")
    cfg.write_text(txt.replace("precision: fp32", f"precision: {precision}"))
    log = tmp_path / "tb" / "run.log"
    os.makedirs(log.parent)
    r = subprocess.run([trainer, "breakout.bin", str(log), str(tmp_path), "grp", str(cfg)], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Rollout 4 of 4" in r.stdout and "Success" in r.stdout
    files = [f for f in os.listdir(log.parent) if f.startswith("run.tfevents.")]
    assert len(files) == 1
    payloads = list(read_events(str(log.parent / files[0])))
    assert b"brain.Event:2" in payloads[0]
    blob = b"".join(payloads)
    for tag in (b"mean_loss", b"mean_clipped_gradient", b"mean_value_loss", b"mean_entropy", b"mean_ratio",
                b"learning_rate", b"clipped_gradients"):
        assert tag in blob, tag
    assert len(payloads) >= 1 + 4 * 7
