"""helper of test_gated_replay_equals_launch_per_slot_replay (run as a script): sha256 over every stored plane of a few
replayed rollouts at edge shapes; ALEPPO_REPLAY_GATED is read once per process, hence the subprocesses"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hashfill as hf  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
from test_gpu_at_size import DeviceBytes, _flags  # noqa: E402

pkg = load_package()
h = hashlib.sha256()
for (E, T, A, H, prec) in [(4, 1, 4, 64, pkg.FP32), (5, 2, 6, 512, pkg.BF16), (300, 3, 4, 512, pkg.BF16), (1, 4, 18, 64, pkg.FP32),
                           (128, 16, 4, 512, pkg.BF16)]:
    dev = DeviceBytes(hf.hf_bytes(31, (T, E, 84, 84)))
    te, tr, st = _flags(32, T, E)
    rew = hf.hf_range(33, (T, E), -2, 2)
    eng = pkg.Engine(E, T, A, H, precision=prec, seed=3)
    eng.load_params(hf.fill_params(34, H, A))
    for r in range(3):
        eng.replay_rollout(dev.addr, pkg.FRAMES_84, E * 7056, rew, te, tr, st)
        eng.finish_rollout()
        for k in ("observations", "actions", "values", "logits", "advantages", "returns", "masks"):
            h.update(eng.read_batch(k).tobytes())
    eng.close()
    dev.free()
print("digest", h.hexdigest())
