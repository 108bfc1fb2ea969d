"""Helper process of test_other_contexts_and_operators_never_wait_for_a_parked_stream (tests/test_gpu_at_size.py).

Runs with GPU_MAX_HW_QUEUES=16: the HIP runtime multiplexes streams onto that many hardware queues (default 4), and a
kernel that lands in the queue of a PARKED stream waits behind its gate whatever the library does (parkprobe: "kernels on 6
new streams").  With a queue per stream what is left is what the library controls: which runtime calls its entry points
make while another context's stream is parked."""
import ctypes
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hashfill as hf  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
from test_gpu_at_size import _flags  # noqa: E402

pkg = load_package()
pkg.lib()
import threading
E, T, A, H = 8, 4, 4, 32
a = pkg.Engine(E, T, A, H, seed=3)
a.load_params(hf.fill_params(2710, H, A))
fbuf, sbuf = a.host_alloc(E * 7056), a.host_alloc(E)
a.act()
a.arm_step(fbuf, sbuf, pkg.FRAMES_84)  # A's stream is parked from here on
done, err = threading.Event(), []

prog = []  # what the other thread was doing last (named in the failure message)

def other():
    try:
        prog.append("aleppo_create")
        b = pkg.Engine(E, T, A, H, seed=4)  # aleppo_create next to a parked stream
        prog.append("load_params")
        b.load_params(hf.fill_params(2711, H, A))
        frames = hf.hf_bytes(2712, (T, E, 84, 84))
        te, tr, st = _flags(2713, T, E)
        rew = hf.hf_range(2714, (T, E), -2, 2)
        for t in range(T):
            prog.append(f"act {t}")
            b.act()
            prog.append(f"step {t}")
            b.step(frames[t], rew[t], te[t], tr[t], st[t])
        prog.append("finish_rollout")
        b.finish_rollout()
        for q in ("observations", "actions", "values", "logits", "advantages", "returns", "masks", "rewards",
                  "terminals", "truncations", "log_probs", "next_values", "current_obs"):
            prog.append("read_batch " + q)
            b.read_batch(q)
        prog.append("train 2x2")
        b.train(2.5e-4, 2, 2)   # metric storage allocated
        prog.append("train 3x4")
        b.train(2.5e-4, 3, 4)   # ... and outgrown (used to hipFree the old planes)
        prog.append("forward 3")
        b.forward(hf.hf_bytes(2715, (3, 4, 84, 84)))
        prog.append("forward E")
        b.forward(hf.hf_bytes(2716, (E, 4, 84, 84)))  # staging outgrown
        prog.append("export_params")
        b.export_params()
        z = np.zeros((2, 3), np.uint8)
        prog.append("gae operator")
        adv = pkg.gae.gae(np.zeros((2, 3), np.float32), np.ones((2, 3), np.float32), np.full((2, 3), .5, np.float32),
                          np.full(2, .5, np.float32), z, z, z, .99, .95)
        assert np.isfinite(adv).all()
        prog.append("resize operator")
        pkg.vision.resize_frame_stacked_grayscale_images(np.ones((1, 210, 160), np.float32))
        prog.append("done")
        err.append(b)  # (closed below, after A's release: aleppo_destroy frees memory and that does wait)
    except Exception as e:  # noqa: BLE001
        err.append(e)
    done.set()

th = threading.Thread(target=other, daemon=True)
th.start()
finished = done.wait(60)
stuck = prog[-1:]  # (what B was inside of when the wait ran out)
# release A whatever happened, so that a regression fails the test instead of hanging the box
ctypes.memmove(fbuf, hf.hf_bytes(2717, (E, 84, 84)).ctypes.data, E * 7056)
ctypes.memset(sbuf, 0, E)
a.release_step(np.zeros(E, np.float32), np.zeros(E, np.uint8), np.zeros(E, np.uint8))
th.join(120)
assert finished, f"context B / a stateless operator waited for context A's parked stream in: {stuck}"
assert err and not isinstance(err[0], Exception), err
a.act()  # A is intact: the armed slot ran after the release
err[0].close()
a.host_free(fbuf)
a.host_free(sbuf)
a.close()
print("parked-contexts ok")
