"""acting micro-bench helper (not a test)"""
import os, sys, time, json
import numpy as np
_T = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/
sys.path.insert(0, _T)
sys.path.insert(0, os.path.dirname(_T))
import hashfill as hf
from __graft_entry__ import load_package
pkg = load_package()
import ctypes as C
E, T, A, H = 128, 128, 4, 512
eng = pkg.Engine(E, T, A, H, precision=pkg.BF16, max_minibatch=4096)
eng.load_params(hf.fill_params(310, H, A))
frames = np.zeros((E, 84, 84), np.uint8); z = np.zeros(E, np.uint8); r = np.zeros(E, np.float32)
for nopub in (0, 1):
    pkg.lib().aleppo_set_option(eng._ctx, 1, nopub)
    for rep in range(2):
        t0 = time.perf_counter()
        for t in range(T):
            eng.act(); eng.step(frames, r, z, z, z)
        eng.finish_rollout()
        dt = time.perf_counter() - t0
    eng.profile(True); eng.profile_reset()
    for t in range(T):
        eng.act(); eng.step(frames, r, z, z, z)
    eng.finish_rollout()
    res = {k: round(eng.profile_read(k)[0] * 1e3, 1) for k in ("ingest", "conv1_fwd", "fc_fwd", "infer_head")}
    eng.profile(False)
    print(json.dumps({"no_publish": nopub, "slot_us": round(dt / T * 1e6, 1), **res}))
