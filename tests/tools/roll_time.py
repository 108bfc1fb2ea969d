"""developer tool: wall time of the 128-slot replay rollout alone (E = 128, bf16, BASELINE configs[1] acting shape), raw frame
pairs or 84x84 frames resident in HBM: `python tests/tools/roll_time.py [raw|84] [reps]`.  Run under `rocprofv3 --kernel-trace`
+ tests/tools/slot_timeline.py for the per-slot kernel timeline."""
import ctypes, json, os, sys, time
import numpy as np
_T = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _T)
sys.path.insert(0, os.path.dirname(_T))
import hashfill as hf
from __graft_entry__ import load_package
pkg = load_package()
kind = sys.argv[1] if len(sys.argv) > 1 else "raw"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
E, T, A, H = 128, 128, 4, 512
per_env = 2 * 210 * 160 if kind == "raw" else 84 * 84
eng = pkg.Engine(E, T, A, H, precision=pkg.BF16, max_minibatch=4096)
eng.load_params(hf.fill_params(310, H, A))
hip = ctypes.CDLL("libamdhip64.so")
dev = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(dev), ctypes.c_size_t(T * E * per_env)) == 0
host = np.random.default_rng(0).integers(0, 128, T * E * per_env, dtype=np.uint8) * 2
assert hip.hipMemcpy(dev, host.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(host.nbytes), 1) == 0
z = np.zeros((T, E), np.uint8)
st = z.copy(); st[0] = 1
rew = np.zeros((T, E), np.float32)
fk = pkg.FRAMES_RAW_PAIR if kind == "raw" else pkg.FRAMES_84
ts = []
for i in range(reps + 3):
    t0 = time.perf_counter()
    eng.replay_rollout(dev.value, fk, E * per_env, rew, z, z, st)
    t1 = time.perf_counter()
    eng.finish_rollout()
    if i >= 3:
        ts.append(t1 - t0)
ts.sort()
print(json.dumps({"tag": os.environ.get("KB_TAG", ""), "frames": kind, "rollout_ms_median": round(ts[len(ts) // 2] * 1e3, 3),
                  "per_slot_us": round(ts[len(ts) // 2] * 1e6 / T, 2)}))
eng.close()
