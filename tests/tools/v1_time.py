"""developer tool: the reference's configs/v1.yaml shape (4096 envs x T = 5, 1 epoch x 16 minibatches of 1280, bf16, 84x84 frames
in HBM) - rollout + update wall time per step, like bench.py's v1_shape leg but without torch: `python tests/tools/v1_time.py`."""
import ctypes, json, os, sys, time
import numpy as np
_T = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _T)
sys.path.insert(0, os.path.dirname(_T))
import hashfill as hf
from __graft_entry__ import load_package
pkg = load_package()
E, T, A, H, epochs, M = 4096, 5, 4, 512, 1, 16
eng = pkg.Engine(E, T, A, H, precision=pkg.BF16, clip_param=0.2, value_loss_coef=0.4, seed=7, max_minibatch=E * T // M)
eng.load_params(hf.fill_params(310, H, A))
hip = ctypes.CDLL("libamdhip64.so")
dev = ctypes.c_void_p()
nb = T * E * 84 * 84
assert hip.hipMalloc(ctypes.byref(dev), ctypes.c_size_t(nb)) == 0
host = np.random.default_rng(7).integers(0, 256, nb, dtype=np.uint8)
assert hip.hipMemcpy(dev, host.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(nb), 1) == 0
rng = np.random.default_rng(7)
rew = np.where(rng.random((T, E)) < 0.05, 1.0, 0.0).astype(np.float32)
z = np.zeros((T, E), np.uint8)
st0 = np.ones((T, E), np.uint8); st0[1:] = 0
slot = E * 84 * 84
split = [0.0, 0.0, 0.0]
def one(first, count=False):
    st = st0 if first else z
    ra, za, sa = rew.ctypes.data, z.ctypes.data, st.ctypes.data
    t0 = time.perf_counter()
    for t in range(T):
        eng.act_fast()
        eng.step_ptr(dev.value + t * slot, pkg.DEVICE, pkg.FRAMES_84, ra + 4 * E * t, za + E * t, za + E * t, sa + E * t)
    t1 = time.perf_counter()
    eng.finish_rollout()
    t2 = time.perf_counter()
    eng.train(2.5e-4, epochs, M)
    t3 = time.perf_counter()
    if count:
        split[0] += t1 - t0; split[1] += t2 - t1; split[2] += t3 - t2
one(True)
for _ in range(3):
    one(False)
steps = 12
eng.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    one(False, True)
eng.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"tag": os.environ.get("KB_TAG", ""), "env_steps_per_s": round(E * T * steps / dt), "ms_per_step": round(dt / steps * 1e3, 3),
                  "slots_ms": round(split[0] / steps * 1e3, 3), "finish_ms": round(split[1] / steps * 1e3, 3),
                  "train_ms": round(split[2] / steps * 1e3, 3)}))
eng.close()
