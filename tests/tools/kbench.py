"""kernel micro-bench helper (not a test): per-kernel times of one update at the C1 minibatch shape."""
import os, sys, json
import numpy as np
_T = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/
sys.path.insert(0, _T)
sys.path.insert(0, os.path.dirname(_T))
import hashfill as hf
from __graft_entry__ import load_package
pkg = load_package()
E, T, A, H, M = 128, 128, int(os.environ.get("KB_A", "4")), 512, int(os.environ.get("KB_M", "4"))
prec = pkg.FP32 if os.environ.get("KB_FP32") else pkg.BF16
eng = pkg.Engine(E, T, A, H, precision=prec, max_minibatch=E * T // M)
eng.load_params(hf.fill_params(310, H, A))
rng = np.random.default_rng(0)
N = E * T
obs = rng.integers(0, 256, (N, 4, 84, 84), dtype=np.uint8)
eng.set_batch(obs, rng.integers(0, A, N), np.full((N, A), -np.log(A), np.float32), rng.standard_normal(N).astype(np.float32),
              rng.standard_normal(N).astype(np.float32), np.ones(N, np.uint8))
if not os.environ.get("KB_COSCHED"):  # isolated per-kernel times: every kernel alone on the main stream
    pkg.lib().aleppo_set_option(eng._ctx, pkg.OPT_SERIAL_UPDATE, 1)
eng.train(2.5e-4, 1, M)
eng.profile(True); eng.profile_reset()
eng.train(2.5e-4, 2, M)
res = {k: round(eng.profile_read(k)[0] * 1e3, 1) for k in pkg.KERNEL_CLASSES if eng.profile_read(k)[1]}
print(json.dumps({"tag": os.environ.get("KB_TAG", ""), "sum_us": round(sum(res.values()) + res.get("reduce", 0), 1), **res}))
