"""developer tool: wall time of the PPO update alone at the BASELINE configs[1] shape (4 epochs x 4 minibatches of 4096, bf16),
no per-kernel profiling (no event records between the kernels): `python tests/tools/upd_time.py [reps]`.  A/B switches come from
the environment (ALEPPO_SCHED, ALEPPO_WG_GRID, ALEPPO_BWD_STREAMS, ...); run it under `rocprofv3 --kernel-trace` +
tests/tools/timeline.py for the kernel timeline of one minibatch."""
import json, os, sys, time
import numpy as np
_T = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _T)
sys.path.insert(0, os.path.dirname(_T))
import hashfill as hf
from __graft_entry__ import load_package
pkg = load_package()
E, T, A, H, M, EP = 128, 128, int(os.environ.get("KB_A", "4")), 512, int(os.environ.get("KB_M", "4")), 4
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng = pkg.Engine(E, T, A, H, precision=pkg.BF16, max_minibatch=E * T // M)
eng.load_params(hf.fill_params(310, H, A))
if os.environ.get("KB_FORCE_COMM"):  # the data-parallel schedule through a 1-rank RCCL communicator
    eng.comm_init(pkg.Engine.comm_unique_id())
    eng.set_option(pkg.OPT_FORCE_COMM, 1)
rng = np.random.default_rng(0)
N = E * T
obs = rng.integers(0, 256, (N, 4, 84, 84), dtype=np.uint8)
eng.set_batch(obs, rng.integers(0, A, N), np.full((N, A), -np.log(A), np.float32), rng.standard_normal(N).astype(np.float32),
              rng.standard_normal(N).astype(np.float32), np.ones(N, np.uint8))
for _ in range(3):
    eng.train(2.5e-4, EP, M)
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    eng.train(2.5e-4, EP, M)
    ts.append(time.perf_counter() - t0)
ts.sort()
print(json.dumps({"tag": os.environ.get("KB_TAG", ""), "update_ms_median": round(ts[len(ts) // 2] * 1e3, 3),
                  "update_ms_min": round(ts[0] * 1e3, 3), "per_minibatch_us": round(ts[len(ts) // 2] * 1e6 / (EP * M), 1)}))
eng.close()
