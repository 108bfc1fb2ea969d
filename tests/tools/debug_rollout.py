"""debug helper (not a test)"""
import sys, os
import numpy as np
_T = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/
sys.path.insert(0, _T)
sys.path.insert(0, os.path.dirname(_T))
import hashfill as hf
import oracle_lib as orc
from __graft_entry__ import load_package
pkg = load_package()
E, T, A, H = 6, 9, 6, 64
params = hf.fill_params(610, H, A)
eng = pkg.Engine(E, T, A, H)
eng.load_params(params)
rng = np.random.default_rng(3)
start = np.ones(E, np.uint8); rewards = np.zeros(E, np.float32)
for t in range(T):
    eng.act(rng.exponential(size=(E, A)).astype(np.float32))
    frames = hf.hf_bytes(2000 + t, (E, 84, 84))
    u = rng.random(E)
    term = ((u < 0.15) & (start == 0)).astype(np.uint8)
    rewards = np.where(start == 1, rewards, rng.integers(-3, 4, E)).astype(np.float32)
    eng.step(frames, rewards, term, np.zeros(E, np.uint8), start)
    start = term.copy()
eng.finish_rollout(rng.exponential(size=(E, A)).astype(np.float32))
b = {k: eng.read_batch(k) for k in pkg.FIELDS if k != "current_obs"}
N = E * T
args = (b["observations"].reshape(N, 4, 84, 84), b["actions"].ravel(), b["log_probs"].reshape(N, A),
        b["advantages"].ravel(), b["returns"].ravel(), b["masks"].ravel())
names = ["c1w", "c1b", "c2w", "c2b", "c3w", "c3b", "fcw", "fcb", "aw", "ab", "vw", "vb"]
offs = orc.param_offsets(H, A)
def cmp(tag, m, w, g):
    print(tag, "loss", m["loss"].ravel(), w["loss"].ravel(), "norm", m["grad_norm"].ravel(), w["grad_norm"].ravel())
    wg = w["last_grads"]
    for k in range(12):
        a, c = g[offs[k]:offs[k + 1]], wg[offs[k]:offs[k + 1]]
        print(f"   {names[k]:4s} |ours|={np.linalg.norm(a):10.6f} |ref|={np.linalg.norm(c):10.6f} maxdiff={np.abs(a - c).max():.3e}")
for M in (1, 3):
    e2 = pkg.Engine(E, T, A, H); e2.load_params(params); e2.set_batch(*args)
    m = e2.train(1e-3, 1, M); w = orc.train(params, H, A, *args, 1, M, lr=1e-3)
    cmp(f"set_batch M={M}", m, w, e2.export_grads())
    print("   params maxdiff", np.abs(e2.export_params() - w["params"]).max())
    e2.close()
m = eng.train(1e-3, 1, 3); w = orc.train(params, H, A, *args, 1, 3, lr=1e-3)
cmp("rollout M=3", m, w, eng.export_grads())
