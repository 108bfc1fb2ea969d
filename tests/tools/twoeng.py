"""Experiment (not a test): does interleaving two independent half-size updates on separate streams beat one full-size
update?  Two engines (E/2 envs each, minibatch B/2) trained from two threads vs one engine (E envs, minibatch B)."""
import os, sys, time, threading
import numpy as np
_T = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _T); sys.path.insert(0, os.path.dirname(_T))
import hashfill as hf
from __graft_entry__ import load_package
pkg = load_package()
T, A, H, M, EP = 128, 4, 512, 4, 4

def make(E):
    eng = pkg.Engine(E, T, A, H, precision=pkg.BF16, max_minibatch=E * T // M)
    eng.load_params(hf.fill_params(310, H, A))
    rng = np.random.default_rng(E)
    N = E * T
    obs = rng.integers(0, 256, (N, 4, 84, 84), dtype=np.uint8)
    eng.set_batch(obs, rng.integers(0, A, N), np.full((N, A), -np.log(A), np.float32), rng.standard_normal(N).astype(np.float32),
                  rng.standard_normal(N).astype(np.float32), np.ones(N, np.uint8))
    eng.train(2.5e-4, 1, M)
    return eng

def timed(engs, reps=5):
    best = 1e9
    for _ in range(reps):
        ths = [threading.Thread(target=lambda e=e: e.train(2.5e-4, EP, M)) for e in engs]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3

one = make(128)
print("one engine  E=128 B=4096: %.2f ms per update" % timed([one]))
del one
for k in (2, 4):
    engs = [make(128 // k) for _ in range(k)]
    print("%d engines E=%d B=%d concurrently: %.2f ms for the same samples" % (k, 128 // k, 128 // k * T // M, timed(engs)))
    print("   one of them alone: %.2f ms" % timed(engs[:1]))
    del engs
