import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(HERE, "golden", "ref_golden.npz"), allow_pickle=False)


# the six known-answer inputs of the reference's test/ai/gae-test.cc (:6,:41,:80,:114,:148,:205),
# re-typed from the NUMBERS in that file: (rewards, values, next_values, terminals, truncations, starts)
GAE_KATS = [
    ([[1, 1, 1]], [[.5, .5, .5]], [.5], [[0, 0, 0]], [[0, 0, 0]], [[0, 0, 0]]),
    ([[1, 1, 1]], [[.5, .5, .5]], [0.], [[0, 0, 1]], [[0, 0, 0]], [[0, 0, 0]]),
    ([[1, 1, 1]], [[.5, .5, .5]], [0.], [[0, 0, 0]], [[0, 0, 1]], [[0, 0, 0]]),
    ([[1, 1, 1]], [[.5, .5, .5]], [.5], [[0, 0, 0]], [[0, 0, 0]], [[0, 1, 0]]),
    ([[1, 1, 1], [.5, .5, .5]], [[.5, .5, .5], [.3, .3, .3]], [.5, .3], [[0, 0, 0], [0, 0, 1]],
     [[0, 0, 0], [0, 0, 0]], [[0, 1, 0], [0, 0, 0]]),
    ([[1, 1, 1, 1, 1]], [[.5] * 5], [.5], [[0, 0, 1, 0, 0]], [[0, 0, 0, 0, 1]], [[0, 0, 0, 1, 0]]),
]


def gae_kat_expected(r, v, nv, term, trunc, start, gamma=0.99, lam=0.95):
    """the scalar loop the reference's tests use to compute their expected values (gae-test.cc:182-201)"""
    r, v = np.asarray(r, np.float32), np.asarray(v, np.float32)
    E, T = r.shape
    out = np.zeros((E, T), np.float32)
    g, l = np.float32(gamma), np.float32(lam)
    for j in range(E):
        last = np.float32(0)
        n = np.float32(nv[j])
        for i in range(T - 1, -1, -1):
            delta = r[j, i] + g * n - v[j, i]
            if start[j][i]:
                a = np.float32(0)
            elif term[j][i]:
                a = r[j, i] - v[j, i]
            elif trunc[j][i]:
                a = delta
            else:
                a = delta + g * l * last
            out[j, i] = a
            last = a
            n = v[j, i]
    return out
