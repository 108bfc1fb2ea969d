"""numpy restatement of oracle/hashfill.h (closed-form integer-hash fills).

Golden fixtures (tests/golden/ref_golden.npz) hold only OUTPUTS of the reference; the matching
inputs are regenerated here from (seed, index), bit-identically to the C harness.
"""
import numpy as np

_M = np.uint32(0xFFFFFFFF)


def hf_u32(seed, n_or_idx):
    if isinstance(n_or_idx, tuple):
        raise TypeError("hf_u32 / hf_unit take a count or an index array, not a shape: use hf_range / reshape")
    idx = np.arange(n_or_idx, dtype=np.uint32) if np.isscalar(n_or_idx) else np.asarray(n_or_idx, dtype=np.uint32)
    with np.errstate(over="ignore"):
        x = idx * np.uint32(0x9E3779B1) + np.uint32(seed) * np.uint32(0x85EBCA77) + np.uint32(0x165667B1)
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return x


def hf_unit(seed, n_or_idx):
    return (hf_u32(seed, n_or_idx) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def hf_range(seed, shape, lo, hi):
    n = int(np.prod(shape))
    lo = np.float32(lo)
    hi = np.float32(hi)
    return (lo + (hi - lo) * hf_unit(seed, n)).astype(np.float32).reshape(shape)


def hf_bytes(seed, shape):
    n = int(np.prod(shape))
    return (hf_u32(seed, n) >> np.uint32(24)).astype(np.uint8).reshape(shape)


# ---- parameter fill shared with ref_harness.cc::fill_params
def param_shapes(H, A):
    return [(32, 4, 8, 8), (32,), (64, 32, 4, 4), (64,), (64, 64, 3, 3), (64,), (H, 3136), (H,), (A, H), (A,),
            (1, H), (1,)]


def fill_params(seed_base, H, A):
    """flat float32 vector in libtorch parameters() order."""
    fan_in = [256, 0, 512, 0, 576, 0, 3136, 0, H, 0, H, 0]
    out = []
    for k, shp in enumerate(param_shapes(H, A)):
        b = np.float32(0.05) if k % 2 == 1 else np.sqrt(np.float32(6.0) / np.float32(fan_in[k])).astype(np.float32)
        out.append(hf_range(seed_base + k, shp, -b, b).ravel())
    return np.concatenate(out).astype(np.float32)


def g1_inputs():
    """inputs of fixture G1 (ref_harness.cc gen(), 'G1'): env-major [E,T] arrays."""
    E = T = 128
    idx = np.arange(E * T, dtype=np.uint32)
    r = np.where(hf_unit(101, idx) < np.float32(0.3), hf_range(102, (E * T,), -3.0, 3.0), np.float32(0)).astype(
        np.float32).reshape(E, T)
    v = hf_range(103, (E, T), -1.0, 1.0)
    nv = hf_range(104, (E,), -1.0, 1.0)
    u = hf_unit(105, idx).reshape(E, T)
    s0 = hf_unit(106, E) < np.float32(0.25)
    term = np.zeros((E, T), np.uint8)
    trunc = np.zeros((E, T), np.uint8)
    start = np.zeros((E, T), np.uint8)
    for e in range(E):
        prev_end = False
        for t in range(T):
            st = (t == 0 and s0[e]) or prev_end
            te = tr = False
            if not st:
                te = u[e, t] < np.float32(0.03)
                tr = (not te) and u[e, t] < np.float32(0.04)
            prev_end = te or tr
            start[e, t], term[e, t], trunc[e, t] = st, te, tr
    return r, v, nv, term, trunc, start
