"""ctypes binding of oracle/liboracle.so - TEST INFRASTRUCTURE (the checker, never the product)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        src = os.path.join(ROOT, "oracle", "oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        _LIB = C.CDLL(path)
        _LIB.oracle_param_count.restype = C.c_size_t
        _LIB.oracle_acts_per_sample.restype = C.c_size_t
        _LIB.oracle_ppo_loss.restype = C.c_float
        _LIB.oracle_clip_grad_norm.restype = C.c_float
    return _LIB


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def c8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def cf(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def param_count(H, A):
    return lib().oracle_param_count(C.c_int(H), C.c_int(A))


def param_offsets(H, A):
    o = (C.c_size_t * 13)()
    lib().oracle_param_offsets(H, A, o)
    return list(o)


def gae(r, v, nv, term, trunc, start, gamma=0.99, lam=0.95):
    r, v, nv = cf(r), cf(v), cf(nv)
    E, T = r.shape
    adv = np.zeros((E, T), np.float32)
    rc = lib().oracle_gae(_p(adv, C.c_float), _p(r, C.c_float), _p(v, C.c_float), _p(nv, C.c_float),
                          _p(c8(term), C.c_uint8), _p(c8(trunc), C.c_uint8), _p(c8(start), C.c_uint8), E, T,
                          C.c_float(gamma), C.c_float(lam))
    if rc:
        raise ValueError("Episode starts, terminals, and truncations must be mutually exclusive.")
    return adv


def buffer_get(r, v, nv, term, trunc, start, gamma=0.99, lam=0.95):
    r = cf(r).copy()
    v, nv = cf(v), cf(nv)
    E, T = r.shape
    adv = np.zeros((E, T), np.float32)
    ret = np.zeros((E, T), np.float32)
    masks = np.zeros((E, T), np.uint8)
    term, trunc, start = c8(term), c8(trunc), c8(start)
    rc = lib().oracle_buffer_get(_p(r, C.c_float), _p(v, C.c_float), _p(nv, C.c_float), _p(term, C.c_uint8),
                                 _p(trunc, C.c_uint8), _p(start, C.c_uint8), _p(adv, C.c_float),
                                 _p(ret, C.c_float), _p(masks, C.c_uint8), E, T, C.c_float(gamma), C.c_float(lam))
    if rc:
        raise ValueError("flags overlap")
    return dict(rewards=r, advantages=adv, returns=ret, masks=masks)


def update_observations(obs, frames, start):
    obs = c8(obs).copy()
    frames, start = c8(frames), c8(start)
    E, S = obs.shape[:2]
    lib().oracle_update_observations(_p(obs, C.c_uint8), _p(frames, C.c_uint8), _p(start, C.c_uint8), E, S)
    return obs


def area_resize(img):
    img = cf(img)
    n = int(np.prod(img.shape[:-2]))
    out = np.zeros(img.shape[:-2] + (84, 84), np.float32)
    lib().oracle_area_resize_f32(_p(img, C.c_float), _p(out, C.c_float), n, img.shape[-2], img.shape[-1], 84, 84)
    return out


def rgb_to_gray(img):
    img = cf(img)  # [..., 3, h, w]
    n = int(np.prod(img.shape[:-3]))
    P = img.shape[-1] * img.shape[-2]
    out = np.zeros(img.shape[:-3] + img.shape[-2:], np.float32)
    lib().oracle_rgb_to_gray(_p(img, C.c_float), _p(out, C.c_float), n, P)
    return out


def preprocess(raw, lut=None):
    raw = c8(raw)  # [E, nf, 210, 160]
    E, nf = raw.shape[:2]
    out = np.zeros((E, 84, 84), np.uint8)
    lutc = None if lut is None else c8(lut)
    lib().oracle_preprocess_u8(_p(raw, C.c_uint8), _p(lutc, C.c_uint8), _p(out, C.c_uint8), E, nf)
    return out


def log_softmax(z):
    z = cf(z)
    out = np.zeros_like(z)
    lib().oracle_log_softmax(_p(z, C.c_float), _p(out, C.c_float), z.shape[0], z.shape[1])
    return out


def softmax(z):
    z = cf(z)
    out = np.zeros_like(z)
    lib().oracle_softmax(_p(z, C.c_float), _p(out, C.c_float), z.shape[0], z.shape[1])
    return out


def sample(probs, q):
    probs, q = cf(probs), cf(q)
    a = np.zeros(probs.shape[0], np.int64)
    lib().oracle_sample(_p(probs, C.c_float), _p(q, C.c_float), _p(a, C.c_int64), probs.shape[0], probs.shape[1])
    return a


def net_forward(params, H, A, obs, want_acts=False):
    params, obs = cf(params), c8(obs)
    N = obs.shape[0]
    logits = np.zeros((N, A), np.float32)
    values = np.zeros(N, np.float32)
    acts = np.zeros((N, lib().oracle_acts_per_sample(H)), np.float32) if want_acts else None
    lib().oracle_net_forward(_p(params, C.c_float), H, A, _p(obs, C.c_uint8), N, _p(logits, C.c_float),
                             _p(values, C.c_float), _p(acts, C.c_float))
    return (logits, values, acts) if want_acts else (logits, values)


def net_backward(params, H, A, acts, dlogits, dvalues):
    params, acts, dlogits, dvalues = cf(params), cf(acts), cf(dlogits), cf(dvalues)
    g = np.zeros(param_count(H, A), np.float32)
    lib().oracle_net_backward(_p(params, C.c_float), H, A, acts.shape[0], _p(acts, C.c_float),
                              _p(dlogits, C.c_float), _p(dvalues, C.c_float), _p(g, C.c_float))
    return g


def ppo_loss(logits, old_logp, actions, adv, values, returns, masks, clip=0.1, c_v=0.5, c_e=0.01, n_mask=0.0):
    logits, old_logp = cf(logits), cf(old_logp)
    B, A = logits.shape
    actions = np.ascontiguousarray(actions, np.int64)
    adv, values, returns, masks = cf(adv), cf(values), cf(returns), c8(masks)
    o = {k: np.zeros(B, np.float32) for k in ("clipped", "value_losses", "entropies", "total_losses", "ratio",
                                               "dvalues")}
    o["dlogits"] = np.zeros((B, A), np.float32)
    o["loss"] = lib().oracle_ppo_loss(
        _p(logits, C.c_float), _p(old_logp, C.c_float), _p(actions, C.c_int64), _p(adv, C.c_float),
        _p(values, C.c_float), _p(returns, C.c_float), _p(masks, C.c_uint8), B, A, C.c_float(clip), C.c_float(c_v),
        C.c_float(c_e), C.c_float(n_mask), _p(o["clipped"], C.c_float), _p(o["value_losses"], C.c_float),
        _p(o["entropies"], C.c_float), _p(o["total_losses"], C.c_float), _p(o["ratio"], C.c_float),
        _p(o["dlogits"], C.c_float), _p(o["dvalues"], C.c_float))
    return o


def adv_norm(adv, masks):
    """masked-sample advantage normalisation (extension, parity unpinned: the reference has none)"""
    a = cf(adv).copy()
    m = c8(masks)
    lib().oracle_adv_norm.restype = C.c_long
    lib().oracle_adv_norm(_p(a, C.c_float), _p(m, C.c_uint8), C.c_long(a.size))
    return a


def clip_grad_norm(g, H, A, max_norm=0.5):
    g = cf(g).copy()
    n = lib().oracle_clip_grad_norm(_p(g, C.c_float), H, A, C.c_float(max_norm))
    return n, g


def adam_step(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-5):
    p, m, v = cf(p).copy(), cf(m).copy(), cf(v).copy()
    g = cf(g)
    lib().oracle_adam_step(_p(p, C.c_float), _p(g, C.c_float), _p(m, C.c_float), _p(v, C.c_float),
                           C.c_size_t(p.size), C.c_double(lr), C.c_double(beta1), C.c_double(beta2), C.c_double(eps),
                           C.c_int64(step))
    return p, m, v


def train(params, H, A, obs, actions, old_logp, adv, returns, masks, epochs, M, lr=2.5e-4, clip=0.1, c_v=0.5,
          c_e=0.01, max_norm=0.5, adam=None):
    """returns dict(params, loss[e,M], grad_norm[e,M], per-sample [e,M,B] arrays, last_grads, adam)"""
    params = cf(params).copy()
    obs = c8(obs)
    N = obs.shape[0]
    B = N // M
    actions = np.ascontiguousarray(actions, np.int64)
    old_logp, adv, returns, masks = cf(old_logp), cf(adv), cf(returns), c8(masks)
    if adam is None:
        adam = dict(m=np.zeros_like(params), v=np.zeros_like(params), step=0)
    m, v = cf(adam["m"]).copy(), cf(adam["v"]).copy()
    step = C.c_int64(adam["step"])
    out = {k: np.zeros((epochs, M, B), np.float32) for k in ("total_losses", "ratio", "entropies", "value_losses",
                                                              "clipped")}
    loss = np.zeros((epochs, M), np.float32)
    gn = np.zeros((epochs, M), np.float32)
    lg = np.zeros_like(params)
    rc = lib().oracle_train(_p(params, C.c_float), _p(m, C.c_float), _p(v, C.c_float), C.byref(step), H, A,
                            _p(obs, C.c_uint8), _p(actions, C.c_int64), _p(old_logp, C.c_float), _p(adv, C.c_float),
                            _p(returns, C.c_float), _p(masks, C.c_uint8), N, epochs, M, C.c_double(lr),
                            C.c_float(clip), C.c_float(c_v), C.c_float(c_e), C.c_float(max_norm),
                            _p(loss, C.c_float), _p(gn, C.c_float), _p(out["total_losses"], C.c_float),
                            _p(out["ratio"], C.c_float), _p(out["entropies"], C.c_float),
                            _p(out["value_losses"], C.c_float), _p(out["clipped"], C.c_float), _p(lg, C.c_float))
    if rc:
        raise RuntimeError("Batch size must be divisible by num_mini_batches")
    out.update(params=params, loss=loss, grad_norm=gn, last_grads=lg, adam=dict(m=m, v=v, step=step.value))
    return out
