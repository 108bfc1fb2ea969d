"""ale-libtorch-ppo_amd: host-side mirror (Python, ctypes) of the PPO-over-ALE hot path whose
compute lives in libaleppo.so (hand-written HIP for gfx950, C ABI in include/aleppo.h).

The directory name has a hyphen, so import it through ``__graft_entry__.load_package()``
(module name ``ale_libtorch_ppo_amd``).

Nothing in this package computes on the CPU: every operator calls into the HIP library and
raises if the library or a gfx950 device is missing.  The CPU oracle under ``oracle/`` is test
infrastructure and is never imported from here.

Sub-namespaces mirror the reference's C++ namespaces for this path:
  gae.gae                      <- ai::gae::gae                       (src/ai/gae.h:4-7)
  vision.*                     <- ai::vision::*                      (src/ai/vision.h:6-17)
  losses.compute               <- ai::ppo::losses::compute           (src/ai/ppo/losses.h:22-27)
  Engine                       <- Network + Adam + Rollout/Buffer + ppo::train::train as driven by
                                  main() (src/bin/train.cc:358-458)
"""
import ctypes as C
import os
from types import SimpleNamespace

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (ALEPPO_LIB_PATH: developer A/B runs of two builds inside one gpurun call; never set by tests or bench)
LIB_PATH = os.environ.get("ALEPPO_LIB_PATH") or os.path.join(_HERE, "libaleppo.so")

OK = 0
ERR_INVALID_ARGUMENT, ERR_RUNTIME, ERR_HIP, ERR_NO_DEVICE = -1, -2, -3, -4
FP32, BF16 = 0, 1
HOST, DEVICE, HOST_MAPPED = 0, 1, 2
FRAMES_84, FRAMES_RAW_PAIR = 0, 1
ROLLOUT_FP32, ROLLOUT_FP16 = 0, 1
ABI_VERSION = 2
UNIQUE_ID_BYTES = 128

FIELDS = dict(observations=0, actions=1, rewards=2, masks=3, logits=4, values=5, advantages=6, returns=7,
              log_probs=8, terminals=9, truncations=10, current_obs=11, next_values=12)
METRIC_FIELDS = dict(total_losses=0, clipped_losses=1, value_losses=2, entropies=3, ratio=4)
KERNEL_CLASSES = dict(ingest=0, gae=1, head=2, adam=3, conv1_fwd=4, conv2_fwd=5, conv3_fwd=6, fc_fwd=7, fc_dgrad=8,
                      fc_wgrad=9, conv3_dgrad=10, conv3_wgrad=11, conv2_dgrad=12, conv2_wgrad=13, conv1_wgrad=14,
                      reduce=15, infer_head=16, act_fused=17, conv_fwd=18, conv_bwd=19)

# every symbol include/aleppo.h declares (checked by tests/test_abi.py against the header text)
# aleppo_set_option keys (include/aleppo.h)
OPT_GENERIC_CONV, OPT_DEBUG_NO_PUBLISH, OPT_FORCE_COMM, OPT_SERIAL_UPDATE = 0, 1, 2, 3
OPT_FC_PIPE, OPT_FUSED_ACT, OPT_UPDATE_GRAPH = 4, 6, 7
OPT_GATE_TIMEOUT_MS = 9
OPT_FUSED_FWD = 10
OPT_FUSED_BWD = 11

EXPORTS = [
    "aleppo_abi_version", "aleppo_create", "aleppo_destroy", "aleppo_last_error", "aleppo_param_count",
    "aleppo_load_params", "aleppo_export_params", "aleppo_export_grads", "aleppo_act", "aleppo_push_frames",
    "aleppo_set_gray_lut", "aleppo_record_step", "aleppo_step", "aleppo_finish_rollout", "aleppo_train",
    "aleppo_read_train_metric", "aleppo_set_batch", "aleppo_read_batch", "aleppo_forward", "aleppo_comm_unique_id",
    "aleppo_comm_init", "aleppo_gae", "aleppo_vision_resize_area", "aleppo_vision_rgb_to_gray", "aleppo_preprocess",
    "aleppo_update_observations", "aleppo_ppo_loss", "aleppo_sample", "aleppo_profile_enable", "aleppo_profile_read",
    "aleppo_profile_reset", "aleppo_synchronize", "aleppo_set_option", "aleppo_export_optimizer",
    "aleppo_import_optimizer", "aleppo_replay_rollout", "aleppo_get_option",
    "aleppo_host_alloc", "aleppo_host_free", "aleppo_arm_step", "aleppo_release_step", "aleppo_device_check",
]


class AleppoError(RuntimeError):
    """ALEPPO_ERR_RUNTIME / HIP / NO_DEVICE (std::runtime_error in the reference)."""


class AleppoInvalidArgument(ValueError):
    """ALEPPO_ERR_INVALID_ARGUMENT (std::invalid_argument in the reference)."""


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device_ordinal", C.c_int32), ("world_size", C.c_int32),
                ("rank", C.c_int32), ("num_envs", C.c_int32), ("horizon", C.c_int32), ("num_actions", C.c_int32),
                ("hidden_size", C.c_int32), ("frame_stack", C.c_int32), ("precision", C.c_int32),
                ("advantage_norm", C.c_int32), ("max_minibatch", C.c_int32), ("rollout_precision", C.c_int32),
                ("gamma", C.c_float),
                ("lambda_", C.c_float), ("clip_param", C.c_float), ("value_loss_coef", C.c_float),
                ("entropy_coef", C.c_float), ("max_gradient_norm", C.c_float), ("adam_beta1", C.c_float),
                ("adam_beta2", C.c_float), ("adam_eps", C.c_float), ("seed", C.c_uint64)]


class MinibatchMetrics(C.Structure):
    _fields_ = [("loss", C.c_float), ("grad_norm", C.c_float), ("clipped_loss", C.c_float),
                ("value_loss", C.c_float), ("entropy", C.c_float), ("ratio", C.c_float), ("mask_count", C.c_float)]


_lib = None


def lib():
    """Load libaleppo.so; fail loudly when the HIP extension has not been built (no CPU fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AleppoError(f"HIP extension missing: {LIB_PATH} (run __graft_entry__.build()); "
                              "this package has no CPU fallback")
        _lib = C.CDLL(LIB_PATH)
        _lib.aleppo_last_error.restype = C.c_char_p
        _lib.aleppo_last_error.argtypes = [C.c_void_p]
        for name in EXPORTS:
            getattr(_lib, name)  # AttributeError here = ABI drift
    return _lib


def _check(rc, ctx=None):
    if rc == OK:
        return
    msg = lib().aleppo_last_error(ctx)
    msg = msg.decode() if msg else f"aleppo error {rc}"
    if rc == ERR_INVALID_ARGUMENT:
        raise AleppoInvalidArgument(msg)
    raise AleppoError(msg)


def device_check(device=0):
    """aleppo_device_check: raises AleppoError ("no CPU fallback") unless HIP device `device` is a gfx950"""
    _check(lib().aleppo_device_check(C.c_int(device)))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------------------- stateless operators
def _gae(advantages, rewards, values, next_values, terminals, truncations, episode_starts, gamma, lam, device=0):
    """ai::gae::gae (src/ai/gae.h:4-7): writes `advantages` in place; same argument checks/errors."""
    rewards, values, next_values = _f32(rewards), _f32(values), _f32(next_values)
    if rewards.ndim != 2 or values.ndim != 2 or next_values.ndim != 1 or np.ndim(terminals) != 2 or \
            np.ndim(truncations) != 2 or np.ndim(episode_starts) != 2:
        raise AleppoInvalidArgument("All input tensors must be 2D except next_values which must be 1D.")
    E, T = rewards.shape
    te, tr, st = _u8(terminals), _u8(truncations), _u8(episode_starts)
    if values.shape[0] != E or te.shape[0] != E or tr.shape[0] != E or st.shape[0] != E or next_values.shape[0] != E:
        raise AleppoInvalidArgument("Input tensors must have compatible dimensions.")
    out = np.zeros((E, T), np.float32)
    _check(lib().aleppo_gae(C.c_int(device), _ptr(out), _ptr(rewards), _ptr(values), _ptr(next_values), _ptr(te),
                            _ptr(tr), _ptr(st), C.c_int64(E), C.c_int64(T), C.c_float(gamma), C.c_float(lam)))
    advantages[...] = out
    return advantages


def _resize_frame_stacked_grayscale_images(images, device=0):
    """ai::vision::resize_frame_stacked_grayscale_images (vision.cc:22-32): f32 [B,S,210,160] -> [B,S,84,84]"""
    images = _f32(images)
    assert images.shape[-2:] == (210, 160)
    lead = images.shape[:-2]
    n = int(np.prod(lead)) if lead else 1
    out = np.zeros((n, 84, 84), np.float32)
    _check(lib().aleppo_vision_resize_area(C.c_int(device), _ptr(images), _ptr(out), C.c_int64(n)))
    return out.reshape(lead + (84, 84))


def _rgb_to_grayscale_frame_stacked_images(images, device=0):
    """ai::vision::rgb_to_grayscale_frame_stacked_images (vision.cc:71-84): f32 [B,S,3,84,84] -> [B,S,84,84]"""
    images = _f32(images)
    assert images.shape[-3:] == (3, 84, 84)
    lead = images.shape[:-3]
    n = int(np.prod(lead)) if lead else 1
    out = np.zeros((n, 84, 84), np.float32)
    _check(lib().aleppo_vision_rgb_to_gray(C.c_int(device), _ptr(images), _ptr(out), C.c_int64(n)))
    return out.reshape(lead + (84, 84))


def _preprocess(raw_pairs, lut=None, device=0):
    """fused device preprocessing: u8 [n,2,210,160] (+256-entry LUT) -> u8 [n,84,84]"""
    raw = _u8(raw_pairs)
    assert raw.shape[1:] == (2, 210, 160)
    out = np.zeros((raw.shape[0], 84, 84), np.uint8)
    l = None if lut is None else _u8(lut)
    _check(lib().aleppo_preprocess(C.c_int(device), _ptr(raw), _ptr(l), _ptr(out), C.c_int64(raw.shape[0])))
    return out


def _update_observations(observations, frames, episode_start, device=0):
    """Rollout::update_observations (rollout.cc:184-196) on u8 [E,4,84,84]; returns the new stack."""
    obs = _u8(observations).copy()
    _check(lib().aleppo_update_observations(C.c_int(device), _ptr(obs), _ptr(_u8(frames)), _ptr(_u8(episode_start)),
                                            C.c_int64(obs.shape[0])))
    return obs


def _losses_compute(logits, old_log_probabilities, actions, advantages, values, returns, masks, clip_param,
                    value_loss_coef, entropy_coef, device=0):
    """ai::ppo::losses::compute on normalize_logits(logits) (losses.cc:4-47) + its gradients.
    Returns the reference's Metrics fields plus dlogits / dvalues."""
    logits, olp = _f32(logits), _f32(old_log_probabilities)
    B, A = logits.shape
    o = SimpleNamespace(loss=np.zeros(1, np.float32), clipped_losses=np.zeros(B, np.float32),
                        value_losses=np.zeros(B, np.float32), entropies=np.zeros(B, np.float32),
                        total_losses=np.zeros(B, np.float32), ratio=np.zeros(B, np.float32),
                        dlogits=np.zeros((B, A), np.float32), dvalues=np.zeros(B, np.float32))
    _check(lib().aleppo_ppo_loss(C.c_int(device), _ptr(logits), _ptr(olp), _ptr(_i64(actions)),
                                 _ptr(_f32(advantages)), _ptr(_f32(values)), _ptr(_f32(returns)), _ptr(_u8(masks)),
                                 C.c_int64(B), C.c_int64(A), C.c_float(clip_param), C.c_float(value_loss_coef),
                                 C.c_float(entropy_coef), _ptr(o.loss), _ptr(o.clipped_losses), _ptr(o.value_losses),
                                 _ptr(o.entropies), _ptr(o.total_losses), _ptr(o.ratio), _ptr(o.dlogits),
                                 _ptr(o.dvalues)))
    o.masks = _u8(masks)
    return o


def _sample(probs, q, device=0):
    """torch::multinomial(probs, 1, true) given its Exp(1) noise q (train.cc:374-375): argmax(p/q)."""
    probs, q = _f32(probs), _f32(q)
    a = np.zeros(probs.shape[0], np.int64)
    _check(lib().aleppo_sample(C.c_int(device), _ptr(probs), _ptr(q), _ptr(a), C.c_int64(probs.shape[0]),
                               C.c_int64(probs.shape[1])))
    return a


gae = SimpleNamespace(gae=_gae)
vision = SimpleNamespace(resize_frame_stacked_grayscale_images=_resize_frame_stacked_grayscale_images,
                         rgb_to_grayscale_frame_stacked_images=_rgb_to_grayscale_frame_stacked_images,
                         preprocess=_preprocess)
losses = SimpleNamespace(compute=_losses_compute)
sampling = SimpleNamespace(multinomial_with_noise=_sample)
rollout = SimpleNamespace(update_observations=_update_observations)


# ---------------------------------------------------------------------------- engine
def env_shard(total_environments, world_size, rank):
    """env index e -> (rank = e // E_g, e_local = e % E_g): contiguous env blocks per rank."""
    if total_environments % world_size:
        raise AleppoInvalidArgument("total_environments must be divisible by world_size")
    eg = total_environments // world_size
    return range(rank * eg, (rank + 1) * eg)


class Engine:
    """One rank's hot path: network + Adam + rollout buffer + PPO update, all on one MI355X.

    Mirrors what main() wires together (src/bin/train.cc:358-458).  Per slot:
    ``act`` -> caller steps its emulators -> ``step`` (or push_frames + record_step);
    then ``finish_rollout`` and ``train``."""

    def __init__(self, num_envs, horizon, num_actions=4, hidden_size=512, precision=FP32, gamma=0.99, lam=0.95,
                 clip_param=0.1, value_loss_coef=0.5, entropy_coef=0.01, max_gradient_norm=0.5, device=0,
                 world_size=1, rank=0, seed=42, advantage_norm=False, max_minibatch=0, rollout_precision=ROLLOUT_FP32):
        self.cfg = Config(ABI_VERSION, device, world_size, rank, num_envs, horizon, num_actions, hidden_size, 4,
                          precision, int(advantage_norm), max_minibatch, rollout_precision, gamma, lam, clip_param,
                          value_loss_coef, entropy_coef, max_gradient_norm, 0.0, 0.0, 0.0, seed)
        self._ctx = C.c_void_p()
        _check(lib().aleppo_create(C.byref(self.cfg), C.byref(self._ctx)))
        self.E, self.T, self.A, self.H = num_envs, horizon, num_actions, hidden_size
        n = C.c_size_t()
        _check(lib().aleppo_param_count(self._ctx, C.byref(n)), self._ctx)
        self.param_count = n.value
        self._f_act = lib().aleppo_act
        self._f_step = lib().aleppo_step
        self._f_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p]
        self._act_ptr = C.POINTER(C.c_int64)()
        self._act_out = C.byref(self._act_ptr)
        self.actions = None  # view of the pinned action buffer, valid after the first act

    def close(self):
        if getattr(self, "_ctx", None):
            lib().aleppo_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def _c(self, rc):
        _check(rc, self._ctx)

    # -- parameters (libtorch parameters() order) --
    def load_params(self, flat):
        flat = _f32(flat).ravel()
        self._c(lib().aleppo_load_params(self._ctx, _ptr(flat), C.c_size_t(flat.size)))

    def export_params(self):
        out = np.zeros(self.param_count, np.float32)
        self._c(lib().aleppo_export_params(self._ctx, _ptr(out), C.c_size_t(out.size)))
        return out

    def export_grads(self):
        out = np.zeros(self.param_count, np.float32)
        self._c(lib().aleppo_export_grads(self._ctx, _ptr(out), C.c_size_t(out.size)))
        return out

    # -- checkpoint / resume --
    def state_dict(self):
        m = np.zeros(self.param_count, np.float32)
        v = np.zeros(self.param_count, np.float32)
        step = C.c_int64()
        self._c(lib().aleppo_export_optimizer(self._ctx, _ptr(m), _ptr(v), C.byref(step), C.c_size_t(m.size)))
        return dict(params=self.export_params(), exp_avg=m, exp_avg_sq=v, step=np.int64(step.value))

    def load_state_dict(self, sd):
        self.load_params(sd["params"])  # resets Adam, then restore it
        m, v = _f32(sd["exp_avg"]).ravel(), _f32(sd["exp_avg_sq"]).ravel()
        self._c(lib().aleppo_import_optimizer(self._ctx, _ptr(m), _ptr(v), C.c_int64(int(sd["step"])),
                                              C.c_size_t(m.size)))

    # -- rollout --
    def act(self, noise=None):
        p = C.POINTER(C.c_int64)()
        n = None if noise is None else _f32(noise)
        self._c(lib().aleppo_act(self._ctx, _ptr(n), C.byref(p)))
        self.actions = np.ctypeslib.as_array(p, shape=(self.E,))  # view of the pinned buffer
        return self.actions

    def push_frames(self, frames, episode_start, kind=FRAMES_84, device_ptr=None):
        st = _u8(episode_start)
        if device_ptr is not None:
            self._c(lib().aleppo_push_frames(self._ctx, C.c_void_p(device_ptr), kind, DEVICE, _ptr(st)))
        else:
            self._c(lib().aleppo_push_frames(self._ctx, _ptr(_u8(frames)), kind, HOST, _ptr(st)))

    def record_step(self, rewards, terminated, truncated, episode_start):
        self._c(lib().aleppo_record_step(self._ctx, _ptr(_f32(rewards)), _ptr(_u8(terminated)), _ptr(_u8(truncated)),
                                         _ptr(_u8(episode_start))))

    def step(self, frames, rewards, terminated, truncated, episode_start, kind=FRAMES_84, device_ptr=None):
        fr = C.c_void_p(device_ptr) if device_ptr is not None else _ptr(_u8(frames))
        loc = DEVICE if device_ptr is not None else HOST
        self._c(lib().aleppo_step(self._ctx, fr, kind, loc, _ptr(_f32(rewards)), _ptr(_u8(terminated)),
                                  _ptr(_u8(truncated)), _ptr(_u8(episode_start))))

    def arm_step(self, frames_addr, start_addr, kind=FRAMES_84, noise_next=None):
        """aleppo_arm_step: enqueue the next step (frames / episode-start bytes at mapped host addresses the emulators are
        about to fill) and the next slot's acting kernels behind the release word"""
        n = None if noise_next is None else _f32(noise_next)
        self._c(lib().aleppo_arm_step(self._ctx, C.c_void_p(frames_addr), kind, C.c_void_p(start_addr), _ptr(n)))

    def release_step(self, rewards, terminated, truncated):
        """aleppo_release_step: the emulators are done - release the stream, record the slot's scalars"""
        self._c(lib().aleppo_release_step(self._ctx, _ptr(_f32(rewards)), _ptr(_u8(terminated)), _ptr(_u8(truncated))))

    # -- low-overhead variants for tight host loops: raw addresses, no numpy conversions --
    def act_fast(self):
        """aleppo_act with the built-in RNG; returns nothing (read self.actions, a view of the pinned buffer)"""
        rc = self._f_act(self._ctx, None, self._act_out)
        if rc:
            self._c(rc)

    def step_ptr(self, frames_addr, location, kind, rewards_addr, term_addr, trunc_addr, start_addr):
        """aleppo_step on raw addresses (host arrays must stay alive and be C-contiguous of the right dtype)"""
        rc = self._f_step(self._ctx, frames_addr, kind, location, rewards_addr, term_addr, trunc_addr, start_addr)
        if rc:
            self._c(rc)

    def replay_rollout(self, frames_addr, kind, slot_stride_bytes, rewards, terminated, truncated, episode_start,
                       noise=None, location=DEVICE):
        """aleppo_replay_rollout: the T-slot act/step loop over a recorded trace (frames in device or mapped
        page-locked host memory, host [T][E] scalars, optional [T][E][A] sampling noise)"""
        r, te, tr, st = _f32(rewards), _u8(terminated), _u8(truncated), _u8(episode_start)
        for a in (r, te, tr, st):
            if a.shape != (self.T, self.E):
                raise AleppoInvalidArgument("replay_rollout: scalars must be [T][E]")
        nz = None
        if noise is not None:
            nz = _f32(noise)
            if nz.shape != (self.T, self.E, self.A):
                raise AleppoInvalidArgument("replay_rollout: noise must be [T][E][A]")
        self._c(lib().aleppo_replay_rollout(self._ctx, C.c_void_p(frames_addr), int(kind), int(location),
                                            C.c_size_t(slot_stride_bytes), _ptr(r), _ptr(te), _ptr(tr), _ptr(st),
                                            _ptr(nz)))

    def host_alloc(self, nbytes):
        """mapped page-locked host memory for frame buffers (pass its address with location=HOST_MAPPED)"""
        p = C.c_void_p()
        self._c(lib().aleppo_host_alloc(self._ctx, C.c_size_t(nbytes), C.byref(p)))
        return p.value

    def host_free(self, addr):
        self._c(lib().aleppo_host_free(self._ctx, C.c_void_p(addr)))

    def set_gray_lut(self, lut):
        self._c(lib().aleppo_set_gray_lut(self._ctx, _ptr(_u8(lut))))

    def finish_rollout(self, noise=None):
        n = None if noise is None else _f32(noise)
        self._c(lib().aleppo_finish_rollout(self._ctx, _ptr(n)))

    # -- update --
    def train(self, lr, epochs, num_mini_batches):
        out = (MinibatchMetrics * (epochs * num_mini_batches))()
        self._c(lib().aleppo_train(self._ctx, C.c_double(lr), epochs, num_mini_batches, out))
        keys = [f[0] for f in MinibatchMetrics._fields_]
        return {k: np.array([getattr(m, k) for m in out], np.float32).reshape(epochs, num_mini_batches) for k in keys}

    def read_train_metric(self, name, epochs, M, B):
        out = np.zeros((epochs, M, B), np.float32)
        self._c(lib().aleppo_read_train_metric(self._ctx, METRIC_FIELDS[name], _ptr(out), C.c_size_t(out.size)))
        return out

    def set_batch(self, observations, actions, log_probabilities, advantages, returns, masks):
        obs = _u8(observations)
        self._c(lib().aleppo_set_batch(self._ctx, _ptr(obs), _ptr(_i64(actions)), _ptr(_f32(log_probabilities)),
                                       _ptr(_f32(advantages)), _ptr(_f32(returns)), _ptr(_u8(masks)),
                                       C.c_int64(obs.shape[0])))

    def forward(self, observations):
        obs = _u8(observations)
        n = obs.shape[0]
        logits = np.zeros((n, self.A), np.float32)
        values = np.zeros(n, np.float32)
        self._c(lib().aleppo_forward(self._ctx, _ptr(obs), C.c_int64(n), _ptr(logits), _ptr(values)))
        return logits, values

    def read_batch(self, name):
        E, T, A = self.E, self.T, self.A
        shapes = dict(observations=((E, T, 4, 84, 84), np.uint8), actions=((E, T), np.int64),
                      rewards=((E, T), np.float32), masks=((E, T), np.uint8), logits=((E, T, A), np.float32),
                      values=((E, T), np.float32), advantages=((E, T), np.float32), returns=((E, T), np.float32),
                      log_probs=((E, T, A), np.float32), terminals=((E, T), np.uint8),
                      truncations=((E, T), np.uint8), current_obs=((E, 4, 84, 84), np.uint8),
                      next_values=((E,), np.float32))
        shp, dt = shapes[name]
        out = np.zeros(shp, dt)
        self._c(lib().aleppo_read_batch(self._ctx, FIELDS[name], _ptr(out), C.c_size_t(out.nbytes)))
        return out

    # -- multi GPU --
    @staticmethod
    def comm_unique_id():
        buf = (C.c_uint8 * UNIQUE_ID_BYTES)()
        _check(lib().aleppo_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id):
        buf = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        self._c(lib().aleppo_comm_init(self._ctx, buf))

    # -- measurement --
    def profile(self, on=True):
        self._c(lib().aleppo_profile_enable(self._ctx, int(on)))

    def profile_reset(self):
        self._c(lib().aleppo_profile_reset(self._ctx))

    def profile_read(self, name):
        ms = C.c_double()
        n = C.c_int64()
        self._c(lib().aleppo_profile_read(self._ctx, KERNEL_CLASSES[name], C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def set_option(self, option, value):
        """aleppo_set_option: per-context A/B switches (OPT_*)"""
        self._c(lib().aleppo_set_option(self._ctx, int(option), int(value)))

    def get_option(self, option):
        v = C.c_int64()
        self._c(lib().aleppo_get_option(self._ctx, int(option), C.byref(v)))
        return v.value

    def set_generic_conv(self, on):
        """A/B switch: run bf16 convolutions on the generic gather-GEMM kernels (this context only)."""
        self.set_option(OPT_GENERIC_CONV, on)

    def synchronize(self):
        self._c(lib().aleppo_synchronize(self._ctx))


def learning_rate(lr0, rollout_index, num_rollouts):
    """linear anneal of main() (src/bin/train.cc:424-428)"""
    return lr0 * (1.0 - rollout_index / float(num_rollouts))
