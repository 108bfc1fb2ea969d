"""Pin the CPU oracle (oracle/oracle.c) before trusting it: the reference's own known-answer tests
and golden outputs of the reference's compiled hot-path sources (tests/golden/ref_golden.npz,
made by oracle/make_golden.py).  CPU-only."""
import numpy as np
import pytest

import hashfill as hf
import oracle_lib as orc
from conftest import GAE_KATS, gae_kat_expected


@pytest.mark.parametrize("k", range(6))
def test_gae_kats(golden, k):
    r, v, nv, te, tr, st = GAE_KATS[k]
    adv = orc.gae(r, v, nv, te, tr, st)
    np.testing.assert_allclose(adv, gae_kat_expected(r, v, nv, te, tr, st), atol=1e-5)  # gae-test.cc EXPECT_NEAR 1e-5
    np.testing.assert_allclose(adv, golden[f"kat{k}_adv"], atol=1e-6)


def test_gae_rejects_overlapping_flags():
    with pytest.raises(ValueError):
        orc.gae([[1, 1]], [[0, 0]], [0], [[1, 0]], [[0, 0]], [[1, 0]])


def test_g1_buffer_get(golden):
    r, v, nv, te, tr, st = hf.g1_inputs()
    assert te.sum() > 100 and tr.sum() > 20 and st.sum() > 100 and (np.abs(r) > 1).sum() > 1000
    o = orc.buffer_get(r, v, nv, te, tr, st)
    np.testing.assert_array_equal(o["rewards"], golden["g1_rewards_clamped"])
    np.testing.assert_array_equal(o["masks"], golden["g1_masks"])
    np.testing.assert_allclose(o["advantages"], golden["g1_adv"], atol=2e-6)
    np.testing.assert_allclose(o["returns"], golden["g1_returns"], atol=2e-6)


def test_vision_constants():
    # test/ai/vision-test.cc:5-31: constant images are preserved exactly, shape [.,84,84]
    out = orc.area_resize(np.ones((1, 210, 160), np.float32))
    assert out.shape == (1, 84, 84) and (out == 1).all()
    inp = np.stack([np.full((210, 160), i, np.float32) for i in range(4)])[None]
    out = orc.area_resize(inp)
    assert out.shape == (1, 4, 84, 84) and out.dtype == np.float32
    for i in range(4):
        assert (out[0, i] == i).all()


def test_g7_vision(golden):
    inp = hf.hf_bytes(701, (1, 2, 210, 160)).astype(np.float32)
    np.testing.assert_allclose(orc.area_resize(inp), golden["g7_area"], rtol=1e-6, atol=1e-4)
    rgb = hf.hf_bytes(702, (1, 1, 3, 84, 84)).astype(np.float32)
    np.testing.assert_allclose(orc.rgb_to_gray(rgb), golden["g7_luma"], rtol=1e-6, atol=1e-4)


def test_preprocess_matches_area_then_round_then_max():
    raw = hf.hf_bytes(711, (3, 2, 210, 160))
    lut = ((np.arange(256) * 7 + 3) % 256).astype(np.uint8)
    want = np.rint(orc.area_resize(lut[raw].astype(np.float32))).astype(np.uint8).max(axis=1)
    np.testing.assert_array_equal(orc.preprocess(raw, lut), want)
    np.testing.assert_array_equal(orc.preprocess(raw), np.rint(orc.area_resize(raw.astype(np.float32))).max(axis=1))


@pytest.mark.parametrize("A", [4, 6])
def test_g2_losses(golden, A):
    B = 256
    logits = hf.hf_range(201 + A, (B, A), -2, 2)
    actions = (hf.hf_u32(205 + A, B) % np.uint32(A)).astype(np.int64)
    masks = (hf.hf_unit(209, B) >= np.float32(0.05)).astype(np.uint8)
    adv = hf.hf_range(206, (B,), -2, 2)
    values = hf.hf_range(207, (B,), -1, 1)
    returns = hf.hf_range(208, (B,), -1.5, 1.5)
    s = f"g2_A{A}_"
    old_lp = golden[s + "old_logp"]
    # our own log-softmax of the same old logits agrees with the reference's normalize_logits
    old_logits = logits + hf.hf_range(203 + A, (B, A), -0.6, 0.6)
    np.testing.assert_allclose(orc.log_softmax(old_logits), old_lp, atol=2e-6)
    o = orc.ppo_loss(logits, old_lp, actions, adv, values, returns, masks, 0.1, 0.5, 0.01)
    ratio = golden[s + "ratio"]
    assert (ratio > 1.1).sum() > 20 and (ratio < 0.9).sum() > 20 and (masks == 0).sum() > 5
    np.testing.assert_allclose(o["loss"], golden[s + "loss"][0], atol=1e-6)
    for k in ("clipped", "value_losses", "entropies", "total_losses", "ratio"):
        np.testing.assert_allclose(o[k], golden[s + k], atol=2e-6, err_msg=k)
    np.testing.assert_allclose(o["dlogits"], golden[s + "dlogits"], atol=1e-7)
    np.testing.assert_allclose(o["dvalues"], golden[s + "dvalues"], atol=1e-8)


@pytest.mark.parametrize("H,A", [(32, 4), (32, 6), (512, 4), (512, 6)])
def test_g3_forward(golden, H, A):
    obs = hf.hf_bytes(301, (8, 4, 84, 84))
    params = hf.fill_params(310, H, A)
    assert params.size == orc.param_count(H, A)
    logits, values = orc.net_forward(params, H, A, obs)
    np.testing.assert_allclose(logits, golden[f"g3_H{H}_A{A}_logits"], atol=2e-5)
    np.testing.assert_allclose(values, golden[f"g3_H{H}_A{A}_values"], atol=2e-5)


def _g4_inputs(golden):
    N, A = 64, 4
    obs = hf.hf_bytes(401, (N, 4, 84, 84))
    actions = (hf.hf_u32(402, N) % np.uint32(A)).astype(np.int64)
    masks = (hf.hf_unit(406, N) >= np.float32(0.1)).astype(np.uint8)
    adv = hf.hf_range(404, (N,), -1, 1)
    ret = hf.hf_range(405, (N,), -1, 1)
    return obs, actions, golden["g4_old_logp"], adv, ret, masks


@pytest.mark.parametrize("name,epochs,M", [("a", 1, 1), ("b", 2, 4)])
def test_g4_train(golden, name, epochs, M):
    H, A = 32, 4
    obs, actions, old_lp, adv, ret, masks = _g4_inputs(golden)
    params = hf.fill_params(410, H, A)
    o = orc.train(params, H, A, obs, actions, old_lp, adv, ret, masks, epochs, M)
    s = f"g4{name}_"
    np.testing.assert_allclose(o["loss"], golden[s + "loss"], atol=1e-4)
    np.testing.assert_allclose(o["grad_norm"], golden[s + "grad_norm"], rtol=1e-4)
    for k in ("total_losses", "ratio", "entropies", "value_losses", "clipped"):
        np.testing.assert_allclose(o[k], golden[s + k], atol=1e-4, err_msg=k)
    offs = orc.param_offsets(H, A)
    for k in range(12):
        p = o["params"][offs[k]:offs[k + 1]].astype(np.float64)
        g = o["last_grads"][offs[k]:offs[k + 1]].astype(np.float64)
        idx = (hf.hf_u32(499, 64) % np.uint32(p.size)).astype(np.int64)
        np.testing.assert_allclose(p[idx], golden[s + "param_samples"][k], atol=2e-5, err_msg=f"param {k}")
        np.testing.assert_allclose([p.sum(), (p * p).sum()], golden[s + "param_sums"][k], rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(g[idx], golden[s + "grad_samples"][k], atol=2e-5, rtol=1e-3, err_msg=f"grad {k}")
        np.testing.assert_allclose((g * g).sum(), golden[s + "grad_sums"][k][1], rtol=2e-3, atol=1e-9)
    shapes = hf.param_shapes(H, A)
    for k in (0, 1, 3, 5, 7, 8, 9, 10, 11):
        np.testing.assert_allclose(o["params"][offs[k]:offs[k + 1]].reshape(shapes[k]), golden[s + f"param{k}"],
                                   atol=2e-5)


@pytest.mark.parametrize("A", [4, 6])
def test_g5_sampling_bit_exact(golden, A):
    s = f"g5_A{A}_"
    np.testing.assert_array_equal(orc.sample(golden[s + "probs"], golden[s + "q"]), golden[s + "actions"])
    E = golden[s + "probs"].shape[0]
    np.testing.assert_allclose(orc.softmax(hf.hf_range(501 + A, (E, A), -3, 3)), golden[s + "probs"], atol=1e-6)


def _planes(obs):
    o = obs.reshape(obs.shape[0], obs.shape[1], -1).astype(np.int64)
    w = np.arange(1, o.shape[2] + 1, dtype=np.int64)
    return np.stack([o.sum(-1), (o * w).sum(-1)], -1)


def test_g6_frame_stack(golden):
    E = 4
    obs = np.zeros((E, 4, 84, 84), np.uint8)
    for step in range(6):
        frames = hf.hf_bytes(601 + step, (E, 84, 84))
        start = np.array([step == 0 or (step == 3 and e == 1) or (step == 4 and e == 2) for e in range(E)], np.uint8)
        obs = orc.update_observations(obs, frames, start)
        np.testing.assert_array_equal(_planes(obs), golden["g6_checksums"][step])
    np.testing.assert_array_equal(obs[1], golden["g6_final_obs_env1"])


def test_clip_and_adam_closed_form():
    H, A = 32, 4
    n = orc.param_count(H, A)
    g = hf.hf_range(801, (n,), -1, 1)
    norm, gc = orc.clip_grad_norm(g, H, A, 0.5)
    np.testing.assert_allclose(norm, np.sqrt((g.astype(np.float64) ** 2).sum()), rtol=1e-6)
    np.testing.assert_allclose(gc, g * np.float32(0.5 / (norm + 1e-6)), rtol=1e-6)
    p = hf.hf_range(802, (n,), -1, 1)
    p1, m1, v1 = orc.adam_step(p, gc, np.zeros(n), np.zeros(n), 2.5e-4, 1)
    # first Adam step in closed form: m/(1-b1) = g, sqrt(v/(1-b2)) = |g|  ->  p - lr * g / (|g| + eps)
    np.testing.assert_allclose(p1, p - 2.5e-4 * gc / (np.abs(gc) + 1e-5), rtol=0, atol=1e-6)
