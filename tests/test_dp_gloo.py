"""world_size-2 CPU test (gloo) of the data-parallel decomposition used by aleppo_train (SURVEY 8e,
DESIGN.md section 5): every rank scales its per-sample gradients by 1/N_m(GLOBAL), the flat gradients are
all-reduced with SUM, clip + Adam run replicated.  Claim: this equals the 1-rank update on the batch whose
minibatch k is the concatenation over ranks of the local minibatch k.  The compute on each rank is done by
the CPU oracle (the checker) - the product's HIP path cannot run here; what is tested is the host-side
sharding logic and the algebra the RCCL path relies on."""
import os
import socket

import numpy as np
import pytest

import hashfill as hf
import oracle_lib as orc
from __graft_entry__ import load_package

H, A, EG, T, M = 32, 4, 2, 8, 2  # 2 envs per rank, 16 samples per rank, 2 minibatches
WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _global_batch():
    N = WORLD * EG * T
    obs = hf.hf_bytes(901, (N, 4, 84, 84))
    actions = (hf.hf_u32(902, N) % np.uint32(A)).astype(np.int64)
    old_lp = orc.log_softmax(hf.hf_range(903, (N, A), -1, 1))
    adv = hf.hf_range(904, (N,), -1, 1)
    ret = hf.hf_range(905, (N,), -1, 1)
    masks = (hf.hf_unit(906, N) >= np.float32(0.25)).astype(np.uint8)  # uneven mask counts across ranks
    return obs, actions, old_lp, adv, ret, masks


def _worker(rank, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    pkg = load_package()
    envs = pkg.env_shard(WORLD * EG, WORLD, rank)  # contiguous env block of this rank
    full = _global_batch()
    rows = np.concatenate([np.arange(e * T, (e + 1) * T) for e in envs])
    obs, actions, old_lp, adv, ret, masks = [x[rows] for x in full]
    params = hf.fill_params(910, H, A)
    m = np.zeros_like(params)
    v = np.zeros_like(params)
    B = len(rows) // M
    # mask counts of all minibatches, all-reduced once (aleppo_train does the same with RCCL)
    counts = torch.tensor([masks[k * B:(k + 1) * B].sum() for k in range(M)], dtype=torch.float32)
    dist.all_reduce(counts)
    losses = []
    # the launcher broadcasts rank 0's 128-byte communicator id exactly like bench.py does
    uid = [bytes(range(128)) if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    assert uid[0] == bytes(range(128)) and len(uid[0]) == pkg.UNIQUE_ID_BYTES
    for k in range(M):
        sl = slice(k * B, (k + 1) * B)
        logits, values, acts = orc.net_forward(params, H, A, obs[sl], want_acts=True)
        o = orc.ppo_loss(logits, old_lp[sl], actions[sl], adv[sl], values, ret[sl], masks[sl], n_mask=float(counts[k]))
        g = torch.from_numpy(orc.net_backward(params, H, A, acts, o["dlogits"], o["dvalues"]))
        loss = torch.tensor([o["loss"]])  # = local masked sum / global count
        dist.all_reduce(g)
        dist.all_reduce(loss)
        norm, gc = orc.clip_grad_norm(g.numpy(), H, A, 0.5)
        params, m, v = orc.adam_step(params, gc, m, v, 2.5e-4, k + 1)
        losses.append((float(loss), float(norm)))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), params=params, losses=np.array(losses))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_update_equals_one_rank_on_interleaved_batch(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.start_processes(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True, start_method="spawn")
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(r0["params"], r1["params"])  # replicated clip + Adam stay in lock-step
    # 1-rank oracle on the batch whose minibatch k = concat over ranks of local minibatch k
    full = _global_batch()
    B = EG * T // M
    order = np.concatenate([np.concatenate([np.arange(r * EG * T + k * B, r * EG * T + (k + 1) * B)
                                            for r in range(WORLD)]) for k in range(M)])
    w = orc.train(hf.fill_params(910, H, A), H, A, *[x[order] for x in full], 1, M)
    np.testing.assert_allclose(r0["losses"][:, 0], w["loss"].ravel(), atol=1e-5)
    np.testing.assert_allclose(r0["losses"][:, 1], w["grad_norm"].ravel(), rtol=1e-5)
    np.testing.assert_allclose(r0["params"], w["params"], atol=1e-6)


def test_env_shard_and_lr_anneal():
    pkg = load_package()
    assert list(pkg.env_shard(1024, 8, 3)) == list(range(384, 512))
    with pytest.raises(pkg.AleppoInvalidArgument):
        pkg.env_shard(10, 4, 0)
    # lr0 * (1 - i / num_rollouts), src/bin/train.cc:424-428
    assert pkg.learning_rate(2.5e-4, 0, 10) == 2.5e-4
    assert abs(pkg.learning_rate(2.5e-4, 5, 10) - 1.25e-4) < 1e-12
